"""Host-side silence trim (behaviour of reference ``mars5/trim.py:110-178`` -- librosa's ``effects.trim`` carried
over to torch -- with its defaults: frame_length 2048, hop_length 512, centred REFLECT padding, mono = mean over
channels, power in dB relative to the loudest frame, fp32) and weight-norm removal (reference ``mars5/utils.py:45-62``).
CPU post-processing after the vocoder -- out of the accelerated path; checked against the reference's
output in ``tests/test_oracle_golden.py::test_trim_matches_reference``."""
from __future__ import annotations

import logging
from typing import Tuple

import numpy as np
import torch
import torch.nn.functional as F


def _frame_power(y: torch.Tensor, frame_length: int, hop_length: int) -> torch.Tensor:
    """Mean power of every analysis frame of the (mono) signal, centred frames with reflect padding
    (reference ``rms`` with center=True, trim.py:255-263, squared)."""
    pad = frame_length // 2
    yp = F.pad(y[None, None], (pad, pad), mode="reflect")[0, 0]
    frames = yp.unfold(0, frame_length, hop_length)          # (n_frames, frame_length)
    return torch.mean(frames.abs() ** 2, dim=-1)


def trim(y, top_db: float = 60, ref=torch.max, frame_length: int = 2048, hop_length: int = 512) -> Tuple[torch.Tensor, torch.Tensor]:
    """Trim leading / trailing silence: frames whose power is more than `top_db` dB below `ref` (of the frame
    powers; a callable or a number) are silent.  y: (n,) or (channels, n).  Returns (y[..., start:end], [start, end])."""
    yt = y if isinstance(y, torch.Tensor) else torch.as_tensor(np.asarray(y))
    mono = torch.mean(yt, dim=0) if yt.dim() > 1 else yt
    power = _frame_power(mono.to(torch.float32), frame_length, hop_length)
    amin = torch.tensor(1e-10, device=power.device)
    ref_value = ref(power) if callable(ref) else torch.abs(torch.as_tensor(ref, dtype=torch.float32))
    db = 10.0 * torch.log10(torch.maximum(amin, power)) - 10.0 * torch.log10(torch.maximum(amin, ref_value))
    nz = torch.nonzero(db > -top_db).reshape(-1)
    if nz.numel() > 0:
        start = int(nz[0]) * hop_length                      # end goes one frame past the last non-silent one
        end = min(int(yt.shape[-1]), (int(nz[-1]) + 1) * hop_length)
    else:
        start, end = 0, 0
    return yt[..., start:end], torch.tensor([start, end], device="cpu")


def trim_device(y: torch.Tensor, top_db: float = 60, frame_length: int = 2048, hop_length: int = 512) -> Tuple[torch.Tensor, torch.Tensor]:
    """``trim`` for a waveform that lives on the GPU (the vocoder's output): frame powers and the [start, end) search run
    in two kernels (``m5_trim_bounds``), only the two indices come back to the host.  Same semantics as ``trim`` with
    ref = max; y: (n,) or (channels, n) fp32 on a cuda device."""
    from . import ops
    mono = (torch.mean(y, dim=0) if y.dim() > 1 else y).to(torch.float32).contiguous()
    if mono.shape[0] <= frame_length // 2:                 # shorter than the reflect padding: nothing to analyse
        return y, torch.tensor([0, int(y.shape[-1])], device="cpu")
    b = ops.trim_bounds(mono, top_db, frame_length, hop_length).cpu()
    start, end = int(b[0]), int(b[1])
    return y[..., start:end], torch.tensor([start, end], device="cpu")


def nuke_weight_norm(module) -> None:
    """Recursively remove weight normalisation (Encodec / Vocos only)."""
    try:
        torch.nn.utils.remove_weight_norm(module)
        logging.debug(f"Removed weight norm from {module.__class__.__name__}")
    except ValueError:
        pass
    for child in module.children():
        nuke_weight_norm(child)

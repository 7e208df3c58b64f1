"""Host-side silence trim (behaviour of ``librosa.effects.trim`` as used by reference
``mars5/trim.py:110-178`` with its defaults: frame_length 2048, hop_length 512, centred
reflect padding, ref = max) and weight-norm removal (reference ``mars5/utils.py:45-62``).
CPU post-processing after the vocoder -- out of the accelerated path."""
from __future__ import annotations

import logging
from typing import Tuple

import numpy as np
import torch


def _frame_rms(y: np.ndarray, frame_length: int, hop_length: int) -> np.ndarray:
    pad = frame_length // 2
    yp = np.pad(y, [(0, 0)] * (y.ndim - 1) + [(pad, pad)], mode="constant")
    n = 1 + (yp.shape[-1] - frame_length) // hop_length
    idx = np.arange(frame_length)[None, :] + hop_length * np.arange(n)[:, None]
    frames = yp[..., idx]                                   # (..., n, frame_length)
    return np.sqrt(np.mean(np.abs(frames) ** 2, axis=-1))   # (..., n)


def trim(y, top_db: float = 60, ref=np.max, frame_length: int = 2048, hop_length: int = 512, aggregate=np.max) -> Tuple[torch.Tensor, np.ndarray]:
    """Trim leading / trailing silence: frames whose RMS is more than `top_db` dB below `ref`
    (of the RMS curve) are silent.  Returns (trimmed signal, [start, end] sample interval)."""
    is_tensor = isinstance(y, torch.Tensor)
    arr = y.detach().cpu().numpy() if is_tensor else np.asarray(y)
    mse = _frame_rms(arr.astype(np.float64), frame_length, hop_length)
    amin = 1e-5
    magnitude = np.abs(mse)
    ref_value = np.abs(ref(magnitude)) if callable(ref) else np.abs(ref)
    db = 20.0 * np.log10(np.maximum(amin, magnitude)) - 20.0 * np.log10(np.maximum(amin, ref_value))
    non_silent = db > -top_db
    if non_silent.ndim > 1:
        non_silent = np.apply_over_axes(aggregate, non_silent, range(non_silent.ndim - 1)).reshape(-1)
    nz = np.flatnonzero(non_silent)
    if nz.size > 0:
        start = int(nz[0] * hop_length)
        end = min(arr.shape[-1], int((nz[-1] + 1) * hop_length))
    else:
        start, end = 0, 0
    out = arr[..., start:end]
    return (torch.from_numpy(np.ascontiguousarray(out)) if is_tensor else out), np.asarray([start, end])


def nuke_weight_norm(module) -> None:
    """Recursively remove weight normalisation (Encodec / Vocos only)."""
    try:
        torch.nn.utils.remove_weight_norm(module)
        logging.debug(f"Removed weight norm from {module.__class__.__name__}")
    except ValueError:
        pass
    for child in module.children():
        nuke_weight_norm(child)

"""Host-side constant tables (built once, uploaded): sine positional table, RoPE cos/sin,
sinusoidal timestep inputs, multinomial-diffusion schedule constants, EOS-penalty table.

These are weight-independent constants the reference also builds on the host at module
construction; they are evaluated here with the same torch expressions so the device
kernels consume bit-identical tables (feed kernels the table built the
reference's way rather than recomputing sin/cos in-kernel).
"""
from __future__ import annotations

import math
from typing import List

import numpy as np
import torch

# Every table is built on the HOST whatever the ambient default device is (`with torch.device(dev)`,
# `torch.set_default_device`): the reference builds them at module construction on the CPU, and callers
# upload the result.  A factory call without `device=` here is a bug (tests/test_host_cpu.py runs every
# function under a non-CPU default device).
CPU = torch.device("cpu")


def sine_pe(n: int, dim: int) -> torch.Tensor:
    """reference nn_future.py:51-76 (``SinePositionalEmbedding.extend_pe``)."""
    pe = torch.zeros(n, dim, device=CPU)
    position = torch.arange(0, n, dtype=torch.float32, device=CPU).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, dim, 2, dtype=torch.float32, device=CPU) * -(math.log(10000.0) / dim))
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe


def rope_table(head_dim: int, n_pos: int, theta: float = 10000.0) -> torch.Tensor:
    """reference nn_future.py:194-198 (``precompute_freqs_cis``) as (n_pos, head_dim/2, 2)
    fp32 [cos, sin]."""
    freqs = 1.0 / (theta ** (torch.arange(0, head_dim, 2, device=CPU)[: (head_dim // 2)].float() / head_dim))
    t = torch.arange(n_pos, device=CPU)
    freqs = torch.outer(t, freqs).float()
    return torch.view_as_real(torch.polar(torch.ones_like(freqs), freqs)).contiguous()


def timestep_inputs(times: List[int], dim: int, max_period: int = 10000) -> torch.Tensor:
    """reference model.py:18-35 (``timestep_embedding``): rows cos || sin for each t."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(start=0, end=half, device=CPU) / half)
    args = torch.tensor(times, dtype=torch.long, device=CPU)[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def diffusion_log_tables(timesteps: int = 200, s: float = 0.008):
    """reference diffuser.py:63-109: (log_alpha, log_1_min_alpha, log_cumprod_alpha,
    log_1_min_cumprod_alpha), each fp32 (timesteps,)."""
    steps = timesteps + 1
    x = torch.linspace(0, timesteps, steps, device=CPU)
    ac = torch.cos(((x / timesteps) + s) / (1 + s) * torch.pi * 0.5) ** 2
    ac = ac / ac[0]
    alphas = torch.sqrt(torch.clamp(ac[1:] / ac[:-1], 0.001, 1.0)).to(torch.float64)
    la = alphas.log()
    lca = torch.cumsum(la, dim=-1)
    l1ma = torch.log((1 - la.exp()).clamp_(min=1e-30))
    l1mca = torch.log((1 - lca.exp()).clamp_(min=1e-30))
    return la.float(), l1ma.float(), lca.float(), l1mca.float()


def nar_step_consts(times: List[int], num_classes: int = 1025, timesteps: int = 200, tables=None) -> torch.Tensor:
    """Per reverse step (in schedule order) the eight scalars m5_nar_sample consumes:
    [lca[t-1], l1mca[t-1]-lnK, la[t], l1ma[t]-lnK, lca[t], l1mca[t]-lnK, t, 0].
    The ``- np.log(K)`` is done on fp32 tensors with a python double exactly as in
    diffuser.py:130-133,169-172.  `tables` = (log_alpha, log_1_min_alpha, log_cumprod_alpha,
    log_1_min_cumprod_alpha) of the caller's ``MultinomialDiffusion`` (the reference reads them from `diff` in
    q_pred / q_posterior, diffuser.py:118-206); None = the default schedule (`timesteps`, s = 0.008)."""
    if tables is None:
        tables = diffusion_log_tables(timesteps)
    la, l1ma, lca, l1mca = [t.detach().to("cpu", torch.float32) for t in tables]
    assert all(0 <= t < la.shape[0] for t in times), f"schedule step outside the diffusion's {la.shape[0]} timesteps"
    lnK = np.log(num_classes)
    # one gather per column (was a python loop of 200 x 8 scalar tensors: 3 ms of host time in front of every utterance's decode);
    # fp32 tensor - python double rounds per element exactly as the scalar form did
    t_idx = torch.tensor(list(times), dtype=torch.long, device=CPU)
    tm1 = (t_idx - 1).clamp_(min=0)
    cols = [lca[tm1], l1mca[tm1] - lnK, la[t_idx], l1ma[t_idx] - lnK, lca[t_idx], l1mca[t_idx] - lnK,
            t_idx.to(torch.float32), torch.zeros(len(times), dtype=torch.float32, device=CPU)]
    return torch.stack(cols, dim=1).float().contiguous()


def log_eps() -> float:
    """log(clamp(0, 1e-7)) as torch computes it in fp32 (diffuser.py:45)."""
    return float(torch.log(torch.tensor(1e-7, dtype=torch.float32, device=CPU)))


def eos_penalty_table(n_est: int, decay: float, factor: float) -> torch.Tensor:
    """samplers.py:47-56: modifier(n) = factor * max(n_est - n, 1) ** decay for n = 0..n_est
    (python double, then rounded to fp32 as the in-place fp32 subtraction does)."""
    vals = [factor * (max(n_est - n, 1) ** decay) for n in range(n_est + 1)]
    return torch.tensor(vals, dtype=torch.float64, device=CPU).to(torch.float32)


def reverse_schedule(T: int) -> List[int]:
    """reference diffuser.py:318-333 (``get_schedule``) at the shipped jump_len =
    jump_n_sample = 1: T-1, ..., 0 (every transition is a reverse step)."""
    return list(range(T - 1, -1, -1))

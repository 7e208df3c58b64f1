"""Drop-in for the inference part of reference ``mars5/diffuser.py``: ``MultinomialDiffusion``
(schedule tables, :62-95), ``DSH`` (:302-315), ``get_schedule`` (:318-333) and
``perform_simple_inference`` (:398-472).  The reverse-diffusion loop runs on the MI355X
engine (``nar_engine``).  Training-only pieces (compute_Lt, kl_prior, ...) and the RePaint
forward-jump branch (dead at the shipped jump_len = jump_n_sample = 1) are out of scope."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Optional, Union

import torch
from torch import Tensor

from .nar_engine import NARConfig, NARSession
from .tables import diffusion_log_tables


class MultinomialDiffusion:
    def __init__(self, num_classes, timesteps=100, diffusion_s=0.008, loss_type='vb_stochastic', parametrization='x0',
                 dtype=torch.float32, device='cpu'):
        assert loss_type in ('vb_stochastic',)
        assert parametrization in ('x0', 'direct')
        self.num_classes = num_classes
        self.loss_type = loss_type
        self.num_timesteps = timesteps
        self.parametrization = parametrization
        la, l1ma, lca, l1mca = diffusion_log_tables(timesteps, diffusion_s)
        self.log_alpha = la.to(dtype).to(device)
        self.log_1_min_alpha = l1ma.to(dtype).to(device)
        self.log_cumprod_alpha = lca.to(dtype).to(device)
        self.log_1_min_cumprod_alpha = l1mca.to(dtype).to(device)


@dataclass
class DSH():
    jump_len: int = 1
    jump_n_sample: int = 1
    last_greedy: bool = False          # never forwarded by the reference loop (diffuser.py:461)
    x_0_temp: float = 1.0
    guidance_w: float = 1.0
    enable_kevin_scaled_inference: bool = True
    T_override: Union[None, int] = None
    deep_clone: bool = False
    q0_override_steps: int = 0
    progress: bool = False


def get_schedule(t_T, jump_len=10, jump_n_sample=10):
    jumps = {j: jump_n_sample - 1 for j in range(0, t_T - jump_len, jump_len)}
    t, ts = t_T, []
    while t >= 1:
        t -= 1
        ts.append(t)
        if jumps.get(t, 0) > 0:
            jumps[t] -= 1
            for _ in range(jump_len):
                t += 1
                ts.append(t)
    ts.append(-1)
    return ts


@torch.inference_mode()
def perform_simple_inference(model, batch: tuple, diff: MultinomialDiffusion, T, dtype=torch.float16,
                             retain_quant0: bool = True, dsh=DSH,
                             uniform: Optional[Callable[[tuple], Tensor]] = None, randint: Optional[Callable] = None,
                             use_graph: bool = True, div_mode: int = 0, n_steps: Optional[int] = None) -> Tensor:
    """batch = (c_text (1,Lt), c_codes (1,Lc,8), c_text_lengths, c_codes_lengths, x (1,Lx,8),
    x_padding_mask); returns (1, S - offset, 8) int64.  RNG draws follow the reference order:
    randint(0,K,(1,Lx,8)) then per step rand (1,S,8,K) x2 (x1 at t = 0).
    `uniform(shape)` / `randint(shape)` override the device generator (parity tests)."""
    c_text, c_codes, c_text_lengths, c_codes_lengths, x, x_padding_mask = batch
    assert c_text.shape[0] == 1, "batch size 1 per call (the reference breaks for bs > 1, SURVEY App. B-9)"
    assert retain_quant0, "retain_quant0=False is not a shipped configuration (inference.py:298)"
    if dsh.jump_len != 1 or dsh.jump_n_sample != 1:
        raise NotImplementedError("RePaint resampling (jump_len/jump_n_sample != 1) is outside the shipped inference path")
    eng = model.engine()
    dev = eng.dev
    K = diff.num_classes
    assert K == eng.shape.n_quant
    times = get_schedule(T, jump_n_sample=dsh.jump_n_sample, jump_len=dsh.jump_len)[:-1]
    cfg = NARConfig(T=T, x_0_temp=float(dsh.x_0_temp), guidance_w=float(dsh.guidance_w), deep_clone=bool(dsh.deep_clone),
                    q0_override_steps=int(dsh.q0_override_steps), div_mode=div_mode)
    sess = NARSession(eng, cfg)
    with torch.cuda.stream(sess.stream):
        x = x.to(dev)
        c_codes = c_codes.to(dev)
        assert int(x.max()) < K, f'Error: {int(x.max())} >= {K}'           # diffuser.py:36
        x_quant0 = x[0, :, 0].clone()
        if randint is None:
            xr = torch.randint(0, K, x.shape, dtype=x.dtype, device=dev)[0]
        else:
            xr = randint(tuple(x.shape)).to(dev)[0]
        xr[:, 0] = x_quant0
        x_known = torch.zeros_like(xr)
        x_known[:, 0] = xr[:, 0]
        m = torch.zeros_like(xr, dtype=torch.uint8)
        m[:, 0] = 1
        offset = 0
        if dsh.deep_clone:
            prompt = c_codes[0]
            xr = torch.cat((prompt, xr), dim=0)
            x_known = torch.cat((prompt, x_known), dim=0)
            m = torch.cat((torch.ones_like(prompt, dtype=torch.uint8), m), dim=0)
            offset = int(c_codes_lengths[0])
        if uniform is None:
            uniform = lambda shape: torch.rand(shape, dtype=torch.float32, device=dev)   # noqa: E731
    sess.prepare(c_text[0], c_codes[0], xr, x_known, m, offset, times)
    out = sess.run(uniform, use_graph=use_graph, n_steps=n_steps)
    return out[None, offset:].clone()

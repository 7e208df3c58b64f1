#!/usr/bin/env python
"""Effective shader clock while a GEMM runs: workgroup 0 records clock64() (shader cycles) and
wall_clock64() (100 MHz) around its main loop."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("M5_HIP_TOOLS", "1")      # tools run on libmars5_hip_tools.so (knobs, probes; csrc/common.h)
import torch
import mars5_tts_amd as pkg            # noqa
from mars5_tts_amd import ops, _lib as L

dev = torch.device("cuda:0")
dt = torch.bfloat16
buf = torch.zeros(16, dtype=torch.int64, device=dev)
for (M, N, K, epi, cfg) in ((2816, 6144, 1024, L.EPI_SWIGLU, 1), (2816, 6144, 1024, L.EPI_SWIGLU, 0), (2816, 3072, 1024, L.EPI_DT, 0),
                            (15600, 6144, 1024, L.EPI_SWIGLU, 0), (15600, 6144, 1024, L.EPI_SWIGLU, 1), (2816, 1024, 3072, L.EPI_RESIDUAL, 3)):
    os.environ["M5_GEMM_CFG"] = str(cfg)
    a = (torch.randn(M, K) * 0.5).to(dev, dt)
    w = (torch.randn(N, K) / K ** 0.5).to(dev, dt)
    out = torch.zeros(M, N // 2 if epi == L.EPI_SWIGLU else N, dtype=torch.float32 if epi == L.EPI_RESIDUAL else dt, device=dev)
    for _ in range(20):
        ops.gemm(a, w, out, epi)
    torch.cuda.synchronize()
    L.check(L.lib.m5_debug_gemm_clock(buf.data_ptr()))
    for _ in range(5):
        ops.gemm(a, w, out, epi)
    torch.cuda.synchronize()
    L.check(L.lib.m5_debug_gemm_clock(None))
    b = buf.cpu().tolist()
    cyc, wall = b[2] - b[0], (b[3] - b[1]) * 10e-9
    e0, e1 = ops.Event(), ops.Event()
    st = torch.cuda.current_stream().cuda_stream
    e0.record(st)
    for _ in range(20):
        ops.gemm(a, w, out, epi)
    e1.record(st)
    torch.cuda.synchronize()
    tot = e0.elapsed_ms(e1) * 1e3 / 20
    print(f"M={M} N={N} K={K} cfg={cfg}: kernel {tot:.1f} us (eager back-to-back); WG 0: prologue+main loop {wall * 1e6:.2f} us "
          f"({cyc / wall / 1e9:.2f} GHz, {cyc / (K // 64):.0f} cyc/K-step), epilogue {(b[5] - b[3]) * 0.01:.2f} us; "
          f"last WG: starts {(b[9] - b[1]) * 0.01:+.2f} us after WG 0, prologue+main {(b[11] - b[9]) * 0.01:.2f} us, epilogue {(b[13] - b[11]) * 0.01:.2f} us", flush=True)

#!/usr/bin/env python
"""Fused query-projection + cross-attention launch vs the two separate launches at the NAR shape, with the phase
stamps of workgroup 0 (entry / end of main loop / q staged / end)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("M5_HIP_TOOLS", "1")      # tools run on libmars5_hip_tools.so (knobs, probes; csrc/common.h)
import torch
import mars5_tts_amd as pkg            # noqa
from mars5_tts_amd import ops, _lib as L
from mars5_tts_amd.blocks import CrossMemory, SeqWorkspace, cross_memory_table

os.environ["M5_GEMM_XATTN"] = "1"
dev = torch.device("cuda:0")
dt = torch.bfloat16
H, D, Sr, nb, Le, T = 16, 1024, 1408, 2, 39, 8
M = nb * Sr
ws = SeqWorkspace(nb, 1349, D, 3072, dt, dev, row_pad=64)
xn = (torch.randn(M, D) * 0.5).to(dev, dt)
w = (torch.randn(D, D) / 32).to(dev, dt)
bias = torch.randn(D, device=dev)
Lep = 64
k = torch.randn(T * nb, H, Le, 64).to(dev, dt)
vt = torch.zeros(T * nb, H, 64, Lep, device=dev, dtype=dt)
vt[..., :Le] = torch.randn(T * nb, H, 64, Le).to(dev, dt)
mem = CrossMemory(k, vt, Le, Lep, nb)
tab, max_le = cross_memory_table(mem, dev)
step = torch.tensor([3], dtype=torch.int32, device=dev)
st = torch.cuda.current_stream().cuda_stream
buf = torch.zeros(16, dtype=torch.int64, device=dev)


def separate():
    ops.gemm(xn, w, None, L.EPI_QKV, bias=bias, scatter=ws.scatter(True, False, False))
    a = L.AttnArgs(q=ws.q.data_ptr(), q_bs=H * Sr * 64, q_hs=Sr * 64, q_rs=64, k=k.data_ptr(), k_bs=H * Le * 64, k_hs=Le * 64, k_rs=64,
                   vt=vt.data_ptr(), vt_bs=H * 64 * Lep, vt_hs=64 * Lep, vt_ds=Lep, o=ws.att.data_ptr(), o_bs=Sr * D, o_rs=D, B=nb, H=H,
                   Sq=1349, Sk=Le, key_len=None, causal=0, scale=0.125, kv_index=step.data_ptr(), kv_index_stride_k=nb * H * Le * 64,
                   kv_index_stride_v=nb * H * 64 * Lep)
    ops.attention(dt, a)


def fused():
    assert ops.gemm_q_cross_attn(xn, w, bias, H, tab, max_le, Sr, step, 0.125, ws.att)


for name, fn in (("separate (q GEMM + attention)", separate), ("fused", fused)):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = ops.Event(), ops.Event()
    e0.record(st)
    for _ in range(50):
        fn()
    e1.record(st)
    torch.cuda.synchronize()
    print(f"{name:32s} {e0.elapsed_ms(e1) * 1e3 / 50:7.2f} us per call", flush=True)
L.check(L.lib.m5_debug_gemm_clock(buf.data_ptr()))
for _ in range(3):
    fused()
torch.cuda.synchronize()
L.check(L.lib.m5_debug_gemm_clock(None))
b = buf.cpu().tolist()
print(f"fused, WG 0: prologue+main loop {(b[3] - b[1]) * 0.01:.2f} us, q staging {(b[6] - b[3]) * 0.01:.2f} us, attention + stores {(b[5] - b[6]) * 0.01:.2f} us; "
      f"last WG starts {(b[9] - b[1]) * 0.01:+.2f} us after WG 0, ends {(b[13] - b[1]) * 0.01:.2f} us after WG 0's start")

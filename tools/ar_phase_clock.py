#!/usr/bin/env python
"""Where does a batch-1 decode GEMV launch spend its time?  Workgroup 0 stamps the 100 MHz wall clock at
entry / after issuing its weight loads / after the prologue hand-off / after the dot products; HIP events
give the whole-launch time (eager, back to back)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("M5_HIP_TOOLS", "1")      # tools run on libmars5_hip_tools.so (knobs, probes; csrc/common.h)
import torch
import mars5_tts_amd as pkg            # noqa
from mars5_tts_amd import ops, _lib as L

dev = torch.device("cuda:0")
dt = torch.bfloat16
D, F, H = 1536, 3584, 24
g = torch.Generator().manual_seed(0)
def w(n, k): return (torch.randn(n, k, generator=g) / k ** 0.5).to(dev, dt)
wqkv, wo, w13, w2 = w(3 * D, D), w(D, D), w(2 * F, D), w(D, F)
x = torch.randn(D, generator=g).to(dev)
nw = torch.ones(D, device=dev)
hbuf = torch.randn(F, generator=g).to(dev, dt)
state = torch.tensor([500, 10, 0, 510, -1, 0, 0, 0], dtype=torch.int32, device=dev)
rope = torch.zeros(8192, 32, 2, device=dev); rope[..., 0] = 1
kc = torch.zeros(H, 1000, 64, dtype=dt, device=dev); vc = torch.zeros_like(kc)
qbuf = torch.zeros(D, dtype=dt, device=dev)
part = torch.zeros(H, 8, L.ATTN_PART, device=dev); part[..., 65] = 1
dbg = torch.zeros(8, dtype=torch.int64, device=dev)
st = torch.cuda.current_stream().cuda_stream

def args(**kw):
    a = L.GemvArgs()
    for k, v in kw.items():
        setattr(a, k, v.data_ptr() if isinstance(v, torch.Tensor) else v)
    return a

cases = [
    ("qkv+rope (14.2 MB)", L.PRO_RMS, L.GEPI_QKV_ROPE, dict(W=wqkv, ldw=D, N=3 * D, K=D, x_f32=x, norm_w=nw, eps=1e-5, rope=rope, state=state,
                                                          kcache=kc, vcache=vc, qbuf=qbuf, w_alloc=1000, window=3000, dim=D)),
    ("wo+residual (4.7 MB)", L.PRO_ATTN, L.GEPI_RESIDUAL, dict(W=wo, ldw=D, N=D, K=D, part=part, nsplit=8, n_heads=H, xres=x, state=state)),
    ("w13+swiglu (22 MB)", L.PRO_RMS, L.GEPI_SWIGLU, dict(W=w13, ldw=D, N=2 * F, K=D, x_f32=x, norm_w=nw, eps=1e-5, y_dt=hbuf, state=state)),
    ("w2+residual (11 MB)", L.PRO_DT, L.GEPI_RESIDUAL, dict(W=w2, ldw=F, N=D, K=F, x_dt=hbuf, xres=x, state=state)),
]
for name, pro, epi, kw in cases:
    a = args(**kw)
    for _ in range(10):
        ops.ar_gemv(dt, pro, epi, a)
    torch.cuda.synchronize()
    e0, e1 = ops.Event(), ops.Event()
    e0.record(st)
    for _ in range(50):
        ops.ar_gemv(dt, pro, epi, a)
    e1.record(st)
    torch.cuda.synchronize()
    tot = e0.elapsed_ms(e1) * 1e3 / 50
    a.dbg = dbg.data_ptr()
    ops.ar_gemv(dt, pro, epi, a)
    torch.cuda.synchronize()
    b = dbg.cpu().tolist()
    print(f"{name:24s} launch {tot:5.2f} us eager back-to-back | WG0: entry->loads issued {(b[1]-b[0])*0.01:5.2f}  "
          f"->prologue done {(b[2]-b[1])*0.01:5.2f}  ->dots+reduce done {(b[3]-b[2])*0.01:5.2f} us", flush=True)

#!/usr/bin/env python
"""m5_nar_uniforms at the bench shape (S = 1349, 8 codebooks, 1025 classes; merged draw over the deep-clone mask): us per launch,
and a checksum of the buffer (two builds of the kernel must agree bit for bit)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("M5_HIP_TOOLS", "1")
import torch
from mars5_tts_amd import ops, _lib as L
from mars5_tts_amd.nar_engine import _philox_geometry, _magic_div

dev = torch.device("cuda:0")
S, Q, K, off = 1349, 8, 1025, 450
n = S * Q * K
G, inc = _philox_geometry(n, dev)
buf = torch.empty(n, dtype=torch.float32, device=dev)
rng = torch.tensor([1234, 8], dtype=torch.int64, device=dev)
mm = torch.zeros(S, Q, dtype=torch.uint8); mm[:, 0] = 1; mm[:off] = 1
mm = mm.to(dev)
consts = torch.ones(200, L.M5_NAR_CONSTS if hasattr(L, "M5_NAR_CONSTS") else 8, dtype=torch.float32, device=dev)
step = torch.zeros(1, dtype=torch.int32, device=dev)
km, ks = _magic_div(K, n)
a = L.NarUniformArgs(out=buf.data_ptr(), n=n, K=K, k_magic=km, k_shift=ks, m=mm.data_ptr(), rng=rng.data_ptr(), inc=inc, grid_threads=G,
                     step=step.data_ptr(), consts=consts.data_ptr())
st = torch.cuda.current_stream().cuda_stream
for _ in range(5):
    ops.nar_uniforms(a, stream=st)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200):
    ops.nar_uniforms(a, stream=st)
e1.record()
torch.cuda.synchronize()
print(f"{os.environ.get('M5_HIP_TOOLS_LIB', 'default').split('/')[-1]:40s} {e0.elapsed_time(e1) * 1e3 / 200:7.2f} us per launch   checksum {int(buf.view(torch.int32).long().sum())}")

#!/usr/bin/env python
"""Price of one all-gather dependency edge inside a persistent launch on MI355X: 256 co-resident workgroups publish an
n-vector as 8-byte {value, tag} granules and every workgroup sweeps it (m5_debug_edge_probe), idle and under a weight stream.
Compare with a dependent launch: ~1.2-1.5 us boundary + ~2.5 us fill / drain per decode GEMV (tools/launch_floor.py)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("M5_HIP_TOOLS", "1")
import torch
import mars5_tts_amd as pkg            # noqa
from mars5_tts_amd import ops, _lib as L

dev = torch.device("cuda:0")
stream = torch.cuda.Stream()
st = stream.cuda_stream
wbuf = torch.zeros(4096 * 64 * 1024, dtype=torch.uint8, device=dev)       # 256 MiB: 4096 chunks of up to 64 KiB
base = 0
for (n, per, threads) in ((1536, 6, 256), (1536, 6, 512), (3584, 14, 512), (4608, 18, 512), (192 * 66, 66, 256)):
    blocks = 256 if n != 192 * 66 else 192
    for skb in (0, 16, 32, 64):
        if skb * 1024 > threads * 128:
            continue
        gran = torch.zeros(2 * n, dtype=torch.int64, device=dev)
        err = torch.zeros(4, dtype=torch.int32, device=dev)
        sums = torch.zeros(blocks, dtype=torch.float32, device=dev)
        res = []
        for iters in (100, 1100):
            with torch.cuda.stream(stream):
                for rep in range(2):                      # first pass warms, second is timed
                    e0, e1 = ops.Event(), ops.Event()
                    e0.record(st)
                    L.check(L.lib.m5_debug_edge_probe(gran.data_ptr(), n, per, blocks, threads, iters, base, wbuf.data_ptr(), skb,
                                                      err.data_ptr(), sums.data_ptr(), st))
                    e1.record(st)
                    stream.synchronize()
                    base += iters + 8
            res.append(e0.elapsed_ms(e1) * 1e3)
        per_edge = (res[1] - res[0]) / 1000.0
        exp = sum(float((1099 + i) & 1023) for i in range(n))      # last iteration's vector sum is part of the total; just report
        print(f"n {n:5d} per {per:3d} blocks {blocks} x {threads:3d} stream {skb:2d} KiB/WG: {per_edge:6.2f} us/edge "
              f"(gave up: {int(err[0])}, sums equal across WGs: {bool((sums == sums[0]).all())})", flush=True)

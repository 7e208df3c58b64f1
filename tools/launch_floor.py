#!/usr/bin/env python
"""Per-launch floor of a dependent kernel chain on MI355X: eager (one C call enqueues n launches) vs
hipGraph replay, trivial kernels with and without a global read-modify-write."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("M5_HIP_TOOLS", "1")      # tools run on libmars5_hip_tools.so (knobs, probes; csrc/common.h)
import torch
import mars5_tts_amd as pkg            # noqa
from mars5_tts_amd import ops, _lib as L

dev = torch.device("cuda:0")
buf = torch.zeros(16, dtype=torch.int32, device=dev)
stream = torch.cuda.Stream()
st = stream.cuda_stream
N = 200
for blocks, threads, touch in ((1, 64, 0), (1, 64, 1), (256, 256, 0), (256, 256, 1), (1024, 256, 0)):
    with torch.cuda.stream(stream):
        L.check(L.lib.m5_debug_launch_chain(buf.data_ptr(), N, blocks, threads, touch, st))
        stream.synchronize()
        e0, e1 = ops.Event(), ops.Event()
        e0.record(st)
        L.check(L.lib.m5_debug_launch_chain(buf.data_ptr(), N, blocks, threads, touch, st))
        e1.record(st)
        stream.synchronize()
        eager = e0.elapsed_ms(e1) * 1e3 / N
        ops.Graph.begin(st)
        L.check(L.lib.m5_debug_launch_chain(buf.data_ptr(), N, blocks, threads, touch, st))
        g = ops.Graph().end(st)
        g.launch(st); stream.synchronize()
        e0, e1 = ops.Event(), ops.Event()
        e0.record(st)
        for _ in range(5):
            g.launch(st)
        e1.record(st)
        stream.synchronize()
        graph = e0.elapsed_ms(e1) * 1e3 / (5 * N)
    print(f"blocks {blocks:5d} x {threads:3d}  touch={touch}:  eager {eager:6.2f} us/launch   hipGraph {graph:6.2f} us/launch", flush=True)

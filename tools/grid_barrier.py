#!/usr/bin/env python
"""Cost of a device-wide barrier inside one launch on MI355X (agent-scope atomic arrive + acquire spin), for
workgroup counts around the CU count; compare with the 1.6 us dependent-launch floor (tools/launch_floor.py)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("M5_HIP_TOOLS", "1")      # tools run on libmars5_hip_tools.so (knobs, probes; csrc/common.h)
import torch
import mars5_tts_amd as pkg            # noqa
from mars5_tts_amd import ops, _lib as L

dev = torch.device("cuda:0")
stream = torch.cuda.Stream()
st = stream.cuda_stream
scratch = torch.zeros(2048, dtype=torch.int32, device=dev)
for blocks, threads in ((64, 256), (128, 256), (256, 64), (256, 256), (256, 512), (512, 256)):
    for mode in (0, 1):
        res = []
        for iters in (50, 550):
            with torch.cuda.stream(stream):
                L.check(L.lib.m5_debug_grid_barrier(scratch.data_ptr(), blocks, threads, iters, mode, st))
                stream.synchronize()
                e0, e1 = ops.Event(), ops.Event()
                e0.record(st)
                L.check(L.lib.m5_debug_grid_barrier(scratch.data_ptr(), blocks, threads, iters, mode, st))
                e1.record(st)
                stream.synchronize()
                res.append(e0.elapsed_ms(e1) * 1e3)
        errs = scratch[1:3].tolist()
        per = (res[1] - res[0]) / 500.0
        print(f"blocks {blocks:4d} x {threads:3d} mode {mode}: {per:6.2f} us/barrier  (timeouts {errs[0]}, stale reads {errs[1]})", flush=True)

#!/usr/bin/env python
"""Does an XCD's L2 (and the Infinity Cache) keep lines across a kernel boundary?  (m5_debug_l2_touch)
Pairs of launches on one stream, the second timed by HIP events inside a hipGraph of REP pairs:
  same      warm(shift 0) -> timed(shift 0): the chunk sits where the SAME workgroup id (same XCD) read it
  neighbour warm(shift 1) -> timed(shift 0): the chunk was read by the neighbouring XCD
  far       warm(shift 4) -> timed(shift 0)
  cold      warm over ANOTHER 512 MB buffer -> timed(shift 0): from HBM
for buffers of 8 / 16 / 64 MB (1 / 2 / 8 MB per XCD; L2 = 4 MB per XCD, Infinity Cache 256 MB)."""
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
sink = torch.zeros(4, device=dev)
big = torch.randint(0, 255, (512 << 20,), dtype=torch.uint8, device=dev)
REP = 20


def touch(buf, chunk, blocks, shift, nt):
    L.check(L.lib.m5_debug_l2_touch(buf.data_ptr(), chunk, blocks, 256, shift, nt, sink.data_ptr(), st))


def pair_time(buf, chunk, blocks, warm_shift, nt_warm, nt_timed, cold=False):
    with torch.cuda.stream(stream):
        def body():
            if cold:
                touch(big, (512 << 20) // 256, 256, 0, 0)
            else:
                touch(buf, chunk, blocks, warm_shift, nt_warm)
            touch(buf, chunk, blocks, 0, nt_timed)
        # time warm+timed pairs and warm-only launches under a graph; the difference is the timed launch
        def run(fn):
            ops.Graph.begin(st)
            for _ in range(REP):
                fn()
            g = ops.Graph().end(st)
            g.launch(st); stream.synchronize()
            e0, e1 = ops.Event(), ops.Event()
            e0.record(st); g.launch(st); e1.record(st); stream.synchronize()
            return e0.elapsed_ms(e1) * 1e3 / REP
        both = run(body)
        only = run((lambda: touch(big, (512 << 20) // 256, 256, 0, 0)) if cold else (lambda: touch(buf, chunk, blocks, warm_shift, nt_warm)))
    return both - only, only


for mb in (8, 16, 64):
    buf = torch.randint(0, 255, (mb << 20,), dtype=torch.uint8, device=dev)
    blocks = 256
    chunk = (mb << 20) // blocks
    for nt_timed in (0, 1):
        row = []
        for name, sh, cold in (("same", 0, False), ("neighbour", 1, False), ("far", 4, False), ("cold", 0, True)):
            t, w = pair_time(buf, chunk, blocks, sh, 0, nt_timed, cold)
            row.append(f"{name} {t:6.2f} us")
        print(f"{mb:3d} MB ({mb / 8:.0f} MB per XCD), timed loads {'nt' if nt_timed else 'plain'}: " + "   ".join(row), flush=True)

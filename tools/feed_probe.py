#!/usr/bin/env python
"""Operand feed rate per CU from L2-resident panels into LDS: LDS-DMA vs register staging vs loads only."""
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
sink = torch.zeros(4, device=dev)
ROWB = [int(x) for x in os.environ.get("ROWB", "2048").split(",")]
for panel_kb, row_bytes in [(pk, rb) for pk in (512,) for rb in ROWB]:
    src = torch.randint(0, 255, (8 * panel_kb * 1024 + 65536,), dtype=torch.uint8, device=dev)
    for (blocks, threads) in ((256, 256), (512, 256), (768, 256), (256, 1024)):
        for mode, name in ((0, "lds-dma"),):
            iters = 200
            with torch.cuda.stream(stream):
                L.check(L.lib.m5_debug_feed_probe(src.data_ptr(), panel_kb * 1024, 5, row_bytes, mode, blocks, threads, sink.data_ptr(), st))
                stream.synchronize()
                e0, e1 = ops.Event(), ops.Event()
                e0.record(st)
                L.check(L.lib.m5_debug_feed_probe(src.data_ptr(), panel_kb * 1024, iters, row_bytes, mode, blocks, threads, sink.data_ptr(), st))
                e1.record(st)
                stream.synchronize()
            ms = e0.elapsed_ms(e1)
            total = blocks * iters * panel_kb * 1024
            print(f"rowB {row_bytes:5d} panel {panel_kb:5d} KiB/XCD  grid {blocks:4d} x {threads:3d}  {name:14s}: {total / ms / 1e9:8.2f} TB/s chip  "
                  f"{total / ms / 1e6 / 256:7.1f} GB/s per CU", flush=True)

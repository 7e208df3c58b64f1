#!/usr/bin/env python
"""Where does the dispatcher place the workgroups of a grid?  (m5_debug_census)"""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("M5_HIP_TOOLS", "1")      # tools run on libmars5_hip_tools.so (knobs, probes; csrc/common.h)
import torch
import mars5_tts_amd as pkg            # noqa
from mars5_tts_amd import _lib as L

def census(nblocks, threads, lds, spin=200000):
    out = torch.zeros(nblocks, 6, dtype=torch.int32, device="cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    L.check(L.lib.m5_debug_census(out.data_ptr(), nblocks, threads, lds, spin, st))
    torch.cuda.synchronize()
    o = out.cpu().numpy().astype("uint32")
    xcc = o[:, 0] & 0xf
    hw = o[:, 1]
    cu = (hw >> 8) & 0xf
    sh = (hw >> 12) & 0x1
    se = (hw >> 13) & 0x7
    keys = collections.Counter(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist()))
    per = collections.Counter(keys.values())
    t0 = o[:, 2].astype("uint64") | (o[:, 3].astype("uint64") << 32)
    late = int(((t0 - t0.min()) > spin // 2).sum())
    print(f"grid {nblocks:5d} x {threads:4d} thr, LDS {lds:6d} B: distinct CUs {len(keys):3d}, WGs-per-CU histogram {dict(sorted(per.items()))}, "
          f"XCC histogram {dict(sorted(collections.Counter(xcc.tolist()).items()))}, started late (2nd round) {late}")
    xmap = [int(x) for x in xcc[:16]]
    print("      xcc of blocks 0..15:", xmap)

if __name__ == "__main__":
    census(352, 256, 48 * 1024)
    census(352, 256, 96 * 1024)
    census(176, 256, 64 * 1024)
    census(240, 512, 144 * 1024)
    census(528, 256, 64 * 1024)
    census(256, 192, 8 * 1024)
    census(704, 128, 48 * 1024)

#!/usr/bin/env python
"""GPU idle time of one bench utterance from a rocprofv3 kernel trace (CSV): the largest gaps between consecutive kernels (end of the
running maximum -> next start) with the kernels on either side, and busy / idle totals per phase (before the first ar_mega_kernel,
decode, between decode and the first nar step, NAR loop, after).  usage: python tools/trace_gaps.py <kernel_trace.csv> [utterance index]"""
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# utterances = runs separated by the largest gaps are not needed: take the LAST utterance = from the last prefill's first kernel;
# find indices of ar_mega kernels, split into contiguous decode runs
mega = [i for i, r in enumerate(rows) if "ar_mega_kernel" in r[2]]
runs, cur = [], [mega[0]]
for a, b in zip(mega, mega[1:]):
    if rows[b][0] - rows[a][1] > 5_000_000:      # > 5 ms without a persistent launch: next utterance
        runs.append(cur); cur = []
    cur.append(b)
runs.append(cur)
which = int(sys.argv[2]) if len(sys.argv) > 2 else len(runs) - 1
run = runs[which]
prev_end = rows[runs[which - 1][-1]][1] if which > 0 else rows[0][0]
# start of this utterance: first kernel after the previous utterance's NAR loop = search backwards from run[0] for a gap > 1 ms ... simpler: window = (end of previous utterance's last nar_sample, this utterance's last nar_sample)
samp = [i for i, r in enumerate(rows) if "nar_sample_kernel" in r[2]]
last_samp = max(i for i in samp if i > run[-1] and (which + 1 >= len(runs) or i < runs[which + 1][0]))
prev_samp = max([i for i in samp if i < run[0]], default=-1)
i0, i1 = prev_samp + 1, last_samp
seg = rows[i0:i1 + 1]
first_samp = min(i for i in samp if i > run[-1])
marks = {"pre-decode": (i0, run[0]), "decode": (run[0], run[-1] + 3), "decode -> first NAR sample": (run[-1] + 3, first_samp), "NAR loop (after step 1)": (first_samp, i1)}
print(f"utterance {which}: {len(seg)} kernels, wall {(seg[-1][1] - seg[0][0]) / 1e6:.2f} ms from first kernel start to last kernel end")
for name, (a, b) in marks.items():
    s = rows[a:b + 1]
    if not s:
        continue
    busy, end = 0, s[0][0]
    for st, en, _ in s:
        busy += max(0, en - max(st, end))
        end = max(end, en)
    wall = end - s[0][0]
    print(f"  {name:28s} {len(s):6d} kernels  wall {wall / 1e6:8.3f} ms  busy {busy / 1e6:8.3f} ms  idle {(wall - busy) / 1e6:7.3f} ms")
gaps, end, endk = [], seg[0][1], seg[0][2]
for st, en, k in seg[1:]:
    if st > end:
        gaps.append((st - end, endk, k, (st - seg[0][0]) / 1e6))
    if en > end:
        end, endk = en, k
gaps.sort(reverse=True)
print("largest idle gaps (us, at ms from start, after kernel -> before kernel):")
for g, a, b, t in gaps[:25]:
    print(f"  {g / 1e3:9.1f} us at {t:8.2f} ms   {a[:60]:60s} -> {b[:60]}")

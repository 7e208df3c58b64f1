#!/usr/bin/env python
"""Summarise rocprofv3 --pmc counter_collection CSVs: mean counter value per kernel name/grid."""
import csv, glob, sys, collections
csv.field_size_limit(10**9)
root = sys.argv[1]
for f in sorted(glob.glob(root + "/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"][:60], r.get("Grid_Size", ""))
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("==", f)
    for k, d in acc.items():
        if not any(t in k[0] for t in ("gemm16", "attn", "skinny", "gemv", "sample_kernel", "ar_mega")):
            continue
        print(k[0], "grid", k[1], "n", len(next(iter(d.values()))))
        for c, v in d.items():
            print("     %-28s %14.0f" % (c, sum(v) / len(v)))

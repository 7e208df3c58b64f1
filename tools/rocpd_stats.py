#!/usr/bin/env python
"""Per-kernel statistics from a rocprofv3 rocpd database (the default output format of this ROCm: `*_results.db`), for runs that were
not made with `--output-format csv`: the columns of rocprofv3's `*_kernel_stats.csv`, plus -- with --resources -- what each kernel was
launched with (grid, workgroup, VGPRs, AGPRs, SGPRs, LDS, scratch).
usage: python tools/rocpd_stats.py run_results.db [--resources] > kernel_stats.csv"""
import csv
import sqlite3
import statistics
import sys


def main() -> None:
    db = sqlite3.connect(sys.argv[1])
    res = "--resources" in sys.argv[2:]
    rows = {}
    for r in db.execute("select name, end - start, grid_x, grid_y, grid_z, workgroup_x, workgroup_y, workgroup_z, vgpr_count, accum_vgpr_count, "
                        "sgpr_count, lds_size, scratch_size from kernels"):
        d = rows.setdefault(r[0], {"t": [], "launch": set()})
        d["t"].append(r[1])
        d["launch"].add(r[2:])
    total = sum(sum(d["t"]) for d in rows.values()) or 1
    w = csv.writer(sys.stdout, quoting=csv.QUOTE_NONNUMERIC)
    head = ["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"]
    w.writerow(head + (["Grids", "Workgroup", "VGPR", "AGPR", "SGPR", "LDS", "Scratch"] if res else []))
    for name, d in sorted(rows.items(), key=lambda kv: -sum(kv[1]["t"])):
        t = d["t"]
        row = [name, len(t), sum(t), round(sum(t) / len(t), 6), round(100.0 * sum(t) / total, 4), min(t), max(t), round(statistics.pstdev(t), 6)]
        if res:
            ls = sorted(d["launch"])
            grids = " ".join(sorted({f"{a[0]}x{a[1]}x{a[2]}" for a in ls}))
            first = ls[0]
            row += [grids, f"{first[3]}x{first[4]}x{first[5]}", first[6], first[7], first[8], max(a[9] for a in ls), max(a[10] for a in ls)]
        w.writerow(row)


if __name__ == "__main__":
    main()

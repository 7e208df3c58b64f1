#!/usr/bin/env python
"""Map the address of a "Memory access fault by GPU ... on address 0x..." message to the allocation of a guard_run.py log.

    python tools/guard_report.py <log dir> <stderr file or 0xADDRESS>

Says whether the address lies past the end / before the start of an allocation (over-run / under-run: which one, its size,
how far), inside a freed one (stale pointer), or nowhere near anything the allocator handed out; with --trace's launch.log
present, also the last launch that was started."""
import os
import re
import sys


def main():
    d, what = sys.argv[1], sys.argv[2]
    if what.startswith("0x"):
        addrs = [int(what, 16)]
    else:
        addrs = [int(m, 16) for m in re.findall(r"on address (0x[0-9a-fA-F]+)", open(what, errors="replace").read())]
    if not addrs:
        print("no fault address found")
        return
    allocs, freed, guard = {}, set(), 2 << 20
    for ln in open(os.path.join(d, "alloc.log")):
        p = ln.split()
        if not p:
            continue
        if p[0] == "I" and "guard=" in ln:
            guard = int(re.search(r"guard=(\d+)", ln).group(1))
        elif p[0] == "A":
            allocs[int(p[1])] = dict(serial=int(p[1]), user=int(p[2], 16), size=int(p[3]), mapped=int(p[4], 16), mapped_bytes=int(p[5]))
        elif p[0] == "F":
            freed.add(int(p[1]))
    print(f"{len(allocs)} allocations logged, {len(freed)} freed, guard {guard} bytes")
    for addr in addrs:
        print(f"fault address {addr:#x}:")
        hit = False
        for r in allocs.values():
            lo, hi = r["mapped"], r["mapped"] + r["mapped_bytes"]
            state = "FREED (stale pointer)" if r["serial"] in freed else "live"
            if lo <= addr < hi:
                print(f"  inside the mapping of allocation #{r['serial']} ({r['size']} bytes, {state}): offset {addr - r['user']} from the tensor's start")
                hit = True
            elif hi <= addr < hi + guard:
                print(f"  {addr - (r['user'] + r['size'])} bytes PAST THE END of allocation #{r['serial']} ({r['size']} bytes, {state}) -> over-run")
                hit = True
            elif lo - guard <= addr < lo:
                print(f"  {r['user'] - addr} bytes BEFORE THE START of allocation #{r['serial']} ({r['size']} bytes, {state}) -> under-run")
                hit = True
        if not hit:
            print("  not in or next to any logged allocation (a wild pointer, or memory the allocator did not hand out)")
    lp = os.path.join(d, "launch.log")
    if os.path.exists(lp):
        lines = open(lp).read().split("\n")
        print("last launches started:", [l for l in lines if l][-4:])


if __name__ == "__main__":
    main()

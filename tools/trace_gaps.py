#!/usr/bin/env python3
"""Reads a rocprofv3 kernel trace (csv) and prints, for the LAST occurrence of a marker kernel sequence, the busy
time per kernel name and the idle time between kernels (development aid).
    python tools/trace_gaps.py <kernel_trace.csv> [t0_fraction]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# second half of the run = the second (warm) build
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
t_lo = int(rows[0]["Start_Timestamp"]); t_hi = int(rows[-1]["End_Timestamp"])
# find the largest idle gap (between the two builds: host-side dataset work) and take what follows it
gaps = [(int(rows[i + 1]["Start_Timestamp"]) - int(rows[i]["End_Timestamp"]), i) for i in range(len(rows) - 1)]
big = max(gaps)[1] if frac == 0.5 else int(len(rows) * frac)
sel = rows[big + 1:]
busy = collections.Counter(); cnt = collections.Counter()
idle = 0; last_end = None; gaplist = []
for r in sel:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "")[:70]
    busy[name] += e - s; cnt[name] += 1
    if last_end is not None and s > last_end:
        idle += s - last_end; gaplist.append((s - last_end, name))
    last_end = max(last_end or e, e)
span = int(sel[-1]["End_Timestamp"]) - int(sel[0]["Start_Timestamp"])
print(f"span {span/1e6:.2f} ms, kernels {len(sel)}, busy {sum(busy.values())/1e6:.2f} ms, idle {idle/1e6:.2f} ms")
for n, t in busy.most_common(25):
    print(f"  {t/1e6:9.3f} ms  x{cnt[n]:<5d} {n}")
print("largest gaps (ms, before kernel):")
for g, n in sorted(gaplist, reverse=True)[:12]:
    print(f"  {g/1e6:8.3f}  {n}")

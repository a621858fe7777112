#!/usr/bin/env python3
"""Per-tick GPU timeline from a rocprofv3 --kernel-trace CSV: kernel time, idle gaps between kernels, tick period.
Usage: tools/trace_gaps.py <kernel_trace.csv> [first_kernel_substring]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
key = sys.argv[2] if len(sys.argv) > 2 else "rg_front_kernel"
rows = [r for r in rows if r["Kernel_Name"].startswith(("rg_", "void rg_"))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ticks, cur = [], []
for r in rows:
    if key in r["Kernel_Name"] and cur:
        ticks.append(cur); cur = []
    cur.append(r)
ticks = ticks[8:]   # skip warm-up
per, busy, gaps = [], [], []
for a, b in zip(ticks[:-1], ticks[1:]):
    t0, t1 = int(a[0]["Start_Timestamp"]), int(b[0]["Start_Timestamp"])
    per.append((t1 - t0) / 1e3)
    busy.append(sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in a) / 1e3)
per.sort(); n = len(per)
print(f"ticks {n}: period us median {per[n//2]:.1f} mean {sum(per)/n:.1f} p10 {per[n//10]:.1f} p90 {per[(9*n)//10]:.1f}; kernel-busy us mean {sum(busy)/n:.1f}; idle per tick mean {(sum(per)-sum(busy))/n:.1f}")
names = {}
for t in ticks:
    for r in t:
        nm = r["Kernel_Name"].split("(")[0][-40:]
        names.setdefault(nm, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for nm, v in names.items():
    v.sort(); print(f"  {nm:42s} n {len(v):4d} mean {sum(v)/len(v):8.1f} median {v[len(v)//2]:8.1f} max {v[-1]:8.1f}")

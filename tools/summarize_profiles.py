#!/usr/bin/env python3
"""Summarise gpurun_out/prof_<tag>/ (tools/collect_profiles.sh) into profiles/<tag>_*: the rocprofv3
kernel stats, per-kernel PMC averages per launch, and the HBM traffic JSON bench.py reads.

HBM bytes per launch follow MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE come from
separate --pmc passes, both are in KiB (x1024); on gfx950 FETCH_SIZE under-reports WIDE (16 B/lane)
coalesced reads by 2x -- the kernels here read 4-8 B per lane, a width the guide calls uncalibrated,
so the raw value is reported and the 2x-corrected value is given as an upper bound."""
import collections, csv, glob, json, os, shutil, sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tag = sys.argv[1] if len(sys.argv) > 1 else "r5"       # r4 (headline) or r4_<workload key>: bench.py profile_tag()
workload_key = tag.split("_", 1)[1] if "_" in tag else "headline"
src = os.path.join("gpurun_out", f"prof_{tag}")
dst = "profiles"
os.makedirs(dst, exist_ok=True)


def short(name):
    for k in ("rg_front_kernel", "rg_qp_resolve_kernel", "rg_qp_fused_kernel", "rg_qp_sched_retry_kernel", "rg_qp_sched_kernel", "rg_swing_ik_kernel", "rg_reset_kernel", "rg_hybrid"):
        if k in name:
            return name[name.index(k):].split("(")[0]
    return None


def newest(pattern):
    """gpurun merges every call's output into gpurun_out/, so a pass directory can hold several runs: take the latest."""
    found = sorted(glob.glob(pattern), key=os.path.getmtime)
    return found[-1:]


stats = newest(os.path.join(src, "trace", "*", "*kernel_stats.csv"))
if stats:
    shutil.copy(stats[0], os.path.join(dst, f"{tag}_kernel_stats.csv"))
pmc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("pmc_fetch", "pmc_write", "pmc_sq1", "pmc_sq2", "pmc_flops"):
    for f in newest(os.path.join(src, d, "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k is None:
                continue
            pmc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = []
traffic = {}
for k in sorted(pmc):
    c = {n: sum(v) / len(v) for n, v in pmc[k].items()}
    rows.append((k, c))
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        traffic[k] = {"fetch_bytes_raw": c["FETCH_SIZE"] * 1024, "write_bytes": c["WRITE_SIZE"] * 1024,
                      "hbm_bytes_per_launch": (c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024,
                      "hbm_bytes_per_launch_if_fetch_x2": (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024}
with open(os.path.join(dst, f"{tag}_pmc_per_launch.csv"), "w") as f:
    names = sorted({n for _, c in rows for n in c})
    f.write("kernel," + ",".join(names) + "\n")
    for k, c in rows:
        f.write(k.replace(",", ";") + "," + ",".join("%.6g" % c.get(n, float("nan")) for n in names) + "\n")
bl = os.path.join(src, "bench_line.json")
meta = json.loads(open(bl).read()) if os.path.exists(bl) and os.path.getsize(bl) else {}
import bench
cmd_file = os.path.join(src, "command.txt")
stamp = os.path.join(src, "evidence_header.txt")   # written on the GPU box by collect_profiles.sh (tools/evidence_guard.py)
header = open(stamp).read().strip() if os.path.exists(stamp) else bench.evidence_header()
json.dump({"tag": tag, "workload_key": workload_key, "source_hash": header.split("kernel sources ")[1].split()[0],
           "commit": header.split("commit ")[1].split()[0], "evidence_header": header,
           "batch": (meta.get("config", {}).get("workload", "batch=4096").split("batch=")[1].split(" ")[0]),
           "command": (open(cmd_file).read().strip() if os.path.exists(cmd_file) else "python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras") + " (under rocprofv3)", "traffic": traffic,
           "bench_line_under_profiler": meta}, open(os.path.join(dst, f"{tag}_traffic.json"), "w"), indent=1)
print(open(os.path.join(dst, f"{tag}_pmc_per_launch.csv")).read())
print(json.dumps(traffic, indent=1))

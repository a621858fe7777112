#!/usr/bin/env python3
"""Compiler-reported resources of every kernel in librg_mpc.so (registers, scratch, occupancy):
`hipcc -Rpass-analysis=kernel-resource-usage` on rg_mpc.hip, one line per kernel.  Usage: tools/resource_usage.py [out.txt]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "robot_gym_amd", "csrc")
res = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-c", "-o", "/dev/null", "rg_mpc.hip",
                      "-Rpass-analysis=kernel-resource-usage"], cwd=src, capture_output=True, text=True)
rows, cur = [], {}
for l in res.stderr.splitlines():
    m = re.search(r"remark: (?:\S+:\d+:\d+:\s+)?(.*?) \[-Rpass", l)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith(("Function Name", "Name:")):
        if cur:
            rows.append(cur)
        cur = {"name": t.split(":", 1)[1].strip()}
    elif ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
if cur:
    rows.append(cur)
def demangle(n):
    m = re.match(r"_Z\d+(rg_[a-z_]+kernel)(I.*?E)?v?P", n)
    if not m:
        return n
    t = re.findall(r"L[ib](\d+)E", m.group(2) or "")
    return m.group(1) + ("<" + ",".join(t) + ">" if t else "")
lines = [f"{demangle(r['name']):44s} VGPR {r.get('VGPRs'):>3s} AGPR {r.get('AGPRs'):>3s} SGPR {r.get('TotalSGPRs'):>3s} scratch B/lane {r.get('ScratchSize [bytes/lane]'):>4s} waves/SIMD {r.get('Occupancy [waves/SIMD]')} LDS {r.get('LDS Size [bytes/block]')}" for r in rows]
out = "\n".join(lines)
print(out)
if len(sys.argv) > 1:
    sys.path.insert(0, ROOT)
    import bench
    open(sys.argv[1], "w").write(f"# hipcc -Rpass-analysis=kernel-resource-usage (tools/resource_usage.py), {bench.evidence_header()}\n" + out + "\n")

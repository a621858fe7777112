#!/usr/bin/env python3
"""What the compiler made of every kernel of librg_mpc.so, from ONE device-only compile of rg_mpc.hip (no GPU needed):
registers / scratch / occupancy (`-Rpass-analysis=kernel-resource-usage`) and, from the ISA of the same compile, the scratch
(spill) operations that sit INSIDE a solver loop -- an innermost loop with at least MIN_FMA f64 FMAs: a sweep's pivot loop,
an ADMM iteration, a row loop of the active-set bodies.  One-time spill frames around the bodies cost nothing that can be
measured; a reload per iteration does.  tests/test_kernel_resources.py holds every kernel to the committed table
(profiles/r6_resource_usage.txt), so that a compiler bump or an innocent edit that tips a loop into scratch fails the CPU
suite instead of waiting for a bench.
Usage: tools/kernel_report.py [out.txt] [--keep-isa file.s]"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "robot_gym_amd", "csrc")
MIN_FMA = 12


def compile_device(isa_path):
    """hipcc -S --cuda-device-only with resource remarks -> stderr text (remarks); the ISA goes to isa_path."""
    res = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-S", "--cuda-device-only", "-o", isa_path,
                          "rg_mpc.hip", "-Rpass-analysis=kernel-resource-usage"], cwd=SRC, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + res.stderr[-4000:])
    return res.stderr


def demangle(n):
    m = re.match(r"_Z\d+(rg_[a-z_0-9]+kernel)(I.*?E)?v?P", n)
    if not m:
        return n
    t = re.findall(r"L[ib](\d+)E", m.group(2) or "")
    return m.group(1) + ("<" + ",".join(t) + ">" if t else "")


def parse_remarks(text):
    rows, cur = [], {}
    for l in text.splitlines():
        m = re.search(r"remark: (?:\S+:\d+:\d+:\s+)?(.*?) \[-Rpass", l)
        if not m:
            continue
        t = m.group(1).strip()
        if t.startswith(("Function Name", "Name:")):
            if cur:
                rows.append(cur)
            cur = {"mangled": t.split(":", 1)[1].strip()}
        elif ":" in t:
            k, v = t.split(":", 1)
            cur[k.strip()] = v.strip()
    if cur:
        rows.append(cur)
    return rows


def loop_scratch(isa_text):
    """{mangled kernel name: (scratch ops inside solver loops, number of solver loops)}"""
    lines = isa_text.split("\n")
    out = {}
    starts = [i for i, l in enumerate(lines) if re.match(r"^(_Z\S+|rg_\w+):\s*(;.*)?$", l) and not l.startswith(".")]
    for st in starts:
        try:
            end = next(i for i in range(st + 1, len(lines)) if lines[i].startswith(".Lfunc_end"))
        except StopIteration:
            continue
        body = lines[st:end]
        labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
        loops = []
        for i, l in enumerate(body):
            m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
            if m and m.group(1) in labels and labels[m.group(1)] < i:
                loops.append((labels[m.group(1)], i))
        ops = nloops = 0
        for a, b in loops:
            if any((a2, b2) != (a, b) and a2 >= a and b2 <= b for a2, b2 in loops):
                continue   # not innermost
            if sum(1 for x in body[a:b] if re.search(r"v_fma_f64|v_fmac_f64", x)) < MIN_FMA:
                continue
            nloops += 1
            ops += sum(1 for x in body[a:b] if "scratch_" in x)
        out[lines[st].split(":")[0]] = (ops, nloops)
    return out


def report(keep_isa=None):
    """[{name, vgpr, agpr, sgpr, scratch, waves, loop_scratch, loops}] for every kernel, in the order the compiler reports them."""
    tmp = keep_isa or tempfile.mktemp(suffix=".s", prefix="rg_mpc_")
    try:
        rows = parse_remarks(compile_device(tmp))
        ls = loop_scratch(open(tmp).read())
    finally:
        if not keep_isa and os.path.exists(tmp):
            os.remove(tmp)
    out = []
    for r in rows:
        ops, nl = ls.get(r["mangled"], (0, 0))
        out.append({"name": demangle(r["mangled"]), "vgpr": int(r.get("VGPRs", -1)), "agpr": int(r.get("AGPRs", -1)), "sgpr": int(r.get("TotalSGPRs", -1)),
                    "scratch": int(r.get("ScratchSize [bytes/lane]", -1)), "waves": int(r.get("Occupancy [waves/SIMD]", -1)), "loop_scratch": ops, "loops": nl})
    return out


def format_rows(rows):
    return "\n".join(f"{r['name']:44s} VGPR {r['vgpr']:3d} AGPR {r['agpr']:3d} SGPR {r['sgpr']:3d} scratch B/lane {r['scratch']:4d} waves/SIMD {r['waves']} "
                     f"scratch ops in solver loops {r['loop_scratch']:2d} (of {r['loops']} loops)" for r in rows)


def parse_table(text):
    """The committed table back into {name: row}."""
    out = {}
    for l in text.splitlines():
        m = re.match(r"(\S+)\s+VGPR\s+(\d+) AGPR\s+(\d+) SGPR\s+(\d+) scratch B/lane\s+(\d+) waves/SIMD (\d+) scratch ops in solver loops\s+(\d+)", l)
        if m:
            out[m.group(1)] = {"vgpr": int(m.group(2)), "agpr": int(m.group(3)), "sgpr": int(m.group(4)), "scratch": int(m.group(5)), "waves": int(m.group(6)),
                               "loop_scratch": int(m.group(7))}
    return out


if __name__ == "__main__":
    args = [a for a in sys.argv[1:]]
    keep = None
    if "--keep-isa" in args:
        k = args.index("--keep-isa"); keep = args[k + 1]; del args[k:k + 2]
    rows = report(keep)
    txt = format_rows(rows)
    print(txt)
    if args:
        sys.path.insert(0, ROOT)
        import bench
        open(args[0], "w").write(f"# hipcc -Rpass-analysis=kernel-resource-usage + ISA loop scan (tools/kernel_report.py), {bench.evidence_header()}\n" + txt + "\n")

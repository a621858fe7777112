#!/usr/bin/env python3
"""Headline benchmark: MPC controller steps/sec (whole node), batch=4096 quadrupeds per GPU,
horizon=10 (BASELINE.json).  One "step" = one rg_mpc_step over one batch of synthetic robot
states already resident in HBM.

  python bench.py --gpus N --steps K --warmup W        (N > 1 without RANK in the environment: starts its own ranks, see self_launch)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W
  python bench.py --gpus 2 --dry-launch                 (CPU: the whole multi-rank protocol over gloo with a stub controller)

Rank 0 prints ONE JSON line.  Extra objects: "roofline" (dominant kernel, hipEvent-timed over
the timed region, algorithmic bytes from DESIGN.md section 5) and, at N=1, "cpu_baseline" (the float64
C oracle = a port of the algorithm, OpenMP over robots, on this box's host cores).

Inputs vary from tick to tick like they do under a live gym (reference gym/robot_gym_env.py:117-129 hands the
controller a new robot state every tick): a ring of RING pre-generated state slabs resident in HBM (velocity,
attitude, rates, foot positions perturbed smoothly per tick; measured contacts following the gait), one slab per
step, outside the timed kernels' critical path.  `--static-inputs` feeds one frozen slab (the round-1 bench); the
default line reports that variant too (config.static_inputs_steps_per_s).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

BATCH_PER_GPU = 4096
HORIZON = 10
EVENT_STRIDE = 4   # per-kernel HIP events are recorded on every 4th step of the timed region
RING = 50          # state slabs in the input ring = ticks of one 0.5 s trot cycle (measured contacts stay gait-consistent)
PROFILE_TAG = "r6"
# algorithmic HBM bytes per controller step (kin_mode 0, all optional outputs off): inputs 320, persistent
# controller state read + written ~550, the swing-IK hand-over (flags 32, target + start angles of ~1.5 swinging legs ~150),
# action row 240
ALGO_BYTES_PER_STEP = 1290
EXACT_WS_BYTES = 80         # exact body: the robot's stored working set (64 one-byte ids + count) read, the new one written
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
F64_VECTOR_PEAK_TFLOPS = 78.6   # 256 CUs x 4 SIMDs x 16 lanes x 2 flop x 2.4 GHz (v_fma_f64 issues at 4 cycles per wave)


def perturb_state(state, k, amp):
    """Smooth per-tick variation of the synthetic state (the bench's stand-in for physics): velocities, attitude,
    rates and foot positions move by a few per cent per tick, so consecutive QPs differ like under a live gym."""
    f32 = np.float32
    st = dict(state)
    st["v_world"] = (state["v_world"] * f32(1.0 + amp * np.sin(0.7 * k))).astype(f32)
    st["rpy_rate"] = (state["rpy_rate"] * f32(1.0 + amp * np.cos(0.45 * k))).astype(f32)
    rpy = state["rpy"].copy()
    rpy[0] += f32(0.2 * amp * np.sin(0.31 * k))
    rpy[1] += f32(0.2 * amp * np.cos(0.23 * k))
    st["rpy"] = rpy
    from robot_gym_amd import synthetic
    st["quat"] = synthetic._quat_from_rpy(rpy[0].astype(np.float64), rpy[1].astype(np.float64), rpy[2].astype(np.float64)).astype(f32)
    st["foot_pos"] = (state["foot_pos"] * f32(1.0 + 0.2 * amp * np.cos(0.3 * k))).astype(f32)
    return st


def chain_geometry(cfg, state):
    """foot_pos / jac of a synthetic batch replaced by the URDF chain's forward kinematics of its joint angles (host twin of
    the device code, controllers/mpc/kinematics.ChainKinematics): what kin_mode 1 computes on the device, handed to kin_mode 0
    as inputs -- both modes then solve the same QPs and their difference is the cost of the on-device kinematics."""
    from robot_gym_amd.controllers.mpc.kinematics import ChainKinematics
    ck = ChainKinematics(cfg)
    B = state["q"].shape[1]
    q = state["q"].astype(np.float64)
    foot, jac = np.zeros((4, 3, B)), np.zeros((4, 3, 3, B))
    for b in range(B):
        for leg in range(4):
            foot[leg, :, b], jac[leg, :, :, b] = ck.foot_position_and_jacobian(leg, q[3 * leg:3 * leg + 3, b])
    state = dict(state)
    state["foot_pos"], state["jac"] = foot.reshape(12, B).astype(np.float32), jac.reshape(36, B).astype(np.float32)
    return state


def make_input_ring(cfg, B, seed, device, ring, amp, fixed_cmd=None, gait=None, schedule=False, chain_geom=False):
    """`ring` input slabs on the device: slab j is the state handed to tick k = j (mod ring)."""
    from robot_gym_amd import synthetic
    state, cmd, t_off = synthetic.make_states(B, cfg, seed=seed, fixed_cmd=fixed_cmd)
    if chain_geom:
        state = chain_geometry(cfg, state)
    names = ("rpy", "rpy_rate", "v_world", "quat", "q", "foot_pos", "jac")
    slabs = []
    for j in range(ring):
        st = perturb_state(state, j, amp) if ring > 1 else state
        if chain_geom:
            st["foot_pos"] = state["foot_pos"]   # the feet follow the joint angles, which the ring does not vary
        dev = {n: torch.from_numpy(np.ascontiguousarray(st[n])).to(device) for n in names}
        dev["contact"] = torch.from_numpy(synthetic.gait_consistent_contacts(cfg, t_off + 0.01 * j, state["_flip"], gait)).to(device)
        if schedule:   # BASELINE config 5: randomised per-step contact schedule, re-drawn every tick
            dev["contact_sched"] = torch.from_numpy(synthetic.contact_schedule(cfg, t_off + 0.01 * j, gait, dropout=0.1, seed=seed, tick=j)).to(device)
        slabs.append(dev)
    return state, cmd, t_off, slabs


def compiled_sources(read=None):
    """Repository-relative paths of the files librg_mpc.so is compiled from: rg_mpc.hip and everything it includes from the
    repository, transitively (system / HIP headers excluded).  A stray .inc / .h left in csrc/ by an experiment is not part
    of the library and must not change the hash that ties committed evidence to kernel sources.
    read(relative path) -> bytes or None: where the file contents come from (default: the working tree)."""
    import re
    if read is None:
        def read(rel):
            try:
                return open(os.path.join(ROOT, rel), "rb").read()
            except OSError:
                return None
    seen, todo = {}, ["robot_gym_amd/csrc/rg_mpc.hip"]
    while todo:
        rel = os.path.normpath(todo.pop())
        if rel in seen:
            continue
        data = read(rel)
        if data is None:
            continue
        seen[rel] = data
        for inc in re.findall(r'^\s*#\s*include\s+"([^"]+)"', data.decode("utf-8", "replace"), flags=re.M):
            todo.append(os.path.join(os.path.dirname(rel), inc))
    return dict(sorted(seen.items()))


def strip_c_comments(data):
    """C / C++ source bytes with every comment removed and every run of white space reduced to one blank (string and
    character literals are left alone): what the compiler sees, as far as a hash needs to know."""
    text = data.decode("utf-8", "replace")
    out, i, n = [], 0, len(text)
    while i < n:
        ch = text[i]
        if ch == '"' or ch == "'":
            j = i + 1
            while j < n and text[j] != ch:
                j += 2 if text[j] == "\\" else 1
            out.append(text[i:j + 1]); i = j + 1
        elif text.startswith("//", i):
            j = text.find("\n", i)
            i = n if j < 0 else j
        elif text.startswith("/*", i):
            j = text.find("*/", i + 2)
            out.append(" "); i = n if j < 0 else j + 2
        else:
            out.append(ch); i += 1
    import re
    return re.sub(r"\s+", " ", "".join(out)).strip().encode()


def source_hash(read=None, raw=False):
    """sha256 over the compiled kernel sources and the ABI header: ties committed profiles to the code they were measured on.
    Comments and white space do NOT count (strip_c_comments): round 5 hashed the raw bytes, and a stale sentence in
    include/rg_mpc.h then could not be fixed without orphaning the evidence that carried the hash.  raw=True: the round-5 form
    (tests/test_evidence.py checks the round-5 files with it).
    read: see compiled_sources (a test hashes the sources of a COMMIT through `git show <commit>:<path>`)."""
    import hashlib
    h = hashlib.sha256()
    for rel, data in compiled_sources(read).items():
        h.update(data if raw else rel.encode() + b"\0" + strip_c_comments(data) + b"\0")
    return h.hexdigest()[:16]


def source_hash_at(commit, raw=False):
    """source_hash() of the tree of a commit (needs git and the history: not available on the GPU box)."""
    import subprocess

    def read(rel):
        r = subprocess.run(["git", "show", f"{commit}:{rel}"], cwd=ROOT, capture_output=True, timeout=30)
        return r.stdout if r.returncode == 0 else None
    return source_hash(read, raw)


STAMP_FILE = os.path.join(ROOT, ".rg_source_commit")   # written by tools/stamp_commit.py before a gpurun call (the GPU box has no .git)


def git_head():
    """(commit, dirty) of the compiled sources: from git where there is a history, else from the stamp file
    tools/stamp_commit.py wrote next to the snapshot -- valid only while its source hash is the tree's."""
    import subprocess
    try:
        head = subprocess.run(["git", "rev-parse", "HEAD"], cwd=ROOT, capture_output=True, text=True, timeout=20)
        if head.returncode == 0 and head.stdout.strip():
            rels = list(compiled_sources())
            dirty = subprocess.run(["git", "status", "--porcelain", "--"] + rels, cwd=ROOT, capture_output=True, text=True, timeout=20).stdout.strip()
            return head.stdout.strip(), bool(dirty)
    except (OSError, subprocess.SubprocessError):
        pass
    try:
        st = json.load(open(STAMP_FILE))
        if st.get("source_hash") == source_hash():
            return st.get("commit"), bool(st.get("dirty"))
    except (OSError, ValueError):
        pass
    return None, None


def evidence_header():
    """First line of every evidence file under profiles/: the kernel-source hash and the commit it belongs to."""
    commit, dirty = git_head()
    return f"kernel sources {source_hash()} commit {commit or 'unknown'}{' +uncommitted changes' if dirty else ''}"


def profile_tag(workload_key):
    """profiles/<tag>_* file prefix of a workload: PROFILE_TAG for the headline, PROFILE_TAG_<key> for the others (tools/collect_profiles.sh)."""
    return PROFILE_TAG if workload_key == "headline" else f"{PROFILE_TAG}_{workload_key}"


def load_profile(batch, workload_key):
    """Committed rocprofv3 summary of this same command (profiles/<tag>_traffic.json), or None when there is none for
    this batch / workload or when it was measured on different kernel sources (then the profile-derived fields of the
    bench line are null instead of stale)."""
    tag = profile_tag(workload_key)
    try:
        prof = json.load(open(os.path.join(ROOT, "profiles", f"{tag}_traffic.json")))
    except (OSError, ValueError):
        return None
    if int(prof.get("batch", -1)) != int(batch) or prof.get("workload_key", "headline") != workload_key:
        return None
    if prof.get("source_hash") != source_hash():
        return None
    return prof


def traffic_from_profile(prof, kernel_name):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of this same command
    (FETCH_SIZE and WRITE_SIZE from separate --pmc runs, KiB -> bytes; FETCH_SIZE raw, see tools/summarize_profiles.py)."""
    if not prof:
        return None
    base = kernel_name.split("<")[0]
    for k, v in prof.get("traffic", {}).items():
        if base in k:
            return v["hbm_bytes_per_launch"]
    return None


def issue_view_from_profile(prof, kernel_name):
    """Secondary (non-HBM) view of the dominant kernel from the committed rocprofv3 SQ counter passes of this same
    command: fraction of SIMD issue cycles with a VALU instruction, LDS pipe busy fraction, occupied wave slots.
    MI355X: 8 XCDs x 32 CUs x 4 SIMDs; SQ_* cycle counters are in quad-cycles, GRBM_GUI_ACTIVE sums the 8 XCDs."""
    if not prof:
        return None
    import csv
    base = kernel_name.split("<")[0]
    path = os.path.join(ROOT, "profiles", f"{prof['tag']}_pmc_per_launch.csv")
    if not os.path.exists(path):
        return None
    for r in csv.DictReader(open(path)):
        if base in r["kernel"]:
            cyc = float(r["GRBM_GUI_ACTIVE"]) / 8.0
            flops = None
            if r.get("SQ_INSTS_VALU_FMA_F64") not in (None, "", "nan"):   # wave-level instruction counts x 64 lanes
                flops = 64.0 * (2.0 * float(r["SQ_INSTS_VALU_FMA_F64"]) + float(r["SQ_INSTS_VALU_ADD_F64"]) + float(r["SQ_INSTS_VALU_MUL_F64"]))
            return {"f64_flop_per_launch": flops, "f64_vector_peak_tflops": F64_VECTOR_PEAK_TFLOPS,
                    "valu_busy_frac": round(4.0 * float(r["SQ_ACTIVE_INST_VALU"]) / (cyc * 1024), 3),
                    "lds_busy_frac": round(4.0 * float(r["SQ_ACTIVE_INST_LDS"]) / (cyc * 256), 3),
                    "wave_slot_occupancy": round(4.0 * float(r["SQ_WAVE_CYCLES"]) / (cyc * 2048), 3),
                    "valu_instructions_per_unit": round(float(r["SQ_INSTS_VALU"]) / float(r["SQ_WAVES"])),
                    "source": f"profiles/{prof['tag']}_pmc_per_launch.csv, kernel sources {prof.get('source_hash')} (2 waves/SIMD by register budget = 2048 wave slots)"}
    return None


def with_f64_rate(view, dur_s):
    """f64 vector FLOP/s of the dominant launch: counted FLOPs (committed counter pass) / its live hipEvent duration."""
    if view and view.get("f64_flop_per_launch") and dur_s > 0:
        view["f64_tflops_achieved"] = round(view["f64_flop_per_launch"] / dur_s / 1e12, 2)
        view["f64_frac_of_vector_peak"] = round(view["f64_tflops_achieved"] / view["f64_vector_peak_tflops"], 3)
    return view


def cpu_baseline(cfg, batch, budget_s=10.0, fixed_cmd=None, gait_seed=None, schedule=False, ring=RING, amp=0.1, gpu_mean_iters=50.0):
    """Time the oracle (port) on the host cores on a bounded sample of the same workload: the first min(batch, 2048)
    robots of the same seeded batch, the same per-tick input variation, for as many ticks as fit the budget."""
    from oracle import oracle as O
    from tests import helpers
    from robot_gym_amd import synthetic
    cores = os.cpu_count() or 1
    Bs = min(int(batch), 2048 if cfg.horizon == 10 else 512)
    state, cmd, t_off = synthetic.make_states(Bs, cfg, seed=0, fixed_cmd=fixed_cmd)
    gait = synthetic.random_gaits(Bs, cfg, seed=gait_seed) if gait_seed is not None else None
    ocfg = helpers.oracle_config(O, cfg)
    coff = helpers.cmd_with_offsets(cfg, cmd)
    inputs = []
    for j in range(min(ring, 8)):
        st = perturb_state(state, j, amp) if ring > 1 else state
        contact = synthetic.gait_consistent_contacts(cfg, t_off + 0.01 * j, state["_flip"], gait)
        sched = synthetic.contact_schedule(cfg, t_off + 0.01 * j, gait, dropout=0.1, seed=0, tick=j) if schedule else None
        inputs.append(helpers.oracle_inputs(O, st, coff, contact, sched))

    def fresh(nthreads):
        ob = O.OracleBatch(ocfg, Bs, 0.0, nthreads, gait=gait)
        for b in range(Bs):
            ob.states[b].reset_time = -float(t_off[b])
        return ob

    # per-thread workspaces (no heap calls in the loop over robots); still, a container's CPU quota or the memory system can make
    # fewer threads faster than all logical CPUs -> measure a few counts first and report the search (`thread_search`, `host`)
    best_threads, best_rate = cores, 0.0
    search = {}
    if Bs >= 64:
        for nthreads in sorted({cores, max(1, cores // 2), max(1, cores // 4), max(1, cores // 8), max(1, cores // 16)}, reverse=True):
            ob = fresh(nthreads)
            ob.step(0.0, inputs[0])
            t0, n = time.perf_counter(), 0
            while n < 2 or time.perf_counter() - t0 < 0.6:   # at least two ticks and 0.6 s per candidate: a two-tick sample was noise-sized
                ob.step(0.01 * (n + 1), inputs[(n + 1) % len(inputs)])
                n += 1
            rate = n * Bs / (time.perf_counter() - t0)
            search[str(nthreads)] = round(rate)
            if rate > best_rate:
                best_threads, best_rate = nthreads, rate
    else:
        best_threads = 1
    cores = best_threads
    ob = fresh(cores)
    ob.step(0.0, inputs[0])  # warm
    t0 = time.perf_counter()
    ticks = 0
    while True:
        ob.step(0.01 * (ticks + 1), inputs[(ticks + 1) % len(inputs)])
        ticks += 1
        el = time.perf_counter() - t0
        if el > budget_s or ticks >= 2000:
            break
    def host_cpus():
        """What the box really gives this process: logical CPUs, the affinity mask, and the cgroup CPU quota (a container can
        see 256 logical CPUs and be allowed 32 CPU-seconds per second: more threads than that only add contention)."""
        info = {"cpu_count": os.cpu_count(), "affinity": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None, "cgroup_cpu_max": None}
        for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
            try:
                info["cgroup_cpu_max"] = open(path).read().strip()
                break
            except OSError:
                pass
        return info

    out = {"value": Bs * ticks / el, "unit": "controller steps/s", "cores": cores, "kind": "port", "host": host_cpus(), "thread_search": search,
           "sample": f"{Bs} robots x {ticks} ticks of this workload (seed 0, same per-tick input variation), float64 C oracle with exact active-set QP, OpenMP over robots"}

    def timed(nthreads, budget):
        ob2 = fresh(nthreads)
        ob2.step(0.0, inputs[0])
        t1, n = time.perf_counter(), 0
        while True:
            ob2.step(0.01 * (n + 1), inputs[(n + 1) % len(inputs)])
            n += 1
            e2 = time.perf_counter() - t1
            if e2 > budget or n >= 500:
                return Bs * n / e2, n

    # BASELINE.md section 2 variants, bounded samples of the same inputs (a few seconds each):
    #   B0  one thread, a plain loop over robots -- the shape of the reference's CPU path (one Python controller object per env,
    #       reference controllers/mpc/mpc_controller.py:102-106 called from gym/robot_gym_env.py:120-121), minus the interpreter
    #   B1  the best thread count of the search above, the SAME over-relaxed ADMM as the GPU kernels at the GPU's mean iteration count (dense Cholesky
    #       of P + rho I, two triangular solves and the pyramid projection per iteration) instead of the exact solver
    try:
        v0, n0 = timed(1, 3.0)
        O.set_qp_mode(1, max(1, int(round(gpu_mean_iters))), cfg.admm_rho, cfg.admm_relax)
        all_cores = cores if Bs >= 64 else 1   # the searched thread count (os.cpu_count() oversubscribes a cgroup-limited box: 256 threads on 16 CPUs ran at a third of the 32-thread rate)
        v1, n1 = timed(all_cores, 3.0)
        out["variants"] = {"B0_one_thread_exact_qp": {"value": v0, "cores": 1, "sample": f"{Bs} robots x {n0} ticks"},
                           "B1_all_cores_fixed_count_admm": {"value": v1, "cores": all_cores, "admm_iterations": int(round(gpu_mean_iters)),
                                                             "sample": f"{Bs} robots x {n1} ticks, rho {cfg.admm_rho} relax {cfg.admm_relax}"}}
    finally:
        O.set_qp_mode(0)
    return out


def gpu_sclk_mhz(device_index=0):
    """Current shader clock in MHz of THE GPU this process computes on, from sysfs, or None when the box does not expose it.
    A multi-GPU host shows every card under /sys/class/drm whatever the container may use, so the card is picked by the HIP
    device's PCI address; then hwmon's freq1_input (the current frequency), else the active `*` level of pp_dpm_sclk."""
    import glob
    import re
    want = None
    try:
        pr = torch.cuda.get_device_properties(device_index)
        want = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}"
    except Exception:
        pass
    cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device"))
    if want is None and len([c for c in cards if glob.glob(os.path.join(c, "hwmon", "hwmon*", "freq1_input")) or os.path.exists(os.path.join(c, "pp_dpm_sclk"))]) > 1:
        return None   # several cards expose a clock and torch does not say which one this device is: no guess
    if want is not None:
        match = [c for c in cards if os.path.basename(os.path.realpath(c)).lower().startswith(want)]
        if not match:
            return None
        cards = match
    for dev in cards:
        for path in sorted(glob.glob(os.path.join(dev, "hwmon", "hwmon*", "freq1_input"))):
            try:
                return int(int(open(path).read().strip()) / 1e6)
            except (OSError, ValueError):
                pass
        try:
            for line in open(os.path.join(dev, "pp_dpm_sclk")):
                if line.rstrip().endswith("*"):
                    m = re.search(r"(\d+)\s*[Mm][Hh]z", line)
                    if m:
                        return int(m.group(1))
        except OSError:
            continue
    return None


def dropin_latency(cfg, calls=300):
    """End-to-end latency of the drop-in plugin class, `MPCController.get_action()` at batch 1 (BASELINE configs[0]; the
    reference's bar is the playground's 10 ms control tick, playground/playground.py:122-126): the state gather through the
    reference's Robot getter names (a stub robot serving a synthetic state: PyBullet's own getter time is NOT in this number),
    four Jacobian callbacks, the tick's three launches reading the pinned host slab and writing the action row into pinned
    host memory themselves (zero copy, the default), and the stream synchronisation.  `copy_path`: the same with one upload
    and one download around the launches (rg_mpc_step_host), the form of rounds 4-5."""
    from robot_gym_amd import synthetic
    from robot_gym_amd.controllers.mpc.mpc_controller import MPCController
    from tests.fake_envs import StubRobot
    state, cmd, _ = synthetic.make_states(1, cfg, seed=0)

    def measure(zero_copy):
        clock = [0.0]
        ctl = MPCController(StubRobot(cfg, state, 0), lambda: clock[0], config=cfg, zero_copy=zero_copy)
        ctl.update_controller_params((0.3, 0.0, 0.0))
        lat = []
        for k in range(calls + 20):
            clock[0] = 0.01 * k
            t0 = time.perf_counter()
            ctl.get_action()
            lat.append(time.perf_counter() - t0)
        t0 = time.perf_counter()
        for k in range(calls):
            ctl._gather_state()
        gather = (time.perf_counter() - t0) / calls * 1e6
        ctl._batched.close()
        lat = np.array(lat[20:]) * 1e6
        return {"mean": round(float(lat.mean()), 1), "p50": round(float(np.median(lat)), 1), "p99": round(float(np.percentile(lat, 99)), 1),
                "state_gather_mean": round(gather, 1)}
    out = measure(True)
    out.update({"calls": calls, "copy_path": measure(False),
                "what": "MPCController.get_action() wall time per call, stub robot getters (no PyBullet): 3 launches on the pinned host slab (zero copy) + sync; copy_path: H2D, 3 launches, D2H, sync.  state_gather_mean: the part of it spent in the robot's getters and the four calculateJacobian callbacks before the library is called (three quarters of that is the STUB building an 18-column Jacobian as Python lists, tests/fake_envs.py -- host Python, neither the library nor PyBullet)"})
    return out


def self_launch(args, argv):
    """`python bench.py --gpus N` with N > 1 and no RANK in the environment: start the N ranks ourselves as a FRESH child
    process tree -- `python -m torch.distributed.run --nproc-per-node N bench.py <same arguments>` -- BEFORE this process
    has made any GPU call (never an exec of a process that touched the GPU), relay rank 0's JSON line and exit with the
    child's code.  One process per GPU over RCCL (gloo under --dry-launch)."""
    import socket
    import subprocess
    if not args.dry_launch:
        have = torch.cuda.device_count()   # this process makes no other GPU call; the ranks are a fresh child process either way
        if have < args.gpus:
            raise SystemExit(f"bench.py --gpus {args.gpus}: this node has {have} GPU(s)")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + [a for a in argv if a != "--force-launcher"]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it on this driver
    # RCCL: no algorithm / protocol knob is forced.  The one exchange is the optional all-gather of 0.98 MB per rank; RCCL's
    # all-gather is a ring (7 serial hops, each bound by one xGMI link), and the alternative -- every rank sends its slab
    # straight to its 7 peers over the pairwise links -- is chosen at the torch level (core/sharding.py, schedule "direct":
    # one grouped batch of point-to-point operations), not by an environment variable.  Both are timed below whenever there
    # is more than one rank.  NCCL_DEBUG=WARN only makes a failing rendezvous say why.
    env.setdefault("NCCL_DEBUG", "WARN")
    res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for l in res.stdout.splitlines():
        if l.startswith("{") and '"metric"' in l:
            line = l
        else:
            print(l, file=sys.stderr)
    if line is not None:
        print(line, flush=True)
    elif res.returncode == 0:
        print("bench.py: the ranks exited cleanly but rank 0 printed no JSON line", file=sys.stderr)
        sys.exit(1)
    sys.exit(res.returncode)


class DryController:
    """--dry-launch stand-in for BatchedMPCController on CPU: same calls, no GPU, no solver.  It exists so that the
    multi-rank protocol of this file (rendezvous, barriers, MAX-over-ranks timing, the action all-gather, the per-rank
    gathers, the JSON contract) runs in the CPU test suite over gloo at world size 2."""

    class _Handle:
        def profile_stride(self, n): pass
        def profile_begin(self, n): self.n = n
        def profile_end(self, stream=None): return 1, [0.001, 0.01, 0.001, 0.0, 0.0, 0.012], [0, 0, 1, 0, 0]
        def profile_window_names(self): return ["rg_front_kernel", "rg_qp_fused_kernel", "rg_qp_resolve_kernel", "-", "-", "step_total"]

    def __init__(self, batch):
        self.batch = batch
        self.action = torch.zeros(batch, 60, dtype=torch.float32)
        self._handle = self._Handle()

    def _stream(self): return None
    def set_gait(self, **kw): pass
    def reset_at(self, t0): pass
    def update_controller_params(self, cmd): pass
    def get_action(self, t, state):
        self.action.add_(1.0)
        return self.action
    def solver_stats(self): return {"iters_sum": 0, "iters_max": 0, "qp_robots": self.batch, "retried_exact": 0, "failures": 0, "iters_mean": 0.0}
    def audit_stats(self, reset=False): return {"audited": 0, "audit_over_tol": 0, "audit_max_rel": 0.0, "audit_max_rel_elem": 0.0, "audit_exact_failures": 0, "audit_dropped": 0}
    def close(self): pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=BATCH_PER_GPU, help="robots per GPU")
    ap.add_argument("--total-batch", type=int, default=None, help="robots over ALL ranks, sharded with core.sharding.shard_bounds (shards may differ by one robot); overrides --batch")
    ap.add_argument("--allgather", action="store_true", help="`value` includes the RCCL all-gather of the action slab inside the timed step (default: further timed passes report it as config.with_allgather_steps_per_s / with_allgather_direct_steps_per_s)")
    ap.add_argument("--allgather-schedule", choices=("ring", "direct"), default="ring", help="schedule of the all-gather that --allgather puts into `value` (core/sharding.py)")
    ap.add_argument("--dry-launch", action="store_true", help="CPU dry run of the multi-rank protocol: gloo, a stub controller, no GPU")
    ap.add_argument("--force-launcher", action="store_true", help="start the ranks through self_launch even for --gpus 1 (tests the launcher on a 1-GPU box)")
    ap.add_argument("--kin-mode", type=int, default=0, help="1 = foot positions / Jacobians from joint angles on the device (chain kinematics replacing controllers/mpc/kinematics.py)")
    ap.add_argument("--chain-geometry", action="store_true", help="synthetic foot positions / Jacobians = forward kinematics of the synthetic joint angles (what kin_mode 1 computes on the device), so that --kin-mode 0 and 1 solve the same QPs")
    ap.add_argument("--robot", default="ghost")
    ap.add_argument("--audit-k", type=int, default=None, help="audit lane picks per tick (0 = off)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--admm-iters", type=int, default=None, help="fixed ADMM iteration count (disables the convergence test)")
    ap.add_argument("--solver", type=int, default=None, help="0 = ADMM only, 1 = exact active set for every robot, 2 = ADMM with the exact re-solve behind it, 3 = hybrid (default): exact for one / two stance legs, ADMM for three / four")
    ap.add_argument("--cold-start", action="store_true", help="start ADMM from scratch every tick (default: warm start from the robot's previous-tick iterate)")
    ap.add_argument("--cap", type=int, default=None, help="ADMM iteration cap (keeps the convergence test)")
    ap.add_argument("--rho", type=float, default=None)
    ap.add_argument("--relax", type=float, default=None)
    ap.add_argument("--tol", type=float, default=None)
    ap.add_argument("--extrap", type=float, default=None, help="geometric-extrapolation convergence guard (admm_extrap; 0 = off)")
    ap.add_argument("--check", type=int, default=None, help="convergence vote period")
    ap.add_argument("--accel", type=int, default=None, help="first iteration at which a vote may extrapolate the iterate (admm_accel; 0 = off)")
    ap.add_argument("--rho34", type=float, default=None, help="admm_rho34_scale: first-stage rho of the wrench-space ADMM body = rho x this")
    ap.add_argument("--rho-sched", type=float, default=None, help="admm_rho_sched_scale: first-stage rho of the schedule body = rho x this")
    ap.add_argument("--rho2", type=float, default=None, help="second-stage ADMM rho of the contact-schedule body (0 = single stage)")
    ap.add_argument("--switch", type=int, default=None, help="first-stage iteration count of the contact-schedule body")
    ap.add_argument("--no-kernel-events", action="store_true", help="diagnostic: do not record per-kernel HIP events in the timed region (roofline.kernel_ms is then empty)")
    ap.add_argument("--lane-grid", type=int, default=None, help="lanes per robot of the default plan's QP launch at horizon 10: 1 = one wave, 2 = 256 lanes, 0 / unset = by batch size (rg_mpc_config.lane_grid)")
    ap.add_argument("--horizon", type=int, default=HORIZON, help="MPC horizon (10 = the headline workload; 20 = BASELINE configs[4] shape)")
    ap.add_argument("--lookahead", action="store_true", help="opt-in contact-schedule extension (per-step contacts from the open-loop gait)")
    ap.add_argument("--random-schedule", action="store_true", help="BASELINE configs[4]: per-robot duty ~ U(0.5, 0.8) and a caller-supplied contact schedule with 10 %% drop-outs, re-drawn every tick (implies --lookahead)")
    ap.add_argument("--fixed-cmd", action="store_true", help="BASELINE configs[1]: fixed forward-velocity command (0.3, 0, 0) instead of randomised commands")
    ap.add_argument("--static-inputs", action="store_true", help="feed one frozen state slab every tick (the round-1 bench) instead of the input ring")
    ap.add_argument("--ring", type=int, default=RING, help="state slabs in the input ring")
    ap.add_argument("--jitter", type=float, default=0.1, help="amplitude of the per-tick input variation")
    ap.add_argument("--no-extras", action="store_true", help="skip the static-input and PCIe-inclusive side measurements")
    ap.add_argument("--sustain-s", type=float, default=5.0, help="seconds of back-to-back ticks behind config.sustained_steps_per_s (0 = skip; part of the extras)")
    args = ap.parse_args()
    if args.random_schedule:
        args.lookahead = True
    ring = 1 if args.static_inputs else max(1, args.ring)

    if "RANK" not in os.environ and (args.gpus > 1 or args.force_launcher):
        self_launch(args, sys.argv[1:])   # does not return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dry = args.dry_launch
    sync = (lambda: None) if dry else torch.cuda.synchronize
    dist = None
    if "RANK" in os.environ and "WORLD_SIZE" in os.environ:   # launched by torch.distributed.run (any N, also N = 1)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if dry:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    elif not dry:
        torch.cuda.set_device(0)
    device = torch.device("cpu") if dry else torch.device("cuda", torch.cuda.current_device())

    from robot_gym_amd.core.config import MPCConfig
    from robot_gym_amd.controllers.mpc.batched import BatchedMPCController
    from robot_gym_amd import synthetic
    over = {} if args.admm_iters is None else {"admm_iters": args.admm_iters, "admm_tol": 0.0}
    if args.solver is not None:
        over["solver"] = args.solver
    if args.cap is not None:
        over["admm_iters"] = args.cap
    if args.cold_start:
        over["warm_start"] = 0
    if args.rho is not None:
        over["admm_rho"] = args.rho
    if args.relax is not None:
        over["admm_relax"] = args.relax
    if args.tol is not None:
        over["admm_tol"] = args.tol
    if args.extrap is not None:
        over["admm_extrap"] = args.extrap
    if args.check is not None:
        over["admm_check"] = args.check
    if args.accel is not None:
        over["admm_accel"] = args.accel
    if args.rho2 is not None:
        over["admm_rho2"] = args.rho2
    if args.rho34 is not None:
        over["admm_rho34_scale"] = args.rho34
    if args.rho_sched is not None:
        over["admm_rho_sched_scale"] = args.rho_sched
    if args.switch is not None:
        over["admm_switch"] = args.switch
    if args.lookahead:
        over["contact_lookahead"] = 1
    if args.kin_mode:
        over["kin_mode"] = args.kin_mode
    if args.audit_k is not None:
        over["audit_k"] = args.audit_k
    if args.lane_grid is not None:
        over["lane_grid"] = args.lane_grid
    cfg = MPCConfig.for_robot(args.robot, horizon=args.horizon, **over)
    B = args.batch
    total_robots = world * B
    if args.total_batch is not None:   # uneven shards: rank r owns shard_bounds(total, r, world)
        from robot_gym_amd.core.sharding import shard_bounds
        lo_, hi_ = shard_bounds(args.total_batch, rank, world)
        B, total_robots = hi_ - lo_, args.total_batch
    fixed_cmd = (0.3, 0.0, 0.0) if args.fixed_cmd else None
    # the robot batch shards trivially: rank r owns robots [r*B, (r+1)*B) -- different seed per shard
    gait = synthetic.random_gaits(B, cfg, seed=rank) if args.random_schedule else None
    state, cmd, t_off, slabs = make_input_ring(cfg, B, rank, device, ring, args.jitter, fixed_cmd, gait, args.random_schedule, args.chain_geometry)
    gathered = torch.empty(total_robots, 60, dtype=torch.float32, device=device) if dist is not None else None
    gather_bufs = None
    if dist is not None and args.total_batch is not None:   # uneven shards travel padded: staged in persistent buffers, nothing allocated in the timed loop
        from robot_gym_amd.core.sharding import GatherBuffers
        gather_bufs = GatherBuffers(args.total_batch, world, 60, torch.float32, device)

    def run(slab_list, steps, warmup, events, cfg_run=None, allgather=None, clock_probe=None):
        """`warmup` untimed then `steps` timed ticks on a fresh controller; returns (seconds, handle-side profile, stats).
        allgather: "ring" / "direct" -- every step also all-gathers the [B, 60] action slab over the process group (RCCL over
        xGMI) with that schedule (core/sharding.py)."""
        from robot_gym_amd.core.sharding import all_gather_actions
        ctl = DryController(B) if dry else BatchedMPCController(B, cfg_run or cfg, device=device, extra_outputs=False)
        if gait is not None:
            ctl.set_gait(**gait)
        ctl.reset_at(-t_off)
        ctl.update_controller_params(torch.from_numpy(cmd.T.copy()).to(device))
        nslab = len(slab_list)

        def one_step(k):
            act = ctl.get_action(0.01 * k, slab_list[k % nslab])
            if allgather:
                all_gather_actions(act, out=gathered, schedule=allgather, total=(args.total_batch if args.total_batch is not None else None), buffers=gather_bufs)

        for k in range(warmup):
            one_step(k)
        sync()
        if dist is not None:
            dist.barrier()
        sync()
        if events:
            # per-kernel HIP events on every 4th step of the timed region (an event record costs ~4-5 us of stream time)
            ctl._handle.profile_stride(EVENT_STRIDE)
            ctl._handle.profile_begin(steps)
        t0 = time.perf_counter()
        for k in range(steps):
            one_step(warmup + k)
            if clock_probe is not None and k in (steps // 100, steps // 2, steps - 1):   # (the host runs ahead of the GPU by its queue depth only)
                clock_probe.append(gpu_sclk_mhz(device.index if device.type == "cuda" else 0))
        sync()
        if dist is not None:
            dist.barrier()
        sync()
        el = time.perf_counter() - t0
        prof = ctl._handle.profile_end(ctl._stream()) if events else (0, [0.0] * 6, [0] * 5)
        wn = ctl._handle.profile_window_names()
        stats = ctl.solver_stats()
        stats["audit"] = ctl.audit_stats()
        return el, prof, stats, wn, ctl

    def max_over_ranks(x):
        if dist is None:
            return x
        tt = torch.tensor([x], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    use_ag = bool(args.allgather and dist is not None)
    elapsed, (nprof, kms, robots), stats, wn, ctl = run(slabs, args.steps, args.warmup, not args.no_kernel_events, allgather=(args.allgather_schedule if use_ag else None))
    elapsed = max_over_ranks(elapsed)
    # SURVEY.md 8e asks for both rates -- without the all-gather and with it -- and the exchange has two schedules (ring /
    # direct, core/sharding.py): further timed passes of the same steps, only when there is a process group to gather over
    ag_elapsed = {}
    plain_elapsed = None if use_ag else elapsed
    ag_note = None
    if dist is not None:
        if use_ag:
            ag_elapsed[args.allgather_schedule] = elapsed
        if "ring" not in ag_elapsed:
            ctl.close()
            e_, _, _, _, ctl = run(slabs, args.steps, args.warmup, False, allgather="ring")
            ag_elapsed["ring"] = max_over_ranks(e_)
        if use_ag:
            ctl.close()
            e_, _, _, _, ctl = run(slabs, args.steps, args.warmup, False)
            plain_elapsed = max_over_ranks(e_)
    # per-rank kernel times (ms): every rank reports its own hipEvent averages
    per_rank_kms = None
    if dist is not None:
        mine = torch.tensor(list(kms[:6]), dtype=torch.float64, device=device)
        allk = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allk, mine)
        per_rank_kms = [[round(float(v), 4) for v in t.tolist()] for t in allk]

    # Side measurements (reported separately, never `value`), single GPU only:
    #  - the same steps with ONE frozen input slab (round 1's bench; flatters the cost-class launch order, whose
    #    prediction from the previous tick is then perfect)
    #  - PCIe-inclusive rate: the gym side holds the robot state on the host -- pinned buffers, one upload of all inputs
    #    and one download of the action slab per tick
    pcie_value = static_value = cold_value = steady_value = dropin_us = audit_off_value = sustained = None
    if world == 1 and dist is None and not args.no_extras and not dry:
        from robot_gym_amd.controllers.mpc.batched import PackedState
        names_io = ("rpy", "rpy_rate", "v_world", "quat", "q", "foot_pos", "jac", "contact")
        ps = PackedState(B, device)     # what MPCVecEnv uses: one pinned slab -> one H2D copy per tick
        act_host = torch.empty(B, 60, dtype=torch.float32).pin_memory()
        nio = max(5, min(args.steps, 20))
        host_slabs = [{n: slabs[j % len(slabs)][n].cpu() for n in names_io} for j in range(min(len(slabs), 4))]
        sched_slabs = [slabs[j % len(slabs)].get("contact_sched") for j in range(len(host_slabs))]
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for k in range(nio):
            hs = host_slabs[k % len(host_slabs)]
            for n in names_io:
                ps.host[n].copy_(hs[n])       # host-side gather of the robot state into the pinned slab
            sdev = dict(ps.upload())
            if sched_slabs[k % len(host_slabs)] is not None:
                sdev["contact_sched"] = sched_slabs[k % len(host_slabs)]
            act_host.copy_(ctl.get_action(0.01 * (args.warmup + args.steps + k), sdev), non_blocking=True)
        torch.cuda.synchronize()
        pcie_value = B * nio / (time.perf_counter() - t1)
        if ring > 1:
            ctl.close()
            el_s, _, _, _, ctl = run(slabs[:1], args.steps, args.warmup, False)
            static_value = B * args.steps / el_s
        if cfg.warm_start and not args.lookahead:
            import dataclasses
            ctl.close()
            el_c, _, _, _, ctl = run(slabs, args.steps, args.warmup, False, dataclasses.replace(cfg, warm_start=0))
            cold_value = B * args.steps / el_c
        # the driver's default line times the first ticks after a cold start with per-kernel events on every 4th tick; the
        # steady state -- warm starts and cost-class predictions settled, no events -- over 200 ticks:
        # (best of three for both 200-tick figures: one such run is 30 ms at the headline, and a single host hiccup once
        # turned an audit-off pass into "15 M steps/s" next to 28 M with the audit on)
        def best_of(n, cfg_run=None):
            nonlocal ctl
            best = float("inf")
            for _ in range(n):
                ctl.close()
                el_, _, _, _, ctl = run(slabs, 200, 20, False, cfg_run)
                best = min(best, el_)
            return best
        steady_value = B * 200 / best_of(3)
        # what the audit lane costs this workload: the same 200 ticks without it
        if cfg.audit_k > 0:
            import dataclasses
            audit_off_value = B * 200 / best_of(3, dataclasses.replace(cfg, audit_k=0))
        # sustained load: >= args.sustain_s seconds of back-to-back ticks (clocks and power settle on that time scale, and the
        # driver's SMI sampler sees the GPU busy), with the shader clock read at the start, in the middle and at the end
        if args.sustain_s > 0:
            nticks = int(min(2_000_000, max(200, args.sustain_s * 1.05 / (B / steady_value))))
            ctl.close()
            clocks = []
            el_su, _, _, _, ctl = run(slabs, nticks, 20, False, clock_probe=clocks)
            sustained = {"steps_per_s": B * nticks / el_su, "seconds": round(el_su, 2), "ticks": nticks,
                         "vs_value_pct": None, "sclk_mhz_sysfs": clocks or None}   # (this GPU's sysfs shader clock at three instants of the run: hwmon freq1_input, else the active pp_dpm_sclk level; not a clock trace)
        if B == 1:
            try:
                dropin_us = dropin_latency(cfg)
            except Exception as e:   # a reported extra
                dropin_us = {"error": f"{type(e).__name__}: {e}"}

    def emit_line():
        total_units = total_robots * args.steps
        value = total_units / elapsed
        names = wn[:5]
        if "fused" in wn[1] or "sched" in wn[1]:   # one QP launch over all stance-leg counts, then the exact re-solve launches
            units = [B, robots[1] + robots[2] + robots[3] + robots[4], stats["retried_exact"], 0, 0]
        else:
            units = [B, robots[1], robots[2], robots[3], robots[4]]
        dom = int(np.argmax(kms[:5]))
        dur_s = kms[dom] * 1e-3
        # algorithmic bytes of the dominant launch: the per-step I/O of DESIGN.md section 5, plus -- with the warm start -- the
        # previous-tick ADMM iterate (z, y as float32, read and written: 16 B per QP variable, 3 * legs * horizon variables)
        warm_on = bool(cfg.warm_start) and not args.lookahead
        exact_upto = (4 if cfg.solver == 1 else 2 if cfg.solver == 3 else 0) if (args.horizon == 10 and not args.lookahead) else 0   # stance-leg counts on an exact body
        warm_bytes = sum((EXACT_WS_BYTES if nc <= exact_upto else 16 * 3 * args.horizon * nc) * robots[nc] for nc in range(1, 5))
        algo_bytes_launch = ALGO_BYTES_PER_STEP * units[dom] + (warm_bytes if (warm_on and dom == 1) else 0)
        achieved = (algo_bytes_launch / dur_s) / 1e9 if dur_s > 0 else 0.0
        if args.random_schedule:
            wl, wkey = f"batch={B} quadrupeds per GPU, horizon={args.horizon}, per-robot duty U(0.5,0.8), randomised contact schedule with 10% drop-outs re-drawn per tick (BASELINE configs[4])", "config5"
        elif args.fixed_cmd:
            wl, wkey = f"batch={B} quadrupeds per GPU, horizon={args.horizon}, fixed forward-velocity command (BASELINE configs[1])", "config2"
        else:
            wl, wkey = f"batch={B} quadrupeds per GPU, horizon={args.horizon}{' with gait-driven contact schedule' if args.lookahead else ''}, randomised (vx,vy,wz) commands (BASELINE configs[2])", ("headline" if args.horizon == HORIZON and not args.lookahead else f"h{args.horizon}{'la' if args.lookahead else ''}")
            if B != BATCH_PER_GPU:
                wkey = f"b{B}" if wkey == "headline" else f"{wkey}_b{B}"
        if args.kin_mode:
            wl += f", kin_mode {args.kin_mode} (foot positions, Jacobians and IK from joint angles on the device)"
            wkey = f"kin{args.kin_mode}" if wkey == "headline" else f"{wkey}_kin{args.kin_mode}"
        if args.chain_geometry and not args.kin_mode:
            wl += ", foot positions / Jacobians = chain forward kinematics of the joint angles (the geometry kin_mode 1 computes)"
            wkey = "kin0chain" if wkey == "headline" else f"{wkey}_kin0chain"
        if args.random_schedule and args.cap is not None:
            wkey = f"config5_cap{args.cap}"
        if args.lane_grid:
            wl += f", lane grid {args.lane_grid} ({'one wave' if args.lane_grid == 1 else '256 lanes'} per robot)"
            wkey = f"grid{args.lane_grid}" if wkey == "headline" else f"{wkey}_grid{args.lane_grid}"
        if args.solver is not None and args.solver != MPCConfig.for_robot(args.robot).solver:   # another solver plan than the default: its own profile key
            wl += f", solver plan {args.solver}"
            wkey = f"s{args.solver}" if wkey == "headline" else f"{wkey}_s{args.solver}"
        prof = load_profile(B, wkey)
        audit = stats.pop("audit", None)
        out = {
            "metric": f"MPC controller steps/sec (whole node), batch={B} quadrupeds, horizon={args.horizon}",
            "value": value, "unit": "controller steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": wl,
                       "input_schedule": ("one frozen state slab (static inputs)" if ring == 1 else
                                          f"ring of {ring} state slabs resident in HBM, one per tick: v_world / rpy_rate scaled by 1 +- {args.jitter}, roll/pitch +- {0.2 * args.jitter:.3g} rad, foot positions +- {20 * args.jitter:.3g} %, measured contacts following the gait"),
                       "static_inputs": ring == 1, "static_inputs_steps_per_s": static_value, "cold_start_steps_per_s": cold_value,
                       "steady_state_steps_per_s": steady_value, "audit_off_steady_state_steps_per_s": audit_off_value,
                       "audit_cost_pct": (round(100.0 * (1.0 - steady_value / audit_off_value), 2) if (audit_off_value and steady_value) else None),
                       "sustained_steps_per_s": (sustained["steps_per_s"] if sustained else None),
                       "sustained": (dict(sustained, vs_value_pct=round(100.0 * sustained["steps_per_s"] / value, 1)) if sustained else None),
                       "dropin_get_action_latency_us": dropin_us,
                       "robot": cfg.robot,
                       "solver": {0: "ADMM only", 1: "exact active set for every robot", 2: "ADMM + exact re-solve", 3: "hybrid: exact active set (1-2 stance legs), ADMM + exact re-solve (3-4)"}[cfg.solver]
                                 + f"; admm rho={cfg.admm_rho} (x{cfg.admm_rho34_scale} wrench body) relax={cfg.admm_relax} tol={cfg.admm_tol} check={cfg.admm_check} cap={cfg.admm_iters}" + (f" second stage rho={cfg.admm_rho2} after {cfg.admm_switch}" if args.lookahead else ""),
                       "admm_iterations": stats,
                       "warm_start": warm_on, "kin_mode": cfg.kin_mode, "allgather": use_ag,
                       "with_allgather_steps_per_s": (total_units / ag_elapsed["ring"] if "ring" in ag_elapsed else None),
                       "with_allgather_direct_steps_per_s": (total_units / ag_elapsed["direct"] if "direct" in ag_elapsed else None),
                       "allgather_schedule": (args.allgather_schedule if use_ag else None), "allgather_note": ag_note,
                       "without_allgather_steps_per_s": total_units / plain_elapsed,
                       "rccl_ranks": (dist.get_world_size() if dist is not None else 1), "backend": (dist.get_backend() if dist is not None else None),
                       "kernel_ms_per_rank": per_rank_kms, "dry_launch": dry,
                       "audit": audit,
                       "pcie_inclusive_steps_per_s": pcie_value,
                       "sharding": (f"{world} x {B} robots" if args.total_batch is None else f"{total_robots} robots over {world} ranks (shard_bounds: {total_robots // world} or {total_robots // world + 1} each)") + ", no data-path collective",
                       "kernel_sources": source_hash(), "commit": git_head()[0]},
            "roofline": {"bound": "hbm", "kernel": names[dom], "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic_from_profile(prof, names[dom]),
                         "units_per_launch": units[dom], "algorithmic_bytes_per_unit": round(algo_bytes_launch / max(units[dom], 1), 1),
                         "avg_launch_ms": kms[dom],
                         "limiter": "f64 VALU issue on a per-robot dependent chain (see issue_view), not HBM",
                         "issue_view": with_f64_rate(issue_view_from_profile(prof, names[dom]), dur_s),
                         "profile": (f"profiles/{prof['tag']}_* (same kernel sources)" if prof else "no committed rocprof summary for these kernel sources / this workload: traffic and issue_view are null"),
                         "kernel_ms": {n: round(x, 4) for n, x in zip(names + ["step_total"], kms) if n != "-"},
                         "kernel_ms_note": "hipEvent windows of the profiled ticks: step_total spans four event records (~1 us of stream time each), which the timed region behind ms_per_step does not carry -- it can exceed ms_per_step by a few us",
                         "robots_per_stance_count": robots,
                         "note": "path is instruction-issue/latency-bound, not HBM-bound (SURVEY.md 7.3-2): see issue_view and DESIGN.md section 5"},
        }
        if world == 1 and not args.no_cpu_baseline and not dry:
            try:
                out["cpu_baseline"] = cpu_baseline(cfg, B, fixed_cmd=fixed_cmd, gait_seed=(0 if args.random_schedule else None),
                                                   schedule=args.random_schedule, ring=ring, amp=args.jitter,
                                                   gpu_mean_iters=stats.get("iters_mean", 50.0))
            except Exception as e:  # the baseline is a reported extra, never the product path
                out["cpu_baseline"] = {"value": None, "unit": "controller steps/s", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {type(e).__name__}: {e}"}
        print(json.dumps(out), flush=True)

    # The direct (point-to-point) all-gather schedule has only ever run over gloo (no multi-GPU node was reachable): it is timed
    # LAST and under a watchdog, so that a hang in it cannot cost the run its line -- after the deadline rank 0 prints the line
    # without that number and every rank leaves with a NON-ZERO code (a run that lost a measurement to a hang did not succeed;
    # self_launch relays the line and the code).  The line is printed exactly once (lock + flag), and the watchdog stays armed
    # until the process group is gone: a rank whose own pass failed can still block in destroy_process_group on hung peers.
    import threading
    emit_lock, emitted = threading.Lock(), [False]

    def emit_once():
        with emit_lock:
            if not emitted[0]:
                emitted[0] = True
                emit_line()

    finished = threading.Event()
    if dist is not None and "direct" not in ag_elapsed and os.environ.get("RG_BENCH_DIRECT_ALLGATHER", "1") != "0":
        def watchdog():
            if not finished.wait(float(os.environ.get("RG_BENCH_DIRECT_TIMEOUT_S", "120"))):
                nonlocal ag_note
                ag_note = "the direct all-gather pass (or the shutdown after it) did not finish within its deadline"
                if rank == 0:
                    emit_once()
                os._exit(3)

        threading.Thread(target=watchdog, daemon=True).start()
        try:
            ctl.close()
            e_, _, _, _, ctl = run(slabs, args.steps, args.warmup, False, allgather="direct")
            ag_elapsed["direct"] = max_over_ranks(e_)
        except Exception as e:   # reported, never fatal for the line
            ag_note = f"direct all-gather pass failed: {type(e).__name__}: {e}"
    if rank == 0:
        emit_once()
    ctl.close()
    if dist is not None:
        dist.destroy_process_group()
    finished.set()


if __name__ == "__main__":
    main()

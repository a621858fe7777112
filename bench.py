#!/usr/bin/env python3
"""Headline benchmark: MPC controller steps/sec (whole node), batch=4096 quadrupeds per GPU,
horizon=10 (BASELINE.json).  One "step" = one rg_mpc_step over one batch of synthetic robot
states already resident in HBM.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

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
PROFILE_TAG = "r2"
# algorithmic HBM bytes per controller step (kin_mode 0, all optional outputs off)
ALGO_BYTES_PER_STEP = 1110
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


def make_input_ring(cfg, B, seed, device, ring, amp, fixed_cmd=None, gait=None, schedule=False):
    """`ring` input slabs on the device: slab j is the state handed to tick k = j (mod ring)."""
    from robot_gym_amd import synthetic
    state, cmd, t_off = synthetic.make_states(B, cfg, seed=seed, fixed_cmd=fixed_cmd)
    names = ("rpy", "rpy_rate", "v_world", "quat", "q", "foot_pos", "jac")
    slabs = []
    for j in range(ring):
        st = perturb_state(state, j, amp) if ring > 1 else state
        dev = {n: torch.from_numpy(np.ascontiguousarray(st[n])).to(device) for n in names}
        dev["contact"] = torch.from_numpy(synthetic.gait_consistent_contacts(cfg, t_off + 0.01 * j, state["_flip"], gait)).to(device)
        if schedule:   # BASELINE config 5: randomised per-step contact schedule, re-drawn every tick
            dev["contact_sched"] = torch.from_numpy(synthetic.contact_schedule(cfg, t_off + 0.01 * j, gait, dropout=0.1, seed=seed, tick=j)).to(device)
        slabs.append(dev)
    return state, cmd, t_off, slabs


def source_hash():
    """sha256 over the kernel sources and the ABI header: ties committed profiles to the code they were measured on."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "robot_gym_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".inc", ".h")):
            h.update(open(os.path.join(d, f), "rb").read())
    h.update(open(os.path.join(ROOT, "include", "rg_mpc.h"), "rb").read())
    return h.hexdigest()[:16]


def profile_tag(workload_key):
    """profiles/<tag>_* file prefix of a workload: r2 for the headline, r2_<key> for the others (tools/collect_profiles.sh)."""
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


def cpu_baseline(cfg, batch, budget_s=10.0, fixed_cmd=None, gait_seed=None, schedule=False, ring=RING, amp=0.1):
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

    # the port allocates per step; on many-core hosts fewer threads can be faster -> pick the best count first
    best_threads, best_rate = cores, 0.0
    if Bs >= 64:
        for nthreads in sorted({cores, max(1, cores // 2), max(1, cores // 4), max(1, cores // 8)}, reverse=True):
            ob = fresh(nthreads)
            ob.step(0.0, inputs[0])
            t0 = time.perf_counter()
            ob.step(0.01, inputs[1 % len(inputs)])
            rate = Bs / (time.perf_counter() - t0)
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
    return {"value": Bs * ticks / el, "unit": "controller steps/s", "cores": cores, "kind": "port",
            "sample": f"{Bs} robots x {ticks} ticks of this workload (seed 0, same per-tick input variation), float64 C oracle with exact active-set QP, OpenMP over robots"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=BATCH_PER_GPU, help="robots per GPU")
    ap.add_argument("--allgather", action="store_true", help="also all-gather the action slab over RCCL inside the timed step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--admm-iters", type=int, default=None, help="fixed ADMM iteration count (disables the convergence test)")
    ap.add_argument("--solver", type=int, default=None, help="0 = ADMM, 1 = exact active set")
    ap.add_argument("--cold-start", action="store_true", help="start ADMM from scratch every tick (default: warm start from the robot's previous-tick iterate)")
    ap.add_argument("--cap", type=int, default=None, help="ADMM iteration cap (keeps the convergence test)")
    ap.add_argument("--rho", type=float, default=None)
    ap.add_argument("--relax", type=float, default=None)
    ap.add_argument("--tol", type=float, default=None)
    ap.add_argument("--extrap", type=float, default=None, help="geometric-extrapolation convergence guard (admm_extrap; 0 = off)")
    ap.add_argument("--check", type=int, default=None, help="convergence vote period")
    ap.add_argument("--accel", type=int, default=None, help="first iteration at which a vote may extrapolate the iterate (admm_accel; 0 = off)")
    ap.add_argument("--rho2", type=float, default=None, help="second-stage ADMM rho of the contact-schedule body (0 = single stage)")
    ap.add_argument("--switch", type=int, default=None, help="first-stage iteration count of the contact-schedule body")
    ap.add_argument("--no-kernel-events", action="store_true", help="diagnostic: do not record per-kernel HIP events in the timed region (roofline.kernel_ms is then empty)")
    ap.add_argument("--horizon", type=int, default=HORIZON, help="MPC horizon (10 = the headline workload; 20 = BASELINE configs[4] shape)")
    ap.add_argument("--lookahead", action="store_true", help="opt-in contact-schedule extension (per-step contacts from the open-loop gait)")
    ap.add_argument("--random-schedule", action="store_true", help="BASELINE configs[4]: per-robot duty ~ U(0.5, 0.8) and a caller-supplied contact schedule with 10 %% drop-outs, re-drawn every tick (implies --lookahead)")
    ap.add_argument("--fixed-cmd", action="store_true", help="BASELINE configs[1]: fixed forward-velocity command (0.3, 0, 0) instead of randomised commands")
    ap.add_argument("--static-inputs", action="store_true", help="feed one frozen state slab every tick (the round-1 bench) instead of the input ring")
    ap.add_argument("--ring", type=int, default=RING, help="state slabs in the input ring")
    ap.add_argument("--jitter", type=float, default=0.1, help="amplitude of the per-tick input variation")
    ap.add_argument("--no-extras", action="store_true", help="skip the static-input and PCIe-inclusive side measurements")
    args = ap.parse_args()
    if args.random_schedule:
        args.lookahead = True
    ring = 1 if args.static_inputs else max(1, args.ring)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    dist = None
    if "RANK" in os.environ and "WORLD_SIZE" in os.environ:   # launched by torch.distributed.run (any N, also N = 1)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    device = torch.device("cuda", torch.cuda.current_device())

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
    if args.switch is not None:
        over["admm_switch"] = args.switch
    if args.lookahead:
        over["contact_lookahead"] = 1
    cfg = MPCConfig.for_robot("ghost", horizon=args.horizon, **over)
    B = args.batch
    fixed_cmd = (0.3, 0.0, 0.0) if args.fixed_cmd else None
    # the robot batch shards trivially: rank r owns robots [r*B, (r+1)*B) -- different seed per shard
    gait = synthetic.random_gaits(B, cfg, seed=rank) if args.random_schedule else None
    state, cmd, t_off, slabs = make_input_ring(cfg, B, rank, device, ring, args.jitter, fixed_cmd, gait, args.random_schedule)
    gathered = torch.empty(world * B, 60, dtype=torch.float32, device=device) if (args.allgather and dist is not None) else None

    def run(slab_list, steps, warmup, events, cfg_run=None):
        """`warmup` untimed then `steps` timed ticks on a fresh controller; returns (seconds, handle-side profile, stats)."""
        ctl = BatchedMPCController(B, cfg_run or cfg, device=device, extra_outputs=False)
        if gait is not None:
            ctl.set_gait(**gait)
        ctl.reset_at(-t_off)
        ctl.update_controller_params(torch.from_numpy(cmd.T.copy()).to(device))
        nslab = len(slab_list)

        def one_step(k):
            act = ctl.get_action(0.01 * k, slab_list[k % nslab])
            if gathered is not None:
                dist.all_gather_into_tensor(gathered, act)

        for k in range(warmup):
            one_step(k)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        if events:
            # per-kernel HIP events on every 4th step of the timed region (an event record costs ~4-5 us of stream time)
            ctl._handle.profile_stride(EVENT_STRIDE)
            ctl._handle.profile_begin(steps)
        t0 = time.perf_counter()
        for k in range(steps):
            one_step(warmup + k)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        prof = ctl._handle.profile_end(ctl._stream()) if events else (0, [0.0] * 6, [0] * 5)
        wn = ctl._handle.profile_window_names()
        stats = ctl.solver_stats()
        return el, prof, stats, wn, ctl

    elapsed, (nprof, kms, robots), stats, wn, ctl = run(slabs, args.steps, args.warmup, not args.no_kernel_events)
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # Side measurements (reported separately, never `value`), single GPU only:
    #  - the same steps with ONE frozen input slab (round 1's bench; flatters the cost-class launch order, whose
    #    prediction from the previous tick is then perfect)
    #  - PCIe-inclusive rate: the gym side holds the robot state on the host -- pinned buffers, one upload of all inputs
    #    and one download of the action slab per tick
    pcie_value = static_value = cold_value = None
    if world == 1 and dist is None and not args.no_extras:
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

    if rank == 0:
        total_units = world * B * args.steps
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
        algo_bytes_launch = ALGO_BYTES_PER_STEP * units[dom] + (16 * 3 * args.horizon * sum(nc * robots[nc] for nc in range(1, 5)) if (warm_on and dom == 1) else 0)
        achieved = (algo_bytes_launch / dur_s) / 1e9 if dur_s > 0 else 0.0
        if args.random_schedule:
            wl, wkey = f"batch={B} quadrupeds per GPU, horizon={args.horizon}, per-robot duty U(0.5,0.8), randomised contact schedule with 10% drop-outs re-drawn per tick (BASELINE configs[4])", "config5"
        elif args.fixed_cmd:
            wl, wkey = f"batch={B} quadrupeds per GPU, horizon={args.horizon}, fixed forward-velocity command (BASELINE configs[1])", "config2"
        else:
            wl, wkey = f"batch={B} quadrupeds per GPU, horizon={args.horizon}{' with gait-driven contact schedule' if args.lookahead else ''}, randomised (vx,vy,wz) commands (BASELINE configs[2])", ("headline" if args.horizon == HORIZON and not args.lookahead else f"h{args.horizon}{'la' if args.lookahead else ''}")
            if B != BATCH_PER_GPU:
                wkey = f"b{B}" if wkey == "headline" else f"{wkey}_b{B}"
        prof = load_profile(B, wkey)
        out = {
            "metric": f"MPC controller steps/sec (whole node), batch={B} quadrupeds, horizon={args.horizon}",
            "value": value, "unit": "controller steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": wl,
                       "input_schedule": ("one frozen state slab (static inputs)" if ring == 1 else
                                          f"ring of {ring} state slabs resident in HBM, one per tick: v_world / rpy_rate scaled by 1 +- {args.jitter}, roll/pitch +- {0.2 * args.jitter:.3g} rad, foot positions +- {20 * args.jitter:.3g} %, measured contacts following the gait"),
                       "static_inputs": ring == 1, "static_inputs_steps_per_s": static_value, "cold_start_steps_per_s": cold_value,
                       "robot": "ghost", "solver": f"admm rho={cfg.admm_rho} relax={cfg.admm_relax} tol={cfg.admm_tol} check={cfg.admm_check} cap={cfg.admm_iters}" + (f" second stage rho={cfg.admm_rho2} after {cfg.admm_switch}" if args.lookahead else ""), "admm_iterations": stats,
                       "warm_start": warm_on, "kin_mode": cfg.kin_mode, "allgather": bool(gathered is not None),
                       "pcie_inclusive_steps_per_s": pcie_value, "sharding": f"{world} x {B} robots, no data-path collective",
                       "kernel_sources": source_hash()},
            "roofline": {"bound": "hbm", "kernel": names[dom], "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic_from_profile(prof, names[dom]),
                         "units_per_launch": units[dom], "algorithmic_bytes_per_unit": round(algo_bytes_launch / max(units[dom], 1), 1),
                         "avg_launch_ms": kms[dom],
                         "limiter": "f64 VALU issue on a per-robot dependent chain (see issue_view), not HBM",
                         "issue_view": with_f64_rate(issue_view_from_profile(prof, names[dom]), dur_s),
                         "profile": (f"profiles/{prof['tag']}_* (same kernel sources)" if prof else "no committed rocprof summary for these kernel sources / this workload: traffic and issue_view are null"),
                         "kernel_ms": {n: round(x, 4) for n, x in zip(names + ["step_total"], kms) if n != "-"},
                         "robots_per_stance_count": robots,
                         "note": "path is instruction-issue/latency-bound, not HBM-bound (SURVEY.md 7.3-2): see issue_view and DESIGN.md section 5"},
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(cfg, B, fixed_cmd=fixed_cmd, gait_seed=(0 if args.random_schedule else None),
                                                   schedule=args.random_schedule, ring=ring, amp=args.jitter)
            except Exception as e:  # the baseline is a reported extra, never the product path
                out["cpu_baseline"] = {"value": None, "unit": "controller steps/s", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {type(e).__name__}: {e}"}
        print(json.dumps(out), flush=True)
    ctl.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

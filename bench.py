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
# algorithmic HBM bytes per controller step (kin_mode 0, all optional outputs off)
ALGO_BYTES_PER_STEP = 1110
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
F64_VECTOR_PEAK_TFLOPS = 78.6   # 256 CUs x 4 SIMDs x 16 lanes x 2 flop x 2.4 GHz (v_fma_f64 issues at 4 cycles per wave)


def make_device_state(cfg, B, seed, device):
    from robot_gym_amd import synthetic
    state, cmd, t_off = synthetic.make_states(B, cfg, seed=seed)
    contact = synthetic.gait_consistent_contacts(cfg, t_off, state["_flip"])
    dev = {n: torch.from_numpy(np.ascontiguousarray(state[n])).to(device)
           for n in ("rpy", "rpy_rate", "v_world", "quat", "q", "foot_pos", "jac")}
    dev["contact"] = torch.from_numpy(contact).to(device)
    return state, cmd, t_off, contact, dev


def traffic_from_profile(kernel_name, batch):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of this same
    command (profiles/r1_traffic.json: FETCH_SIZE and WRITE_SIZE from separate --pmc runs, KiB -> bytes;
    FETCH_SIZE raw, see tools/summarize_profiles.py).  None when no profile of this batch size is committed."""
    try:
        prof = json.load(open(os.path.join(ROOT, "profiles", "r1_traffic.json")))
        if int(prof.get("batch", -1)) != int(batch):
            return None
        nc = kernel_name.split("nc=")[1][0] if "nc=" in kernel_name else None
        for k, v in prof["traffic"].items():
            if (nc and f"tile_kernel<{nc}," in k and "true>" not in k) or (nc is None and kernel_name.split("<")[0] in k):
                return v["hbm_bytes_per_launch"]
    except Exception:
        pass
    return None


def issue_view_from_profile(kernel_name, batch):
    """Secondary (non-HBM) view of the dominant kernel from the committed rocprofv3 SQ counter passes of this same
    command: fraction of SIMD issue cycles with a VALU instruction, LDS pipe busy fraction, occupied wave slots.
    MI355X: 8 XCDs x 32 CUs x 4 SIMDs; SQ_* cycle counters are in quad-cycles, GRBM_GUI_ACTIVE sums the 8 XCDs."""
    try:
        import csv
        meta = json.load(open(os.path.join(ROOT, "profiles", "r1_traffic.json")))
        if int(meta.get("batch", -1)) != int(batch):
            return None
        base = kernel_name.split("<")[0]
        for r in csv.DictReader(open(os.path.join(ROOT, "profiles", "r1_pmc_per_launch.csv"))):
            if base in r["kernel"] and "true>" not in r["kernel"]:
                cyc = float(r["GRBM_GUI_ACTIVE"]) / 8.0
                flops = None
                if r.get("SQ_INSTS_VALU_FMA_F64") not in (None, "", "nan"):   # wave-level instruction counts x 64 lanes
                    flops = 64.0 * (2.0 * float(r["SQ_INSTS_VALU_FMA_F64"]) + float(r["SQ_INSTS_VALU_ADD_F64"]) + float(r["SQ_INSTS_VALU_MUL_F64"]))
                return {"f64_flop_per_launch": flops, "f64_vector_peak_tflops": F64_VECTOR_PEAK_TFLOPS,
                        "valu_busy_frac": round(4.0 * float(r["SQ_ACTIVE_INST_VALU"]) / (cyc * 1024), 3),
                        "lds_busy_frac": round(4.0 * float(r["SQ_ACTIVE_INST_LDS"]) / (cyc * 256), 3),
                        "wave_slot_occupancy": round(4.0 * float(r["SQ_WAVE_CYCLES"]) / (cyc * 2048), 3),
                        "valu_instructions_per_unit": round(float(r["SQ_INSTS_VALU"]) / float(r["SQ_WAVES"])),
                        "source": "profiles/r1_pmc_per_launch.csv (2 waves/SIMD by register budget = 2048 wave slots)"}
    except Exception:
        pass
    return None


def with_f64_rate(view, dur_s):
    """f64 vector FLOP/s of the dominant launch: counted FLOPs (committed counter pass) / its live hipEvent duration."""
    if view and view.get("f64_flop_per_launch") and dur_s > 0:
        view["f64_tflops_achieved"] = round(view["f64_flop_per_launch"] / dur_s / 1e12, 2)
        view["f64_frac_of_vector_peak"] = round(view["f64_tflops_achieved"] / view["f64_vector_peak_tflops"], 3)
    return view


def cpu_baseline(cfg, budget_s=10.0):
    """Time the oracle (port) on the host cores on a bounded sample of the same workload."""
    from oracle import oracle as O
    from tests import helpers
    from robot_gym_amd import synthetic
    cores = os.cpu_count() or 1
    Bs = 2048
    state, cmd, t_off = synthetic.make_states(Bs, cfg, seed=0)
    ocfg = helpers.oracle_config(O, cfg)
    coff = helpers.cmd_with_offsets(cfg, cmd)
    contact = synthetic.gait_consistent_contacts(cfg, t_off, state["_flip"])
    inp = helpers.oracle_inputs(O, state, coff, contact)

    def fresh(nthreads):
        ob = O.OracleBatch(ocfg, Bs, 0.0, nthreads)
        for b in range(Bs):
            ob.states[b].reset_time = -float(t_off[b])
        return ob

    # the port allocates per step; on many-core hosts fewer threads can be faster -> pick the best count first
    best_threads, best_rate = cores, 0.0
    for nthreads in sorted({cores, max(1, cores // 2), max(1, cores // 4), max(1, cores // 8)}, reverse=True):
        ob = fresh(nthreads)
        ob.step(0.0, inp)
        t0 = time.perf_counter()
        ob.step(0.01, inp)
        rate = Bs / (time.perf_counter() - t0)
        if rate > best_rate:
            best_threads, best_rate = nthreads, rate
    cores = best_threads
    ob = fresh(cores)
    ob.step(0.0, inp)  # warm
    t0 = time.perf_counter()
    ticks = 0
    while True:
        ob.step(0.01 * (ticks + 1), inp)
        ticks += 1
        el = time.perf_counter() - t0
        if el > budget_s or ticks >= 200:
            break
    return {"value": Bs * ticks / el, "unit": "controller steps/s", "cores": cores, "kind": "port",
            "sample": f"{Bs} robots x {ticks} ticks of the batch=4096 workload (seed 0), float64 C oracle with exact active-set QP, OpenMP over robots"}


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
    ap.add_argument("--warm-start", action="store_true", help="opt-in ADMM warm start from the previous tick (not the headline configuration)")
    ap.add_argument("--cap", type=int, default=None, help="ADMM iteration cap (keeps the convergence test)")
    ap.add_argument("--rho", type=float, default=None)
    ap.add_argument("--relax", type=float, default=None)
    ap.add_argument("--tol", type=float, default=None)
    ap.add_argument("--no-kernel-events", action="store_true", help="diagnostic: do not record per-kernel HIP events in the timed region (roofline.kernel_ms is then empty)")
    ap.add_argument("--horizon", type=int, default=HORIZON, help="MPC horizon (10 = the headline workload; 20 = BASELINE configs[4] shape)")
    ap.add_argument("--lookahead", action="store_true", help="opt-in contact look-ahead extension (per-step contact schedule from the open-loop gait)")
    ap.add_argument("--reserved0", type=int, default=0, help="tuning bits passed to rg_mpc_config.reserved0")
    args = ap.parse_args()

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
    over = {} if args.admm_iters is None else {"admm_iters": args.admm_iters, "admm_tol": 0.0}
    over["reserved0"] = args.reserved0
    if args.solver is not None:
        over["solver"] = args.solver
    if args.cap is not None:
        over["admm_iters"] = args.cap
    if args.warm_start:
        over["warm_start"] = 1
    if args.rho is not None:
        over["admm_rho"] = args.rho
    if args.relax is not None:
        over["admm_relax"] = args.relax
    if args.tol is not None:
        over["admm_tol"] = args.tol
    if args.lookahead:
        over["contact_lookahead"] = 1
    cfg = MPCConfig.for_robot("ghost", horizon=args.horizon, **over)
    B = args.batch
    # the robot batch shards trivially: rank r owns robots [r*B, (r+1)*B) -- different seed per shard
    state, cmd, t_off, contact, dev = make_device_state(cfg, B, seed=rank, device=device)
    ctl = BatchedMPCController(B, cfg, device=device, extra_outputs=False)
    ctl.reset_at(-t_off)
    ctl.update_controller_params(torch.from_numpy(cmd.T.copy()).to(device))
    gathered = torch.empty(world * B, 60, dtype=torch.float32, device=device) if (args.allgather and world > 1) else None

    def one_step(k):
        act = ctl.get_action(0.01 * k, dev)
        if gathered is not None:
            dist.all_gather_into_tensor(gathered, act)

    for k in range(args.warmup):
        one_step(k)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    if not args.no_kernel_events:
        # per-kernel HIP events on every 4th step of the timed region (an event record costs ~4-5 us of stream time)
        ctl._handle.profile_stride(EVENT_STRIDE)
        ctl._handle.profile_begin(args.steps)
    t0 = time.perf_counter()
    for k in range(args.steps):
        one_step(args.warmup + k)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    nprof, kms, robots = ctl._handle.profile_end(ctl._stream())
    stats = ctl.solver_stats()
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # PCIe-inclusive rate (reported separately, never `value`): the same steps when the gym side holds the
    # robot state on the host -- pinned buffers, one upload of all inputs and one download of the action slab per tick.
    pcie_value = None
    if world == 1:
        from robot_gym_amd.controllers.mpc.batched import PackedState
        names_io = ("rpy", "rpy_rate", "v_world", "quat", "q", "foot_pos", "jac", "contact")
        ps = PackedState(B, device)     # what MPCVecEnv uses: one pinned slab -> one H2D copy per tick
        for n in names_io:
            ps.host[n].copy_(dev[n].cpu())
        act_host = torch.empty(B, 60, dtype=torch.float32).pin_memory()
        nio = max(5, min(args.steps, 20))
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for k in range(nio):
            sdev = ps.upload()
            act_host.copy_(ctl.get_action(0.01 * (args.warmup + args.steps + k), sdev), non_blocking=True)
        torch.cuda.synchronize()
        pcie_value = B * nio / (time.perf_counter() - t1)

    if rank == 0:
        total_units = world * B * args.steps
        value = total_units / elapsed
        wn = ctl._handle.profile_window_names()
        names = wn[:5]
        if "fused" in wn[1]:   # one QP launch over all stance-leg counts, then the exact re-solve launches
            units = [B, robots[1] + robots[2] + robots[3] + robots[4], stats["retried_exact"], 0, 0]
        else:
            units = [B, robots[1], robots[2], robots[3], robots[4]]
        dom = int(np.argmax(kms[:5]))
        dur_s = kms[dom] * 1e-3
        achieved = (ALGO_BYTES_PER_STEP * units[dom] / dur_s) / 1e9 if dur_s > 0 else 0.0
        out = {
            "metric": f"MPC controller steps/sec (whole node), batch={B} quadrupeds, horizon={args.horizon}",
            "value": value, "unit": "controller steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"batch={B} quadrupeds per GPU, horizon={args.horizon}{' with contact look-ahead' if args.lookahead else ''}, randomised (vx,vy,wz) commands (BASELINE configs[2])",
                       "robot": "ghost", "solver": f"admm rho={cfg.admm_rho} relax={cfg.admm_relax} tol={cfg.admm_tol} check={cfg.admm_check} cap={cfg.admm_iters}", "admm_iterations": stats,
                       "warm_start": bool(cfg.warm_start), "kin_mode": cfg.kin_mode, "allgather": bool(gathered is not None),
                       "pcie_inclusive_steps_per_s": pcie_value, "sharding": f"{world} x {B} robots, no data-path collective"},
            "roofline": {"bound": "hbm", "kernel": names[dom], "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic_from_profile(names[dom], B),
                         "units_per_launch": units[dom], "algorithmic_bytes_per_unit": ALGO_BYTES_PER_STEP,
                         "avg_launch_ms": kms[dom],
                         "issue_view": with_f64_rate(issue_view_from_profile(names[dom], B), dur_s),
                         "kernel_ms": {n: round(x, 4) for n, x in zip(names + ["step_total"], kms) if n != "-"},
                         "robots_per_stance_count": robots,
                         "note": "path is instruction-issue/latency-bound, not HBM-bound (SURVEY.md 7.3-2): see issue_view and DESIGN.md section 5"},
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(cfg)
            except Exception as e:  # the baseline is a reported extra, never the product path
                out["cpu_baseline"] = {"value": None, "unit": "controller steps/s", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {e}"}
        print(json.dumps(out), flush=True)
    ctl.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

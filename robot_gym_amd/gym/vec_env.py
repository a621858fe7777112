"""Batched VecEnv wrapper: B reference-style envs, ONE batched controller call per tick.

API of the reference's BatchEnv (agents/ppo/tools/batch_env.py:18-115): `BatchEnv(envs, blocking)`, space-equality check at
construction, `len`, indexing, attribute forwarding to the first env (PPO reads `observation_space` / `action_space`
through it), `step(actions)` with `action_space.contains` validation -> stacked (observ, reward, done, info),
`reset(indices)`, `close()` closing every sub-env.  Per-env tick order of RobotGymEnv.step
(gym/robot_gym_env.py:117-129): command -> controller action -> ApplyStepAction -> (update_equip) -> observation /
reward / termination.

Each sub-env keeps its OWN `step()`: task-level logic such as GoEnv.step's action clipping, on-target standing action and
camera hook (gym/envs/go_to/go_env.py:272-296) is not re-implemented here.  Every sub-env is built with
`BatchSlotController` as its controller class; a tick then runs in three phases:

  1. every env runs up to the point where the reference asks the controller for its action.  Envs that provide the
     two-half protocol
         pre_step(action, **kwargs) -> (command, kwargs)      everything before controller.update_controller_params
         post_step(motor_action, **kwargs) -> (obs, reward, done, info)   from simulation.ApplyStepAction on
     run their pre-controller code ONCE, and so do envs built by `split_step.one_pass(TaskEnv, BaseEnv)`: a generic
     interceptor placed after the task env in the MRO suspends the step where the task env's own step() calls
     `super().step(action, **kwargs)` and later resumes `BaseEnv.step` with those arguments -- no env code is restated.
     Any other env is driven through its unmodified step() twice: the first pass is unwound at the controller call
     (`get_action` raises StepSuspended, caught here), the second pass gets the action.  Code in front of the controller
     call therefore runs twice for such an env -- harmless when it only derives the command (the slot controller refuses
     a second pass that derives a different one, before anything is applied), NOT harmless when it has side effects of
     its own (GoEnv(show_plot=True)._update_plot, RNG draws, counters): build those envs with `one_pass`.
  2. the wrapper gathers every robot's state, its own clock and its command into ONE pinned slab, uploads it once, runs
     rg_mpc_step for all envs and downloads the [B, 60] action slab once.  Pending per-env resets (the env's
     Simulation.reset() -> controller.reset()) are applied first, each with that env's own clock value.
  3. every env finishes its step with its action row; ApplyStepAction, observation, reward and termination are the env's
     own code.

`blocking=False` (reference batch_env.py:80-84 steps `ExternalProcess` workers concurrently, wrappers.py:294-458): the envs
live in worker PROCESSES -- `workers` of them, each hosting a contiguous slice of the batch and built there from
`constructors` (callables, like ExternalProcess's `constructor`).  Phases 1 and 3 of all slices run in parallel; the
workers write their state columns straight into one shared [82, B] slab and read their action rows from a shared [B, 60]
slab; the parent makes the one rg_mpc_step in between.  One process per env, as in the reference, does not scale to the
1024-4096 envs the GPU side is built for; a slice per worker does.

`devices=[d0, d1, ...]` (the gym side of BASELINE configs[3], batch sharded over the GPUs of one node, SURVEY.md 8e "one handle +
one stream per device"): the batch is cut into contiguous shards (core/sharding.shard_bounds), each with its own
BatchedMPCController (handle) on its own stream of its device; the host keeps ONE pinned state buffer -- shard s owns the
contiguous [82, n_s] block behind the blocks of the shards before it -- and ONE pinned [B, 60] action slab whose rows
[lo_s, hi_s) shard s downloads into.  The host needs every action row anyway (PyBullet steps on the CPU), so one process
driving N GPUs needs NO collective at all; all uploads and launches of a tick are enqueued before the first wait.  The same
device may appear more than once (two handles and two streams on one GPU: how the CI box with one GPU tests this).
One process per GPU instead: every rank builds an MPCVecEnv over ITS shard of the envs (shard_bounds(total, rank, world)) and
calls core/sharding.all_gather_actions only if it wants every rank's rows (tests/test_host_logic.py, world size 2, gloo).

Physics stays per-env on the CPU (PyBullet).  Clocks: robot b is stepped at env b's own GetTimeSinceReset() (per-robot
clock array of the C-ABI), exactly like B separate reference controllers -- a partial reset never shifts another env.
"""
import multiprocessing
import sys
import traceback
from multiprocessing import shared_memory

import numpy as np
import torch

from robot_gym_amd.controllers.mpc.batched import BatchedMPCController, PackedState, STATE_FIELDS
from robot_gym_amd.controllers.mpc.slot_controller import BatchSlotController, StepSuspended

SLAB_WORDS = 2 + sum(c for _, c, _ in STATE_FIELDS) + 3   # PackedState layout: clock (2 rows), the state fields, command


def _slab_views(slab):
    """Views into a [SLAB_WORDS, B] float32 slab laid out like PackedState: ({field: array}, clock float64 [B], cmd [3, B])."""
    views, row = {}, 2
    for name, comps, dt in STATE_FIELDS:
        v = slab[row:row + comps]
        views[name] = v if dt == torch.float32 else v.view(np.int32)
        row += comps
    return views, slab[0:2].reshape(-1).view(np.float64), slab[row:row + 3]


def _default_jacobian(env, leg):
    return env.simulation.controller.kinematics_model.leg_jacobian(leg)


class _EnvGroup:
    """A contiguous slice [lo, lo + n) of the batch living in ONE process: phases 1 and 3 of a tick and the state gather
    for its envs.  The in-process wrapper owns one group over the whole batch; every worker process owns one over its slice."""

    def __init__(self, envs, lo, views, clock, cmd, cfg, jacobian_fn, base=None):
        """lo: first column of this slice in `views`; base: batch index of its first env (default lo: the views span the batch)."""
        self.envs, self.lo, self.cfg = list(envs), lo, cfg
        self.base = lo if base is None else base
        self.h, self.clock, self.cmd = views, clock, cmd
        self.slots = [env.simulation.controller for env in self.envs]
        for b, ctl in enumerate(self.slots):
            if not isinstance(ctl, BatchSlotController):
                raise TypeError(f"env {self.base + b}: simulation.controller is {type(ctl).__name__}; MPCVecEnv needs envs built with "
                                "controller_class=BatchSlotController (one slot of the batched GPU controller per env)")
        self.split = [hasattr(env, "pre_step") and hasattr(env, "post_step") for env in self.envs]
        self.one_pass = [not sp and hasattr(env, "resume_step") for sp, env in zip(self.split, self.envs)]   # split_step.one_pass classes
        self.jacobian_fn = jacobian_fn or _default_jacobian
        self.offsets = np.array([cfg.vx_offset, cfg.vy_offset, cfg.wz_offset], dtype=np.float32).reshape(3, 1)
        self.kwargs = [None] * len(self.envs)

    def pre(self, actions):
        """Phase 1: every env up to its controller call."""
        for b, (env, a) in enumerate(zip(self.envs, actions)):
            ctl = self.slots[b]
            if self.split[b]:
                command, self.kwargs[b] = env.pre_step(a)
                ctl.update_controller_params(command)
                continue
            ctl.phase = "capture"
            try:
                env.step(a)
            except StepSuspended:
                pass
            else:
                raise RuntimeError(f"env {self.base + b}: step() returned without asking its controller for an action")
            finally:
                ctl.phase = "idle"

    def gather(self):
        """Robot state, clock and offset-corrected command of every env into the slab columns of this slice: one pass over
        the envs per reference getter, one array assignment per field (not 11 small slice writes per env).
        Returns the resets the batch has not applied yet, [(batch index, clock value at the reset)]."""
        lo, hi, h = self.lo, self.lo + len(self.envs), self.h
        robots = [env.simulation.robot for env in self.envs]
        h["rpy"][:, lo:hi] = np.asarray([rb.GetBaseRollPitchYaw() for rb in robots], dtype=np.float32).T
        h["rpy_rate"][:, lo:hi] = np.asarray([rb.GetBaseRollPitchYawRate() for rb in robots], dtype=np.float32).T
        h["v_world"][:, lo:hi] = np.asarray([rb.GetBaseVelocity() for rb in robots], dtype=np.float32).T
        h["quat"][:, lo:hi] = np.asarray([rb.GetTrueBaseOrientation() for rb in robots], dtype=np.float32).T
        h["q"][:, lo:hi] = np.asarray([rb.GetMotorAngles() for rb in robots], dtype=np.float32).T
        h["contact"][:, lo:hi] = np.asarray([rb.GetFootContacts() for rb in robots], dtype=np.int32).T
        if self.cfg.kin_mode == 0:
            h["foot_pos"][:, lo:hi] = np.asarray([rb.GetFootPositionsInBaseFrame() for rb in robots], dtype=np.float32).reshape(len(robots), 12).T
            h["jac"][:, lo:hi] = np.asarray([[self.jacobian_fn(env, leg) for leg in range(4)] for env in self.envs], dtype=np.float32).reshape(len(robots), 36).T
        self.clock[lo:hi] = [env.simulation.GetTimeSinceReset() for env in self.envs]
        # lin = [vx + VX_OFFSET, vy + VY_OFFSET, 0], ang = wz + WZ_OFFSET (reference mpc_controller.py:90-95), float32
        self.cmd[:, lo:hi] = np.asarray([ctl.command for ctl in self.slots], dtype=np.float32).T + self.offsets
        resets = [(self.base + b, ctl.reset_clock) for b, ctl in enumerate(self.slots) if ctl.reset_clock is not None]
        for ctl in self.slots:
            ctl.reset_clock = None
        return resets

    def post(self, actions, rows):
        """Phase 3: every env from its controller call on, with its action row."""
        transitions = []
        for b, (env, a) in enumerate(zip(self.envs, actions)):
            ctl = self.slots[b]
            if self.split[b]:
                transitions.append(env.post_step(np.array(rows[b], dtype=np.float32), **(self.kwargs[b] or {})))
                continue
            if self.one_pass[b]:
                transitions.append(env.resume_step(rows[b]))   # BaseEnv.step with the arguments the task env's step() ended in
                continue
            ctl.begin_replay(rows[b])
            try:
                transitions.append(env.step(a))
            finally:
                ctl.phase, ctl.action = "idle", None
        return transitions

    def reset(self, local_indices):
        observs = []
        for i in local_indices:
            env, ctl = self.envs[i], self.slots[i]
            ctl.reset_clock = None
            observs.append(np.asarray(env.reset()))
            if ctl.reset_clock is None:   # the env did not route through Simulation.reset() -> controller.reset()
                ctl.reset()
        return observs


# ---- worker processes (blocking=False) -------------------------------------------------------------------------------------
_STEP, _ACT, _RESET, _ATTRIBUTE, _CLOSE, _READY, _TRANSITION, _OBSERV, _VALUE, _EXCEPTION = range(10)


def _worker_main(conn, constructors, lo, batch, shm_state, shm_act, cfg, jacobian_fn):
    """One slice of the batch in its own process (the reference's ExternalProcess._worker, wrappers.py:419-456, for n envs
    and with the controller call cut out of the middle of the step)."""
    try:
        s1, s2 = shared_memory.SharedMemory(name=shm_state), shared_memory.SharedMemory(name=shm_act)
        slab = np.ndarray((SLAB_WORDS, batch), dtype=np.float32, buffer=s1.buf)
        act = np.ndarray((batch, 60), dtype=np.float32, buffer=s2.buf)
        views, clock, cmd = _slab_views(slab)
        envs = [c() for c in constructors]
        group = _EnvGroup(envs, lo, views, clock, cmd, cfg or envs[0].simulation.controller.config, jacobian_fn)
        conn.send((_READY, (envs[0].observation_space, envs[0].action_space, [e.observation_space == envs[0].observation_space and e.action_space == envs[0].action_space for e in envs],
                            group.cfg)))
        while True:
            try:
                if not conn.poll(0.1):
                    continue
                message, payload = conn.recv()
            except (EOFError, KeyboardInterrupt):
                break
            if message == _STEP:
                actions = payload
                group.pre(actions)
                conn.send((_READY, group.gather()))
                message, _ = conn.recv()
                if message == _CLOSE:   # the parent gave up on this tick (another slice failed): leave without finishing it
                    break
                if message != _ACT:
                    raise KeyError(f"expected the action message, got {message}")
                # one message of stacked arrays per slice, not n pickled tuples (the parent's serial unpickling of 4096 small
                # objects was most of the tick)
                observs, rewards, dones, infos = zip(*group.post(actions, act[lo:lo + len(envs)]))
                conn.send((_TRANSITION, (np.stack(observs), np.stack(rewards), np.stack(dones), list(infos))))
            elif message == _RESET:
                conn.send((_OBSERV, group.reset(payload)))
            elif message == _ATTRIBUTE:
                conn.send((_VALUE, getattr(envs[0], payload)))
            elif message == _CLOSE:
                for env in envs:
                    if hasattr(env, "close"):
                        env.close()
                break
            else:
                raise KeyError(f"Received message of unknown type {message}")
    except Exception:  # pylint: disable=broad-except
        conn.send((_EXCEPTION, "".join(traceback.format_exception(*sys.exc_info()))))
    finally:
        conn.close()


class _Worker:
    def __init__(self, ctx, constructors, lo, batch, shm_state, shm_act, cfg, jacobian_fn):
        self.lo, self.n = lo, len(constructors)
        self.conn, child = ctx.Pipe()
        self.process = ctx.Process(target=_worker_main, args=(child, constructors, lo, batch, shm_state, shm_act, cfg, jacobian_fn), daemon=True)
        self.process.start()

    def receive(self, expected):
        message, payload = self.conn.recv()
        if message == _EXCEPTION:
            raise Exception(payload)   # re-raised in the main process, like reference wrappers.py:411-413
        if message != expected:
            raise KeyError("Received message of unexpected type {}".format(message))
        return payload


class _Shard:
    """Rows [lo, hi) of the batch on one device: its controller handle, its block of the pinned state buffer, its stream."""

    def __init__(self, lo, hi, controller, state, stream):
        self.lo, self.hi, self.controller, self.state, self.stream = lo, hi, controller, state, stream


class MPCVecEnv:
    def __init__(self, envs=None, blocking=True, device=None, config=None, jacobian_fn=None, constructors=None, workers=None, devices=None):
        """envs: RobotGymEnv-like objects (`.simulation` with robot / controller / GetTimeSinceReset / ApplyStepAction,
        `step`, `reset`, `observation_space`, `action_space`) whose controller is a BatchSlotController -- stepped in this
        process (`blocking=True`, the reference's name for "one after another").
        blocking=False: pass `constructors` instead, one callable per env (picklable: the workers are spawned, never forked
        from a process that holds a GPU context); `workers` processes (default min(8, B)) each build and step a slice.
        devices: HIP devices to shard the batch over, one controller handle and stream each (module docstring); default: the
        one `device`."""
        self._blocking = bool(blocking)
        self._workers, self._shm = [], []
        if self._blocking:
            if not envs:
                raise ValueError("need at least one env")
            self._envs = list(envs)
            B = len(self._envs)
            # reference agents/ppo/tools/batch_env.py:37-42
            observ_space = self._envs[0].observation_space
            if not all(env.observation_space == observ_space for env in self._envs):
                raise ValueError("All environments must use the same observation space.")
            action_space = self._envs[0].action_space
            if not all(env.action_space == action_space for env in self._envs):
                raise ValueError("All environments must use the same action space.")
            self._action_space = action_space
            self._check_slots(self._envs)
            self.cfg = config or self._envs[0].simulation.controller.config
        else:
            if not constructors:
                raise ValueError("blocking=False needs `constructors`: one callable per env, run inside the worker processes")
            self._envs = None
            B = len(constructors)
            nw = max(1, min(int(workers or 8), B))
            ctx = multiprocessing.get_context("spawn")
            s1 = shared_memory.SharedMemory(create=True, size=4 * SLAB_WORDS * B)
            s2 = shared_memory.SharedMemory(create=True, size=4 * 60 * B)
            self._shm = [s1, s2]
            self._shared_slab = np.ndarray((SLAB_WORDS, B), dtype=np.float32, buffer=s1.buf)
            self._shared_act = np.ndarray((B, 60), dtype=np.float32, buffer=s2.buf)
            self._shared_slab[:] = 0
            bounds = [(w * B) // nw for w in range(nw + 1)]
            try:
                self._workers = [_Worker(ctx, list(constructors[bounds[w]:bounds[w + 1]]), bounds[w], B, s1.name, s2.name, config, jacobian_fn)
                                 for w in range(nw)]
                hello = [w.receive(_READY) for w in self._workers]
            except Exception:
                self.close()
                raise
            observ_space, action_space = hello[0][0], hello[0][1]
            if not all(h[0] == observ_space and h[1] == action_space and all(h[2]) for h in hello):
                self.close()
                raise ValueError("All environments must use the same observation space.")
            self._observation_space, self._action_space = observ_space, action_space
            self.cfg = config or hello[0][3]
        self._batch = B
        # the GPU context is created only now, after the workers were started
        from robot_gym_amd.core.sharding import shard_bounds
        devs = list(devices) if devices else [device]
        if len(devs) > B:
            raise ValueError(f"{len(devs)} devices for {B} envs")
        pin = torch.cuda.is_available()
        # ONE pinned host buffer for the state of every shard (shard s: a contiguous [82, n_s] block) and ONE action slab
        self._host_buffer = torch.zeros(SLAB_WORDS * B, dtype=torch.float32, pin_memory=pin)
        self._act_host = torch.zeros(B, 60, dtype=torch.float32, pin_memory=pin)
        self._shards = []
        for s, dv in enumerate(devs):
            lo, hi = shard_bounds(B, s, len(devs))
            ctl = BatchedMPCController(hi - lo, self.cfg, device=dv, extra_outputs=False)
            state = PackedState(hi - lo, ctl.device, pin, host_storage=self._host_buffer[SLAB_WORDS * lo:SLAB_WORDS * hi])
            stream = torch.cuda.Stream(device=ctl.device) if (ctl.device.type == "cuda" and len(devs) > 1) else None
            self._shards.append(_Shard(lo, hi, ctl, state, stream))
        self.controllers = [sh.controller for sh in self._shards]
        self._dev, self._state = self.controller.device, self._shards[0].state
        if self._blocking:
            self._groups = []
            for sh in self._shards:
                views = {n: t.numpy() for n, t in sh.state.host.items()}
                self._groups.append(_EnvGroup(self._envs[sh.lo:sh.hi], 0, views, sh.state.host_clock.numpy(), sh.state.host_cmd.numpy(), self.cfg, jacobian_fn, base=sh.lo))
            self._group = self._groups[0]
            self._slots = [sl for g in self._groups for sl in g.slots]
        self.batched_calls = 0
        self._broken = None   # set when a tick failed half-way: the batch is then in no defined state

    @property
    def controller(self):
        """The first shard's controller (the only one unless `devices` was given)."""
        return self._shards[0].controller

    @controller.setter
    def controller(self, ctl):
        self._shards[0].controller = ctl
        self.controllers[0] = ctl

    @staticmethod
    def _check_slots(envs):
        for b, env in enumerate(envs):
            ctl = env.simulation.controller
            if not isinstance(ctl, BatchSlotController):
                raise TypeError(f"env {b}: simulation.controller is {type(ctl).__name__}; MPCVecEnv needs envs built with "
                                "controller_class=BatchSlotController (one slot of the batched GPU controller per env)")

    def __len__(self):
        return self._batch

    def __getitem__(self, index):
        if not self._blocking:
            raise TypeError("the envs of a non-blocking MPCVecEnv live in worker processes; use attribute forwarding")
        return self._envs[index]

    def __getattr__(self, name):
        """Forward unimplemented attributes to the first env (reference batch_env.py:52-61 forwards every name; with worker
        processes the request goes to the first worker, like ExternalProcess.__getattr__, wrappers.py:343-356)."""
        if name in ("_envs", "_blocking", "_workers", "_shm", "_group", "_groups", "_shards", "_batch", "_broken"):   # not set yet: no recursion during __init__
            raise AttributeError(name)
        if self._blocking:
            return getattr(self._envs[0], name)
        if name == "observation_space":
            return self._observation_space
        if name == "action_space":
            return self._action_space
        w = self._workers[0]
        w.conn.send((_ATTRIBUTE, name))
        return w.receive(_VALUE)

    # ---------------------------------------------------------------------------------------------------------
    def _controller_call(self, resets):
        """Phase 2: pending resets, ONE upload, ONE rg_mpc_step, ONE download -- per shard, everything enqueued on every
        shard's stream before the first wait.  Returns the [B, 60] host action rows."""
        import contextlib
        for sh in self._shards:
            mine = [(b - sh.lo, t) for b, t in resets if sh.lo <= b < sh.hi]
            with (torch.cuda.stream(sh.stream) if sh.stream is not None else contextlib.nullcontext()):
                if mine:
                    sh.controller.reset_at([t for _, t in mine], [b for b, _ in mine])
                dev = sh.state.upload(with_clock=True, with_cmd=True)
                act = sh.controller.get_action(0.0, dev)      # per-robot clocks travel in dev["t_robot"]
                self._act_host[sh.lo:sh.hi].copy_(act, non_blocking=True)
        self.batched_calls += 1
        for sh in self._shards:
            if sh.controller.device.type == "cuda":
                (sh.stream if sh.stream is not None else torch.cuda.current_stream(sh.controller.device)).synchronize()
        return self._act_host.numpy()

    def step(self, action):
        """action: batch of per-env actions (whatever the envs' action_space holds, e.g. (vx, wz)).
        Returns stacked (observ, reward, done, info) like reference batch_env.py:63-93."""
        actions = action
        if self._broken is not None:
            raise RuntimeError(f"this MPCVecEnv is unusable: an earlier step() failed half-way ({self._broken}); build a new one")
        if len(actions) != self._batch:
            raise ValueError(f"expected {self._batch} actions, got {len(actions)}")
        for index, a in enumerate(actions):
            if not self._action_space.contains(a):
                raise ValueError("Invalid action at index {}: {}".format(index, a))
        # An exception from here on leaves the batch half-stepped -- some envs have run their pre-controller code or applied
        # their action, the batched controller may have advanced every robot, workers may be blocked waiting for the action
        # message -- so it is fatal for the whole wrapper: the workers are shut down and every later step() raises.
        try:
            return self._step_unchecked(actions)
        except BaseException as e:
            self._broken = f"{type(e).__name__}: {e}".splitlines()[0][:200]
            self._abort_workers()
            raise

    def _abort_workers(self):
        for w in self._workers:
            try:
                w.conn.send((_CLOSE, None))
            except (IOError, OSError, ValueError):
                pass
        for w in self._workers:
            w.process.join(timeout=2)
            if w.process.is_alive():
                w.process.terminate()

    def _step_unchecked(self, actions):
        if self._blocking:
            for g, sh in zip(self._groups, self._shards):
                g.pre(actions[sh.lo:sh.hi])
            rows = self._controller_call([r for g in self._groups for r in g.gather()])
            transitions = [tr for g, sh in zip(self._groups, self._shards) for tr in g.post(actions[sh.lo:sh.hi], rows[sh.lo:sh.hi])]
        else:
            try:
                batch_actions = np.asarray(actions)          # one contiguous block per worker instead of n small pickles
                if batch_actions.dtype == object or batch_actions.shape[0] != self._batch:
                    batch_actions = None
            except ValueError:
                batch_actions = None
            for w in self._workers:   # phase 1 of every slice runs concurrently
                w.conn.send((_STEP, batch_actions[w.lo:w.lo + w.n] if batch_actions is not None else [actions[i] for i in range(w.lo, w.lo + w.n)]))
            resets = [r for w in self._workers for r in w.receive(_READY)]
            for sh in self._shards:   # shared slab -> the shards' blocks of the pinned buffer (1.3 MB at B = 4096)
                sh.state.host_slab.numpy()[:] = self._shared_slab[:, sh.lo:sh.hi]
            self._shared_act[:] = self._controller_call(resets)
            for w in self._workers:   # ... and phase 3
                w.conn.send((_ACT, None))
            parts = [w.receive(_TRANSITION) for w in self._workers]
            return (np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts]), np.concatenate([p[2] for p in parts]),
                    tuple(i for p in parts for i in p[3]))
        observs, rewards, dones, infos = zip(*transitions)
        return np.stack(observs), np.stack(rewards), np.stack(dones), tuple(infos)

    def reset(self, indices=None):
        """Reset the envs `indices` (default all) and return their stacked observations (reference batch_env.py:95-109)."""
        if indices is None:
            indices = np.arange(self._batch)
        indices = [int(i) for i in indices]
        if self._blocking:
            got = {}
            for g, sh in zip(self._groups, self._shards):
                mine = [i for i in indices if sh.lo <= i < sh.hi]
                for i, o in zip(mine, g.reset([i - sh.lo for i in mine])):
                    got[i] = o
            return np.stack([got[i] for i in indices])
        per = {}
        for w in self._workers:   # non-blocking like the reference: every worker resets its share concurrently
            mine = [i - w.lo for i in indices if w.lo <= i < w.lo + w.n]
            if mine:
                w.conn.send((_RESET, mine))
                per[w] = mine
        got = {}
        for w, mine in per.items():
            for i, o in zip(mine, w.receive(_OBSERV)):
                got[w.lo + i] = o
        return np.stack([got[i] for i in indices])

    def close(self):
        """Close every sub-env (reference batch_env.py:111-115), the worker processes and the batched controller."""
        if self._blocking:
            for env in self._envs or []:
                if hasattr(env, "close"):
                    env.close()
        for w in self._workers:
            try:
                w.conn.send((_CLOSE, None))
                w.conn.close()
            except (IOError, OSError):
                pass
        for w in self._workers:
            w.process.join(timeout=10)
            if w.process.is_alive():
                w.process.terminate()
        self._workers = []
        self.__dict__.pop("_shared_slab", None)   # views into the shared memory must go before it can be closed
        self.__dict__.pop("_shared_act", None)
        for s in self._shm:
            try:
                s.close()
                s.unlink()
            except FileNotFoundError:
                pass
        self._shm = []
        for ctl in self.__dict__.get("controllers", []):
            ctl.close()
        self.__dict__["controllers"] = []

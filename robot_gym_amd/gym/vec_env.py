"""Batched VecEnv wrapper: B reference-style envs, ONE batched controller call per tick.

API of the reference's BatchEnv (agents/ppo/tools/batch_env.py:18-115): space-equality check at construction, `len`,
indexing, attribute forwarding to the first env (PPO reads `observation_space` / `action_space` through it),
`step(actions)` with `action_space.contains` validation -> stacked (observ, reward, done, info), `reset(indices)`,
`close()` closing every sub-env.  Per-env tick order of RobotGymEnv.step (gym/robot_gym_env.py:117-129): command ->
controller action -> ApplyStepAction -> (update_equip) -> observation / reward / termination.

Each sub-env keeps its OWN `step()`: task-level logic such as GoEnv.step's action clipping, on-target standing action
and camera hook (gym/envs/go_to/go_env.py:272-296) is not re-implemented here.  Every sub-env is built with
`BatchSlotController` as its controller class; a tick then runs in three phases:

  1. every env runs its step() up to the point where the reference asks the controller for its action -- the slot
     controller records the command (`update_controller_params`) and suspends the step (`get_action` raises
     StepSuspended, caught here).  Envs that provide the explicit two-half protocol
         pre_step(action, **kwargs) -> (command, kwargs)      everything before controller.update_controller_params
         post_step(motor_action, **kwargs) -> (obs, reward, done, info)   from simulation.ApplyStepAction on
     are driven through it instead, and their pre-controller code runs once rather than twice.
  2. the wrapper gathers every robot's state, its own clock and its command into ONE pinned slab, uploads it once, runs
     rg_mpc_step for all envs and downloads the [B, 60] action slab once.  Pending per-env resets (the env's
     Simulation.reset() -> controller.reset()) are applied first, each with that env's own clock value.
  3. every env's step() is entered again and now gets its action row; ApplyStepAction, observation, reward and
     termination are the env's own code.

Physics stays per-env on the CPU (PyBullet).  Clocks: robot b is stepped at env b's own GetTimeSinceReset() (per-robot
clock array of the C-ABI), exactly like B separate reference controllers -- a partial reset never shifts another env.
"""
import numpy as np
import torch

from robot_gym_amd.controllers.mpc.batched import BatchedMPCController, PackedState
from robot_gym_amd.controllers.mpc.slot_controller import BatchSlotController, StepSuspended


class MPCVecEnv:
    def __init__(self, envs, device=None, config=None, jacobian_fn=None):
        """envs: RobotGymEnv-like objects (`.simulation` with robot / controller / GetTimeSinceReset / ApplyStepAction,
        `step`, `reset`, `observation_space`, `action_space`) whose controller is a BatchSlotController."""
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
        self._slots = [env.simulation.controller for env in self._envs]
        for b, ctl in enumerate(self._slots):
            if not isinstance(ctl, BatchSlotController):
                raise TypeError(f"env {b}: simulation.controller is {type(ctl).__name__}; MPCVecEnv needs envs built with "
                                "controller_class=BatchSlotController (one slot of the batched GPU controller per env)")
        self._split = [hasattr(env, "pre_step") and hasattr(env, "post_step") for env in self._envs]
        self.cfg = config or self._slots[0].config
        self.controller = BatchedMPCController(B, self.cfg, device=device, extra_outputs=False)
        self._dev = self.controller.device
        self._jacobian_fn = jacobian_fn or (lambda env, leg: env.simulation.controller.kinematics_model.leg_jacobian(leg))
        pin = torch.cuda.is_available()
        self._state = PackedState(B, self._dev, pin)   # one pinned slab, one device slab, one copy per tick
        self._host = {n: t.numpy() for n, t in self._state.host.items()}
        self._clock = self._state.host_clock.numpy()
        self._cmd = self._state.host_cmd.numpy()
        self._offsets = np.array([self.cfg.vx_offset, self.cfg.vy_offset, self.cfg.wz_offset], dtype=np.float32)
        self._act_host = torch.zeros(B, 60, dtype=torch.float32, pin_memory=pin)
        self.batched_calls = 0

    def __len__(self):
        return len(self._envs)

    def __getitem__(self, index):
        return self._envs[index]

    def __getattr__(self, name):
        """Forward unimplemented attributes to the first env (reference batch_env.py:52-61)."""
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self._envs[0], name)

    # ---------------------------------------------------------------------------------------------------------
    def _apply_pending_resets(self):
        idx = [b for b, ctl in enumerate(self._slots) if ctl.reset_clock is not None]
        if idx:
            self.controller.reset_at([self._slots[b].reset_clock for b in idx], idx)
            for b in idx:
                self._slots[b].reset_clock = None

    def _gather(self):
        h = self._host
        kin0 = self.cfg.kin_mode == 0
        for b, env in enumerate(self._envs):
            rb = env.simulation.robot
            h["rpy"][:, b] = rb.GetBaseRollPitchYaw()
            h["rpy_rate"][:, b] = rb.GetBaseRollPitchYawRate()
            h["v_world"][:, b] = rb.GetBaseVelocity()
            h["quat"][:, b] = rb.GetTrueBaseOrientation()
            h["q"][:, b] = rb.GetMotorAngles()
            h["contact"][:, b] = np.asarray(rb.GetFootContacts(), dtype=np.int32)
            if kin0:
                h["foot_pos"][:, b] = np.asarray(rb.GetFootPositionsInBaseFrame()).reshape(12)
                h["jac"][:, b] = np.stack([self._jacobian_fn(env, leg) for leg in range(4)]).reshape(36)
            self._clock[b] = env.simulation.GetTimeSinceReset()
            # lin = [vx + VX_OFFSET, vy + VY_OFFSET, 0], ang = wz + WZ_OFFSET (reference mpc_controller.py:90-95), float32
            self._cmd[:, b] = np.asarray(self._slots[b].command, dtype=np.float32) + self._offsets
        return self._state.upload(with_clock=True, with_cmd=True)

    def step(self, action):
        """action: batch of per-env actions (whatever the envs' action_space holds, e.g. (vx, wz)).
        Returns stacked (observ, reward, done, info) like reference batch_env.py:63-93."""
        actions = action
        if len(actions) != len(self._envs):
            raise ValueError(f"expected {len(self._envs)} actions, got {len(actions)}")
        for index, (env, a) in enumerate(zip(self._envs, actions)):
            if not env.action_space.contains(a):
                raise ValueError("Invalid action at index {}: {}".format(index, a))
        # ---- phase 1: every env up to its controller call
        kwargs = [None] * len(self._envs)
        for b, (env, a) in enumerate(zip(self._envs, actions)):
            ctl = self._slots[b]
            if self._split[b]:
                command, kwargs[b] = env.pre_step(a)
                ctl.update_controller_params(command)
                continue
            ctl.phase = "capture"
            try:
                env.step(a)
            except StepSuspended:
                pass
            else:
                raise RuntimeError(f"env {b}: step() returned without asking its controller for an action")
            finally:
                ctl.phase = "idle"
        # ---- phase 2: ONE batched controller call
        self._apply_pending_resets()
        dev = self._gather()
        act = self.controller.get_action(0.0, dev)      # per-robot clocks travel in dev["t_robot"]
        self.batched_calls += 1
        self._act_host.copy_(act, non_blocking=True)
        if self._dev.type == "cuda":
            torch.cuda.current_stream(self._dev).synchronize()
        rows = self._act_host.numpy()
        # ---- phase 3: every env from its controller call on
        transitions = []
        for b, (env, a) in enumerate(zip(self._envs, actions)):
            ctl = self._slots[b]
            if self._split[b]:
                transitions.append(env.post_step(rows[b].copy(), **(kwargs[b] or {})))
                continue
            captured = ctl.command
            ctl.phase, ctl.action = "replay", rows[b]
            try:
                transitions.append(env.step(a))
                if ctl.command != captured:
                    raise RuntimeError(f"env {b}: step() derived a different command on re-entry ({ctl.command} vs {captured}); "
                                       "its pre-controller code is not repeatable -- give it pre_step/post_step")
            finally:
                ctl.phase, ctl.action = "idle", None
        observs, rewards, dones, infos = zip(*transitions)
        return np.stack(observs), np.stack(rewards), np.stack(dones), tuple(infos)

    def reset(self, indices=None):
        """Reset the envs `indices` (default all) and return their stacked observations (reference batch_env.py:95-109)."""
        if indices is None:
            indices = np.arange(len(self._envs))
        observs = []
        for index in indices:
            env, ctl = self._envs[index], self._slots[index]
            ctl.reset_clock = None
            observs.append(np.asarray(env.reset()))
            if ctl.reset_clock is None:   # the env did not route through Simulation.reset() -> controller.reset()
                ctl.reset()
        return np.stack(observs)

    def close(self):
        """Close every sub-env (reference batch_env.py:111-115) and the batched controller."""
        for env in self._envs:
            if hasattr(env, "close"):
                env.close()
        self.controller.close()

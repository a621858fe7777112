"""Batched MPC controller: B quadrupeds per call, HIP kernels underneath.

Host-side mirror of MPCController (reference controllers/mpc/mpc_controller.py:14-113) with
every per-robot scalar promoted to a [.., B] device tensor.  PyTorch-ROCm is used only for
device buffers and the current stream; all arithmetic happens in librg_mpc.so.
"""
import ctypes as C

import torch

from robot_gym_amd.core import mpc_abi
from robot_gym_amd.core.config import MPCConfig

STATE_FIELDS = (("rpy", 3, torch.float32), ("rpy_rate", 3, torch.float32), ("v_world", 3, torch.float32),
                ("quat", 4, torch.float32), ("q", 12, torch.float32), ("foot_pos", 12, torch.float32),
                ("jac", 36, torch.float32), ("contact", 4, torch.int32))


class PackedState:
    """Host-resident robot state staged for ONE upload per tick: every STATE_FIELDS array is a view into a single
    pinned [82, B] 32-bit host slab and a matching device slab -- rows 0-1 are the per-robot float64 clock
    (`t_robot`, first so that it stays 8-byte aligned for any B), then the fields in STATE_FIELDS order
    (`contact` is the int32 view of its rows), then the offset-corrected command (`cmd`, 3 rows) -- so the gym
    side pays one H2D copy instead of ten."""

    WORDS = 2 + sum(c for _, c, _ in STATE_FIELDS) + 3

    def __init__(self, batch, device, pin=None, host_storage=None):
        """host_storage: a flat float32 host tensor of WORDS * batch elements to use as the host slab (a slice of ONE pinned
        buffer shared by the shards of a multi-device MPCVecEnv) instead of allocating one."""
        words = self.WORDS
        pin = torch.cuda.is_available() if pin is None else pin
        if host_storage is not None:
            if host_storage.dtype != torch.float32 or host_storage.numel() != words * batch or not host_storage.is_contiguous():
                raise ValueError(f"host_storage must be a contiguous float32 tensor of {words * batch} elements")
            self.host_slab = host_storage.view(words, batch)
        else:
            self.host_slab = torch.zeros(words, batch, dtype=torch.float32, pin_memory=pin)
        self.dev_slab = torch.zeros(words, batch, dtype=torch.float32, device=device)
        self.host, self.dev = {}, {}
        # the two clock rows hold B float64 values back to back (not one value per column)
        self.host_clock = self.host_slab[0:2].view(-1).view(torch.float64)
        self.dev_clock = self.dev_slab[0:2].view(-1).view(torch.float64)
        row = 2
        for name, comps, dt in STATE_FIELDS:
            h, d = self.host_slab[row:row + comps], self.dev_slab[row:row + comps]
            self.host[name] = h if dt == torch.float32 else h.view(dt)
            self.dev[name] = d if dt == torch.float32 else d.view(dt)
            row += comps
        self.host_cmd, self.dev_cmd = self.host_slab[row:row + 3], self.dev_slab[row:row + 3]

    def upload(self, with_clock=False, with_cmd=False):
        """One host->device copy of the whole slab.  with_clock: the returned dict also carries the per-robot clock
        (`t_robot`), for envs that are not in lock-step; with_cmd: and the command rows (`cmd`, robot offsets already
        added by the caller)."""
        self.dev_slab.copy_(self.host_slab, non_blocking=True)
        if not (with_clock or with_cmd):
            return self.dev
        out = dict(self.dev)
        if with_clock:
            out["t_robot"] = self.dev_clock
        if with_cmd:
            out["cmd"] = self.dev_cmd
        return out


def command_with_offsets(params, offsets, batch):
    """(vx, wz) or (vx, vy, wz) per robot -> component-major [3,B] command with the robot's
    offsets added: lin = [vx + VX_OFFSET, vy + VY_OFFSET, 0], ang = wz + WZ_OFFSET
    (reference controllers/mpc/mpc_controller.py:83-95).  Works on any torch device."""
    p = params
    if p.dim() == 1:
        p = p.unsqueeze(0).expand(batch, -1)
    if p.shape[0] != batch or p.shape[1] not in (2, 3):
        raise ValueError(f"params must be [B,2] or [B,3], got {tuple(p.shape)}")
    if p.shape[1] == 2:
        vx, wz = p[:, 0], p[:, 1]
        vy = torch.zeros_like(vx)
    else:
        vx, vy, wz = p[:, 0], p[:, 1], p[:, 2]
    return torch.stack([vx, vy, wz], 0) + offsets.view(3, 1)


class BatchedMPCController:
    """update_controller_params / get_action / reset for a batch of robots on one GPU."""

    MOTOR_CONTROL_MODE = 3  # simple_motor.MOTOR_CONTROL_HYBRID, reference mpc_controller.py:16

    def __init__(self, batch, cfg: MPCConfig = None, device=None, extra_outputs=True):
        if not torch.cuda.is_available():
            raise RuntimeError("BatchedMPCController needs a HIP device (no CPU fallback)")
        self.cfg = cfg or MPCConfig.for_robot("ghost")
        self.batch = int(batch)
        dev = torch.device("cuda") if device is None else torch.device(device)
        if dev.type != "cuda":
            raise ValueError(f"BatchedMPCController runs on a HIP device, not {dev}")
        # normalised to an explicit index: state tensors report cuda:N, and 'cuda' != 'cuda:0' for torch.device
        self.device = torch.device("cuda", torch.cuda.current_device() if dev.index is None else dev.index)
        self._handle = mpc_abi.MpcHandle(self.cfg, self.batch, self.device.index)
        B, dev = self.batch, self.device
        self.cmd = torch.zeros(3, B, dtype=torch.float32, device=dev)
        self.action = torch.zeros(B, 60, dtype=torch.float32, device=dev)
        self.extra = {}
        if extra_outputs:
            self.extra = dict(
                grf=torch.zeros(B, 12, dtype=torch.float32, device=dev),
                tau_stance=torch.zeros(B, 12, dtype=torch.float32, device=dev),
                leg_state=torch.zeros(B, 4, dtype=torch.int32, device=dev),
                desired_state=torch.zeros(B, 4, dtype=torch.int32, device=dev),
                phase=torch.zeros(B, 4, dtype=torch.float32, device=dev),
                foot_target=torch.zeros(B, 12, dtype=torch.float32, device=dev),
                v_body=torch.zeros(B, 3, dtype=torch.float32, device=dev),
            )
        self._out = mpc_abi.COutPtrs()
        self._out.action = self.action.data_ptr()
        for k, v in self.extra.items():
            setattr(self._out, k, v.data_ptr())
        self._offsets = torch.tensor([self.cfg.vx_offset, self.cfg.vy_offset, self.cfg.wz_offset],
                                     dtype=torch.float32, device=dev).view(3, 1)
        self._set_command_called = False

    # -- stream plumbing -------------------------------------------------------------
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # -- plugin surface, batched -----------------------------------------------------
    def update_controller_params(self, params):
        """params: [B,2] (vx, wz) or [B,3] (vx, vy, wz) -- reference mpc_controller.py:83-100."""
        p = torch.as_tensor(params, dtype=torch.float32, device=self.device)
        self.cmd.copy_(command_with_offsets(p, self._offsets, self.batch))
        self._handle.set_command(self.cmd.data_ptr(), self._stream())
        self._set_command_called = True

    def set_raw_command(self, cmd_3xB):
        """cmd already offset-corrected, component-major [3,B]."""
        self.cmd.copy_(cmd_3xB)
        self._handle.set_command(self.cmd.data_ptr(), self._stream())
        self._set_command_called = True

    def get_action(self, t, state):
        """state: dict of component-major device tensors (STATE_FIELDS; optional `contact_sched` int32 [4,B] and
        `t_robot` float64 [B], the per-robot clock that replaces the scalar t).  Returns action [B,60]
        (device tensor, overwritten by the next call) -- reference mpc_controller.py:102-106."""
        sp = mpc_abi.CStatePtrs()
        for name, comps, dt in STATE_FIELDS:
            tsr = state.get(name)
            if tsr is None:
                if name in ("foot_pos", "jac") and self.cfg.kin_mode == 1:
                    continue
                raise KeyError(f"state is missing {name!r}")
            if tsr.dtype != dt or tuple(tsr.shape) != (comps, self.batch) or not tsr.is_contiguous() or tsr.device != self.device:
                raise ValueError(f"state[{name!r}] must be contiguous {dt} [{comps},{self.batch}] on {self.device}")
            setattr(sp, name, tsr.data_ptr())
        cmd = state.get("cmd")   # offset-corrected [3,B] command travelling with the state; None = update_controller_params' copy
        if cmd is not None:
            if cmd.dtype != torch.float32 or tuple(cmd.shape) != (3, self.batch) or not cmd.is_contiguous() or cmd.device != self.device:
                raise ValueError(f"state['cmd'] must be contiguous float32 [3,{self.batch}] on {self.device}")
            sp.cmd = cmd.data_ptr()
        t_robot = state.get("t_robot")
        if t_robot is not None:
            if t_robot.dtype != torch.float64 or tuple(t_robot.shape) != (self.batch,) or not t_robot.is_contiguous() or t_robot.device != self.device:
                raise ValueError(f"state['t_robot'] must be contiguous float64 [{self.batch}] on {self.device}")
            sp.t_robot = t_robot.data_ptr()
        sched = state.get("contact_sched")
        if sched is not None:
            if sched.dtype != torch.int32 or tuple(sched.shape) != (4, self.batch) or not sched.is_contiguous() or sched.device != self.device:
                raise ValueError(f"state['contact_sched'] must be contiguous int32 [4,{self.batch}] on {self.device}")
            sp.contact_sched = sched.data_ptr()
        self._handle.step(t, sp, self._out, self._stream())
        return self.action

    def bind_host_state(self, packed: "PackedState", action_host, zero_copy=False):
        """Prepare the host-resident fast path (get_action_host): validate ONCE that `packed` (a PackedState of this batch
        on this device, pinned) and `action_host` (pinned float32 [B,60]) fit, and build the pointer structs the per-tick
        call reuses -- the per-tick path then is one call across the C-ABI (rg_mpc_step_host).
        zero_copy: the kernels read the state straight from the pinned host slab and write the action row straight into
        `action_host` (pinned host memory is device-accessible through its own address on ROCm): no upload, no download --
        for a batch of one or a few robots the two copy operations cost more stream time (~6 us each) than the 568 bytes
        take over PCIe inside the kernels.  The tick is then rg_mpc_step + a stream synchronisation."""
        if packed.dev_slab.device != self.device or tuple(packed.dev_slab.shape[1:]) != (self.batch,) or not packed.host_slab.is_pinned():
            raise ValueError("bind_host_state: PackedState of another batch / device, or not pinned")
        if action_host.dtype != torch.float32 or tuple(action_host.shape) != (self.batch, 60) or not action_host.is_pinned() or not action_host.is_contiguous():
            raise ValueError(f"bind_host_state: action_host must be a pinned contiguous float32 [{self.batch},60] tensor")
        sp = mpc_abi.CStatePtrs()
        for name, comps, dt in STATE_FIELDS:
            if name in ("foot_pos", "jac") and self.cfg.kin_mode == 1:
                continue
            setattr(sp, name, packed.dev[name].data_ptr())
        self._host_bound = (sp, packed, action_host, packed.host_slab.data_ptr(), packed.dev_slab.data_ptr(),
                            packed.host_slab.numel() * packed.host_slab.element_size(), action_host.data_ptr())
        self._zero_copy = None
        if zero_copy:
            spz = mpc_abi.CStatePtrs()
            for name, comps, dt in STATE_FIELDS:
                if name in ("foot_pos", "jac") and self.cfg.kin_mode == 1:
                    continue
                setattr(spz, name, packed.host[name].data_ptr())
            outz = mpc_abi.COutPtrs()
            outz.action = action_host.data_ptr()
            for k, v in self.extra.items():
                setattr(outz, k, v.data_ptr())
            self._zero_copy = (spz, outz)

    def get_action_host(self, t):
        """One tick from the bound host slab: upload, step, download, wait (rg_mpc_step_host).  Returns the bound pinned
        action tensor (overwritten by the next call)."""
        if getattr(self, "_host_bound", None) is None:
            raise RuntimeError("get_action_host: call bind_host_state(packed_state, pinned_action) first")
        sp, packed, action_host, hptr, dptr, nbytes, aptr = self._host_bound
        if self._zero_copy is not None:
            spz, outz = self._zero_copy
            self._handle.step(t, spz, outz, self._stream())
            torch.cuda.current_stream(self.device).synchronize()
            return action_host
        self._handle.step_host(t, hptr, dptr, nbytes, sp, self._out, aptr, self._stream())
        return action_host

    def set_gait(self, stance_duration=None, duty_factor=None, init_phase=None, init_state=None):
        """Per-robot gait timing, [4,B] each (float64; init_state int32, optional): the OpenloopGaitGenerator arguments of
        reference mpc_controller.py:30-35, one row per robot.  All None returns to the config-wide gait.  Call before reset()."""
        if stance_duration is None and duty_factor is None and init_phase is None and init_state is None:
            self._handle.set_gait(None, None, None, None, self._stream())
            return
        f = lambda a: torch.as_tensor(a, dtype=torch.float64).to(self.device).contiguous()
        sd, du, ph = f(stance_duration), f(duty_factor), f(init_phase)
        ist = None if init_state is None else torch.as_tensor(init_state, dtype=torch.int32).to(self.device).contiguous()
        for a in (sd, du, ph) + (() if ist is None else (ist,)):
            if tuple(a.shape) != (4, self.batch):
                raise ValueError(f"gait arrays must be [4,{self.batch}]")
        self._handle.set_gait(sd.data_ptr(), du.data_ptr(), ph.data_ptr(), None if ist is None else ist.data_ptr(), self._stream())
        torch.cuda.current_stream(self.device).synchronize()   # the library copied from these temporaries

    def reset(self, idx=None, t0=0.0):
        """LocomotionController.reset for robots idx (None = all) -- reference mpc_controller.py:108-109."""
        if idx is not None:
            idx = [int(i) for i in (idx.tolist() if hasattr(idx, "tolist") else idx)]
        self._handle.reset(idx, t0, self._stream())

    def reset_at(self, t0s, idx=None):
        """Per-robot reset clock values (phase offsets between sub-envs)."""
        self._handle.reset_at(list(t0s), None if idx is None else list(idx), self._stream())

    def hybrid_to_torque(self, action, q, qd, out=None):
        """Motor model, HYBRID branch (reference model/robots/simple_motor.py:128-140).
        q, qd [12,B] -> tau [B,12]; or, for the S sub-steps of one control tick (the action-repeat loop of reference
        core/simulation.py:175-179), q, qd [S,12,B] -> tau [S,B,12] in one launch."""
        if q.shape != qd.shape or q.dtype != torch.float32 or qd.dtype != torch.float32 or not (q.is_contiguous() and qd.is_contiguous()):
            raise ValueError("q and qd must be contiguous float32 tensors of one shape")
        if q.device != self.device or qd.device != self.device:
            raise ValueError(f"q and qd must live on {self.device}")
        # the kernel reads action[b * 60 + 5 j ..] and writes out[(s * B + b) * 12 + j] through raw pointers: a wrong shape,
        # dtype, stride or device here is an out-of-bounds device access, not an exception -- so check
        if not torch.is_tensor(action) or action.dtype != torch.float32 or tuple(action.shape) != (self.batch, 60) or not action.is_contiguous() or action.device != self.device:
            raise ValueError(f"action must be a contiguous float32 [{self.batch},60] tensor on {self.device}")
        if q.dim() == 2 and tuple(q.shape) == (12, self.batch):
            steps, shape = None, (self.batch, 12)
        elif q.dim() == 3 and tuple(q.shape[1:]) == (12, self.batch):
            steps, shape = int(q.shape[0]), (int(q.shape[0]), self.batch, 12)
        else:
            raise ValueError(f"q must be [12,{self.batch}] or [S,12,{self.batch}]")
        if out is None:
            out = torch.empty(*shape, dtype=torch.float32, device=self.device)
        elif not torch.is_tensor(out) or out.dtype != torch.float32 or tuple(out.shape) != shape or not out.is_contiguous() or out.device != self.device:
            raise ValueError(f"out must be a contiguous float32 {list(shape)} tensor on {self.device}")
        self._handle.hybrid_to_torque(action.data_ptr(), q.data_ptr(), qd.data_ptr(), out.data_ptr(), self._stream(), substeps=steps)
        return out

    def solver_stats(self):
        return self._handle.last_solver_stats(self._stream())

    def audit_stats(self, reset=False):
        """Audit lane counters (core/mpc_abi.MpcHandle.audit_stats)."""
        return self._handle.audit_stats(reset, self._stream())

    def bin_counts(self):
        return self._handle.last_bin_counts(self._stream())

    @staticmethod
    def get_standing_action():
        return 0., 0.  # reference mpc_controller.py:111-113

    def close(self):
        self._handle.close()

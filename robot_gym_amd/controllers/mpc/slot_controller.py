"""`BatchSlotController`: the per-env face of ONE batched GPU controller.

A vectorised env (robot_gym_amd/gym/vec_env.py) steps B reference-style envs with a single `rg_mpc_step`.  Each
sub-env still owns a controller object, because the reference wires one into every `Simulation`
(core/simulation.py:113-127: `controller_class.MOTOR_CONTROL_MODE`, `controller_class(robot, GetTimeSinceReset)`,
`controller.reset()`) and its `Robot` reaches the kinematics callbacks through it
(model/robots/robot.py:94-102 -> `simulation.controller.kinematics_model`).  This class has that plugin surface
(reference controllers/controller.py:4-28, controllers/mpc/mpc_controller.py:14-113) but holds NO GPU handle: it
records the command and the reset, and hands out the action row the vectorised env computed for its slot.  Register it
next to "mpc" in the reference's util/cli/mapper.py:7-9 to build envs for `MPCVecEnv`.
"""
import numpy as np

from robot_gym_amd.controllers.controller import Controller
from robot_gym_amd.controllers.mpc.kinematics import ChainKinematics, PybulletKinematics
from robot_gym_amd.controllers.mpc.mpc_controller import MOTOR_CONTROL_HYBRID, config_from_robot


class StepSuspended(Exception):
    """Raised by get_action() while the vectorised env is collecting commands: the env's own step() is unwound at the
    exact point where the reference asks the controller for its action (gym/robot_gym_env.py:121), and re-entered once
    the batched action exists."""


class BatchSlotController(Controller):
    MOTOR_CONTROL_MODE = MOTOR_CONTROL_HYBRID   # read before construction, reference core/simulation.py:113

    def __init__(self, robot, get_time_since_reset, config=None):
        super().__init__(robot, get_time_since_reset)
        self.config = config or config_from_robot(robot)
        self._kinematics = PybulletKinematics(robot, ChainKinematics(self.config))   # host-side, no GPU handle
        self.command = (0.0, 0.0, 0.0)      # (vx, vy, wz) BEFORE the robot offsets (the batch adds them on the device)
        self.reset_clock = None             # clock value of a reset the batch has not applied yet
        self.phase = "idle"                 # "capture": get_action suspends the env's step; "replay": returns `action`
        self.action = None
        self._captured = None
        self.reset()

    @property
    def kinematics_model(self):
        return self._kinematics

    @staticmethod
    def setup_ui_params(pybullet_client):
        return tuple(pybullet_client.addUserDebugParameter(n, -2., 2., 0.) for n in ("Vx", "Vy", "Wz"))

    @staticmethod
    def read_ui_params(pybullet_client, ui):
        return tuple(pybullet_client.readUserDebugParameter(i) for i in ui)

    @staticmethod
    def get_standing_action():
        return 0., 0.   # reference mpc_controller.py:111-113

    def update_controller_params(self, params):
        """(vx, wz) or (vx, vy, wz), reference mpc_controller.py:83-88."""
        p = [float(x) for x in params]
        if len(p) == 2:
            self.command = (p[0], 0.0, p[1])
        elif len(p) == 3:
            self.command = (p[0], p[1], p[2])
        else:
            raise ValueError("params must be (vx, wz) or (vx, vy, wz)")

    def begin_replay(self, action_row):
        """Second pass of a two-pass tick (envs without pre_step / post_step): get_action() will hand out `action_row`,
        after checking that this pass derived the same command as the first one did."""
        self._captured = self.command
        self.phase, self.action = "replay", action_row

    def get_action(self):
        if self.phase == "capture":
            raise StepSuspended()
        if self.phase == "replay" and self.action is not None:
            # checked HERE, where the reference asks for the action (gym/robot_gym_env.py:121) and before the env applies
            # anything: a step() whose pre-controller code is not repeatable must not reach ApplyStepAction
            if self.command != self._captured:
                raise RuntimeError(f"step() derived a different command on re-entry ({self.command} vs {self._captured}): its "
                                   "pre-controller code is not repeatable -- give the env pre_step / post_step (robot_gym_amd/gym/split_step.py)")
            return np.array(self.action, dtype=np.float32)
        raise RuntimeError("BatchSlotController.get_action() outside MPCVecEnv.step(): this controller is one slot of a "
                           "batched GPU controller and has no action of its own (use MPCController for a single env)")

    def reset(self):
        """LocomotionController.reset (reference mpc_controller.py:108-109): remember the clock value; the batch applies
        it to this slot (rg_mpc_reset_at) before its next step."""
        self.reset_clock = float(self.get_time_since_reset())

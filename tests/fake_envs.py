"""Stand-ins for the reference's PyBullet envs (no pybullet in any image): a stub Robot serving one column of a synthetic
state batch through the reference's getter names, and fake envs whose step() has the reference's statement order --
RobotGymEnv.step (gym/robot_gym_env.py:117-129) and GoEnv.step's pre-processing (gym/envs/go_to/go_env.py:272-296:
clip the action, replace it with the controller's standing action when on target, update_equip hook)."""
import types

import numpy as np


class Box:
    """Minimal gym.spaces.Box (gym is not installed): bounds, equality, contains."""

    def __init__(self, low, high):
        self.low, self.high = np.asarray(low, dtype=np.float32), np.asarray(high, dtype=np.float32)

    def __eq__(self, other):
        return isinstance(other, Box) and np.array_equal(self.low, other.low) and np.array_equal(self.high, other.high)

    def contains(self, x):
        x = np.asarray(x, dtype=np.float32)
        return x.shape == self.low.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))


class StubRobot:
    """Serves one column of a synthetic state batch through the reference's Robot getter names."""

    def __init__(self, cfg, state, b):
        self.cfg, self.state, self.b = cfg, state, b
        self.contact = np.ones(4, dtype=bool)
        self.pybullet_client = self
        self.GetRobotId = 1
        self.GetFootLinkIds = [10, 11, 12, 13]
        self.equipment_updates = 0

    # reference model/robots/robot.py getters
    def GetBaseRollPitchYaw(self): return self.state["rpy"][:, self.b]
    def GetBaseRollPitchYawRate(self): return self.state["rpy_rate"][:, self.b]
    def GetBaseVelocity(self): return self.state["v_world"][:, self.b]
    def GetTrueBaseOrientation(self): return self.state["quat"][:, self.b]
    def GetMotorAngles(self): return self.state["q"][:, self.b]
    def GetFootPositionsInBaseFrame(self): return self.state["foot_pos"][:, self.b].reshape(4, 3)
    def GetFootContacts(self): return list(self.contact)
    def update_equipment(self): self.equipment_updates += 1
    @property
    def GetJointStates(self): return [(float(a), 0.0) for a in self.state["q"][:, self.b]]

    def calculateJacobian(self, robot_id, link_id, local, q, qd, qdd):
        leg = link_id - 10
        jv = np.zeros((3, 18))
        jv[:, 6 + 3 * leg:9 + 3 * leg] = self.state["jac"][:, self.b].reshape(4, 3, 3)[leg]
        return jv.tolist(), None

    def GetCtrlConstants(self):
        c = self.cfg
        return types.SimpleNamespace(MPC_BODY_MASS=c.mass, MPC_BODY_INERTIA=c.inertia, MPC_BODY_HEIGHT=c.body_height,
                                     STANCE_DURATION_SECONDS=list(c.stance_duration), DUTY_FACTOR=list(c.duty_factor),
                                     INIT_PHASE_FULL_CYCLE=list(c.init_phase), INIT_LEG_STATE=c.init_state,
                                     VX_OFFSET=c.vx_offset, VY_OFFSET=c.vy_offset, WZ_OFFSET=c.wz_offset)

    def GetConstants(self):
        return types.SimpleNamespace(DEFAULT_HIP_POSITIONS=np.array(self.cfg.hip).reshape(4, 3).tolist(), NUM_LEG=4)

    def GetMotorConstants(self):
        c = self.cfg
        return types.SimpleNamespace(MOTOR_POSITION_GAINS=list(c.motor_kp), MOTOR_VELOCITY_GAINS=np.array(c.motor_kd),
                                     MOTOR_DIRECTION=np.array(c.motor_dir), MOTOR_OFFSET=np.array(c.motor_off), NUM_MOTORS=12)


class FakeSimulation:
    """The slice of reference core/simulation.py the controller path touches: robot, controller, clock, ApplyStepAction."""
    ACTION_REPEAT, TIME_STEP = 10, 0.001   # core/sim_constants.py:7,11

    def __init__(self, robot, controller_class, **controller_kwargs):
        self.robot = robot
        self._step_counter = 0
        self.applied = []
        assert controller_class.MOTOR_CONTROL_MODE == 3          # read before construction, simulation.py:113
        self.controller = controller_class(robot, self.GetTimeSinceReset, **controller_kwargs)   # simulation.py:117
        self.reset()

    def GetTimeSinceReset(self):
        return self._step_counter * self.TIME_STEP               # simulation.py:141-142

    def ApplyStepAction(self, action):
        self.applied.append(np.array(action))
        self._step_counter += self.ACTION_REPEAT                 # simulation.py:175-179

    def reset(self):
        self._step_counter = 0
        self.controller.reset()                                  # simulation.py:123-127


class FakeRobotGymEnv:
    """step() in the statement order of reference gym/robot_gym_env.py:117-129."""

    def __init__(self, cfg, state, b, controller_class, **controller_kwargs):
        self.simulation = FakeSimulation(StubRobot(cfg, state, b), controller_class, **controller_kwargs)
        self.observation_space = Box([-np.inf] * 2, [np.inf] * 2)
        self.action_space = Box([-2.0, -2.0, -2.0], [2.0, 2.0, 2.0])
        self.pre_controller_runs = 0
        self.closed = False

    def step(self, action, **kwargs):
        self.pre_controller_runs += 1
        self.simulation.controller.update_controller_params(action)
        action = self.simulation.controller.get_action()
        self.simulation.ApplyStepAction(action)
        if "update_equip" in kwargs:
            self.simulation.robot.update_equipment()
        return np.array(self.get_observation()), self.reward(), *self.termination()

    def get_observation(self): return [self.simulation.GetTimeSinceReset(), float(len(self.simulation.applied))]
    def reward(self): return 1.0
    def termination(self): return False, {}

    def reset(self):
        self.simulation.reset()
        return self.get_observation()

    def close(self):
        self.closed = True


class FakeGoEnv(FakeRobotGymEnv):
    """The pre-processing of reference gym/envs/go_to/go_env.py:272-296 (debug/agent-feedback branch): clip (vx, wz),
    standing action when on target, update_equip kwarg when a camera follows -- then the parent's step."""

    def __init__(self, *args, on_target=False, follow_camera=False, **kw):
        super().__init__(*args, **kw)
        self.action_space = Box([-1.0, -1.0], [1.0, 1.0])       # go_env.py:101-103 (2-d action)
        self.on_target, self.follow_camera = on_target, follow_camera

    def step(self, action, **kwargs):
        action = max(0, min(action[0], 0.35)), max(-0.4, min(action[1], 0.4))       # go_env.py:280
        if self.follow_camera:
            kwargs = {"update_equip": True}                                          # go_env.py:289
        if self.on_target:
            action = self.simulation.controller.get_standing_action()               # go_env.py:291-292
        return super().step(action, **kwargs)


class SplitGoEnv(FakeGoEnv):
    """The same env offering the explicit two-half protocol of MPCVecEnv (what a maintainer's refactor of step() looks like)."""

    def pre_step(self, action, **kwargs):
        self.pre_controller_runs += 1
        action = max(0, min(action[0], 0.35)), max(-0.4, min(action[1], 0.4))
        if self.follow_camera:
            kwargs = {"update_equip": True}
        if self.on_target:
            action = self.simulation.controller.get_standing_action()
        return action, kwargs

    def post_step(self, motor_action, **kwargs):
        self.simulation.ApplyStepAction(motor_action)
        if "update_equip" in kwargs:
            self.simulation.robot.update_equipment()
        return np.array(self.get_observation()), self.reward(), *self.termination()


def make_fake_env(kind, robot, seed, batch, b, **kw):
    """Picklable env constructor for MPCVecEnv(blocking=False): runs INSIDE a worker process (the reference's
    ExternalProcess takes such a `constructor`, agents/ppo/tools/wrappers.py:306-327) and rebuilds the seeded synthetic
    state there."""
    from robot_gym_amd.core.config import MPCConfig
    from robot_gym_amd import synthetic
    from robot_gym_amd.controllers.mpc.slot_controller import BatchSlotController
    cfg = MPCConfig.for_robot(robot)
    state, _, _ = synthetic.make_states(batch, cfg, seed=seed)
    cls = {"go": FakeGoEnv, "split": SplitGoEnv, "base": FakeRobotGymEnv}[kind]
    return cls(cfg, state, b, BatchSlotController, **kw)

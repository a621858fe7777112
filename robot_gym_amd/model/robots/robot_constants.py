"""Per-robot constants the MPC path is configured from (data only).

Values restate the reference modules cited per field:
  ctrl   robot_gym/model/robots/<robot>/ctrl_constants.py:8-41
  motor  robot_gym/model/robots/<robot>/motor_constants.py:5-19
  geom   robot_gym/model/robots/<robot>/constants.py:4-43
  chain  util/pybullet_data/robots/<robot>.urdf via tools/extract_urdf_chain.py -> chain.json
"""
import json
import os
from dataclasses import dataclass, field
from typing import Tuple


from robot_gym_amd.controllers.mpc.gait import LegState

_HERE = os.path.dirname(os.path.abspath(__file__))


@dataclass(frozen=True)
class RobotConstants:
    name: str
    # ctrl_constants.py
    mpc_body_mass: float
    mpc_body_inertia: Tuple[float, ...]
    mpc_body_height: float
    stance_duration_seconds: Tuple[float, ...]
    duty_factor: Tuple[float, ...]
    init_phase_full_cycle: Tuple[float, ...]
    init_leg_state: Tuple[int, ...]
    vx_offset: float
    vy_offset: float
    wz_offset: float
    # constants.py
    default_hip_positions: Tuple[Tuple[float, float, float], ...]
    init_motor_angles: Tuple[float, ...]
    start_pos: Tuple[float, float, float]
    # motor_constants.py
    num_motors: int = 12
    motor_position_gains: Tuple[float, ...] = (220.0,) * 12
    motor_velocity_gains: Tuple[float, ...] = (1.0, 2.0, 2.0) * 4
    motor_direction: Tuple[float, ...] = (1.0,) * 12
    motor_offset: Tuple[float, ...] = (0.0,) * 12
    chain: dict = field(default_factory=dict, compare=False, hash=False)


def _chain(name):
    with open(os.path.join(_HERE, name, "chain.json")) as f:
        return json.load(f)


_TROT_STATE = (LegState.SWING, LegState.STANCE, LegState.STANCE, LegState.SWING)

GHOST = RobotConstants(
    name="ghost",
    mpc_body_mass=190 / 9.8,                                          # ghost/ctrl_constants.py:8
    mpc_body_inertia=(0.07335, 0, 0, 0, 0.25068, 0, 0, 0, 0.25447),   # :9
    mpc_body_height=0.42,                                             # :10
    stance_duration_seconds=(0.3,) * 4,                               # :13
    duty_factor=(0.6,) * 4,                                           # :28
    init_phase_full_cycle=(0.9, 0, 0, 0.9),                           # :29
    init_leg_state=_TROT_STATE,                                       # :32-37
    vx_offset=0.0, vy_offset=0.08, wz_offset=-0.025,                  # :39-41
    default_hip_positions=((0.22, -0.1, 0), (0.22, 0.1, 0), (-0.22, -0.1, 0), (-0.22, 0.1, 0)),  # ghost/constants.py:31-36
    init_motor_angles=(0, 0.67, -1.25) * 4,                           # ghost/constants.py:8-17
    start_pos=(0, 0, 0.48),                                           # ghost/constants.py:5
    chain=_chain("ghost"),
)

K3LSO = RobotConstants(
    name="k3lso",
    mpc_body_mass=190 / 9.8,                                          # k3lso/ctrl_constants.py:8
    mpc_body_inertia=(0.07335, 0, 0, 0, 0.25068, 0, 0, 0, 0.25447),   # :10
    mpc_body_height=0.38,                                             # :11
    stance_duration_seconds=(0.3,) * 4,
    duty_factor=(0.6,) * 4,
    init_phase_full_cycle=(0.9, 0, 0, 0.9),
    init_leg_state=_TROT_STATE,
    vx_offset=0.0, vy_offset=0.0, wz_offset=0.0,                      # k3lso/ctrl_constants.py:39-41
    default_hip_positions=((0.22, -0.105, 0), (0.22, 0.105, 0), (-0.22, -0.105, 0), (-0.22, 0.105, 0)),  # k3lso/constants.py:32-37
    init_motor_angles=(0, 0.67, -1.25, -0, 0.67, 1.25, 0, -0.67, -1.25, 0, -0.67, 1.25),  # k3lso/constants.py:12-18
    start_pos=(0, 0, 0.48),
    chain=_chain("k3lso"),
)

ROBOTS = {"ghost": GHOST, "k3lso": K3LSO}

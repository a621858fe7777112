"""Leg-state enum of the open-loop gait generator.

Mirrors `mpc_controller.gait_generator.LegState` of motion_imitation==0.0.5 (not in the
reference tree); the reference uses it at model/robots/ghost/ctrl_constants.py:3,32-37.
"""
import enum


class LegState(enum.IntEnum):
    SWING = 0
    STANCE = 1
    EARLY_CONTACT = 2
    LOSE_CONTACT = 3

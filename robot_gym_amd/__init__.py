"""robot_gym_amd -- MI355X-native batched convex-MPC gait controller for robot-gym.

Holds only what the hot path needs:
  csrc/         HIP kernels + the C-ABI shared library (include/rg_mpc.h)
  core/         ctypes shim over the C-ABI, configuration
  controllers/  host-side mirror of robot_gym.controllers (plugin surface)
  gym/          batched VecEnv wrapper
  model/robots/ per-robot constants (data) the controller is configured from
"""
__version__ = "0.1.0"

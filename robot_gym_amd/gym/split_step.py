"""The two halves of the reference's env step() as mixins, for envs stepped by MPCVecEnv.

MPCVecEnv makes ONE batched controller call per tick, in the middle of every env's step.  An env that offers
    pre_step(action, **kwargs) -> (command, kwargs)               everything before controller.update_controller_params
    post_step(motor_action, **kwargs) -> (obs, reward, done, info)  from simulation.ApplyStepAction on
runs its pre-controller code once per tick; an env that does not is stepped twice (robot_gym_amd/gym/vec_env.py), which
repeats whatever that code does besides deriving the command -- GoEnv(show_plot=True)._update_plot, the camera writes of
parse_equipment_ui_params, RNG draws of a user env.  Use:

    class BatchedRobotGymEnv(RobotGymEnvSplitStep, RobotGymEnv): pass
    class BatchedGoEnv(GoEnvSplitStep, GoEnv): pass

The bodies restate the statement order of the reference methods they split (cited per line); step() itself is untouched, so
the same class still works alone with the batch-1 MPCController.
"""
import numpy as np


class RobotGymEnvSplitStep:
    """reference gym/robot_gym_env.py:117-129, cut at the controller call (:120-121)."""

    def pre_step(self, action, **kwargs):
        return action, kwargs

    def post_step(self, motor_action, **kwargs):
        self._simulation.ApplyStepAction(motor_action)                 # :122
        if "update_equip" in kwargs:
            self._simulation.robot.update_equipment()                  # :123-124
        observation = self.get_observation()                           # :125
        reward = self.reward()                                         # :126
        done, info = self.termination()                                # :127
        return np.array(observation), reward, done, info               # :129


class GoEnvSplitStep(RobotGymEnvSplitStep):
    """reference gym/envs/go_to/go_env.py:272-296 in front of RobotGymEnv.step."""

    def pre_step(self, action, **kwargs):
        if self._debug and not self.simulation.read_ui_parameters(self._ui) and not self._policy:
            action = self._read_inputs()                               # :273-275 UI input overrides the agent
        if self._debug and (self.simulation.read_ui_parameters(self._ui) or self._policy):
            action = max(0, min(action[0], 0.35)), max(-0.4, min(action[1], 0.4))   # :278-280
        if self._debug and self.parse_equipment_ui_params():           # :283-289 follower camera
            pos_x, pos_y = self._follower.cam_pos_point.get_xy()
            target_x, target_y = self._follower.cam_target_point.get_xy()
            self.simulation.robot.get_default_camera().position = pos_x, pos_y, 0.095
            self.simulation.robot.get_default_camera().target = target_x, target_y, 0.0
            kwargs = {"update_equip": True}
        if self._on_target():
            action = self.simulation.controller.get_standing_action()  # :291-292
        if self._show_plot:
            self._update_plot()                                        # :294-295 -- once per tick
        return action, kwargs

"""Plugin surface every locomotion controller implements.

Same contract as the reference's robot_gym/controllers/controller.py:4-28
(__init__(robot, get_time_since_reset); update_controller_params; get_action;
setup_ui_params; read_ui_params; reset) so a class written against either is
interchangeable inside core/simulation.py:113-127 of the reference.
"""
import abc


class Controller(abc.ABC):
    MOTOR_CONTROL_MODE = None  # read before construction, reference core/simulation.py:113

    def __init__(self, robot, get_time_since_reset):
        self._robot = robot
        self.get_time_since_reset = get_time_since_reset

    @abc.abstractmethod
    def update_controller_params(self, params):
        ...

    @abc.abstractmethod
    def get_action(self):
        ...

    @abc.abstractmethod
    def setup_ui_params(self, pybullet_client):
        ...

    @abc.abstractmethod
    def read_ui_params(self, pybullet_client, ui):
        ...

    @abc.abstractmethod
    def reset(self):
        ...

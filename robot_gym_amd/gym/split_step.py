"""One-pass stepping for envs inside MPCVecEnv, without restating any env code.

MPCVecEnv makes ONE batched controller call per tick, in the middle of every env's step().  The reference's task envs end
their own step() in `super(TaskEnv, self).step(action, **kwargs)` (gym/envs/go_to/go_env.py:296 -> RobotGymEnv.step,
gym/robot_gym_env.py:117-129, which is where the controller is asked for its action).  `one_pass(TaskEnv, BaseEnv)` builds

    class _Intercept(BaseEnv): ...            # sits AFTER the task env in the MRO
    class OnePassTaskEnv(TaskEnv, _Intercept)

so that the task env's pre-controller code (action clipping, standing action on target, camera hooks, plots -- whatever it
is, none of it is known here) runs exactly ONCE per tick: the call it ends in lands in `_Intercept.step`, which records
`(action, kwargs)`, hands the command to the env's slot controller and suspends (phase 1); phase 3 calls `BaseEnv.step` itself
with the recorded arguments and the batched action row.  Outside MPCVecEnv (a single env with the batch-1 MPCController)
the interceptor is a pass-through.

    BatchedGoEnv = one_pass(GoEnv, RobotGymEnv)
    envs = [BatchedGoEnv(..., controller_class=BatchSlotController) for _ in range(B)]

An env class that is stepped without this (or without its own pre_step / post_step pair) goes through MPCVecEnv's two-pass
path: its whole step() runs twice, which is only right when the code in front of the controller call is repeatable.
"""
from robot_gym_amd.controllers.mpc.slot_controller import BatchSlotController, StepSuspended


def _slot_of(env):
    sim = getattr(env, "simulation", None) or getattr(env, "_simulation", None)
    ctl = getattr(sim, "controller", None)
    return ctl if isinstance(ctl, BatchSlotController) else None


def one_pass(task_env_cls, base_env_cls=None):
    """Subclass of `task_env_cls` whose step() suspends at `base_env_cls.step` -- the class whose step() asks the controller
    for its action; default: the task env's direct base -- while MPCVecEnv collects commands, and resumes from there."""
    if base_env_cls is None:
        base_env_cls = task_env_cls.__mro__[1]
    if not (isinstance(base_env_cls, type) and issubclass(task_env_cls, base_env_cls) and "step" in vars(base_env_cls)):
        raise TypeError(f"{base_env_cls!r} must be a base of {task_env_cls.__name__} that defines step()")

    class _Intercept(base_env_cls):
        def step(self, action, **kwargs):
            ctl = _slot_of(self)
            if ctl is None or ctl.phase != "capture":
                return super().step(action, **kwargs)
            self._intercepted_step = (action, kwargs)
            ctl.update_controller_params(action)   # what base_env_cls.step does first; repeated, unchanged, on resume
            raise StepSuspended()

        def resume_step(self, action_row):
            """Phase 3: base_env_cls.step with the arguments the task env's step() ended in and the batched action row."""
            action, kwargs = self._intercepted_step
            ctl = _slot_of(self)
            ctl.begin_replay(action_row)
            try:
                return base_env_cls.step(self, action, **kwargs)
            finally:
                ctl.phase, ctl.action = "idle", None
                self._intercepted_step = None

    _Intercept.__name__ = f"_Intercept{base_env_cls.__name__}"
    bases = (_Intercept,) if task_env_cls is base_env_cls else (task_env_cls, _Intercept)
    return type(f"OnePass{task_env_cls.__name__}", bases, {"__doc__": f"{task_env_cls.__name__} stepped in one pass by MPCVecEnv (split_step.one_pass)."})

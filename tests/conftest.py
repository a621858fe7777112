import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_addoption(parser):
    parser.addoption("--lane-grid", type=int, default=int(os.environ.get("RG_TEST_LANE_GRID", "0")),
                     help="run every test that does not choose a lane grid itself on this one (rg_mpc_config.lane_grid: 1 = one wave per "
                          "robot, 2 = 256 lanes per robot; 0 = the library's choice by batch size).  The GPU suite is green on both.")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    grid = config.getoption("--lane-grid")
    if grid:
        from robot_gym_amd.core.config import MPCConfig
        plain = MPCConfig.for_robot.__func__

        def for_robot(cls, robot="ghost", **overrides):
            return plain(cls, robot, **{"lane_grid": grid, **overrides})
        MPCConfig.for_robot = classmethod(for_robot)


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import oracle as O
    O.lib()
    return O

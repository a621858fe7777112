"""Study (GPU + oracle): the configuration sweep of tests/test_gpu_parity.py::test_randomised_configurations run in a
shuffled order, several repetitions in ONE process; reports every configuration whose error figures differ between two
runs or exceed the tolerance.  Written to hunt a launch-order dependence (a workgroup vote that let one wave of a
256-lane workgroup read a different result than the others once the launch's timing shifted -- DESIGN.md section 4);
300 configurations x 2 are clean with the vote of rg_qp_common.inc.
Usage: python tests/studies/launch_order_soak.py [configurations] [repetitions]"""
import os
import random
import sys


ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from tests import helpers                        # noqa: E402
from tests.test_gpu_parity import _sweep_case    # noqa: E402
from oracle import oracle as O                   # noqa: E402


def run(seed):
    cfg, B, over, kw = _sweep_case(seed)
    orc = helpers.run_oracle(O, cfg, **kw)
    gpu = helpers.run_gpu(cfg, **kw)
    errs = [helpers.compare_tick(g, o)["tau_rel_max"] for g, o in zip(gpu, orc)]
    return errs, " ".join(f"{e:.1e}" for e in errs) + "  retried " + str([g["solver_stats"]["retried_exact"] for g in gpu])


if __name__ == "__main__":
    O.lib()
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    first, bad = {}, 0
    for rep in range(reps):
        order = list(range(n))
        random.Random(rep).shuffle(order)
        for s in order:
            errs, text = run(s)
            if s in first and first[s] != text:
                print("DIFFERS", s, "|", first[s], "|", text)
                bad += 1
            if max(errs) > 1e-4:
                print("OVER", s, text)
                bad += 1
            first.setdefault(s, text)
    print("configurations", n, "repetitions", reps, "problems", bad)

cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_dbg; mkdir -p $O
python tests/studies/sweep_case_debug.py 24 > $O/case24.log 2>&1
python tests/studies/sweep_case_debug.py 24 '{"admm_tol": 1e-6}' > $O/case24_tol6.log 2>&1
python tests/studies/sweep_case_debug.py 22 > $O/case22.log 2>&1
python tests/studies/sweep_case_debug.py 23 > $O/case23.log 2>&1

set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_exact1; mkdir -p $O
timeout 600 python tests/studies/exact_body_check.py 512 10 > $O/check.log 2>&1
for tag in "hyb:" "hyb7:--tol 1e-7" "auto:--solver 2" "hybc2:--batch 1024 --fixed-cmd" "hybkin1:--kin-mode 1" "hybb1:--batch 1" "hyb32k:--batch 32768 --steps 30" ; do
  name=${tag%%:*}; args=${tag#*:}
  timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras $args > $O/$name.json 2> $O/$name.err
done

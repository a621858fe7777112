set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_base; mkdir -p $O
for tag in "head:" "s1:--solver 1" "s1c2:--solver 1 --batch 1024 --fixed-cmd" "c2:--batch 1024 --fixed-cmd" "kin1:--kin-mode 1" "s1kin1:--solver 1 --kin-mode 1" "b1:--batch 1" "s1b1:--solver 1 --batch 1" "tol7:--tol 1e-7" ; do
  name=${tag%%:*}; args=${tag#*:}
  timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras $args > $O/$name.json 2> $O/$name.err
done

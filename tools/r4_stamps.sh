cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_stamps; mkdir -p $O
timeout 900 python tests/studies/exact_body_check.py 512 8 > $O/check.log 2>&1
timeout 600 python tests/studies/fused_launch_schedule.py 4096 '{"solver": 1}' > $O/b4096.log 2>&1
timeout 600 python tests/studies/fused_launch_schedule.py 1024 '{"solver": 1, "fixed_cmd": 1}' > $O/b1024.log 2>&1
for tag in "s1:--solver 1" "s3:--solver 3" "s1c2:--solver 1 --batch 1024 --fixed-cmd" "s1kin1:--solver 1 --kin-mode 1" "s1b1:--solver 1 --batch 1" "s1b32k:--solver 1 --batch 32768 --steps 30" "s1cold:--solver 1 --cold-start"; do
  name=${tag%%:*}; args=${tag#*:}
  timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras $args > $O/$name.json 2> $O/$name.err
done

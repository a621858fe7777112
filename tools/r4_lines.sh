cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_lines; mkdir -p $O
for tag in "headline:" "config2:--batch 1024 --fixed-cmd" "b1:--batch 1" "b32768:--batch 32768 --steps 30" "kin1:--kin-mode 1" "kin0chain:--chain-geometry" "h20:--horizon 20 --steps 30" "config5:--horizon 20 --random-schedule --steps 30" "s1:--solver 1" "s2:--solver 2"; do
  name=${tag%%:*}; args=${tag#*:}
  timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline $args > $O/$name.json 2> $O/$name.err
done
timeout 600 python tests/studies/fused_launch_schedule.py 4096 '{}' > $O/stamps_b4096.log 2>&1

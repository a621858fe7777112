cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4_lines
python tests/studies/fused_launch_schedule.py 4096 '{}' > gpurun_out/r4_lines/stamps2.log 2>&1
for tag in "headline:" "config2:--batch 1024 --fixed-cmd" "b32768:--batch 32768 --steps 30" "s2:--solver 2"; do
  name=${tag%%:*}; args=${tag#*:}
  timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline $args > gpurun_out/r4_lines/$name.json 2> gpurun_out/r4_lines/$name.err
done

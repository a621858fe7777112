#!/bin/bash
# Runs on the GPU box (tools/stamp_commit.py first): rocprofv3 stats + PMC passes for the workloads named (keys of tools/r5_profiles.sh).
# Usage: tools/r5_partial_profiles.sh headline h20 config5 [config2 b1 b32768 kin1 config2_grid1]
cd ${GRAFT_REPO_ROOT:-$(pwd)}
python3 tools/evidence_guard.py || exit 1
for key in "$@"; do
  mode=full
  case $key in
    headline) tag=r5; args="";;
    h20) tag=r5_h20; args="--horizon 20";;
    config5) tag=r5_config5; args="--horizon 20 --random-schedule";;
    config2) tag=r5_config2; args="--batch 1024 --fixed-cmd";;
    b1) tag=r5_b1; args="--batch 1";;
    b32768) tag=r5_b32768; mode=stats; args="--batch 32768";;
    kin1) tag=r5_kin1; mode=stats; args="--kin-mode 1";;
    config2_grid1) tag=r5_config2_grid1; mode=stats; args="--batch 1024 --fixed-cmd --lane-grid 1";;
    *) echo "unknown key $key"; exit 1;;
  esac
  tools/collect_profiles.sh $tag $mode $args > gpurun_out/collect_$key.log 2>&1
  echo "$key collected"
done

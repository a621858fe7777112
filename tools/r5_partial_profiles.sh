#!/bin/bash
# Runs on the GPU box (tools/stamp_commit.py first): rocprofv3 stats + PMC passes for the workloads named (keys of tools/r5_profiles.sh).
# Usage: tools/r5_partial_profiles.sh headline h20 config5
cd ${GRAFT_REPO_ROOT:-$(pwd)}
python3 tools/evidence_guard.py || exit 1
for key in "$@"; do
  case $key in
    headline) tag=r5; args="";;
    h20) tag=r5_h20; args="--horizon 20";;
    config5) tag=r5_config5; args="--horizon 20 --random-schedule";;
    *) echo "unknown key $key"; exit 1;;
  esac
  tools/collect_profiles.sh $tag full $args > gpurun_out/collect_$key.log 2>&1
  echo "$key collected"
done

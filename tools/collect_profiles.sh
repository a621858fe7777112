#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats + separate PMC passes of bench.py.
# Usage: tools/collect_profiles.sh <tag> [stats|full] [bench args...]   -> gpurun_out/prof_<tag>/...
#   stats: only the --kernel-trace --stats pass (and the bench line);  full (default): + HBM / SQ / f64 counter passes.
# Summarise with tools/summarize_profiles.py <tag>.  The program after `--` is python3 itself (no env/bash hop).
set -u
TAG=${1:-r5}
MODE=${2:-full}
shift; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
python3 $ROOT/tools/evidence_guard.py > $OUT/evidence_header.txt || { cat $OUT/evidence_header.txt; exit 1; }   # the sources must be a commit's
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras $*"
echo "$BENCH" > $OUT/command.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1
grep -h '"metric"' $OUT/trace.log | tail -1 > $OUT/bench_line.json
if [ "$MODE" = "full" ]; then
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $BENCH > $OUT/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS --output-format csv -d $OUT/pmc_sq1 -- $BENCH > $OUT/pmc_sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq2 -- $BENCH > $OUT/pmc_sq2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 --output-format csv -d $OUT/pmc_flops -- $BENCH > $OUT/pmc_flops.log 2>&1
fi
find $OUT -name "*.csv" | head -30

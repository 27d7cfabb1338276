#!/bin/bash
# Run ON THE GPU BOX (via gpurun): rocprofv3 kernel-trace stats + PMC passes of one bench.py workload.
# Usage: tools/collect_profiles.sh <tag> [bench.py args, e.g. --scene soup100000]   -> gpurun_out/prof_<tag>/<tag>_summary.json
# (copy the summaries into profiles/ afterwards and merge them with tools/merge_profiles.py)
set -u
TAG=${1:-r02}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
STEPS=${STEPS:-50}
BENCH="python3 $ROOT/bench.py --steps $STEPS --warmup 5 --no-cpu-baseline --no-host-fb $*"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1
echo "trace rc=$?"
# PMC passes kept separate (TCC slots: FETCH_SIZE 3, WRITE_SIZE 2 -- MI355X_MICROARCH.md) and never combined with tracing
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH > $OUT/pmc_fetch.log 2>&1; echo "fetch rc=$?"
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $BENCH > $OUT/pmc_write.log 2>&1; echo "write rc=$?"
timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc_sq -- $BENCH > $OUT/pmc_sq.log 2>&1; echo "sq rc=$?"
timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_LDS SQ_INSTS_VMEM --output-format csv -d $OUT/pmc_f64 -- $BENCH > $OUT/pmc_f64.log 2>&1; echo "f64 rc=$?"
python3 $ROOT/tools/summarize_profiles.py $OUT $TAG "$*"

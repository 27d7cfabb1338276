#!/bin/bash
# Developer tool (run ON THE GPU BOX via gpurun): memory-pipeline counters of one bench.py workload, to tell whether a kernel is bound by the
# texture-address / L1 path.  Usage: tools/pmc_mem_probe.sh <tag> [bench.py args]  -> gpurun_out/pmcmem_<tag>/*.csv + a printed summary
set -u
TAG=${1:-x}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmcmem_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps ${STEPS:-10} --warmup 3 --no-cpu-baseline --no-host-fb $*"
i=0
# (TA_* counters are left out on purpose: rocprofv3 passes with TA_BUSY / TA_*_STALLED / TA_*_WAVEFRONTS sums never finished on this pool -- they ran into
# the 300 s timeout below each time)
for set in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- $BENCH > $OUT/p$i.log 2>&1; echo "pass $i rc=$?"
done
python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "render_kernel" not in r.get("Kernel_Name", ""): continue
        k = r["Counter_Name"]; tot[k][0] += float(r["Counter_Value"]); tot[k][1] += 1
for k, (v, n) in sorted(tot.items()): print(f"{k:40s} {v / n:18,.0f} per launch ({n} launches)")
PY

#!/bin/bash
# Run ON THE GPU BOX: latency-side PMC passes (outstanding-instruction levels, instruction fetch, per-unit active cycles) of a bench.py workload.
# Usage: tools/pmc_lat.sh <tag> [bench args...]   -> prints per-launch averages of the dominant render kernel
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmclat_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --no-cpu-baseline ${*:---steps 30 --warmup 3}"
timeout -k 10 300 rocprofv3 --pmc SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_IFETCH SQ_IFETCH_LEVEL --output-format csv -d $OUT/p1 -- $BENCH > $OUT/p1.log 2>&1; echo "p1 rc=$?"
timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VALU SQ_INSTS_BRANCH SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES --output-format csv -d $OUT/p2 -- $BENCH > $OUT/p2.log 2>&1; echo "p2 rc=$?"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR --output-format csv -d $OUT/p3 -- $BENCH > $OUT/p3.log 2>&1; echo "p3 rc=$?"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "render_kernel" in r["Kernel_Name"]:
            agg[r["Kernel_Name"][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    n = max(len(x) for x in v.values())
    if n < 10: continue
    print(k[:24], "launches", n)
    for c, x in sorted(v.items()): print(f"  {c:28s} {sum(x)/len(x)/1e6:12.2f} M")
PY

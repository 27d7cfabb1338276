#!/bin/bash
# Developer tool (run ON THE GPU BOX): instruction-cache, scalar-cache and in-flight-level counters of one bench.py workload: is a kernel waiting for
# instructions (I-cache), for scalar data (K$ -> L2) or for issue slots?      tools/pmc_sq_probe.sh <tag> [bench.py args]   -> gpurun_out/pmcsq_<tag>.json
set -u
TAG=${1:-x}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmcsq_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps ${STEPS:-6} --warmup 2 --no-cpu-baseline --no-host-fb --no-first-frame $*"
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_TC_INST_REQ SQ_IFETCH SQ_IFETCH_LEVEL GRBM_GUI_ACTIVE" \
           "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE SQC_TC_DATA_READ_REQ SQC_TC_STALL SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" \
           "SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $set --output-format csv -d "$OUT/p$i" -- $BENCH > "$OUT/p$i.log" 2>&1; echo "pass $i rc=$?"
done
python3 - "$OUT" "$ROOT/gpurun_out/pmcsq_$TAG.json" <<'PY'
import csv, glob, collections, json, sys
tot = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "render_kernel" not in r.get("Kernel_Name", ""): continue
        k = r["Counter_Name"]; tot[k][0] += float(r["Counter_Value"]); tot[k][1] += 1
res = {k: v / n for k, (v, n) in sorted(tot.items())}
for k, v in res.items(): print(f"{k:36s} {v:20,.0f}")
json.dump(res, open(sys.argv[2], "w"), indent=1)
PY

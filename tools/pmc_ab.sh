#!/bin/bash
# Run ON THE GPU BOX: one PMC pass (instruction counts) of the default bench workload for each given build of librrt_hip.so.
# Usage: tools/pmc_ab.sh libA.so libB.so ...   -> prints per-launch averages of the dominant render kernel
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  tag=$(basename $lib .so)
  OUT=$ROOT/gpurun_out/pmcab_$tag
  rm -rf $OUT; mkdir -p $OUT
  RRT_LIB=$ROOT/$lib timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc_sq -- python3 $ROOT/bench.py --no-cpu-baseline ${BENCH_ARGS:---steps 30 --warmup 3} > $OUT/log 2>&1
  python3 - "$OUT" "$tag" <<'PY'
import csv, glob, sys, collections
out, tag = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc_sq/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "render_kernel" in r["Kernel_Name"]:
            agg[r["Kernel_Name"][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    n = len(next(iter(v.values())))
    if n < 10: continue
    print(tag, k[:24], "launches", n, " ".join(f"{c.replace('SQ_','')}={sum(x)/len(x)/1e6:.1f}M" for c, x in sorted(v.items())))
PY
done

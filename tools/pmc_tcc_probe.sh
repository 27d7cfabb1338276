set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/pmctcc_$1; TAG=$1; shift
rm -rf $OUT; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-fb --no-first-frame $*"
i=0
for set in "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" "TCC_EA0_RDREQ_sum TCC_READ_REQ_LATENCY_sum"; do i=$((i+1)); timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- $BENCH > $OUT/p$i.log 2>&1; echo "pass $i rc=$?"; grep -m1 -o "error code [0-9]*[^\"]*" $OUT/p$i.log | head -c 150; done
python3 - $OUT <<'PY'
import csv, glob, collections, sys
tot = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "render_kernel" not in r.get("Kernel_Name", ""): continue
        tot[r["Counter_Name"]][0] += float(r["Counter_Value"]); tot[r["Counter_Name"]][1] += 1
for k, (v, n) in sorted(tot.items()): print(f"{k:36s} {v / n:20,.0f}")
PY

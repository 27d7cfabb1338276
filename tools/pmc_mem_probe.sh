#!/bin/bash
# Developer tool (run ON THE GPU BOX via gpurun): memory-pipeline counters of one bench.py workload -- is a kernel bound by the texture-address / L1
# path (TA, TCP) or by instruction issue?     tools/pmc_mem_probe.sh <tag> [bench.py args]   ->  gpurun_out/pmcmem_<tag>/ + gpurun_out/pmcmem_<tag>.json
#
# Round-2 post-mortem (the records were in gpurun_out/pmcmem_*/p2.log all along): passes that asked for several derived `_sum` counters at once died with
#     rocprofiler_create_counter_config ... error code 38: Request exceeds the capabilities of the hardware to collect
# -- every `_sum` expands into one hardware counter per TA/TCP instance, and a pass holds only so many -- after which rocprofv3 caught SIGABRT and sat in
# its own finalisation until the 300 s timeout killed it.  It was never a TA or kernel hang, and the program under the profiler never ran.  So here:
#   * at most TWO derived `_sum` counters per pass (SQ counters are single registers and can share a pass);
#   * the pass's log is watched: an abort of the profiler itself (error code 38, "Could not construct", caught signal 6) ends the pass at once;
#   * counter names are taken from `rocprofv3 --list-avail` of this box, so a name that does not exist here is skipped instead of failing the pass.
set -u
TAG=${1:-x}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmcmem_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps ${STEPS:-10} --warmup 3 --no-cpu-baseline --no-host-fb --no-first-frame $*"
rocprofv3 --list-avail > "$OUT/avail.txt" 2>&1 || rocprofv3 -L > "$OUT/avail.txt" 2>&1
have() { grep -qw "$1" "$OUT/avail.txt"; }

run_pass() {   # run_pass <n> <counters...>
  local n=$1; shift
  local keep=()
  for c in "$@"; do if have "$c"; then keep+=("$c"); else echo "pass $n: counter $c not on this GPU, skipped"; fi; done
  [ ${#keep[@]} -eq 0 ] && { echo "pass $n: nothing to collect"; return; }
  rocprofv3 --pmc "${keep[@]}" --output-format csv -d "$OUT/p$n" -- $BENCH > "$OUT/p$n.log" 2>&1 &
  local pid=$! t=0
  while kill -0 $pid 2>/dev/null; do
    sleep 2; t=$((t + 2))
    if grep -qE "error code 38|Could not construct|caught signal 6" "$OUT/p$n.log" 2>/dev/null; then
      echo "pass $n (${keep[*]}): the PROFILER aborted ($(grep -m1 -oE 'error code [0-9]+[^\"]*' "$OUT/p$n.log" | head -c 120)); pass dropped"; kill $pid 2>/dev/null; sleep 1; kill -9 $pid 2>/dev/null; wait $pid 2>/dev/null; return
    fi
    if [ $t -ge ${PASS_TIMEOUT:-240} ]; then echo "pass $n (${keep[*]}): timed out after $t s"; kill $pid 2>/dev/null; sleep 2; kill -9 $pid 2>/dev/null; wait $pid 2>/dev/null; return; fi
  done
  wait $pid; echo "pass $n (${keep[*]}) rc=$?"
}

run_pass 1 GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM
run_pass 2 SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT
run_pass 3 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
run_pass 4 TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
run_pass 5 TA_TA_BUSY_sum TA_BUSY_avr
run_pass 6 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
run_pass 7 TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WAVEFRONTS_sum
run_pass 8 TCP_GATE_EN1_sum TCP_GATE_EN2_sum
run_pass 9 TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum

python3 - "$OUT" "$ROOT/gpurun_out/pmcmem_$TAG.json" <<'PY'
import csv, glob, collections, json, sys
out_dir, out_json = sys.argv[1], sys.argv[2]
tot = collections.defaultdict(lambda: [0.0, 0]); dur = []
for f in glob.glob(out_dir + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "render_kernel" not in r.get("Kernel_Name", ""): continue
        k = r["Counter_Name"]; tot[k][0] += float(r["Counter_Value"]); tot[k][1] += 1
res = {k: {"per_launch": v / n, "launches": n} for k, (v, n) in sorted(tot.items())}
for k, v in res.items(): print(f"{k:44s} {v['per_launch']:20,.0f} per launch ({v['launches']} launches)")
json.dump({"counters_per_launch_of_render_kernel": res}, open(out_json, "w"), indent=1)
PY

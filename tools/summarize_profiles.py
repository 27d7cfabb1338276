"""Condenses a tools/collect_profiles.sh output directory into <dir>/<tag>_summary.json (kernel stats + per-launch PMC averages of render_kernel),
stamped with the sha256 of the kernel sources the numbers were measured on (bench.py refuses a summary whose stamp does not match its sources)."""
import collections, csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_source_sha256
out_dir, tag = sys.argv[1], sys.argv[2]
bench_args = sys.argv[3] if len(sys.argv) > 3 else ""
summary = {"tag": tag, "bench_args": bench_args, "source_sha256": kernel_source_sha256(), "kernel": "render_kernel", "pmc_avg_per_launch": {}, "kernel_stats": None}
dominant = None      # render_kernel<false> (lane filter) and render_kernel<true> (bundle filter) both run while the variant is tuned: keep the one that does the frames
for f in glob.glob(os.path.join(out_dir, "trace", "*", "*kernel_stats.csv")):
    for r in csv.DictReader(open(f)):
        if "render_kernel" in r["Name"] and (dominant is None or int(r["Calls"]) > int(summary["kernel_stats"]["Calls"])):
            dominant = r["Name"]
            summary["kernel_stats"] = {k: r[k] for k in ("Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")}
summary["kernel"] = dominant
for f in glob.glob(os.path.join(out_dir, "trace", "*", "*kernel_trace.csv")):   # registers / scratch / LDS of the dominant kernel as dispatched
    for r in csv.DictReader(open(f)):
        if r.get("Kernel_Name") == dominant:
            summary["dispatch"] = {k: r[k] for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Grid_Size", "Workgroup_Size") if k in r}
            break
for f in glob.glob(os.path.join(out_dir, "pmc_*", "*", "*counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"] == dominant or (dominant is None and "render_kernel" in r["Kernel_Name"]):
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        summary["pmc_avg_per_launch"][k] = sum(v) / len(v)
p = summary["pmc_avg_per_launch"]
if "FETCH_SIZE" in p:
    # MI355X_MICROARCH.md (HBM): FETCH_SIZE/WRITE_SIZE are in KiB-like units of 1024 B in rocprofv3's derived counter; on gfx950 FETCH_SIZE reads HALF the bytes of
    # a wide coalesced stream -> the corrected figure doubles it.  This kernel's reads are scalar/narrow (uncalibrated width), so both are reported.
    summary["hbm_bytes_per_launch"] = {"fetch_raw": p["FETCH_SIZE"] * 1024, "fetch_x2_gfx950_correction": p["FETCH_SIZE"] * 2048, "write": p.get("WRITE_SIZE", 0) * 1024}
if "SQ_INSTS_VALU_ADD_F64" in p:
    summary["executed_f64_flop_per_launch_upper_bound"] = 64 * (p["SQ_INSTS_VALU_ADD_F64"] + p["SQ_INSTS_VALU_MUL_F64"] + 2 * p["SQ_INSTS_VALU_FMA_F64"] + p.get("SQ_INSTS_VALU_TRANS_F64", 0))
json.dump(summary, open(os.path.join(out_dir, f"{tag}_summary.json"), "w"), indent=1)
print(json.dumps(summary, indent=1))

"""Developer tool: average the rocprofv3 --pmc counter_collection.csv rows of render_kernel."""
import collections, csv, glob, sys
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "render_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(agg.items()):
        print(f"{k:28s} {sum(v)/len(v):18.0f}  (n={len(v)})")

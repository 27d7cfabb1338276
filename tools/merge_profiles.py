"""Builds profiles/current_summary.json (what bench.py's `roofline.executed` reads) from per-workload summaries under profiles/:
   python tools/merge_profiles.py model2.obj@1920x1080=profiles/r02_teapot_summary.json soup100000@1920x1080=profiles/r02_soup100k_summary.json ...
Every summary must carry the same source_sha256 (tools/summarize_profiles.py stamps it on the GPU box), and it must be that of the sources in this tree."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_source_sha256
sha = kernel_source_sha256()
out = {"source_sha256": sha, "sources": list(__import__("bench").KERNEL_SOURCES), "workloads": {}}
for arg in sys.argv[1:]:
    key, path = arg.split("=", 1)
    s = json.load(open(path))
    if s.get("source_sha256") != sha:
        raise SystemExit(f"{path}: measured on other kernel sources ({str(s.get('source_sha256'))[:12]} != {sha[:12]}): re-collect it")
    out["workloads"][key] = s
json.dump(out, open(os.path.join(ROOT, "profiles", "current_summary.json"), "w"), indent=1)
print("profiles/current_summary.json:", ", ".join(out["workloads"]), "sha", sha[:12])

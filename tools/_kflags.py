"""The kernels' code-generation switches, read from the one place that defines them (rust-ray-tracer_amd/csrc/Makefile: KFLAGS), for the developer
tools that compile render.hip on their own (bbprof.py, kernel_resources.py, build_variant.sh).  RRT_KFLAGS_NAME=KFLAGS_LANE in the environment
selects the lane-filter / ray-walk kernels' switches instead of the bundle-filter kernel's (the product compiles render.hip once per group)."""
import os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kflags(name: str = "") -> list:
    name = name or os.environ.get("RRT_KFLAGS_NAME", "KFLAGS")
    text = open(os.path.join(ROOT, "rust-ray-tracer_amd", "csrc", "Makefile")).read()
    m = re.search(r"^%s\s*\?=\s*(.*)$" % re.escape(name), text, re.M)
    if not m:
        raise SystemExit(f"{name} not found in the Makefile")
    out = []
    for tok in m.group(1).split():
        mm = re.fullmatch(r"\$\((\w+)\)", tok)          # e.g. KFLAGS_LANE ?= $(KFLAGS) ...
        out += kflags(mm.group(1)) if mm else [tok]
    return out


if __name__ == "__main__":
    print(" ".join(kflags(sys.argv[1] if len(sys.argv) > 1 else "")))

"""The kernels' code-generation switches, read from the one place that defines them (rust-ray-tracer_amd/csrc/Makefile: KFLAGS), for the developer
tools that compile render.hip on their own (bbprof.py, kernel_resources.py, build_variant.sh)."""
import os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kflags(name: str = "KFLAGS") -> list:
    text = open(os.path.join(ROOT, "rust-ray-tracer_amd", "csrc", "Makefile")).read()
    m = re.search(r"^%s\s*\?=\s*(.*)$" % re.escape(name), text, re.M)
    if not m:
        raise SystemExit(f"{name} not found in the Makefile")
    return m.group(1).split()


if __name__ == "__main__":
    print(" ".join(kflags(sys.argv[1] if len(sys.argv) > 1 else "KFLAGS")))

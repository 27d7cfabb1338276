"""Developer tool: registers, spills, scratch and occupancy of every kernel of render.hip (hipcc -Rpass-analysis=kernel-resource-usage).
   python tools/kernel_resources.py [extra hipcc flags, e.g. -DRRT_WAVES_LANE=4]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "rust-ray-tracer_amd", "csrc")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _kflags import kflags
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", *kflags(), "-S", "--cuda-device-only", "-o", "/tmp/render.gfx950.s",
       os.path.join(src, "render.hip"), "-Rpass-analysis=kernel-resource-usage"] + sys.argv[1:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
keep = ("VGPRs", "TotalSGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "SGPRs Spill", "VGPRs Spill")
line = ""
for l in out.splitlines():
    m = re.search(r"remark:\s+(.*?) \[-Rpass", l)
    if not m:
        if "error" in l: print(l)
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name"):
        if line: print(line)
        name = subprocess.run(["c++filt", t.split(":", 1)[1].strip()], capture_output=True, text=True).stdout.strip()
        line = name.split("(")[0].replace("rrt::(anonymous namespace)::", "").replace("void ", "") + ": "
    else:
        k, v = t.split(":", 1)
        if k.strip() in keep: line += k.strip().split(" [")[0] + "=" + v.strip() + "  "
if line: print(line)

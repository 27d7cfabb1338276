"""Developer tool: dynamic instruction counts (all kinds) per source line / per region from a tools/bbprof.py run.
   python tools/bbprof_lines.py <counts.txt> <kernel-name-substring> [top-n]"""
import json, collections, re, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
counts, kern = sys.argv[1], sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
m = json.load(open(os.path.join(ROOT, "gpurun_out", "bbprof_map.json")))["kernels"]
src = open(os.path.join(ROOT, "rust-ray-tracer_amd", "csrc", "render.hip")).read().split("\n")
sums, launch = {}, collections.Counter()
for rec in open(counts).read().strip().split("\n"):
    name, vals = rec.split(" ", 1); v = list(map(int, vals.split())); launch[name] += 1
    sums[name] = [a + b for a, b in zip(sums.get(name, [0] * len(v)), v)]
name = [k for k in sums if kern in k][0]
blocks = m[name]["blocks"]; cnt = [c / launch[name] for c in sums[name][:len(blocks)]]
def kind(op):
    return "VALU" if op.startswith("v_") else "SMEM" if op.startswith("s_load") else "wait" if op.startswith(("s_waitcnt", "s_nop")) else "branch" if op.startswith(("s_cbranch", "s_branch")) else \
           "SALU" if op.startswith("s_") else "LDS" if op.startswith("ds_") else "VMEM"
by_line = collections.Counter(); kinds = collections.Counter(); line_kinds = collections.defaultdict(collections.Counter)
for b, c in zip(blocks, cnt):
    for op, l in b["ops"]:
        by_line[l] += c; kinds[kind(op)] += c; line_kinds[l][kind(op)] += c
T = sum(by_line.values())
print(name, f"{launch[name]} launches; instructions per launch {T:,.0f}:", ", ".join(f"{k} {v:,.0f}" for k, v in kinds.most_common()))
# functions: attribute each line to the enclosing "__device__ ... name(" definition
func_of = {}; cur = "?"
for i, l in enumerate(src, 1):
    mm = re.match(r"^(?:template.*\n)?(?:__device__|__global__).*?(\w+)\(", l)
    if mm and not l.startswith(" "): cur = mm.group(1)
    func_of[i] = cur
by_func = collections.Counter()
for l, c in by_line.items(): by_func[func_of.get(l, "(other file / no line)") if l > 0 else "(other file / no line)"] += c
print("by function:")
for f, c in by_func.most_common(25): print(f"  {100 * c / T:5.1f} %  {c:15,.0f}  {f}")
print("top lines:")
for l, c in by_line.most_common(top):
    print(f"  {100 * c / T:5.1f} %  L{l}: {src[l - 1].strip()[:110] if 0 < l <= len(src) else ''}   [{' '.join(f'{k}:{100 * v / c:.0f}%' for k, v in line_kinds[l].most_common(3))}]")

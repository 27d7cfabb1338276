"""Developer tool: attribute the static VALU instructions of render_kernel to source functions/lines (needs a -gline-tables-only .s)."""
import re, collections, sys
asm, srcp = sys.argv[1], sys.argv[2]
lines = open(asm).read().split('\n')
start = next(i for i, l in enumerate(lines) if l.startswith('_ZN3rrt12_GLOBAL__N_113render_kernel') and ':' in l)
end = next(i for i in range(start, len(lines)) if 's_endpgm' in lines[i])
cur = 0
cnt = collections.Counter()
for l in lines[start:end]:
    m = re.match(r'\s*\.loc\s+\d+\s+(\d+)', l)
    if m: cur = int(m.group(1)); continue
    t = l.strip().split()
    if t and t[0].startswith('v_'): cnt[cur] += 1
src = open(srcp).read().split('\n')
funcs = []
for i, l in enumerate(src, 1):
    m = re.match(r'__device__ __forceinline__ .*?(\w+)\(', l) or re.match(r'__global__ .* void (\w+)\(', l)
    if m: funcs.append((i, m.group(1)))
def fn(line):
    name = '?'
    for s, n in funcs:
        if s <= line: name = n
    return name
agg = collections.Counter()
for line, c in cnt.items(): agg[fn(line)] += c
tot = sum(agg.values())
for k, v in agg.most_common(): print(f"{k:24s} {v:6d}  {100*v/tot:5.1f}%")
print('total static VALU', tot)
for f in sys.argv[3:]:
    print('---', f)
    for c, l in sorted(((c, l) for l, c in cnt.items() if fn(l) == f), reverse=True)[:30]: print(f"{c:5d}  L{l}: {src[l-1].strip()[:120]}")

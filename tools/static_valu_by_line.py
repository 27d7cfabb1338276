"""Developer tool: attribute the static VALU instructions of one render_kernel variant to source lines (needs a -gline-tables-only .s).
   python tools/static_valu_by_line.py render_lines.s render.hip [ILb1|ILb0] [lo-hi line ranges to sum ...]"""
import re, collections, sys
asm, srcp = sys.argv[1], sys.argv[2]
variant = sys.argv[3] if len(sys.argv) > 3 else "ILb1"
lines = open(asm).read().split('\n')
start = next(i for i, l in enumerate(lines) if l.startswith('_ZN3rrt12_GLOBAL__N_113render_kernel' + variant) and ':' in l)
end = next(i for i in range(start, len(lines)) if 's_endpgm' in lines[i])
cur = 0
cnt = collections.Counter(); sal = collections.Counter()
for l in lines[start:end]:
    m = re.match(r'\s*\.loc\s+\d+\s+(\d+)', l)
    if m: cur = int(m.group(1)); continue
    t = l.strip().split()
    if t and t[0].startswith('v_'): cnt[cur] += 1
    if t and t[0].startswith('s_') and not t[0].startswith('s_nop'): sal[cur] += 1
src = open(srcp).read().split('\n')
print('total static VALU', sum(cnt.values()), 'SALU', sum(sal.values()))
for r in sys.argv[4:]:
    lo, hi = map(int, r.split('-'))
    print(f"lines {lo}-{hi}: VALU {sum(c for l, c in cnt.items() if lo <= l <= hi)}  SALU {sum(c for l, c in sal.items() if lo <= l <= hi)}   [{src[lo-1].strip()[:80]}]")
if len(sys.argv) <= 4:
    for l in sorted(cnt):
        print(f"{cnt[l]:5d} {sal[l]:5d}  L{l}: {src[l-1].strip()[:130]}")

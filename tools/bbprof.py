"""Developer tool: EXACT dynamic instruction mix of the product kernels by basic-block counting (no PC sampling on this pool).

The product's own device assembly (hipcc -S of render.hip with the product flags, plus -gline-tables-only, which does not change the code) is
patched so that every basic block of the render_kernel variants increments a counter of its own; the patched code object is loaded in place
of the compiled-in kernel by a developer build of the library (-DRRT_DEV_HSACO, env RRT_DEV_HSACO / RRT_DEV_BBPROF_OUT), and the per-block
execution counts x the block's static instructions give the dynamic opcode histogram, per opcode class and per source line.

Counter i lives in lane i%64 of VGPR base+i/64, above the registers the kernel uses (the descriptor's VGPR count is raised; occupancy of the
instrumented run may drop, the counts do not care).  One increment, executed once per wave per block whatever EXEC is:
    v_readlane_b32 s100, vR, L ; s_cselect_b32 s101, 1, 0 (save SCC) ; s_add_u32 s100, s100, 1 ; s_cmp_lg_u32 s101, 0 (restore SCC) ;
    v_writelane_b32 vR, s100, L                     (s100, s101: the kernels use s0..s99; VCC, EXEC, M0 untouched)
At s_endpgm the counters are added to a global buffer whose address the developer library passes in DevScene::tex (unused by the kernels).

    python tools/bbprof.py build                      -> rust-ray-tracer_amd/render_bbprof.hsaco + gpurun_out/bbprof_map.json   (CPU, here)
    python tools/bbprof.py run <bench args...>        -> runs bench.py with the instrumented kernels, writes gpurun_out/bbprof_counts.txt  (GPU box)
    python tools/bbprof.py report [counts] [map]      -> tables
"""
import collections, json, os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "rust-ray-tracer_amd", "csrc")
LLVM = "/opt/rocm/lib/llvm/bin"
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _kflags import kflags
FLAGS = ["-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", *kflags()]   # the product's FLAGS + KFLAGS (csrc/Makefile)
HSACO = os.path.join(ROOT, "rust-ray-tracer_amd", "render_bbprof.hsaco")
MAP = os.path.join(ROOT, "gpurun_out", "bbprof_map.json")
COUNTS = os.path.join(ROOT, "gpurun_out", "bbprof_counts.txt")
TEX_OFFSET = 64          # offsetof(DevScene, tex): 8 pointers before it


def opclass(op: str) -> str:
    op = re.sub(r"_e32$|_e64$|_dpp$|_sdwa$|_e64_dpp$", "", op)
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")): return "lane<->SGPR moves (readlane/writelane: SGPR spills, broadcasts)"
    if op.startswith("v_cmp") or op.startswith("v_cmpx"): return "compares"
    if op.startswith("v_cndmask"): return "selects (v_cndmask)"
    if op.startswith(("v_mov", "v_accvgpr")): return "moves"
    if op.startswith(("v_min", "v_max", "v_med3")): return ("min/max f64" if "f64" in op else "min/max f32" if "f32" in op else "min/max int")
    if op.startswith(("v_cvt", "v_ldexp", "v_frexp", "v_trunc", "v_floor", "v_rndne", "v_fract")): return "conversions / ldexp / rounding"
    if op.startswith(("v_rcp", "v_rsq", "v_sqrt", "v_exp", "v_log", "v_sin", "v_cos")): return "transcendental (" + ("f64" if "f64" in op else "f32") + ")"
    if op.startswith(("v_div_scale", "v_div_fmas", "v_div_fixup")): return "divide helpers (div_scale/fmas/fixup f64)"
    if "f64" in op: return "arithmetic f64 (add/mul/fma)"
    if "f32" in op: return "arithmetic f32 (add/mul/fma)"
    if op.startswith(("v_mbcnt", "v_bfe", "v_bfi", "v_and", "v_or", "v_xor", "v_not", "v_lshl", "v_lshr", "v_ashr", "v_add_u", "v_sub_u", "v_subrev_u", "v_add_co", "v_addc", "v_subb", "v_mul_lo", "v_mul_hi",
                      "v_mad_u", "v_mad_i", "v_add3", "v_lshl_add", "v_lshl_or", "v_and_or", "v_or3", "v_add_lshl", "v_alignbit", "v_perm", "v_bcnt", "v_ffb", "v_sad", "v_sub_co", "v_add_i", "v_sub_i", "v_mul_u", "v_mul_i")): return "integer / bit ops"
    return "other VALU (" + op + ")"


def build():
    os.makedirs(os.path.dirname(MAP), exist_ok=True)
    src_s = "/tmp/render_bbprof_in.s"
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", *FLAGS, "-gline-tables-only", "-S", "--cuda-device-only", "-o", src_s, os.path.join(CSRC, "render.hip")], check=True, stderr=subprocess.DEVNULL)
    lines = open(src_s).read().split("\n")
    out, kernels = [], {}
    i = 0
    fn_re = re.compile(r"^(_ZN3rrt12_GLOBAL__N_113render_kernel\w+):")
    cur = None
    pending_desc = {}            # kernel name -> (base vgpr, n counter regs)
    # pass 1: find kernels, their bodies and their VGPR use
    nfv = {}
    for k, l in enumerate(lines):
        m = re.match(r"\s*\.amdhsa_kernel (\S+)", l)
        if m: cur = m.group(1)
        m = re.match(r"\s*\.amdhsa_next_free_vgpr (\d+)", l)
        if m and cur: nfv[cur] = int(m.group(1))
        m = re.match(r"\s*\.amdhsa_next_free_sgpr (\d+)", l)
        if m and cur and "render_kernel" in cur and int(m.group(1)) > 100: raise SystemExit(f"{cur} uses SGPRs beyond s99: no free temporaries")
    curline = 0
    main_file = next((int(m.group(1)) for m in (re.match(r'\s*\.file\s+(\d+)\s+.*"render\.hip"', l) for l in lines) if m), 1)
    while i < len(lines):
        l = lines[i]
        m = fn_re.match(l)
        if not m:
            out.append(l); i += 1; continue
        name = m.group(1)
        end = next(j for j in range(i, len(lines)) if lines[j].startswith(".Lfunc_end"))
        body = lines[i + 1:end]
        bb_re = re.compile(r"^(\.LBB\d+_\d+):|^; (%bb\.\d+):")          # labelled blocks and fall-through-only blocks (printed as a comment)
        n_bb = 1 + sum(1 for b in body if bb_re.match(b))
        n_regs = (n_bb + 63) // 64
        base = nfv[name]
        vP, vA = base + n_regs, base + n_regs + 1
        pending_desc[name] = (base, n_regs)
        blocks = []
        def counter(idx):
            r, ln = base + idx // 64, idx % 64
            return [f"\tv_readlane_b32 s100, v{r}, {ln}", "\ts_cselect_b32 s101, 1, 0", "\ts_nop 1", "\ts_add_u32 s100, s100, 1", "\ts_cmp_lg_u32 s101, 0", "\ts_nop 1", f"\tv_writelane_b32 v{r}, s100, {ln}"]
        out.append(l)
        # prologue: zero the counters, stash the counter-buffer pointer (kernarg DevScene::tex)
        pro = [f"\tv_mov_b32_e32 v{base + r}, 0" for r in range(n_regs)] + [f"\tv_mov_b32_e32 v{vP}, 0", f"\ts_load_dwordx2 s[100:101], s[0:1], {hex(TEX_OFFSET)}", "\ts_waitcnt lgkmcnt(0)",
                                                                    f"\tv_writelane_b32 v{vP}, s100, 0", f"\tv_writelane_b32 v{vP}, s101, 1", "\ts_nop 1"]
        out += pro + counter(0)
        blocks.append({"id": 0, "label": "entry", "ops": []})
        epi = f".Lbbprof_epi_{len(kernels)}"
        for b in body:
            mm = bb_re.match(b)
            if mm:
                out.append(b)
                if mm.group(2) == "%bb.0": continue                            # the entry block: counted by the prologue
                out += counter(len(blocks))
                blocks.append({"id": len(blocks), "label": mm.group(1) or mm.group(2), "ops": []})
                continue
            ml = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", b)
            if ml: curline = int(ml.group(2)) if int(ml.group(1)) == main_file else -int(ml.group(1))
            t = b.strip().split()
            if t and not t[0].startswith((".", ";")) and not t[0].endswith(":"):
                if t[0] == "s_endpgm":
                    out.append(f"\ts_branch {epi}")
                    blocks[-1]["ops"].append(["s_branch", curline])
                    continue
                blocks[-1]["ops"].append([t[0], curline])
            out.append(b)
        # epilogue
        out.append(f"{epi}:")
        out += ["\ts_mov_b64 exec, -1", f"\tv_readlane_b32 s100, v{vP}, 0", f"\tv_readlane_b32 s101, v{vP}, 1", f"\tv_mbcnt_lo_u32_b32 v{vA}, -1, 0", f"\tv_mbcnt_hi_u32_b32 v{vA}, -1, v{vA}",
                f"\tv_lshlrev_b32_e32 v{vA}, 2, v{vA}", "\ts_nop 4"]
        for r in range(n_regs):
            out.append(f"\tglobal_atomic_add v{vA}, v{base + r}, s[100:101] offset:{256 * r}")
        out += ["\ts_waitcnt vmcnt(0)", "\ts_endpgm"]
        kernels[name] = {"blocks": blocks, "n_regs": n_regs, "base_vgpr": base}
        i = end
    # descriptors / metadata: more VGPRs, s100/s101 in use
    text = "\n".join(out)
    for name, (base, n_regs) in pending_desc.items():
        new = ((base + n_regs + 2 + 7) // 8) * 8
        a = text.index(f".amdhsa_kernel {name}\n"); b = text.index(".end_amdhsa_kernel", a)
        d = text[a:b]
        d = re.sub(r"\.amdhsa_next_free_vgpr \d+", f".amdhsa_next_free_vgpr {new}", d)
        d = re.sub(r"\.amdhsa_next_free_sgpr \d+", ".amdhsa_next_free_sgpr 102", d)
        d = re.sub(r"\.amdhsa_accum_offset \d+", f".amdhsa_accum_offset {new}", d)
        text = text[:a] + d + text[b:]
        a = text.index(f".name:           {name}\n"); b = text.index(".wavefront_size", a)
        d = re.sub(r"\.vgpr_count:\s+\d+", f".vgpr_count:     {new}", text[a:b])
        text = text[:a] + d + text[b:]
    pat_s = "/tmp/render_bbprof.s"
    open(pat_s, "w").write(text)
    subprocess.run([f"{LLVM}/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", pat_s, "-o", "/tmp/render_bbprof.o"], check=True)
    subprocess.run([f"{LLVM}/ld.lld", "-shared", "/tmp/render_bbprof.o", "-o", HSACO], check=True)
    json.dump({"kernels": kernels}, open(MAP, "w"))
    for n, k in kernels.items():
        print(f"{n}: {len(k['blocks'])} blocks, counters in v{k['base_vgpr']}..v{k['base_vgpr'] + k['n_regs'] - 1}")
    print("->", HSACO, MAP)


def run(args):
    # (RRT_BBPROF_LIB / RRT_BBPROF_HSACO: copies under another name -- *_dev.so and *.hsaco are developer artefacts that .gpurunignore keeps off the GPU box)
    env = dict(os.environ, RRT_LIB=os.environ.get("RRT_BBPROF_LIB", os.path.join(ROOT, "rust-ray-tracer_amd", "librrt_hip_dev.so")),
               RRT_DEV_HSACO=os.environ.get("RRT_BBPROF_HSACO", HSACO), RRT_DEV_BBPROF_OUT=COUNTS)
    if os.path.exists(COUNTS): os.remove(COUNTS)
    subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-host-fb", *args], env=env, check=True)
    print("->", COUNTS)


def report(counts_path=COUNTS, map_path=MAP, out_json=None):
    kernels = json.load(open(map_path))["kernels"]
    launches = collections.Counter(); sums = {}
    for rec in open(counts_path).read().strip().split("\n"):
        name, vals = rec.split(" ", 1)
        v = list(map(int, vals.split()))
        launches[name] += 1
        sums[name] = [a + b for a, b in zip(sums.get(name, [0] * len(v)), v)]
    result = {}
    for name, n in launches.most_common():
        blocks = kernels[name]["blocks"]
        cnt = [c / n for c in sums[name][:len(blocks)]]
        cls = collections.Counter(); by_line = collections.Counter(); by_op = collections.Counter(); tot = collections.Counter()
        s_op = collections.Counter(); s_line = collections.Counter(); nop_after = collections.Counter(); prev_op = None; inl = collections.Counter()
        last_line = 0
        for b, c in zip(blocks, cnt):
            for op, line in b["ops"]:
                # Inlined library code (OCML pow / sqrt / divide expansions) carries no line of render.hip (0, or -file): it is charged to the call site,
                # i.e. the last render.hip line seen before it in layout order, and counted under "inlined at" so that nothing stays unattributed.
                if line > 0: last_line = line
                elif op.startswith("v_"): inl[last_line] += c; line = last_line
                kind = "VALU" if op.startswith("v_") else "SALU" if op.startswith("s_") and not op.startswith(("s_load", "s_buffer_load", "s_waitcnt", "s_nop")) else \
                       "SMEM" if op.startswith(("s_load", "s_buffer_load")) else "LDS" if op.startswith("ds_") else "VMEM" if op.startswith(("global_", "scratch_", "flat_", "buffer_")) else "other"
                tot[kind] += c
                if not op.startswith("v_"):
                    s_op[op] += c; s_line[line] += c
                    if op == "s_nop": nop_after[prev_op] += c
                prev_op = op
                if kind == "VALU":
                    cls[opclass(op)] += c; by_line[line] += c; by_op[re.sub(r"_e32$|_e64$", "", op)] += c
        src = open(os.path.join(CSRC, "render.hip")).read().split("\n")
        print(f"\n=== {name}  ({n} launches; per launch)")
        print("  instructions per launch: " + ", ".join(f"{k} {v:,.0f}" for k, v in tot.most_common()))
        V = tot["VALU"] or 1
        print("  VALU by class:")
        for k, v in cls.most_common():
            print(f"    {v:16,.0f}  {100 * v / V:5.1f} %  {k}")
        print("  top opcodes: " + ", ".join(f"{k} {100 * v / V:.1f}%" for k, v in by_op.most_common(24)))
        print("  top source lines (VALU):")
        for line, v in by_line.most_common(40):
            print(f"    {100 * v / V:5.1f} %  L{line}: {src[line - 1].strip()[:120] if 0 < line <= len(src) else ''}")
        print(f"  VALU of inlined library code, by the render.hip line it was inlined at ({100 * sum(inl.values()) / V:.1f} % of VALU):")
        for line, v in inl.most_common(12):
            print(f"    {100 * v / V:5.1f} %  L{line}: {src[line - 1].strip()[:120] if 0 < line <= len(src) else ''}")
        S = sum(s_op.values()) or 1
        print(f"  non-VALU instructions by opcode ({S:,.0f} per launch): " + ", ".join(f"{k} {100 * v / S:.1f}%" for k, v in s_op.most_common(30)))
        print("  s_nop by the instruction before it: " + ", ".join(f"{k} {v:,.0f}" for k, v in nop_after.most_common(10)))
        print("  top source lines (non-VALU):")
        for line, v in s_line.most_common(30):
            print(f"    {100 * v / S:5.1f} %  L{line}: {src[line - 1].strip()[:120] if 0 < line <= len(src) else ''}")
        result[name] = {"launches": n, "scalar_by_opcode": dict(s_op.most_common(60)), "scalar_by_source_line": {str(k): v for k, v in s_line.most_common(80)}, "per_launch": dict(tot), "valu_by_class": dict(cls), "valu_by_opcode": dict(by_op.most_common(60)),
                        "valu_by_source_line": {str(k): v for k, v in by_line.most_common(80)},
                        "valu_inlined_library_code_by_call_line": {str(k): v for k, v in inl.most_common(40)}, "block_counts": cnt}
    if out_json:
        json.dump(result, open(out_json, "w"), indent=1)
    return result


REGION_MARKS = [("walk set-up (fp32 ray, reciprocals, bundle)", "__device__ __forceinline__ void traverse("),
                ("pick node + record load", "const unsigned long long pending = __builtin_amdgcn_ballot_w64(!done);"),
                ("children: fp32 reach filter", "conservative fp32 filter against the TIGHT"),
                ("children: single candidate (exact slab test, leaf triangle)", "if ((reach & (reach - 1u)) == 0u) {"),
                ("children: several candidates (plane quotients, slab tests, leaf triangles, rank)", "double qlx = 0"),
                ("own list: boxes in lanes", "bool in_lanes = kBundle;"),
                ("own list: one box at a time (lane filter)", "a node with a single super-cluster carries its slot range in the node record and skips"),
                ("return / push / unwind", "bool returning;"),
                ("shading + per-ray state machine", "// ------------------------------------------------------------------------------------------------ shading helpers"),
                ("kernel prologue / epilogue (pixel grid, Color::mix, store)", "// ------------------------------------------------------------------------------------------------ kernels")]


def regions(counts_path=COUNTS, map_path=MAP):
    """Dynamic VALU per region of the kernel: a block belongs to the region of the LAST marker line at or before the largest traverse-or-later
    source line among its instructions; blocks made only of inlined helper lines inherit the region of the block before them in layout order."""
    kernels = json.load(open(map_path))["kernels"]
    src = open(os.path.join(CSRC, "render.hip")).read().split("\n")
    marks = [(nm, next(i + 1 for i, l in enumerate(src) if key in l)) for nm, key in REGION_MARKS]
    first = marks[0][1]
    launches = collections.Counter(); sums = {}
    for rec in open(counts_path).read().strip().split("\n"):
        name, vals = rec.split(" ", 1)
        v = list(map(int, vals.split())); launches[name] += 1
        sums[name] = [a + b for a, b in zip(sums.get(name, [0] * len(v)), v)]
    out = {}
    for name, n in launches.most_common():
        blocks = kernels[name]["blocks"]; cnt = [c / n for c in sums[name][:len(blocks)]]
        tot = collections.Counter(); prev = marks[-1][0]
        for b, c in zip(blocks, cnt):
            ls = [l for op, l in b["ops"] if l >= first]
            reg = [nm for nm, ln in marks if ln <= max(ls)][-1] if ls else prev
            prev = reg
            tot[reg] += c * sum(1 for op, l in b["ops"] if op.startswith("v_"))
        T = sum(tot.values()) or 1
        print(f"\n=== {name} ({n} launches): VALU per launch {T:,.0f}")
        for k, v in tot.most_common():
            print(f"   {100 * v / T:5.1f} %  {v:16,.0f}  {k}")
        out[name] = dict(tot)
    return out


if __name__ == "__main__":
    cmd = sys.argv[1] if len(sys.argv) > 1 else "build"
    if cmd == "build": build()
    elif cmd == "run": run(sys.argv[2:])
    elif cmd == "regions":
        a = sys.argv[2:]
        regions(a[0] if a else COUNTS, a[1] if len(a) > 1 else MAP)
    elif cmd == "report":
        a = sys.argv[2:]
        report(a[0] if a else COUNTS, a[1] if len(a) > 1 else MAP, a[2] if len(a) > 2 else None)

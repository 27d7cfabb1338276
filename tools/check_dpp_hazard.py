"""Developer tool / CPU test helper: the wait-state hazards that hipcc's hazard recognizer cannot see because one side is inside inline asm.

   python tools/check_dpp_hazard.py [extra hipcc flags, e.g. -mllvm -amdgpu-sched-strategy=iterative-ilp]

Compiles the three units of render.hip (RRT_TU = 1, 2, 3 with the Makefile's switches for each) to assembly and, per kernel, walks every basic block:
  * a VALU instruction that writes VGPR v followed within 2 instructions (s_nop N counts as N + 1) by a DPP instruction that READS v  -> hazard
    (CDNA3/4 ISA 4.5: "VALU writes VGPR -> VALU DPP reads that VGPR: 2 wait states");
  * a VALU instruction that writes EXEC (v_cmpx*, v_readfirstlane is not one) followed within 5 instructions by any DPP instruction          -> hazard.
Prints every finding with its kernel and line; exit status 1 if any.  The product's inline asm keeps its s_nop INSIDE the block whose first
instruction is the DPP read (render.hip: RRT_DPP_STEP_U32, RRT_DPP12), so a clean run is the expected state whatever the scheduler does."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _kflags import kflags

UNITS = {1: "KFLAGS", 2: "KFLAGS_RAYS", 3: "KFLAGS_LANE"}
VREG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def regs(tok):
    out = set()
    for m in VREG.finditer(tok):
        if m.group(1) is not None: out.add(int(m.group(1)))
        else: out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def compile_unit(tu, extra):
    out = f"/tmp/render_tu{tu}.s"
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", *kflags(UNITS[tu]), f"-DRRT_TU={tu}",
           "-S", "--cuda-device-only", "-o", out, os.path.join(ROOT, "rust-ray-tracer_amd", "csrc", "render.hip"), *extra]
    subprocess.run(cmd, check=True, capture_output=True)
    return out


def scan(path):
    findings, kernel = [], "?"
    window = []   # (slots_ago_weight, written vregs, writes_exec, text, lineno) of the previous instructions of this block, newest last
    for lineno, raw in enumerate(open(path), 1):
        line = raw.split(";")[0].strip()
        if not line or line.startswith((".", "//")):
            continue
        if line.endswith(":"):
            if not line.startswith(".L") and not line.startswith("BB"): kernel = line[:-1]
            window = []
            continue
        parts = line.split(None, 1)
        op, args = parts[0], (parts[1] if len(parts) > 1 else "")
        if op == "s_nop":
            window.append((int(args.strip() or "0", 0) + 1, set(), False, line, lineno))
            continue
        is_valu = op.startswith("v_")
        operands = [a.strip() for a in args.split(",")]
        if is_valu and ("dpp" in op or "row_shr" in args or "row_bcast" in args or "quad_perm" in args or "row_shl" in args or "wave_" in args):
            srcs = set()
            for a in operands[1:]: srcs |= regs(a.split(" ")[0])
            dist = 0
            for w, written, wexec, text, ln in reversed(window):
                if dist < 2 and written & srcs:
                    findings.append((kernel, lineno, f"DPP reads v{sorted(written & srcs)} written {dist} wait states earlier (line {ln}: {text})  <-  {line}"))
                if dist < 5 and wexec:
                    findings.append((kernel, lineno, f"DPP {dist} wait states after a VALU write of EXEC (line {ln}: {text})  <-  {line}"))
                dist += w
                if dist >= 5: break
        written = regs(operands[0]) if is_valu and operands else set()
        wexec = is_valu and op.startswith("v_cmpx")
        window.append((1, written, wexec, line, lineno))
        if op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc", "s_swappc")): window = []
        if len(window) > 8: window = window[-8:]
    return findings


def main(extra):
    bad = 0
    for tu in UNITS:
        path = compile_unit(tu, extra)
        f = scan(path)
        n_dpp = sum(1 for l in open(path) if "_dpp" in l or "row_shr" in l)
        print(f"RRT_TU={tu} ({UNITS[tu]}): {n_dpp} DPP instructions, {len(f)} hazards")
        for k, ln, msg in f[:20]: print(f"  {k}:{ln}: {msg}")
        bad += len(f)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))

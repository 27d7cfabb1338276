"""Code generation check (CPU, no GPU): the wait states that hipcc's hazard recognizer cannot see because one side is inside inline asm.

The kernels use DPP reductions written as inline asm (render.hip: wave_min_u32, wave_min12_f32).  A VALU write followed by a DPP read of the same VGPR
needs two wait states and nothing inserts them around inline asm; round 2 kept the s_nop in a statement of its own, where another scheduler could (and,
with -amdgpu-sched-strategy=iterative-ilp, did) move the producer next to the DPP read: wrong pixels on hardware.  tools/check_dpp_hazard.py compiles the
three units of render.hip with the product's switches and scans every kernel for such pairs."""
import os
import subprocess
import sys

from conftest import ROOT


def test_no_dpp_wait_state_hazard_in_the_product_kernels():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_dpp_hazard.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("0 hazards") == 3 and "DPP instructions" in r.stdout, r.stdout

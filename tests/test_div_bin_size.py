"""The lidar's two divisions by bin_size are evaluated WITHOUT a division in the kernels (gx_device.h:div_bin16: one
multiplication and two fused multiply-adds with the constant's reciprocal).  That is only admissible if it returns the
correctly rounded quotient -- the bits of the checker's `/` -- for every input the kernel sends through it.  This test
is the proof: gcc builds tests/div_bin_size_check.c, which compares the two over EVERY fp32 value of the admitted range
(2^-100 <= |x| <= 2 pi + margin, both signs, and +0: 1.7e9 inputs) and must report zero mismatches; it also shows that
the identity does fail below the bound (so the kernel's guard is necessary)."""
import os
import re
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def test_div_bin16_is_the_correctly_rounded_quotient_on_its_whole_domain():
    exe = os.path.join(tempfile.mkdtemp(), "div_bin_size_check")
    subprocess.check_call(["gcc", "-O2", "-fopenmp", "-ffp-contract=off", "-o", exe,
                           os.path.join(HERE, "div_bin_size_check.c"), "-lm"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    m = re.search(r"checked (\d+) mismatches (\d+) below_bound_mismatches (\d+)", out.stdout)
    assert m, out.stdout + out.stderr
    n, bad, below = map(int, m.groups())
    assert out.returncode == 0 and bad == 0, out.stdout
    assert n > 1_700_000_000          # every value of the range was visited, both signs (103 binades x 2^23 x 2)
    assert below > 0                   # ... and the bound is not decorative


def test_kernel_constants_are_the_ones_proved():
    src = open(os.path.join(ROOT, "guardx_amd", "csrc", "gx_device.h")).read()
    assert "kBinSize16 = 0x1.921fb6p-2f, kInvBinSize16 = 0x1.45f306p+1f" in src
    assert "kDivFastMinBits = 0x0D800000u" in src and "kDivFastBins = 16" in src
    assert float.fromhex("0x1.921fb6p-2") == __import__("numpy").float32(2 * 3.141592653589793 / 16)

"""bench.py quotes profile numbers (PMC traffic, kernel times, VALU counts): they are READ from the committed rocprofv3
summaries under profiles/ (tools/profile_evidence.py), and those summaries must have been taken on THIS build of the
library -- a kernel change without re-running tools/collect_profiles.sh fails here instead of leaving stale evidence."""
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import profile_evidence as pe  # noqa: E402


def test_profiles_were_taken_on_this_build():
    from guardx_amd import build as gx_build
    have, compiler = pe.build_id()
    assert have is not None, f"profiles/{pe.TAG}_build_id.txt is missing: run tools/collect_profiles.sh {pe.TAG} through gpurun"
    assert have == gx_build.source_hash(), (
        f"profiles/{pe.TAG}_* were taken on build {have}; the tree is {gx_build.source_hash()}: the kernels changed, "
        f"re-run tools/collect_profiles.sh {pe.TAG} on the GPU box and copy gpurun_out/{pe.TAG}_* into profiles/")
    assert compiler == gx_build.compiler_id()


def test_quoted_numbers_come_out_of_the_files():
    r = pe.rollout_numbers()
    # one gx_rollout call = one dynamics launch over 32 one-wave workgroups + one observation launch over 400 000 rows
    assert r["dyn_waves"] == 32 and 50 < r["dyn_us"] < 200 and 15 < r["obs_us"] < 80
    algo = 372 * 2000 * 200
    total = r["dyn_bytes"] + r["obs_bytes"]
    assert 0.6 * algo < total < 1.2 * algo, (total, algo)            # no wasted re-reads: traffic ~ algorithmic bytes
    # the slim tape: the dynamics pass writes 36 B per env-step (+ the entry records), not 80 (round 2), 48 (round 3) or
    # 40 (round 4)
    assert 0.9 * 36 * 400_000 < r["dyn_write_kb"] * 1024 < 1.15 * 36 * 400_000, r["dyn_write_kb"]
    s = pe.step_large_numbers()
    assert 370 < s["bytes_per_env"] < 400, s["bytes_per_env"]          # 380 B per env-step, the kernel's own byte count
    n = pe.epoch_valu_instructions()
    assert 1.5e8 < n < 3.5e8
    smp = pe.sampler_numbers()
    assert smp["sample_phase1_kernel_us"] > smp["sample_phase2_kernel_us"] > smp["scan_compact_kernel_us"]


def test_bench_has_no_typed_in_profile_numbers():
    """the literals the round-2 verdict listed (PMC KB, kernel us, 301e6 instructions) are gone from bench.py"""
    src = open(os.path.join(ROOT, "bench.py")).read()
    for lit in ("1742.6", "16662.1", "31500.0", "71875.0", "118.3", "33.8", "301e6"):
        assert lit not in src, lit
    assert "profile_evidence" in src and re.search(r"_evidence\(\"rollout_numbers\"\)", src)


def test_readme_table_of_this_round_is_what_the_files_say():
    """profiles/README.md's table for the current round is generated (python tools/profile_evidence.py --readme): build id,
    kernel microseconds, PMC megabytes, roofline fraction, bench values, test counts are READ from the CSV / JSON / log
    files -- the committed prose cannot go stale without this test failing."""
    text = open(pe.README).read()
    assert pe.BEGIN in text and pe.END in text, "run `python tools/profile_evidence.py --readme`"
    have = text[text.index(pe.BEGIN):text.index(pe.END) + len(pe.END)]
    assert have == pe.readme_block(), "profiles/README.md differs from the evidence files: run `python tools/profile_evidence.py --readme`"
    assert f"`{pe.TAG}_build_id.txt`" in have and f"`{pe.TAG}_gputest_final.log`" in have


def test_design_epoch_table_is_what_the_bench_line_says():
    """DESIGN.md section 5's table of epoch rates is generated from profiles/<round>_bench_driver_style.json too"""
    text = open(pe.DESIGN).read()
    assert pe.D_BEGIN in text and pe.D_END in text
    have = text[text.index(pe.D_BEGIN):text.index(pe.D_END) + len(pe.D_END)]
    assert have == pe.design_epoch_table(), "DESIGN.md section 5 differs from the bench line: run `python tools/profile_evidence.py --readme`"

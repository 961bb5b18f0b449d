#!/usr/bin/env python3
"""The profile numbers bench.py quotes, read from the committed rocprofv3 summaries under profiles/ -- not typed in.

profiles/<TAG>_build_id.txt names the library build the summaries were taken on; tests/test_profiles_evidence.py fails
when that is not the build of the tree (so a kernel change without re-taking the profiles fails CI), and bench.py then
reports `traffic: null` instead of a stale number.  Collected by tools/collect_profiles.sh on the GPU box.
"""
import csv
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROFILES = os.path.join(ROOT, "profiles")
TAG = "r04"


def path(name):
    return os.path.join(PROFILES, f"{TAG}_{name}")


def build_id():
    """(build id, compiler) the TAG profiles were taken on, or (None, None)"""
    try:
        lines = open(path("build_id.txt")).read().splitlines()
        return lines[0].strip(), (lines[1].strip() if len(lines) > 1 else "")
    except OSError:
        return None, None


def kernel_avg_us(csv_name, kernel_substr):
    """average duration (us) and call count of the first kernel whose name contains `kernel_substr` in a
    `rocprofv3 --kernel-trace --stats` kernel_stats.csv"""
    with open(path(csv_name), newline="") as f:
        for r in csv.DictReader(f):
            if kernel_substr in r["Name"]:
                return float(r["AverageNs"]) / 1e3, int(r["Calls"])
    raise KeyError(f"{kernel_substr} not in {csv_name}")


def pmc_mean(csv_name, kernel_substr, counter):
    """per-dispatch mean of `counter` for the first kernel matching `kernel_substr` in a tools/pmc_means.py summary"""
    with open(path(csv_name), newline="") as f:
        for r in csv.DictReader(f):
            if kernel_substr in r["Kernel_Name"] and r["Counter_Name"] == counter:
                return float(r["Mean"])
    raise KeyError(f"{kernel_substr}/{counter} not in {csv_name}")


def hbm_traffic_bytes(prefix, kernel_substr):
    """HBM bytes per dispatch = 2 * FETCH_SIZE + WRITE_SIZE (KB; FETCH_SIZE counts 64-byte units as 32 on gfx950's TCC:
    the x2 of /opt/skills/guides/MI355X_MICROARCH.md) from the two separate --pmc passes"""
    f = pmc_mean(f"{prefix}_pmc_FETCH_SIZE.csv", kernel_substr, "FETCH_SIZE")
    w = pmc_mean(f"{prefix}_pmc_WRITE_SIZE.csv", kernel_substr, "WRITE_SIZE")
    return (2.0 * f + w) * 1024.0, f, w


def rollout_numbers():
    """the two launches of one gx_rollout call at env_num=2000, T=200 (standalone)"""
    pre = "rollout_N2000_T200"
    dyn_us, _ = kernel_avg_us(f"{pre}_kernel_stats.csv", "dyn_tape_kernel")
    obs_us, _ = kernel_avg_us(f"{pre}_kernel_stats.csv", "obs_tape_kernel")
    dyn_b, dyn_f, dyn_w = hbm_traffic_bytes(pre, "dyn_tape_kernel")
    obs_b, obs_f, obs_w = hbm_traffic_bytes(pre, "obs_tape_kernel")
    return dict(dyn_us=dyn_us, obs_us=obs_us, dyn_bytes=dyn_b, obs_bytes=obs_b, dyn_fetch_kb=dyn_f, dyn_write_kb=dyn_w,
                obs_fetch_kb=obs_f, obs_write_kb=obs_w,
                dyn_valu=pmc_mean(f"{pre}_pmc_SQ.csv", "dyn_tape_kernel", "SQ_INSTS_VALU"),
                dyn_salu=pmc_mean(f"{pre}_pmc_SQ.csv", "dyn_tape_kernel", "SQ_INSTS_SALU"),
                dyn_waves=pmc_mean(f"{pre}_pmc_SQ.csv", "dyn_tape_kernel", "SQ_WAVES"),
                obs_valu=pmc_mean(f"{pre}_pmc_SQ.csv", "obs_tape_kernel", "SQ_INSTS_VALU"),
                files=[f"profiles/{TAG}_{pre}_kernel_stats.csv", f"profiles/{TAG}_{pre}_pmc_FETCH_SIZE.csv",
                       f"profiles/{TAG}_{pre}_pmc_WRITE_SIZE.csv", f"profiles/{TAG}_{pre}_pmc_SQ.csv"])


def step_large_numbers():
    """step_kernel at 2^22 envs"""
    pre = "step_N4194304"
    b, f, w = hbm_traffic_bytes(pre, "step_kernel")
    us, _ = kernel_avg_us(f"{pre}_kernel_stats.csv", "step_kernel")
    return dict(us=us, bytes=b, fetch_kb=f, write_kb=w, bytes_per_env=b / (1 << 22),
                files=[f"profiles/{TAG}_{pre}_kernel_stats.csv", f"profiles/{TAG}_{pre}_pmc_FETCH_SIZE.csv",
                       f"profiles/{TAG}_{pre}_pmc_WRITE_SIZE.csv"])


def sampler_numbers():
    """one inline reset(): kernel times and VALU wave-instructions of the layout sampler"""
    out = dict(files=[f"profiles/{TAG}_sampler_kernel_stats.csv", f"profiles/{TAG}_sampler_pmc_SQ.csv"])
    for k in ("sample_phase0_kernel", "sample_phase1_kernel", "sample_phase2_kernel", "scan_compact_kernel", "reset_apply_kernel"):
        out[k + "_us"] = kernel_avg_us("sampler_kernel_stats.csv", k)[0]
        out[k + "_valu"] = pmc_mean("sampler_pmc_SQ.csv", k, "SQ_INSTS_VALU")
    return out


def epoch_valu_instructions():
    """VALU wave-instructions of one headline epoch: the sampler's kernels + the two rollout kernels"""
    s, r = sampler_numbers(), rollout_numbers()
    return sum(v for k, v in s.items() if k.endswith("_valu")) + r["dyn_valu"] + r["obs_valu"]


if __name__ == "__main__":
    import json
    print(json.dumps(dict(build=build_id(), rollout=rollout_numbers(), step_large=step_large_numbers(),
                          sampler=sampler_numbers(), epoch_valu=epoch_valu_instructions()), indent=1))

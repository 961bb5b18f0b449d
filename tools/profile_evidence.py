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
TAG = "r05"


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


# ---------------------------------------------------------------------------------------------------------------------
# profiles/README.md: the round's table is GENERATED from the files (python tools/profile_evidence.py --readme), and
# tests/test_profiles_evidence.py fails when the committed text differs from what the files say.
# ---------------------------------------------------------------------------------------------------------------------
README = os.path.join(PROFILES, "README.md")
BEGIN, END = f"<!-- BEGIN GENERATED {TAG} (tools/profile_evidence.py --readme) -->", f"<!-- END GENERATED {TAG} -->"


def _have(name):
    return os.path.exists(path(name))


def _json_line(name):
    import json
    with open(path(name)) as f:
        return json.loads([ln for ln in f if ln.startswith("{")][-1])


def _kt(csv_name, substr):
    try:
        return f"{kernel_avg_us(csv_name, substr)[0]:.1f}"
    except (OSError, KeyError):
        return "n/a"


def _pmc(csv_name, substr, counter, scale=1.0, fmt="{:.1f}"):
    try:
        return fmt.format(pmc_mean(csv_name, substr, counter) * scale)
    except (OSError, KeyError):
        return "n/a"


def _M(x):
    return f"{x / 1e6:.1f} M"


def readme_rows():
    """[(files, command, what it shows)] for every evidence file of TAG that exists; every number is read from the file"""
    rows = []
    bid, comp = build_id()
    if bid:
        rows.append((f"`{TAG}_build_id.txt`", "`gx_build_id()` / `gx_build_compiler()` of the library the profiles were taken on",
                     f"build `{bid}` ({comp.split(';')[0]}); must equal the tree's build (`tests/test_profiles_evidence.py`)"))
    pre = "rollout_N2000_T200"
    if _have(f"{pre}_kernel_stats.csv"):
        r = rollout_numbers()
        tot = r["dyn_bytes"] + r["obs_bytes"]
        algo = 372 * 2000 * 200
        rows.append((f"`{TAG}_{pre}_kernel_stats.csv`, `..._pmc_{{FETCH_SIZE,WRITE_SIZE,SQ}}.csv`",
                     "`rocprofv3 --kernel-trace --stats` / `--pmc ...` `-- python3 tools/profile_step.py --mode rollout --env-num 2000 "
                     "--launches 200 --repeat 20`",
                     f"one `gx_rollout` call standalone: `dyn_tape_kernel` {r['dyn_us']:.1f} us ({r['dyn_waves']:.0f} waves, "
                     f"{r['dyn_valu'] / r['dyn_waves'] / 200:.0f} VALU + {r['dyn_salu'] / r['dyn_waves'] / 200:.0f} SALU per step) + "
                     f"`obs_tape_kernel` {r['obs_us']:.1f} us; PMC traffic 2 x FETCH + WRITE: dynamics pass "
                     f"{r['dyn_bytes'] / 1e6:.1f} MB, observation pass {r['obs_bytes'] / 1e6:.1f} MB = {tot / 1e6:.1f} MB against "
                     f"{algo / 1e6:.1f} MB algorithmic ({tot / algo:.2f}); `roofline.frac` from these two durations = "
                     f"{algo / ((r['dyn_us'] + r['obs_us']) * 1e-6) / 8e12:.3f}"))
    if _have("step_N4194304_kernel_stats.csv"):
        st = step_large_numbers()
        rows.append((f"`{TAG}_step_N4194304_*`, `{TAG}_thread_rollout_N4194304_K16_pmc_*`",
                     "`... --mode step --env-num 4194304 --launches 20` / `--mode rollout --env-num 4194304 --launches 16`",
                     f"bandwidth regime: `step_kernel` {st['us']:.1f} us under the profiler, {st['bytes_per_env']:.2f} B per env-step "
                     f"(PMC) against 372 algorithmic = {372 * (1 << 22) / (st['us'] * 1e-6) / 8e12:.3f} of peak"))
    parts = []
    for rb, kern in (("swimmer", "dyn_tape_kernel"), ("ant", "group_dyn_tape_kernel"), ("walker", "group_dyn_tape_kernel")):
        f = f"{rb}_rollout_N2000_T200_kernel_stats.csv"
        if _have(f):
            sq = f"{rb}_rollout_N2000_T200_pmc_SQ.csv"
            parts.append(f"{rb.capitalize()} dynamics pass {_kt(f, kern)} us + observation pass {_kt(f, 'obs_tape_kernel')} us per 200 "
                         f"steps ({_pmc(sq, kern, 'SQ_WAVES', 1, '{:.0f}')} waves, "
                         f"{_pmc(sq, kern, 'SQ_INSTS_VALU', 1e-6)} M VALU wave-instructions)")
    if parts:
        rows.append((f"`{TAG}_{{swimmer,ant,walker}}_rollout_N2000_T200_{{kernel_stats,pmc_SQ}}.csv`",
                     "`... --robot xmls/<robot>.xml --repeat 10`", "; ".join(parts)))
    for name, task in (("sampler", "the reference's arena"), ("sampler_config5", "the synthetic config 5 (18 objects, 6 m)")):
        f = f"{name}_kernel_stats.csv"
        if _have(f):
            ph = [(k, _kt(f, k)) for k in ("sample_phase0_kernel", "sample_phase1_kernel", "sample_phase2_kernel", "scan_compact_kernel")]
            valu = [_pmc(f"{name}_pmc_SQ.csv", k, "SQ_INSTS_VALU", 1e-6) for k, _ in ph[:3]]
            rows.append((f"`{TAG}_{name}_kernel_stats.csv`, `{TAG}_{name}_pmc_SQ.csv`",
                         "`... -- python3 tools/profile_reset.py`" + (" `--task Ant_8Hazards_8Pillars_synthetic`" if "config5" in name else ""),
                         f"inline `reset()`, {task}: phases 0 / 1 / 2 / compaction {' / '.join(u for _, u in ph)} us "
                         f"(n/a = the form without that phase), {' / '.join(valu)} M VALU wave-instructions"))
    if _have("bench_noextras_kernel_stats.csv"):
        f = "bench_noextras_kernel_stats.csv"
        rows.append((f"`{TAG}_{f}`", "`... -- python3 bench.py --no-cpu-baseline --no-extras`",
                     f"the headline run under the profiler, per epoch: `sample_phase0/1/2` {_kt(f, 'sample_phase0')} / "
                     f"{_kt(f, 'sample_phase1')} / {_kt(f, 'sample_phase2')} us on the side stream; `reset_apply` {_kt(f, 'reset_apply')}, "
                     f"`dyn_tape` {_kt(f, 'dyn_tape_kernel')}, `obs_tape` {_kt(f, 'obs_tape_kernel')} us on the caller's stream beside them"))
    if _have("policy_widths_kernel_stats.csv"):
        f = "policy_widths_kernel_stats.csv"
        rows.append((f"`{TAG}_{f}`", "`... -- python3 tools/bench_policy_widths.py`",
                     f"closed-loop policy rollout by width: the fused kernels (one launch per 200 steps) `group_rollout_kernel<.., 2>` "
                     f"(64, weights in LDS) {_kt(f, ', true, 2>(')} us, `<.., 3>` (128, weights in registers) {_kt(f, ', true, 3>(')} us, "
                     f"`<.., 192>` / `<.., 256>` (weights streamed from L2) {_kt(f, ', true, 192>(')} / {_kt(f, ', true, 256>(')} us"))
    for name, cmd in (("bench_driver_style.json", "`python bench.py --steps 20 --warmup 5`"), ("bench_full.json", "`python bench.py`")):
        if _have(name):
            l = _json_line(name)
            oth = l.get("other_robots", {})
            reps = l.get("repetitions", {})
            api = l.get("api_step_loop_env_steps_per_s", {})
            wide = l.get("closed_loop_policy_wider_env_steps_per_s", {})
            mg = l.get("multi_gpu_rehearsal", {})
            txt = (f"`value` {_M(l['value'])} env-steps/s = the median of {reps.get('n')} un-preconditioned repetitions "
                   f"(min {_M(reps.get('min', 0))}, max {_M(reps.get('max', 0))}, first {_M(reps.get('first', 0))}); "
                   f"`preconditioned` {_M(l.get('preconditioned', {}).get('value', 0))}; `roofline.frac` {l['roofline'].get('frac')}; "
                   f"other robots " + " / ".join(_M(v['env_steps_per_s']) for v in oth.values() if isinstance(v, dict)) +
                   f"; api loop {_M(api.get('value', 0))} (ring of 8: {_M(api.get('out_ring_8', {}).get('value', 0))}); closed loop "
                   f"{_M(l.get('closed_loop_policy_env_steps_per_s', 0))} (hidden 128 / 256: "
                   f"{_M(wide.get('hidden_128', 0))} / {_M(wide.get('hidden_256', 0))}); rank rehearsal at W = 8: "
                   f"{mg.get('expand_all', {}).get('ms_per_epoch')} ms per epoch; `vs_previous_round.regressions` = "
                   f"{l.get('vs_previous_round', {}).get('regressions')}; `cpu_baseline` "
                   f"{l.get('cpu_baseline', {}).get('value')} env-steps/s on {l.get('cpu_baseline', {}).get('cores')} cores")
            rows.append((f"`{TAG}_{name}`", cmd, txt))
    for name, lbl, W in (("rehearsal_rank0_of_8.json", "Point", 8), ("rehearsal_rank0_of_8_swimmer.json", "Swimmer", 8),
                         ("rehearsal_rank0_of_8_ant.json", "Ant", 8), ("rehearsal_rank0_of_8_walker.json", "Walker", 8),
                         ("rehearsal_rank0_of_4.json", "Point", 4), ("rehearsal_rank0_of_4_ant.json", "Ant", 4),
                         ("rehearsal_rank0_of_2.json", "Point", 2), ("rehearsal_rank0_of_2_ant.json", "Ant", 2)):
        if _have(name):
            import json
            d = json.load(open(path(name)))
            m = d["expand_all"]["model"]
            rows.append((f"`{TAG}_{name}`", f"`python tools/rehearse_rank.py --world {W} --epochs 30" +
                         ("" if lbl == "Point" else f" --robot xmls/{lbl.lower()}.xml") + " --json ...`",
                         f"one GPU playing rank 0 of {W}, {lbl}: {d['expand_all']['ms_per_epoch']} ms per rank epoch (`expand=\"local\"`: "
                         f"{d['expand_local']['ms_per_epoch']}) against {d['one_gpu_own_sampler']['ms_per_epoch']} ms on one GPU; "
                         f"{d['expand_all']['bytes_received_per_epoch'] / 1e6:.1f} MB received per epoch: link-bound below "
                         f"{m.get('link_bound_below_GBps')} GB/s, break-even with one GPU at {m.get('break_even_GBps')} GB/s (model, not a "
                         f"measurement of the link); the other ranks' slots: {d.get('link_stand_in', 'device copies')}"))
    if _have("rccl_one_rank_report.json"):
        import json
        d = json.load(open(path("rccl_one_rank_report.json")))
        th = d.get("tape_handoff", {})
        rows.append((f"`{TAG}_rccl_one_rank_report.json`, `{TAG}_rccl_one_rank_nccl.log`",
                     "`pytest tests/test_rccl_one_rank.py` (child: `tests/rccl_one_rank_child.py`, `GX_FORCE_DIST=1`, `NCCL_DEBUG=INFO`)",
                     f"the N > 1 path over a real one-rank RCCL {d.get('nccl_version')} group: {len(d.get('calls', []))} kinds of calls ran "
                     f"(init, barrier, all_reduce, async all_gather_into_tensor on device receive rings, ShardedReset, destroy); "
                     f"TapeHandoff {th.get('epochs')} epochs, {th.get('blocks_installed')} shard blocks installed, rows bit-equal to the "
                     f"packed rollout and to the CPU checker"))
    if _have("bench_force_dist_one_rank.json"):
        l = _json_line("bench_force_dist_one_rank.json")
        rows.append((f"`{TAG}_bench_force_dist_one_rank.json`", "`GX_FORCE_DIST=1 python bench.py --gpus 1 --steps 4 --warmup 2 --no-extras`",
                     f"the N > 1 bench path over a one-rank `{l.get('forced_dist', {}).get('backend')}` group (a code-path run, not a "
                     f"rate): all legs present -- `value` {_M(l['value'])}, stepping_only {_M(l['stepping_only']['value'])}, "
                     f"unsharded_sampler {_M(l['legs']['unsharded_sampler']['value'])}, local_expand {_M(l['legs']['local_expand']['value'])}"))
    if _have("gputest_final.log"):
        last = [ln.strip() for ln in open(path("gputest_final.log")) if " passed" in ln or " failed" in ln]
        rows.append((f"`{TAG}_gputest_final.log`", "`python -m pytest tests -m gpu -q`", last[-1] if last else "(no summary line)"))
    soaks = []
    for rb in ("point", "swimmer", "ant", "walker"):
        for kind in ("soak", "soak_variants"):
            f = f"{kind}_{rb}.log"
            if _have(f):
                lines = [ln.strip() for ln in open(path(f)) if ln.strip()]
                soaks.append(f"{kind} {rb}: {lines[-1][:110] if lines else 'empty'}")
    if soaks:
        rows.append((f"`{TAG}_soak_{{point,swimmer,ant,walker}}.log`, `{TAG}_soak_variants_*.log`",
                     "`python tests/soak_parity.py <robot> 300000 4096 400`, `python tests/soak_variants.py <robot> 2`",
                     "HIP vs CPU restatement on the final build -- last line of each log: " + "; ".join(soaks)))
    import glob
    longs = sorted(glob.glob(path("soak_long_*.log")))
    if longs:
        res = []
        for f in longs:
            lines = [ln.strip() for ln in open(f) if ln.strip()]
            res.append(os.path.basename(f).replace(TAG + "_soak_long_", "").replace(".log", "") + ": " + (lines[-1] if lines else "empty"))
        rows.append((f"`{TAG}_soak_long_*.log`", "`python tests/soak_parity.py <robot> 1500000 8192 600`",
                     "the same soak, five times the states and twice the envs, for the robots whose solves changed last (the "
                     "pivots' reciprocals) -- " + "; ".join(res)))
    hs = sorted(glob.glob(path("soak_handoff_*.log")))
    if hs:
        res = []
        for f in hs:
            lines = [ln.strip() for ln in open(f) if ln.strip()]
            head = [ln for ln in lines if ln.startswith("soak_handoff ")]
            res.append((head[-1].split(":")[0].replace("soak_handoff ", "") if head else os.path.basename(f)) + ": " + (lines[-1] if lines else "empty"))
        rows.append((f"`{TAG}_soak_handoff_*.log`", "`python tests/soak_handoff.py <robot> <W> <epochs> <N> <T>`",
                     "the N > 1 epoch (sharded sampler, every rank expands every tape; W ranks in one process, the collective played by a "
                     "stream) against ONE engine of W x N envs, every expanded row, reset observation and pool -- " + "; ".join(res)))
    ab = sorted(os.path.basename(f) for f in glob.glob(path("ab_*.log")))
    if ab:
        rows.append((", ".join(f"`{f}`" for f in ab), "`tools/ab/*.sh` (same-box A/B runs of library variants / trees: `tools/build_variant.py`, `tools/ab_epoch.py`)",
                     "the in-epoch A/B logs behind this round's decisions (DESIGN.md section 5): the Swimmer regression bisected to the "
                     "capped observation grid, `reset_apply` at priority, the sampler's fused form, the hand-off stream"))
    for h in (128, 256):
        f = f"example_ppo_fused_hid{h}.log"
        if _have(f):
            rows_ = [ln.split() for ln in open(path(f)) if ln.strip() and ln.split()[0].isdigit()]
            if rows_:
                a, b = rows_[0], rows_[-1]
                rows.append((f"`{TAG}_{f}`", f"`python examples/train_ppo_fused.py --epochs 25 --hid {h}`",
                             f"PPO-clip with ({h},{h}) actor and critic, the collection phase one fused launch per 400k-step epoch "
                             f"({b[5]} ms): return {a[1]} -> {b[1]}, episode length {a[3]} -> {b[3]}, goals reached per env per epoch "
                             f"{a[4]} -> {b[4]} in {len(rows_)} epochs"))
    if _have("mfma_rate_probe.log"):
        first = [ln.strip() for ln in open(path("mfma_rate_probe.log")) if ln.startswith("v_mfma")]
        rows.append((f"`{TAG}_mfma_rate_probe.log`", "`tools/probes/mfma_rate_probe` (one wave per SIMD, independent accumulator chains)",
                     first[0] if first else ""))
    if _have("rcp_exact_probe.log"):
        first = [ln.strip() for ln in open(path("rcp_exact_probe.log")) if ln.startswith(("A ", "D ", "E ", "F "))]
        rows.append((f"`{TAG}_rcp_exact_probe.log`", "`tools/probes/rcp_exact_probe` (all 2^32 inputs)", "; ".join(first)))
    return rows


def readme_block():
    lines = [BEGIN, "", "| file | command | what it shows |", "|---|---|---|"]
    for files, cmd, what in readme_rows():
        lines.append(f"| {files} | {cmd} | {what} |")
    lines += ["", END]
    return "\n".join(lines)


DESIGN = os.path.join(ROOT, "DESIGN.md")
D_BEGIN, D_END = "<!-- BEGIN GENERATED epoch table (tools/profile_evidence.py --readme) -->", "<!-- END GENERATED epoch table -->"


def design_epoch_table():
    """DESIGN.md section 5's table of epoch rates, from the committed driver-style bench line and the previous round's record"""
    import json
    l = _json_line("bench_driver_style.json")
    vp = l.get("vs_previous_round", {}).get("values", {})

    def prev(key):
        return f"{vp[key]['previous'] / 1e6:.1f} M" if key in vp else "n/a"
    reps = l["repetitions"]
    rows = [D_BEGIN, "",
            f"| per 200-step epoch at env_num = 2000 (`profiles/{TAG}_bench_driver_style.json`: `value` and, in a fresh child process "
            f"after 30 warm-up epochs, `other_robots`) | ms | env-steps/s | round {int(l['vs_previous_round']['previous'][1:])} (driver record) |",
            "|---|---|---|---|",
            f"| Goal_Point_8Hazards (`value`: median of {reps['n']} repetitions, {reps['min'] / 1e6:.0f} … {reps['max'] / 1e6:.0f} M; "
            f"`preconditioned` {l['preconditioned']['value'] / 1e6:.0f} M) | {l['ms_per_step']:.3f} | **{l['value'] / 1e6:.1f} M** | "
            f"{prev('value')} (preconditioned) |"]
    for name, v in l["other_robots"].items():
        if isinstance(v, dict):
            rows.append(f"| {name} | {v['ms_per_epoch']:.3f} | {v['env_steps_per_s'] / 1e6:.1f} M | {prev('other_robots.' + name)} |")
    rh = l["reset_done_heavy"]
    rows.append(f"| Point, every env re-initialised ≈ 3 times per epoch (`reset_done_heavy`) | {rh['ms_per_epoch']:.3f} | "
                f"{rh['env_steps_per_s'] / 1e6:.1f} M | {prev('reset_done_heavy')} (builder's line) |")
    api = l["api_step_loop_env_steps_per_s"]
    wide = l["closed_loop_policy_wider_env_steps_per_s"]
    ab_api = ""
    if _have("ab_api.log"):
        import json
        best = {}
        for ln in open(path("ab_api.log")):
            try:
                d = json.loads(ln)
                best.setdefault(d["tag"], []).append(d["best_M"])
            except (ValueError, KeyError):
                pass
        if "r04" in best and "new" in best:
            ab_api = (" (same box, the round-4 tree against this round's: best of seven " + " / ".join(f"{v:.1f}" for v in best["r04"]) +
                      " M vs " + " / ".join(f"{v:.1f}" for v in best["new"]) + " M: what the line compares with is another box's host)")
    rows += ["", f"(`vs_previous_round.regressions` of that line: {l['vs_previous_round']['regressions']}. The Python-driven "
                 f"`step()+reset_done()` loop: {api['value'] / 1e6:.0f} M, ring of 8: {api['out_ring_8']['value'] / 1e6:.0f} M — host-bound, "
                 f"± 10 % between consecutive medians of one library on one box, `profiles/{TAG}_ab_api.log`{ab_api}; `reset_done_heavy` moves ± 2 % "
                 f"from box to box (712–730 M over the round's lines) and not at all between builds on one box, "
                 f"`profiles/{TAG}_ab_rdh_final.log`. Closed loop, hidden 64 / 128 / "
                 f"256: {l['closed_loop_policy_env_steps_per_s'] / 1e6:.0f} / {wide['hidden_128'] / 1e6:.0f} / {wide['hidden_256'] / 1e6:.0f} M.)",
             "", D_END]
    return "\n".join(rows)


def write_design():
    text = open(DESIGN).read()
    if D_BEGIN in text and D_END in text:
        a, b = text.index(D_BEGIN), text.index(D_END) + len(D_END)
        text = text[:a] + design_epoch_table() + text[b:]
        with open(DESIGN, "w") as f:
            f.write(text)


def write_readme():
    text = open(README).read()
    block = readme_block()
    if BEGIN in text and END in text:
        a, b = text.index(BEGIN), text.index(END) + len(END)
        text = text[:a] + block + text[b:]
    else:
        marker = "## Round 4"
        head = f"## Round {int(TAG[1:])}\n\nEvery number in this table is read from the files by `tools/profile_evidence.py --readme`; " \
               f"`tests/test_profiles_evidence.py` fails when this text and the files disagree.\n\n"
        a = text.index(marker)
        text = text[:a] + head + block + "\n\n" + text[a:]
    with open(README, "w") as f:
        f.write(text)


if __name__ == "__main__":
    import sys
    if "--readme" in sys.argv:
        write_readme()
        write_design()
        print(readme_block())
        sys.exit(0)
    import json
    print(json.dumps(dict(build=build_id(), rollout=rollout_numbers(), step_large=step_large_numbers(),
                          sampler=sampler_numbers(), epoch_valu=epoch_valu_instructions()), indent=1))

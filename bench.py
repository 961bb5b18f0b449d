#!/usr/bin/env python3
"""Headline benchmark: env-steps/sec of the GUARD batched environment step,
Goal_Point_8Hazards, env_num=2000 per GPU, random-policy rollout (BASELINE.json).

One bench "step" (--steps K, --warmup W) is ONE EPOCH of the hot path over the batch: reset()
(the reference's 1e6-candidate layout resampling, engine.py:433-467) followed by max_ep_len = 200
passes of Engine.step for all envs + reset_done for the envs that finished (device-side done test;
identical results to the learner's `if done.any(): reset_done()`), SURVEY.md section 8d.  So
`--steps 20 --warmup 5` times 20 epochs = 4000 step passes = 8 M env-steps per GPU after 5 warm-up
epochs.  `value` = env_num * 200 * K * n_gpus / wall.  Inputs (the action tapes) are resident in HBM
before the timed region.  Prints ONE JSON line on rank 0.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

`python bench.py --gpus N` with no WORLD_SIZE in the environment starts the N rank processes itself
(fresh children, spawned before this process touches the GPU) and fails if fewer than N join.
"""
import argparse
import json
import os
import sys
import socket
import subprocess
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

TASK = {   # Goal_Point_8Hazards, safe_rl_libX/guard_utils/safe_rl_env_config.py:59-81
    'robot_base': 'xmls/point.xml', 'task': 'goal', 'goal_size': 0.5,
    'observe_goal_comp': True, 'observe_hazards': True,
    'constrain_hazards': True, 'constrain_indicator': False,
    'lidar_num_bins': 16, 'hazards_num': 8, 'hazards_size': 0.3,
}
ENV_NUM = 2000
EP_LEN = 200
ALGO_BYTES_PER_ENV_STEP = 372     # 124 B read + 248 B written
HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md: 8 TB/s spec


def make_engine(env_num, rank, world, seed=0, n_candidates=1_000_000, robot_base=None):
    from guardx_amd import Engine
    cfg = dict(TASK)
    if robot_base:
        cfg['robot_base'] = robot_base
    cfg.update(env_num=env_num, _seed=seed, num_steps=EP_LEN, device_id=torch.cuda.current_device())
    return Engine(cfg, shard=(rank, world) if world > 1 else None, n_candidates=n_candidates)


def action_tape(T, N, seed, device, act_dim=2):
    g = torch.Generator(device=device).manual_seed(seed)
    return torch.rand(T, N, act_dim, device=device, generator=g) * 2 - 1   # a ~ U(-1,1), myTest.py:28-31


class RolloutHandoff:
    """Per-epoch hand-off of the rollout shard to the learner: ONE all-gather of the packed
    (T, N, obs+act+3) shard per epoch (RCCL over xGMI), issued asynchronously so that it overlaps
    the next epochs' stepping.  The rollout kernel writes the packed layout itself (gx_rollout_packed),
    so there is no pack pass; `depth` gathered buffers are in flight.  On the gloo rehearsal backend
    (several ranks on one GPU, no RCCL) the shard is staged through pinned host memory."""

    def __init__(self, world, depth=3):
        import torch.distributed as dist
        self.world = world
        self.depth = depth
        self.slots = [None] * depth        # (work, packed, gathered)
        self.k = 0
        self.bytes = 0
        self.host = dist.get_backend() != "nccl"

    def out_buffer(self, shape, device):
        """the gathered buffer of the slot about to be used (waits for its previous collective)"""
        i = self.k % self.depth
        s = self.slots[i]
        if s is not None:
            s[0].wait()
            if tuple(s[2].shape[1:]) == tuple(shape):
                return s[2]
        dev = "cpu" if self.host else device
        return torch.empty((self.world,) + tuple(shape), dtype=torch.float32, device=dev,
                           pin_memory=self.host and torch.cuda.is_available())

    def submit(self, packed):
        import torch.distributed as dist
        i = self.k % self.depth
        out = self.out_buffer(packed.shape, packed.device)
        self.k += 1
        T = packed.shape[0]
        src = packed
        if self.host:
            src = packed.to("cpu")          # rehearsal only
        work = dist.all_gather_into_tensor(out.view((self.world * T,) + tuple(packed.shape[1:])), src,
                                           async_op=True)
        self.slots[i] = (work, packed, out)
        self.bytes += packed.numel() * 4 * self.world

    def drain(self):
        for s in self.slots:
            if s is not None:
                s[0].wait()


def run_epochs(env, tapes, epochs, handoff):
    """`epochs` bench steps: reset() + one fused 200-pass rollout each (+ the async hand-off)."""
    for ep in range(epochs):
        env.reset(check=False)           # the layout_size assert is checked once after the loop
        acts = tapes[ep % len(tapes)]
        if isinstance(handoff, RolloutHandoff):
            handoff.submit(env.rollout(acts, packed=True)[4])
        elif handoff is not None:
            handoff.step(acts)           # TapeHandoff: dynamics pass here, observation pass on every rank
        else:
            env.rollout(acts)
    if handoff is not None:
        handoff.drain()
    env.check_layouts()                  # engine.py:444 for every reset above (one sync)


PRECONDITION_MS = 60.0


def precondition_clocks(device, ms=PRECONDITION_MS):
    """Keep the GPU busy for `ms` with work that is NOT the workload (fp32 torch.mm), so that the firmware's clock /
    power ramp is over when the W warm-up epochs start.  Measured on this pool (tools/history/debug/cold_start_probe.py,
    DESIGN.md section 6): after >= 50 ms of idle the first ~25 epochs (13 ms) run 10 % -> 0 % slower than the steady
    state whatever ran before the idle gap, and 30 ms of any sustained compute removes that.  The driver's region
    (5 + 20 epochs = 13 ms) would otherwise sit entirely inside the ramp.  The line reports `cold_start` beside.
    (GX_PRECONDITION=int / GX_PRECONDITION_MS: experiments with another kind and length of filler work.)"""
    ms = float(os.environ.get("GX_PRECONDITION_MS", ms))
    kind = os.environ.get("GX_PRECONDITION", "mm")
    if kind == "int":
        x = torch.arange(1 << 24, device=device, dtype=torch.int32)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 0
        while (time.perf_counter() - t0) * 1e3 < ms:
            for _ in range(8):
                x.mul_(1664525).add_(1013904223)
            n += 8
            torch.cuda.synchronize()
        return {"ms": round((time.perf_counter() - t0) * 1e3, 1), "work": f"{n} x int32 multiply-add over 2^24 elements (not the workload)"}
    a = torch.ones(4096, 4096, device=device)
    b = torch.ones(4096, 4096, device=device)
    c = a @ b                                            # library initialisation happens here, outside the busy loop
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 0
    while (time.perf_counter() - t0) * 1e3 < ms:
        for _ in range(4):
            torch.mm(a, b, out=c)
        n += 4
        torch.cuda.synchronize()
    return {"ms": round((time.perf_counter() - t0) * 1e3, 1), "work": f"{n} x fp32 torch.mm 4096^3 (not the workload)"}


def _fresh_engine(env_num):
    from guardx_amd import ResamplingError
    env = make_engine(env_num, 0, 1, n_candidates=200_000)
    env.set_prefetch(-1)
    try:
        env.reset()
    except ResamplingError:
        # env_num beyond the reference's own limit (engine.py:444 needs layout_size > env_num):
        # the envs are initialised from the pool anyway (drawn with replacement), which is all
        # the kernel timing needs
        assert env_num > ENV_NUM
    return env


def time_launches(launch, nlaunch):
    """Average duration of one kernel launch measured with HIP events on the launch stream
    (torch's current stream is the stream every gx_* call is given)."""
    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    # (a) one event pair per launch: kernel duration (+ event overhead)
    pairs = []
    for _ in range(nlaunch):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); launch(); e1.record()
        pairs.append((e0, e1))
    torch.cuda.synchronize()
    per = np.array([a.elapsed_time(b) for a, b in pairs]) * 1e-3
    # (b) back-to-back launches between two events: launch cadence
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(nlaunch):
        launch()
    e1.record()
    torch.cuda.synchronize()
    cadence = e0.elapsed_time(e1) * 1e-3 / nlaunch
    return float(np.median(per)), float(cadence)


def _evidence(fn_name):
    """Numbers from the committed rocprofv3 summaries (tools/profile_evidence.py), or (None, reason) when the
    summaries are missing or were taken on another build of the library than the one loaded here."""
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import profile_evidence as pe
        from guardx_amd import _native
        have, _ = pe.build_id()
        lib_id = _native.load().gx_build_id().decode()
        if have is None:
            return None, f"no profiles/{pe.TAG}_build_id.txt"
        if have != lib_id:
            return None, f"profiles/{pe.TAG}_* were taken on build {have}, this library is {lib_id}: re-run tools/collect_profiles.sh"
        return getattr(pe, fn_name)(), None
    except Exception as exc:  # noqa: BLE001 - evidence is optional, the measured numbers are not
        return None, f"{type(exc).__name__}: {exc}"[:200]


def roofline_rollout(env_num, T, nlaunch, device):
    """The step path of the headline workload: one gx_rollout call = T fused step+reset_done passes over env_num
    envs = the dynamics-tape kernel + the observation-pass kernel (gx_split_rollout.inl)."""
    env = _fresh_engine(env_num)
    tape = action_tape(T, env_num, 7, device)
    N, D = env_num, env.obs_flat_size
    obs = torch.empty(T, N, D, device=device)
    r, c, d = (torch.empty(T, N, device=device) for _ in range(3))
    from guardx_amd import _native
    import ctypes as C
    lib = _native.load()
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def launch():
        _native.check(lib.gx_rollout(env._h, T, tape.data_ptr(), obs.data_ptr(), r.data_ptr(), c.data_ptr(),
                                     d.data_ptr(), stream))
    per, cadence = time_launches(launch, nlaunch)
    env.close()
    t = min(per, cadence)
    ach = ALGO_BYTES_PER_ENV_STEP * env_num * T / t / 1e9
    out = {"bound": "hbm", "achieved": round(ach, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": round(ach / HBM_PEAK_GBS, 6), "traffic": None,
           "kernel": "gx::dyn_tape_kernel<PointRobot,64,5,true> + gx::obs_tape_kernel<PointRobot,64,5,true> "
                     "(the two launches of one gx_rollout call = 200 fused step+reset_done passes)",
           "env_num": env_num, "steps_per_launch": T,
           "avg_launch_us": round(t * 1e6, 3), "event_pair_us": round(per * 1e6, 3),
           "back_to_back_us": round(cadence * 1e6, 3),
           "algorithmic_bytes_per_env_step": ALGO_BYTES_PER_ENV_STEP,
           "note": "env_num=2000 is latency-bound by construction (0.74 MB of algorithmic traffic per step): the serial "
                   "dynamics pass (32 waves, one per SIMD) takes 3/4 of the call; see roofline_large_batch for "
                   "the bandwidth regime"}
    ev, why = _evidence("rollout_numbers")
    if ev is None:
        out["traffic_note"] = "no PMC traffic figure: " + why
        return out
    scale = (env_num * T) / (2000 * 200)
    out["traffic"] = round((ev["dyn_bytes"] + ev["obs_bytes"]) / 1e9 * scale, 4)
    out["traffic_note"] = ("NOT measured in this run: GB per gx_rollout call from the committed rocprofv3 PMC passes at "
                           "env_num=2000, T=200 (2*FETCH_SIZE + WRITE_SIZE per kernel: dynamics pass "
                           f"{ev['dyn_bytes'] / 1e6:.1f} MB, observation pass {ev['obs_bytes'] / 1e6:.1f} MB) against "
                           f"{ALGO_BYTES_PER_ENV_STEP * 2000 * 200 / 1e6:.1f} MB algorithmic; read from " + ", ".join(ev["files"][1:3]))
    out["kernels_us_rocprof"] = {"dyn_tape_kernel": round(ev["dyn_us"], 1), "obs_tape_kernel": round(ev["obs_us"], 1),
                                 "source": ev["files"][0] + " (standalone)"}
    out["obs_pass_alone"] = {"GBps_pmc_traffic": round(ev["obs_bytes"] / (ev["obs_us"] * 1e-6) / 1e9, 1),
                             "frac_of_peak": round(ev["obs_bytes"] / (ev["obs_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                             "note": "the 400k-row observation pass alone (profiled, not this run)"}
    out["dyn_pass_alone"] = {"valu_per_wave_step": round(ev["dyn_valu"] / ev["dyn_waves"] / 200, 1),
                             "salu_per_wave_step": round(ev["dyn_salu"] / ev["dyn_waves"] / 200, 1),
                             "waves": int(ev["dyn_waves"]), "ns_per_step": round(ev["dyn_us"] * 1e3 / 200, 1)}
    return out


def roofline_step(env_num, nlaunch, device):
    """The thread-per-env step kernel (one launch = one step over env_num envs)."""
    env = _fresh_engine(env_num)
    act = action_tape(1, env_num, 7, device)[0]
    N, D = env_num, env.obs_flat_size
    obs = torch.empty(N, D, device=device)
    r, c, d = (torch.empty(N, device=device) for _ in range(3))
    qacc = torch.empty(N, 3, device=device)
    from guardx_amd import _native
    import ctypes as C
    lib = _native.load()
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def launch():
        _native.check(lib.gx_step(env._h, act.data_ptr(), obs.data_ptr(), r.data_ptr(), c.data_ptr(),
                                  d.data_ptr(), qacc.data_ptr(), stream))
    per, cadence = time_launches(launch, nlaunch)
    env.close()
    t = min(per, cadence)
    ach = ALGO_BYTES_PER_ENV_STEP * env_num / t / 1e9
    out = {"bound": "hbm", "achieved": round(ach, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": round(ach / HBM_PEAK_GBS, 6), "traffic": None,
           "kernel": "gx::step_kernel<PointRobot,64,5,true,true>", "env_num": env_num,
           "avg_launch_us": round(t * 1e6, 3), "event_pair_us": round(per * 1e6, 3),
           "back_to_back_us": round(cadence * 1e6, 3),
           "algorithmic_bytes_per_env_step": ALGO_BYTES_PER_ENV_STEP}
    ev, why = _evidence("step_large_numbers")
    if ev is None:
        out["traffic_note"] = "no PMC traffic figure: " + why
    else:
        out["traffic"] = round(ev["bytes_per_env"] * env_num / 1e9, 4)
        out["traffic_note"] = ("NOT measured in this run: GB per launch from the committed rocprofv3 PMC passes at 2^22 envs "
                               f"(2*FETCH_SIZE + WRITE_SIZE = {ev['bytes_per_env']:.1f} B/env; " + ", ".join(ev["files"][1:]) + ")")
    return out


def large_batch_fused(env_num, K, device, reps=4):
    """Bandwidth regime with K steps fused per launch (SURVEY 8d "K=32 fused steps"; K=16 keeps the
    time-major outputs at 11.5 GB): Engine.rollout on the thread-per-env persistent kernel.  State, layout
    and history stay in registers, so an env-step moves action 8 + obs 172 + reward/cost/done 12 B plus
    1/K of the 180 B state round trip."""
    env = _fresh_engine(env_num)
    tape = action_tape(K, env_num, 3, device)
    for _ in range(3):      # the first two calls pay hipMalloc for the 11.5 GB of time-major outputs
        env.rollout(tape)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        env.rollout(tape)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    env.close()
    del tape
    torch.cuda.empty_cache()
    bytes_step = 192 + (ALGO_BYTES_PER_ENV_STEP - 192) / K
    ach = bytes_step * env_num * K / dt / 1e9
    return {"kernel": "gx::thread_rollout_kernel<PointRobot,64,5,true>", "env_num": env_num, "steps_per_launch": K,
            "us_per_step": round(dt / K * 1e6, 2), "env_steps_per_s": round(env_num * K / dt, 1),
            "algorithmic_bytes_per_env_step": round(bytes_step, 1), "achieved": round(ach, 2), "unit": "GB/s",
            "frac": round(ach / HBM_PEAK_GBS, 4),
            "note": "includes the in-kernel reset_done; VALU/occupancy bound (133 VGPRs), not HBM bound"}


def cpu_baseline(epochs_all=8, epochs_1t=1):
    """The CPU restatement (oracle/, 'port') timed on the host cores on a bounded sample: the same epoch
    (reset over 1e6 layout candidates + 200 x (step, reset_done if any done)) with one thread and with all
    cores.  Both phases are OpenMP-parallel in the checker (candidates in reset, envs in step)."""
    from oracle import gxo
    cfg = dict(TASK)
    cfg.update(env_num=ENV_NUM, _seed=0, num_steps=EP_LEN)
    rng = np.random.RandomState(0)
    acts = rng.uniform(-1, 1, (EP_LEN, ENV_NUM, 2)).astype(np.float32)
    L = gxo.lib()
    visible = int(L.gxo_get_threads())
    share = cpu_share()
    # a GPU box shows all of its host's hardware threads but grants this job a share of them (cgroup quota): more
    # OpenMP threads than the share only time-slice (round 3 timed 128 threads on what was a 16-CPU share)
    ncores = max(1, min(visible, share)) if share else visible

    def run(threads, epochs):
        L.gxo_set_threads(threads)
        ref = gxo.OracleEngine(cfg, n_candidates=1_000_000)
        t_reset = t_step = 0.0
        for _ in range(epochs):
            t0 = time.perf_counter()
            ref.reset()
            t1 = time.perf_counter()
            for t in range(EP_LEN):
                _, _, d, _ = ref.step(acts[t])
                if d.any():
                    ref.reset_done()
            t2 = time.perf_counter()
            t_reset += t1 - t0
            t_step += t2 - t1
        n = epochs * EP_LEN * ENV_NUM
        return n / (t_reset + t_step), n / t_step, t_reset / epochs, t_reset + t_step

    v1, s1, r1, w1 = run(1, epochs_1t)
    va, sa, ra, wa = run(ncores, epochs_all)
    L.gxo_set_threads(0)
    return {"value": round(va, 1), "unit": "env-steps/s", "cores": ncores, "kind": "port",
            "value_all_cores": round(va, 1), "value_1thread": round(v1, 1),
            "stepping_only_all_cores": round(sa, 1), "stepping_only_1thread": round(s1, 1),
            "reset_s_all_cores": round(ra, 3), "reset_s_1thread": round(r1, 3),
            "threads": {"reset_phase": ncores, "step_phase": ncores},
            "host": {"hardware_threads_visible": visible, "cpu_share_of_this_job": share},
            "sample": f"{epochs_all} epochs on {ncores} threads ({wa:.1f} s) and {epochs_1t} epoch on 1 thread "
                      f"({w1:.1f} s): {EP_LEN} steps x {ENV_NUM} envs incl. reset() over 1e6 layout candidates "
                      "and reset_done()",
            "note": "CPU restatement (oracle/, gcc -O2 -fopenmp), not the reference's XLA:CPU program; "
                    "a reported baseline, not the optimisation target"}


def cpu_share():
    """CPUs this process may actually use: the cgroup quota (v2 cpu.max, v1 cfs quota) and the affinity mask, or None"""
    n = None
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:  # noqa: BLE001
        pass
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: None if t.split()[0] == "max" else float(t.split()[0]) / float(t.split()[1])),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", None)):
        try:
            txt = open(path).read().strip()
            if parse is None:
                q = float(txt)
                per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().strip())
                v = None if q <= 0 else q / per
            else:
                v = parse(txt)
            if v:
                n = int(min(n, max(1, round(v)))) if n else int(max(1, round(v)))
            break
        except Exception:  # noqa: BLE001
            continue
    return n


def epoch_breakdown(device):
    """Where one 200-step epoch goes: the rollout kernel alone, reset() with the layout sampler
    inline, and the two overlapped (sampler prefetched on the side stream)."""
    env = make_engine(ENV_NUM, 0, 1)
    tape = action_tape(EP_LEN, ENV_NUM, 11, device)

    def timeit(fn, n):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n

    def epoch():
        env.reset()
        env.rollout(tape)
    env.set_prefetch(-1)
    env.reset()
    t_roll = timeit(lambda: env.rollout(tape), 10)
    t_reset = timeit(env.reset, 5)
    env.set_prefetch(EP_LEN)
    t_epoch = timeit(epoch, 10)
    env.close()
    return {"rollout_kernel_200_steps_us": round(t_roll * 1e6, 1),
            "reset_inline_sampler_us": round(t_reset * 1e6, 1),
            "epoch_overlapped_us": round(t_epoch * 1e6, 1),
            "note": "reset() = exact restatement of the reference's 1e6-candidate rejection sampler "
                    "(6e8 Threefry-2x32 blocks cut to under 3e8 by exact early rejection and lazy evaluation of the draws); it is integer-VALU "
                    "bound and bounds the epoch",
            "valu_issue": valu_issue_floor()}


def valu_issue_floor():
    """the time the vector ALUs need just to issue one epoch's instructions, from the committed counters"""
    n, why = _evidence("epoch_valu_instructions")
    if n is None:
        return {"note": "no counter evidence: " + why}
    ns = 1.73   # profiles/history/r02_probe_threefry_chain.log: 110 ns per 63.5-instruction Threefry block per SIMD
    return {"wave_instructions_per_epoch": round(n), "ns_per_wave_instruction_per_simd": ns, "simds": 1024,
            "issue_floor_us": round(n * ns * 1e-9 / 1024 * 1e6, 1),
            "note": "NOT measured in this run: SQ_INSTS_VALU of every kernel of one epoch (profiles/r04_sampler_pmc_SQ.csv, "
                    "r04_rollout_N2000_T200_pmc_SQ.csv) x the measured issue cost of the sampler's own Threefry code / 1024 "
                    "SIMDs; compare with the headline ms_per_step"}


def closed_loop_rate(device, epochs=50, hidden=64):
    """reset() + rollout_policy per epoch: the (h, h)-tanh actor-critic of trpo_core.py:110-173 (random init) evaluated on
    device.  h = 64 (the reference default): ONE launch per 200-step rollout; wider networks
    (trpo.py:606 --hid): two launches per control step."""
    from guardx_amd import Engine
    env = make_engine(ENV_NUM, 0, 1)
    D = env.obs_flat_size
    torch.manual_seed(0)
    mk = lambda out: torch.nn.Sequential(torch.nn.Linear(D, hidden), torch.nn.Tanh(), torch.nn.Linear(hidden, hidden),  # noqa: E731
                                         torch.nn.Tanh(), torch.nn.Linear(hidden, out))
    params = Engine.pack_actor_critic(mu_net=mk(2), v_net=mk(1), log_std=torch.full((2,), -0.5)).to(device)

    def epoch():
        env.reset()
        env.rollout_policy(params, EP_LEN)
    epoch(); epoch()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(epochs):
        epoch()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    env.close()
    return ENV_NUM * EP_LEN * epochs / dt


OTHER_WARMUP = 30   # untimed warm-up epochs per task in other_robots (17 ms for the Swimmer, 75 ms for the Walker)


def other_robots(device, epochs=50):
    """the same epoch (reset + one 200-step rollout, env_num=2000) for the articulated robots: BASELINE config 3
    (Goal_Swimmer_8Hazards), Goal_Ant_8Hazards / Goal_Walker_8Hazards (contact + joint-limit solver), and BASELINE
    config 5 as a SYNTHETIC task (the reference has no runnable counterpart): Ant + 8 hazards + 8 pillars"""
    from guardx_amd import Engine, configuration
    out = {}
    cases = [("Goal_Swimmer_8Hazards", dict(TASK, robot_base="xmls/swimmer.xml"), None),
             ("Goal_Ant_8Hazards", dict(TASK, robot_base="xmls/ant.xml"), None),
             ("Goal_Walker_8Hazards", dict(TASK, robot_base="xmls/walker.xml"), None),
             ("Ant_8Hazards_8Pillars_synthetic", dict(configuration("Ant_8Hazards_8Pillars_synthetic")),
              "synthetic -- no reference counterpart (BASELINE config 5): ant.xml, goal task, 8 hazards + 8 static "
              "pillar circles with their own lidar and keepout, 6 m x 6 m arena")]
    for name, cfg, label in cases:
        cfg.update(env_num=ENV_NUM, _seed=0, num_steps=EP_LEN, device_id=torch.cuda.current_device())
        env = Engine(cfg)
        tape = action_tape(EP_LEN, ENV_NUM, 0, device, env.action_space.shape[0])

        def epoch():
            env.reset(check=False)
            env.rollout(tape)
        for _ in range(OTHER_WARMUP):    # the process is fresh: the first ~15 ms of any load sit in the firmware's clock ramp
            epoch()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(epochs):
            epoch()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        env.check_layouts()
        env.close()
        out[name] = {"env_steps_per_s": round(ENV_NUM * EP_LEN * epochs / dt, 1), "obs_dim": env.obs_flat_size,
                     "ms_per_epoch": round(dt / epochs * 1e3, 4)}
        if label:
            out[name]["label"] = label
        # the same epochs through the hand-off pipeline in a world of one: the observation pass of epoch k runs on the
        # hand-off's stream during epoch k + 1 (packed rows one epoch late) instead of behind the dynamics pass -- what
        # a rank of the multi-GPU run does; it pays where the dynamics chain, not the sampler, bounds the epoch
        try:
            from guardx_amd.dist import TapeHandoff
            env = Engine(cfg)
            h = TapeHandoff(env, EP_LEN, sharded_sampler=False)

            def epoch_p():
                env.reset(check=False)
                h.step(tape)
            for _ in range(4):
                epoch_p()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(epochs):
                epoch_p()
            h.drain()
            torch.cuda.synchronize()
            dtp = time.perf_counter() - t0
            env.check_layouts()
            env.close()
            out[name]["pipelined_env_steps_per_s"] = round(ENV_NUM * EP_LEN * epochs / dtp, 1)
        except Exception as exc:  # noqa: BLE001
            out[name]["pipelined_env_steps_per_s"] = f"{type(exc).__name__}: {exc}"[:120]
    return out


def reset_done_heavy(device, epochs=50):
    """The headline epoch with the reset_done branch actually taken: with the force-limited Point a random policy
    almost never reaches a 0.5 m goal 3 m away, so the headline's timed region holds next to no reset_done events.
    Here episodes are shorter than the epoch (num_steps = 60: the timeout of engine.py:492 ends every episode on its
    62nd step) and the goal is wide (2.9), so every env is re-initialised ~3 times per epoch inside the rollout
    (layout draw, re-placement, re-initialised observation row).  Parity at exactly this size:
    tests/test_gpu_parity.py::test_bench_workload_reset_done_heavy."""
    from guardx_amd import Engine
    cfg = dict(TASK, goal_size=2.9)
    cfg.update(env_num=ENV_NUM, _seed=0, num_steps=60, device_id=torch.cuda.current_device())
    env = Engine(cfg)
    env.set_prefetch(EP_LEN)
    tape = action_tape(EP_LEN, ENV_NUM, 0, device)

    def epoch():
        env.reset(check=False)
        return env.rollout(tape)
    epoch()
    done = epoch()[3]
    n_done = float(done.sum().item())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(epochs):
        epoch()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    env.check_layouts()
    env.close()
    return {"env_steps_per_s": round(ENV_NUM * EP_LEN * epochs / dt, 1), "ms_per_epoch": round(dt / epochs * 1e3, 4),
            "reset_done_events_per_epoch": n_done, "per_env": round(n_done / ENV_NUM, 2),
            "config": "Goal_Point_8Hazards env_num=2000, goal_size=2.9, num_steps=60 (timeouts), 200-step epochs"}


def multi_gpu_rehearsal(device, world=8, epochs=30):
    """This GPU plays rank 0 of `world` in the default N > 1 epoch (tools/rehearse_rank.py): everything a rank does per
    epoch -- 1/world of the layout sampler for a later reset, its dynamics pass, the install of every rank's export
    block, the observation pass over all (or only its own) tapes -- with device copies standing in for the all-gather.
    Measured GPU time of a rank's epoch + the link as arithmetic = the predicted weak-scaling efficiency (an 8-GPU node
    is the driver's to run).  Run in a FRESH process, as a rank is: inside this one -- a dozen engines and their streams
    created and destroyed by the other extras -- the same rehearsal measures 0.72-0.74 ms per epoch instead of 0.48
    (HIP assigns streams to hardware queues in creation order; a rank process creates its engine and hand-off first)."""
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "rehearsal.json")
        env = dict(os.environ)
        for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
            env.pop(k, None)
        env["HIP_VISIBLE_DEVICES"] = env.get("HIP_VISIBLE_DEVICES", str(device.index if device.index is not None else 0))
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rehearse_rank.py"), "--world", str(world),
                            "--epochs", str(epochs), "--json", out], env=env, capture_output=True, text=True, timeout=600)
        if r.returncode != 0:
            raise RuntimeError("tools/rehearse_rank.py failed: " + r.stderr[-300:])
        with open(out) as f:
            res = json.load(f)
    res["note"] = ("NOT an 8-GPU measurement: one GPU playing rank 0 of 8 in a fresh process; the xGMI transfer enters as "
                   "bytes / bandwidth (model)")
    return res


def in_fresh_process(what, device):
    """run one of the supplementary measurements in a child process started from scratch (`bench.py --child-extra`)"""
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, what + ".json")
        env = dict(os.environ)
        for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GX_FORCE_DIST"):
            env.pop(k, None)
        env["HIP_VISIBLE_DEVICES"] = env.get("HIP_VISIBLE_DEVICES", str(device.index if device.index is not None else 0))
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child-extra", what, "--child-out", out], env=env,
                           capture_output=True, text=True, timeout=900)
        if r.returncode != 0:
            raise RuntimeError(f"bench.py --child-extra {what} failed: " + r.stderr[-300:])
        with open(out) as f:
            return json.load(f)


def api_loop_fresh(device):
    """The learner-driven loop in a process of its own, as an unmodified learner's would be: in the bench process -- after a
    thousand epochs, with the headline engine's pools, three hand-offs' buffers and their streams alive -- the same loop
    measured 197-252 M on boxes where it runs 234-241 M alone (host-bound: 8-9 us of Python + hipLaunchKernel per pair)."""
    env = make_engine(ENV_NUM, 0, 1)
    env.set_prefetch(EP_LEN)
    tape = action_tape(EP_LEN, ENV_NUM, 0, device)
    api_loop_rate(env, tape, 1000)
    res = dict(api_loop_summary(env, tape), process="fresh child process of bench.py")
    env.close()
    return res


CHILD_EXTRAS = {"other_robots": lambda device: dict(other_robots(device), process="fresh child process of bench.py"),
                "api_step_loop": api_loop_fresh}


def api_loop_rate(env, tape, steps):
    """Python-driven Engine.step()/reset_done() loop (what an unmodified learner drives)."""
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s = 0
    while s < steps:
        env.reset()
        for t in range(min(EP_LEN, steps - s)):
            env.step(tape[t])
            env.reset_done()
            s += 1
    torch.cuda.synchronize()
    return env.env_num * steps / (time.perf_counter() - t0)


def api_loop_summary(env, tape, reps=5, steps=2000):
    """the default Engine (out_ring=0: step() outputs are never overwritten, engine.py:495) and the opt-in ring of 8
    reused output sets, side by side"""
    from guardx_amd import Engine
    rates = sorted(api_loop_rate(env, tape, steps) for _ in range(reps))
    ring = Engine(env._ctor_config, **dict(env._ctor_kwargs, out_ring=8))
    ring.set_prefetch(EP_LEN)
    r8 = sorted(api_loop_rate(ring, tape, steps) for _ in range(reps))
    ring.close()
    return {"value": round(rates[reps // 2], 1), "best": round(rates[-1], 1), "out_ring": 0,
            "out_ring_8": {"value": round(r8[reps // 2], 1), "best": round(r8[-1], 1)},
            "unit": "env-steps/s", "note": f"median (and best) of {reps} runs of {steps} step()+reset_done() pairs incl. "
                                           "a reset() every 200"}


def previous_round_values():
    """What the newest committed driver record (BENCH_rNN.json at the repo root: `parsed` + the last 2000 characters of
    the line in `tail`) and, for the keys the truncated tail no longer holds, the builder's own driver-style line of the
    same round (profiles/rNN_bench_driver_style.json) say about the values this line reports.  -> (round, {key: (value,
    source)})"""
    import glob
    import re
    recs = sorted(glob.glob(os.path.join(ROOT, "BENCH_r[0-9][0-9].json")))
    if not recs:
        return None, {}
    rec = recs[-1]
    rnd = os.path.basename(rec)[6:9]
    prev = {}
    try:
        with open(rec) as f:
            d = json.load(f)
        src = os.path.basename(rec)
        if isinstance(d.get("parsed"), dict) and d["parsed"].get("value"):
            prev["value"] = (float(d["parsed"]["value"]), src + ":parsed.value")
        tail = d.get("tail") or ""
        for name in ("Goal_Swimmer_8Hazards", "Goal_Ant_8Hazards", "Goal_Walker_8Hazards", "Ant_8Hazards_8Pillars_synthetic"):
            m = re.search(r'"%s": \{"env_steps_per_s": ([0-9.]+)' % name, tail)
            if m:
                prev["other_robots." + name] = (float(m.group(1)), src + ":tail")
    except (OSError, ValueError):
        pass
    own = os.path.join(ROOT, "profiles", f"{rnd}_bench_driver_style.json")
    try:
        with open(own) as f:
            line = json.loads([ln for ln in f if ln.startswith("{")][0])
        src = os.path.relpath(own, ROOT)
        for key, val in _comparable_values(line).items():
            prev.setdefault(key, (val, src))
    except (OSError, ValueError, IndexError):
        pass
    return rnd, prev


def _comparable_values(line):
    """the rates of a bench line that are compared round over round"""
    out = {}
    if line.get("value"):
        out["value"] = float(line["value"])
    for name, v in (line.get("other_robots") or {}).items():
        if isinstance(v, dict) and v.get("env_steps_per_s"):
            out["other_robots." + name] = float(v["env_steps_per_s"])
    # the Python-driven loop is host-bound and the host is shared: of its five repetitions the BEST is the one least
    # disturbed by the box's other tenants (one run of round 5: 211 M median, 251 M best; the next: 251 / 259) -- round over
    # round the best is compared with the best
    api = line.get("api_step_loop_env_steps_per_s")
    if isinstance(api, dict) and api.get("value"):
        out["api_step_loop"] = float(api.get("best") or api["value"])
        if isinstance(api.get("out_ring_8"), dict) and api["out_ring_8"].get("value"):
            out["api_step_loop.out_ring_8"] = float(api["out_ring_8"].get("best") or api["out_ring_8"]["value"])
    if isinstance(line.get("preconditioned"), dict) and line["preconditioned"].get("value"):
        out["preconditioned"] = float(line["preconditioned"]["value"])
    if isinstance(line.get("closed_loop_policy_env_steps_per_s"), (int, float)):
        out["closed_loop_policy.hidden_64"] = float(line["closed_loop_policy_env_steps_per_s"])
    for k, v in (line.get("closed_loop_policy_wider_env_steps_per_s") or {}).items():
        if isinstance(v, (int, float)):
            out["closed_loop_policy." + k] = float(v)
    rh = line.get("reset_done_heavy")
    if isinstance(rh, dict) and rh.get("env_steps_per_s"):
        out["reset_done_heavy"] = float(rh["env_steps_per_s"])
    return out


HOST_BOUND_THRESHOLD = -0.10


def vs_previous_round(line, threshold=-0.02):
    """every compared rate of this line beside the previous round's, and the list of those more than 2 % below it"""
    rnd, prev = previous_round_values()
    if not prev:
        return {"previous": None, "note": "no BENCH_rNN.json in the tree"}
    now = _comparable_values(line)
    unlike = set()
    if rnd <= "r04" and "value" in prev:
        # rounds 2-4 reported ONE repetition behind a clock-warming prelude as `value`; from round 5 on it is the
        # un-preconditioned median.  Like is compared with like: this line's `preconditioned` sibling against it.
        unlike.add("value")
        if isinstance(line.get("preconditioned"), dict):
            now["preconditioned"] = float(line["preconditioned"]["value"])
            prev["preconditioned"] = prev["value"]
    rows, regress = {}, []
    for key, val in now.items():
        if key not in prev:
            continue
        p, src = prev[key]
        rel = val / p - 1.0
        rows[key] = {"now": round(val, 1), "previous": round(p, 1), "change": round(rel, 4), "source": src}
        # the Python-driven loop is host-bound (one hipLaunchKernel per step): on one box, same library, consecutive medians
        # of it differ by +-10 % (tools/ab_api.py, profiles/r05_ab_api.log) -- its bar is -10 %, and the row says so
        thr = HOST_BOUND_THRESHOLD if key.startswith("api_step_loop") else threshold
        if thr != threshold:
            rows[key]["threshold"] = thr
            rows[key]["statistic"] = "best of the repetitions"
        if key in unlike:
            rows[key]["like_for_like"] = False
        elif rel < thr:
            regress.append(key)
    out = {"previous": rnd, "values": rows, "regressions": regress, "threshold": threshold}
    if unlike:
        out["note"] = ("the previous round's `value` was one repetition behind a clock-warming prelude; this round's is the "
                       "un-preconditioned median: the like-for-like row is `preconditioned`")
    return out


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (this process has not
    touched the GPU), one per GPU, rendezvous on 127.0.0.1, relay rank 0's line, fail if any rank fails."""
    n = args.gpus
    have = torch.cuda.device_count()          # does not initialise the GPU
    if have < n and not os.environ.get("GX_BENCH_FORCE_DEVICE"):
        print(f"bench.py: --gpus {n} but only {have} HIP device(s) are visible", file=sys.stderr)
        return 2
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), GX_BENCH_CHILD="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    deadline = time.time() + float(os.environ.get("GX_BENCH_SPAWN_TIMEOUT", "1500"))
    alive = list(procs)
    while alive:
        for p in list(alive):
            code = p.poll()
            if code is not None:
                alive.remove(p)
                if code != 0:
                    rc = rc or code
        if rc or time.time() > deadline:
            for p in alive:                    # exactly the children started above
                p.terminate()
            for p in alive:
                try:
                    p.wait(timeout=20)
                except subprocess.TimeoutExpired:
                    p.kill()
            return rc or 3
        time.sleep(0.2)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)     # bench steps = 200-pass epochs
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--reps", type=int, default=5,
                    help="repetitions of the [warm-up, timed] region; `value` is their median (default 5)")
    ap.add_argument("--no-precondition", action="store_true",
                    help="skip the `preconditioned` sibling measurement (one repetition behind 60 ms of torch.mm)")
    ap.add_argument("--child-extra", choices=sorted(CHILD_EXTRAS), help=argparse.SUPPRESS)
    ap.add_argument("--child-out", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.child_extra:
        torch.cuda.set_device(0)
        res = CHILD_EXTRAS[args.child_extra](torch.device("cuda", 0))
        with open(args.child_out, "w") as f:
            json.dump(res, f)
        return
    if args.steps < 1 or args.warmup < 0 or args.reps < 1:
        ap.error("--steps >= 1, --warmup >= 0, --reps >= 1")

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))

    from guardx_amd import dist as gxd
    rank, local, world = gxd.init_from_env()
    if world != args.gpus:
        print(f"bench.py: {world} rank(s) joined the process group but --gpus is {args.gpus}", file=sys.stderr)
        sys.exit(2)
    if os.environ.get("GX_BENCH_FORCE_DEVICE"):      # rehearsal: several ranks share one GPU
        local = int(os.environ["GX_BENCH_FORCE_DEVICE"])
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)

    env = make_engine(ENV_NUM, rank, world)
    env.set_prefetch(EP_LEN)
    tapes = [action_tape(EP_LEN, ENV_NUM, 1000 * rank + k, device) for k in range(4)]
    import torch.distributed as tdist
    forced = gxd.forced_dist() and tdist.is_initialized()   # GX_FORCE_DIST=1: the N > 1 path over a one-rank group
    gather = world > 1 or forced
    # the hand-off: "tape" (default) all-gathers the 36-B-per-env-step dynamics tape and expands it on every rank,
    # "packed" all-gathers the 192-B packed rows (what round 1 did; GX_HANDOFF=packed to compare)
    mode = os.environ.get("GX_HANDOFF", "tape")
    if gather and mode == "tape":
        env.reset()                      # sizes the export blocks of the sharded sampler (layout_size)

    def make_handoff(sharded, expand):
        if not gather:
            return None
        if mode != "tape":
            return RolloutHandoff(world)
        return gxd.TapeHandoff(env, EP_LEN, sharded_sampler=sharded, expand=expand)

    def timed_region(handoff, warmup):
        run_epochs(env, tapes, warmup, handoff)
        gxd.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_epochs(env, tapes, args.steps, handoff)
        torch.cuda.synchronize()
        gxd.barrier()
        return gxd.max_over_ranks(time.perf_counter() - t0, device)

    def leg(sharded, expand, warmup, reps=1):
        """`reps` repetitions of [W warm-up epochs, barrier, K timed epochs, barrier] through one hand-off object;
        returns the sorted-by-time middle one (the median for odd reps), all of them, and the hand-off"""
        h = make_handoff(sharded, expand)
        ts = [timed_region(h, warmup) for _ in range(reps)]
        if hasattr(h, "close"):
            h.close()                    # the engine samples for itself again
        return sorted(ts)[len(ts) // 2], ts, h

    # `value` at N > 1: the tape hand-off with the layout sampler sharded over the ranks (each rank samples 1/N of the
    # candidates of the reset after next, the rows ride on the tape all-gather: still ONE collective per epoch) and every
    # rank expanding every rank's tape
    # `value` = the MEDIAN of --reps repetitions of the driver's region (W untimed warm-up epochs, then exactly K timed
    # epochs between barrier + synchronize), started from whatever state the process start left the GPU in: no filler
    # work in front of it.  The first repetition is the cold one (the firmware's clock ramp, ~13 ms on this pool, covers
    # it); it is reported with the others, and min / max give the spread.
    dt, dts, handoff = leg(True, "all", args.warmup, reps=args.reps)

    env_steps = ENV_NUM * world * EP_LEN * args.steps
    value = env_steps / dt
    stepping_only = None
    legs = None
    if gather:
        w2 = max(4, args.warmup)         # a change of the layout source costs up to three inline samplers
        W = env.obs_flat_size + 2 + 3

        def rate(t):
            return {"value": round(env_steps / t, 1), "unit": "env-steps/s", "ms_per_step": round(t / args.steps * 1e3, 6)}
        # the same epochs without the hand-off: what the sharded stepping alone sustains (no collective on the
        # data path, every rank samples all candidates for itself)
        dt1 = timed_region(None, w2)
        stepping_only = dict(rate(dt1),
                             handoff_ms_per_epoch_exposed=round((dt - dt1) / args.steps * 1e3, 6), handoff=mode,
                             handoff_bytes_received_per_rank_per_epoch=int(
                                 (world - 1) * (handoff.n if mode == "tape" else EP_LEN * ENV_NUM * W) * 4),
                             packed_rows_bytes_per_rank_per_epoch=int(EP_LEN * ENV_NUM * W * 4),
                             handoff_queue_probe=getattr(handoff, "queue_probe", None),
                             note="same epochs with the rollout hand-off switched off (two-kernel gx_rollout, every rank "
                                  "samples all 1e6 layout candidates itself); `value` above includes the hand-off")
        if mode == "tape":
            dt2, _, _ = leg(False, "all", w2)
            dt3, _, _ = leg(True, "local", w2)
            legs = {"unsharded_sampler": dict(rate(dt2), note="the round-3 default: tape hand-off, every rank expands every "
                                              "tape, every rank samples all 1e6 layout candidates itself"),
                    "local_expand": dict(rate(dt3), note="as `value`, but a rank expands only its own tape; the other "
                                         "ranks' tapes are held and expanded on demand (TapeHandoff.expand_rank)"),
                    "warmup_epochs_each": w2,
                    "note": "`value` = sharded sampler + expand all; all legs: same engine, same epochs, one collective per epoch"}
    line = {
        "metric": "env-steps/sec", "value": round(value, 1), "unit": "env-steps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 6), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "timed_region_s": round(dt, 6), "env_steps_timed": env_steps,
        "config": {"workload": "Goal_Point_8Hazards env_num=2000/GPU, random-policy rollout (U(-1,1) action "
                               "tape resident in HBM); ONE BENCH STEP = ONE 200-PASS EPOCH: reset() over 1e6 layout "
                               "candidates + 200 x (step + reset_done)"
                               + ((", one async RCCL all-gather of the dynamics tape per epoch, overlapped with the "
                                   "following epoch, every rank expanding all tapes into the packed rollout"
                                   if mode == "tape" else
                                   ", one async RCCL all-gather of the packed rollout shard per epoch, "
                                   "overlapped with the following epochs") if gather else ""),
                   "env_num_per_gpu": ENV_NUM, "max_ep_len": EP_LEN, "step_passes_per_bench_step": EP_LEN,
                   "obs_dim": env.obs_flat_size,
                   "driver": "gx_rollout: two launches per 200-pass epoch (serial dynamics tape, then one thread per "
                             "(step, env) observation row), layout pool of the next epoch prefetched on a side stream",
                   "layout_candidates_per_reset": 1_000_000,
                   "layout_sampler": ("sharded over the ranks: each samples 1/N of the candidates of the reset after next, the "
                                      "valid rows ride in the tail of its tape shard (no second collective)"
                                      if getattr(handoff, "sharded_used", False) else
                                      "every rank samples all candidates (shared key)"),
                   "point_actuators": "mjcf defaults inherited (DESIGN.md 0.1)"},
    }
    try:   # the 8-GPU hand-off as arithmetic (it cannot be measured on a one-GPU box): bytes on the wire vs the epoch
        shard_bytes = int(handoff.n if hasattr(handoff, "n") else sum(env.tape_floats(EP_LEN))) * 4
        line["handoff_model"] = {
            "shard_bytes_per_rank_per_epoch": shard_bytes,
            "packed_rows_bytes_per_rank_per_epoch": int(EP_LEN * ENV_NUM * (env.obs_flat_size + 2 + 3) * 4),
            "received_per_rank_at_8_gpus_bytes": 7 * shard_bytes,
            "allgather_ms_at_8_gpus_310GBps": round(7 * shard_bytes / 310e9 * 1e3, 4),
            "note_n1": "at N = 1 no hand-off runs; the shard here is the bare tape -- at N > 1 each rank's block of valid "
                       "layouts (~0.5 MB) rides in its tail",
            "ms_per_step_this_run": round(dt / args.steps * 1e3, 4),
            "note": "one async all-gather of the dynamics tape per epoch (36 B per env-step: qpos, qvel, action, one word for "
                    "done and the layout rows; at N > 1 plus the rank's export block of valid layouts, ~0.5 MB), overlapped "
                    "with the following epoch; 310 GB/s = a realistic all-gather bus "
                    "bandwidth over 7 xGMI links (537 GB/s peak per direction); arithmetic, not a measurement"}
    except Exception as exc:  # noqa: BLE001
        line["handoff_model"] = {"error": f"{type(exc).__name__}: {exc}"[:200]}
    line["repetitions"] = {
        "n": len(dts), "values": [round(env_steps / t, 1) for t in dts],
        "min": round(env_steps / max(dts), 1), "max": round(env_steps / min(dts), 1),
        "first": round(env_steps / dts[0], 1),
        "note": "`value` = the median of these: each repetition is W untimed warm-up epochs + K timed epochs between "
                "barrier + synchronize (max over ranks), run back to back; the first starts from the idle GPU the "
                "process start leaves (inside the firmware's clock ramp), no filler work precedes any of them"}
    if not args.no_precondition:
        # the round-2..4 headline, kept as a sibling: ONE repetition behind 60 ms of unrelated GPU work (torch.mm)
        precond = precondition_clocks(device)
        dtp, _, _ = leg(True, "all", args.warmup)
        line["preconditioned"] = {"value": round(env_steps / dtp, 1), "unit": "env-steps/s",
                                  "ms_per_step": round(dtp / args.steps * 1e3, 6), "prelude": precond,
                                  "note": "one repetition started right behind unrelated GPU work that holds the clocks "
                                          "up (what rounds 2-4 reported as `value`); NOT the headline any more"}
    if forced:
        line["forced_dist"] = {"backend": tdist.get_backend(), "world_size": world,
                               "note": "GX_FORCE_DIST=1: the N > 1 path (process group, barrier, max-over-ranks, tape hand-off "
                                       "with the collective issued, sharded sampler, all-tapes expansion) over a group of "
                                       "this many ranks -- a code-path run, not a scaling measurement"}
    if dt < 0.010:
        line["warning"] = f"timed region {dt*1e3:.2f} ms < 10 ms: use more --steps for a meaningful rate"
    if stepping_only is not None:
        line["stepping_only"] = stepping_only
    if legs is not None:
        line["legs"] = legs
    if rank == 0:
        try:
            line["roofline"] = roofline_rollout(ENV_NUM, EP_LEN, 30, device)
        except Exception as exc:  # noqa: BLE001
            line["roofline"] = {"error": f"{type(exc).__name__}: {exc}"[:300]}
        if not args.no_extras and world == 1:
            # supplementary measurements; none of them may take the headline line down with it
            def extra(key, fn):
                try:
                    line[key] = fn()
                except Exception as exc:  # noqa: BLE001 - reported in the line instead
                    line[key] = {"error": f"{type(exc).__name__}: {exc}"[:300]}
                    torch.cuda.empty_cache()
            # bandwidth regime: the thread-per-env step kernel at 2^22 envs
            extra("roofline_large_batch", lambda: roofline_step(1 << 22, 30, device))
            extra("large_batch_fused", lambda: large_batch_fused(1 << 22, 32, device))
            # host-bound (one ctypes call per step; the box's host cores are shared with other tenants): median of five
            extra("api_step_loop_env_steps_per_s", lambda: in_fresh_process("api_step_loop", device))
            extra("epoch_breakdown", lambda: epoch_breakdown(device))
            extra("closed_loop_policy_env_steps_per_s", lambda: round(closed_loop_rate(device), 1))
            extra("closed_loop_policy_wider_env_steps_per_s",
                  lambda: {f"hidden_{h}": round(closed_loop_rate(device, 20, h), 1) for h in (128, 256)})
            extra("reset_done_heavy", lambda: reset_done_heavy(device))
            extra("multi_gpu_rehearsal", lambda: multi_gpu_rehearsal(device))
            mg = line.get("multi_gpu_rehearsal", {})
            if "expand_all" in mg and isinstance(line.get("handoff_model"), dict):   # the prediction next to the byte count
                line["handoff_model"]["predicted_at_8_gpus"] = {
                    "rank_epoch_ms_measured_on_one_gpu": mg["expand_all"]["ms_per_epoch"],
                    "one_gpu_epoch_ms": mg["one_gpu_own_sampler"]["ms_per_epoch"],
                    "weak_scaling_efficiency_at_310GBps": mg["expand_all"]["model"]["at_310GBps"]["weak_scaling_efficiency"],
                    "weak_scaling_efficiency_at_200GBps": mg["expand_all"]["model"]["at_200GBps"]["weak_scaling_efficiency"],
                    "bytes_received_per_rank_per_epoch": mg["expand_all"]["bytes_received_per_epoch"],
                    "link_bound_below_GBps": mg["expand_all"]["model"].get("link_bound_below_GBps"),
                    "break_even_GBps": mg["expand_all"]["model"].get("break_even_GBps"),
                    "efficiency_if_gpu_bound": mg["expand_all"]["model"].get("efficiency_if_gpu_bound"),
                    "note": "from `multi_gpu_rehearsal` (this GPU playing rank 0 of 8 in the default N > 1 epoch, fresh "
                            "process): GPU time of a rank's epoch measured, the link as bytes / bandwidth; efficiency = "
                            "one-GPU epoch / max(rank epoch, all-gather time)"}
            # in a fresh process, as each of these tasks would run on its own (same reason as multi_gpu_rehearsal: after the
            # dozen engines and streams of the extras above, HIP's stream -> hardware-queue assignment costs ~3 %)
            extra("other_robots", lambda: in_fresh_process("other_robots", device))
        if world == 1:
            try:
                line["vs_previous_round"] = vs_previous_round(line)
            except Exception as exc:  # noqa: BLE001
                line["vs_previous_round"] = {"error": f"{type(exc).__name__}: {exc}"[:300]}
        if world == 1 and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline()
            except Exception as exc:  # noqa: BLE001
                line["cpu_baseline"] = {"error": f"{type(exc).__name__}: {exc}"[:300]}
        print(json.dumps(line), flush=True)
    gxd.barrier()
    env.close()
    if tdist.is_initialized():
        tdist.destroy_process_group()


if __name__ == "__main__":
    main()

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


SHARDED_RESET = None   # guardx_amd.dist.ShardedReset when GX_SHARD_SAMPLER=1 (optional second collective, default off)


def run_epochs(env, tapes, epochs, handoff):
    """`epochs` bench steps: reset() + one fused 200-pass rollout each (+ the async hand-off)."""
    for ep in range(epochs):
        if SHARDED_RESET is not None and SHARDED_RESET.env is env:
            SHARDED_RESET.reset(check=False)
        else:
            env.reset(check=False)       # the layout_size assert is checked once after the loop
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
    power ramp is over when the W warm-up epochs start.  Measured on this pool (tools/debug/cold_start_probe.py,
    DESIGN.md section 6): after >= 50 ms of idle the first ~25 epochs (13 ms) run 10 % -> 0 % slower than the steady
    state whatever ran before the idle gap, and 30 ms of any sustained compute removes that.  The driver's region
    (5 + 20 epochs = 13 ms) would otherwise sit entirely inside the ramp.  The line reports `cold_start` beside."""
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
    ncores = int(L.gxo_get_threads())

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
            "sample": f"{epochs_all} epochs on {ncores} threads ({wa:.1f} s) and {epochs_1t} epoch on 1 thread "
                      f"({w1:.1f} s): {EP_LEN} steps x {ENV_NUM} envs incl. reset() over 1e6 layout candidates "
                      "and reset_done()",
            "note": "CPU restatement (oracle/, gcc -O2 -fopenmp), not the reference's XLA:CPU program; "
                    "a reported baseline, not the optimisation target"}


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
    ns = 1.73   # profiles/r02_probe_threefry_chain.log: 110 ns per 63.5-instruction Threefry block per SIMD
    return {"wave_instructions_per_epoch": round(n), "ns_per_wave_instruction_per_simd": ns, "simds": 1024,
            "issue_floor_us": round(n * ns * 1e-9 / 1024 * 1e6, 1),
            "note": "NOT measured in this run: SQ_INSTS_VALU of every kernel of one epoch (profiles/r03_sampler_pmc_SQ.csv, "
                    "r03_rollout_N2000_T200_pmc_SQ.csv) x the measured issue cost of the sampler's own Threefry code / 1024 "
                    "SIMDs; compare with the headline ms_per_step"}


def closed_loop_rate(device, epochs=50):
    """reset() + ONE rollout_policy launch per epoch: the (64,64)-tanh actor-critic of
    trpo_core.py:110-173 (random init) evaluated inside the persistent kernel."""
    from guardx_amd import Engine
    env = make_engine(ENV_NUM, 0, 1)
    D = env.obs_flat_size
    torch.manual_seed(0)
    mk = lambda out: torch.nn.Sequential(torch.nn.Linear(D, 64), torch.nn.Tanh(), torch.nn.Linear(64, 64),  # noqa: E731
                                         torch.nn.Tanh(), torch.nn.Linear(64, out))
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
        epoch(); epoch()
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
    ap.add_argument("--no-precondition", action="store_true",
                    help="skip the 60 ms of unrelated GPU work before the warm-up epochs (clock ramp, see precondition_clocks)")
    args = ap.parse_args()
    if args.steps < 1 or args.warmup < 0:
        ap.error("--steps >= 1, --warmup >= 0")

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
    global SHARDED_RESET
    if os.environ.get("GX_SHARD_SAMPLER") == "1":
        # OPTIONAL, off by default: the 1e6-candidate layout sampler split over the ranks + one small all-gather of the
        # valid layouts (a second collective; north_star names one).  Same pools, same results.
        SHARDED_RESET = gxd.ShardedReset(env)
    tapes = [action_tape(EP_LEN, ENV_NUM, 1000 * rank + k, device) for k in range(4)]
    gather = world > 1
    # the hand-off: "tape" (default) all-gathers the 48-B-per-env-step dynamics tape and expands it on every rank,
    # "packed" all-gathers the 192-B packed rows (what round 1 did; GX_HANDOFF=packed to compare)
    mode = os.environ.get("GX_HANDOFF", "tape")
    handoff = None
    if gather:
        if mode == "tape":
            try:
                handoff = gxd.TapeHandoff(env, EP_LEN)
            except Exception as exc:  # noqa: BLE001 - same code on every rank, so every rank falls back together
                print(f"bench.py: tape hand-off unavailable ({type(exc).__name__}: {exc}); using the packed rows",
                      file=sys.stderr)
                mode = "packed"
        if mode != "tape":
            handoff = RolloutHandoff(world)

    def timed_region():
        run_epochs(env, tapes, args.warmup, handoff)
        gxd.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_epochs(env, tapes, args.steps, handoff)
        torch.cuda.synchronize()
        gxd.barrier()
        return gxd.max_over_ranks(time.perf_counter() - t0, device)

    precond = None if args.no_precondition else precondition_clocks(device)
    dt = timed_region()

    env_steps = ENV_NUM * world * EP_LEN * args.steps
    value = env_steps / dt
    stepping_only = None
    if gather:
        # the same epochs without the hand-off: what the sharded stepping alone sustains (no collective on the
        # data path), so the cost of the mandated all-gather can be read off the two numbers
        gxd.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        run_epochs(env, tapes, args.steps, None)
        torch.cuda.synchronize()
        gxd.barrier()
        dt1 = gxd.max_over_ranks(time.perf_counter() - t1, device)
        W = env.obs_flat_size + 2 + 3
        stepping_only = {"value": round(env_steps / dt1, 1), "unit": "env-steps/s",
                         "ms_per_step": round(dt1 / args.steps * 1e3, 6),
                         "handoff_ms_per_epoch_exposed": round((dt - dt1) / args.steps * 1e3, 6),
                         "handoff": mode,
                         "handoff_bytes_received_per_rank_per_epoch":
                             int((world - 1) * (handoff.n if mode == "tape" else EP_LEN * ENV_NUM * W) * 4),
                         "packed_rows_bytes_per_rank_per_epoch": int(EP_LEN * ENV_NUM * W * 4),
                         "note": "same epochs with the rollout hand-off switched off (two-kernel gx_rollout); `value` "
                                 "above includes the hand-off: asynchronous all-gather of the dynamics tape, 3 in "
                                 "flight, and the observation pass over all ranks' tapes on every rank"
                                 if mode == "tape" else
                                 "same epochs with the rollout hand-off switched off; `value` above includes it "
                                 "(asynchronous all-gather of the packed rows, 3 gathered buffers in flight)"}
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
                   "layout_sampler": ("sharded over the ranks + all-gather of the valid layouts (GX_SHARD_SAMPLER=1)"
                                      if SHARDED_RESET is not None else "every rank samples all candidates (shared key)"),
                   "point_actuators": "mjcf defaults inherited (DESIGN.md 0.1)"},
    }
    try:   # the 8-GPU hand-off as arithmetic (it cannot be measured on a one-GPU box): bytes on the wire vs the epoch
        shard_bytes = int(sum(env.tape_floats(EP_LEN))) * 4
        line["handoff_model"] = {
            "shard_bytes_per_rank_per_epoch": shard_bytes,
            "packed_rows_bytes_per_rank_per_epoch": int(EP_LEN * ENV_NUM * (env.obs_flat_size + 2 + 3) * 4),
            "received_per_rank_at_8_gpus_bytes": 7 * shard_bytes,
            "allgather_ms_at_8_gpus_310GBps": round(7 * shard_bytes / 310e9 * 1e3, 4),
            "ms_per_step_this_run": round(dt / args.steps * 1e3, 4),
            "note": "one async all-gather of the dynamics tape per epoch (48 B per env-step: qpos, qvel, action, done, "
                    "two layout-row indices), overlapped with the following epoch; 310 GB/s = a realistic all-gather bus "
                    "bandwidth over 7 xGMI links (537 GB/s peak per direction); arithmetic, not a measurement"}
    except Exception as exc:  # noqa: BLE001
        line["handoff_model"] = {"error": f"{type(exc).__name__}: {exc}"[:200]}
    line["clock_preconditioning"] = (dict(precond, note="unrelated GPU work before the W warm-up epochs so that the "
                                          "firmware clock ramp (13 ms on this pool) is not inside the timed region; "
                                          "`cold_start` below is the same W + K epochs after 1 s of idle without it")
                                     if precond else None)
    if precond and world == 1:
        time.sleep(1.0)
        dtc = timed_region()
        line["cold_start"] = {"value": round(env_steps / dtc, 1), "unit": "env-steps/s",
                              "ms_per_step": round(dtc / args.steps * 1e3, 6),
                              "note": "same W warm-up + K timed epochs started from an idle GPU (1 s sleep), no preconditioning"}
    if dt < 0.010:
        line["warning"] = f"timed region {dt*1e3:.2f} ms < 10 ms: use more --steps for a meaningful rate"
    if stepping_only is not None:
        line["stepping_only"] = stepping_only
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
            # host-bound (one ctypes call per step): best of three, the box's host cores are shared with other tenants
            extra("api_step_loop_env_steps_per_s", lambda: round(max(api_loop_rate(env, tapes[0], 2000) for _ in range(3)), 1))
            extra("epoch_breakdown", lambda: epoch_breakdown(device))
            extra("closed_loop_policy_env_steps_per_s", lambda: round(closed_loop_rate(device), 1))
            extra("reset_done_heavy", lambda: reset_done_heavy(device))
            extra("other_robots", lambda: other_robots(device))
        if world == 1 and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline()
            except Exception as exc:  # noqa: BLE001
                line["cpu_baseline"] = {"error": f"{type(exc).__name__}: {exc}"[:300]}
        print(json.dumps(line), flush=True)
    gxd.barrier()
    env.close()
    if world > 1:
        import torch.distributed as dist
        if dist.is_initialized():
            dist.destroy_process_group()


if __name__ == "__main__":
    main()

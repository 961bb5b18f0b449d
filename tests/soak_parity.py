#!/usr/bin/env python3
"""Soak test (MI355X box): HIP vs CPU restatement on many more random states and longer episodes
than the unit tests, looking for rare floating-point divergences.  Exits non-zero on any mismatch.

    python tests/soak_parity.py [point|swimmer|ant|walker] [n_states] [episode_envs] [episode_steps]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from helpers import task_config, random_state, SWIMMER, ANT, WALKER  # noqa: E402
from guardx_amd import Engine  # noqa: E402
from oracle import gxo  # noqa: E402


def main():
    robot = sys.argv[1] if len(sys.argv) > 1 else "point"
    n_states = int(sys.argv[2]) if len(sys.argv) > 2 else 400000
    N = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
    T = int(sys.argv[4]) if len(sys.argv) > 4 else 600
    extra = {"point": {}, "swimmer": SWIMMER, "ant": ANT, "walker": WALKER}[robot]
    A = {"ant": 8, "walker": 10}.get(robot, 2)
    bad = 0
    t0 = time.time()
    # 1. single steps from random states, both kernel families
    for path in (1, 2):
        cfg = task_config(n_states, seed=1, **extra)
        E = Engine(cfg, n_candidates=20000); E.set_path(path); E.set_prefetch(-1)
        O = gxo.OracleEngine(cfg, n_candidates=20000)
        rng = np.random.default_rng(path)
        for trial in range(3):
            s = random_state(n_states, 8, rng, robot=robot, spread=[2.5, 30.0, 0.3][trial])
            E.set_state(s); O.set_state(s)
            act = (rng.uniform(-1, 1, (n_states, A)) * [1.0, 30.0, 0.01][trial]).astype(np.float32)
            og, rg, dg, ig = E.step(torch.from_numpy(act).cuda())
            oo, ro, do, io = O.step(act)
            for name, a, b in (("obs", og.cpu().numpy(), oo), ("rew", rg.cpu().numpy(), ro),
                               ("done", dg.cpu().numpy(), do), ("cost", ig['cost'].cpu().numpy(), io['cost']),
                               ("qacc", ig['obs']['qacc'].cpu().numpy(), io['qacc'])):
                neq = ~((a == b) | (np.isnan(a) & np.isnan(b)))
                if neq.any():
                    bad += int(neq.sum())
                    print(f"MISMATCH path={path} trial={trial} {name}: {int(neq.sum())} elements")
            sg, so = E.get_state(), O.get_state()
            for k in ('qpos', 'qvel', 'pose0', 'steps', 'done0'):
                if not np.array_equal(sg[k], so[k], equal_nan=True):
                    bad += 1
                    print(f"MISMATCH state {k} path={path} trial={trial}")
        E.close()
        print(f"path {path}: 3 x {n_states} single steps compared  ({time.time() - t0:.1f} s)", flush=True)
    # 2. long fused episodes with resets, on both persistent kernels
    # 3: two-kernel rollout (every robot since round 3); -3: the same with the layout prefetch off, which selects the
    # dynamics pass's alone-on-the-chip form where a robot has one (Swimmer: a quad of lanes per env; Ant / Walker: one or
    # two envs per wave -- the soak's N decides -- instead of four)
    for path in ((2, 1, 3) if robot == 'point' else (2, 1, 3, -3)):
        cfg = task_config(N, seed=7, num_steps=150, goal_size=2.6 if robot == 'point' else 0.8, **extra)
        E = Engine(cfg, n_candidates=1000000); E.set_path(abs(path))
        if path < 0:
            E.set_prefetch(-1)
        O = gxo.OracleEngine(cfg, n_candidates=1000000)
        np.testing.assert_array_equal(E.reset().cpu().numpy(), O.reset())
        rng = np.random.default_rng(9)
        for chunk in range(T // 100):
            acts = rng.uniform(-1, 1, (100, N, A)).astype(np.float32)
            obs, rew, cost, done = E.rollout(torch.from_numpy(acts).cuda())
            obs, rew, cost, done = (x.cpu().numpy() for x in (obs, rew, cost, done))
            for t in range(100):
                o, r, d, info = O.step(acts[t])
                o = O.reset_done()
                for name, a, b in (("obs", obs[t], o), ("rew", rew[t], r), ("done", done[t], d), ("cost", cost[t], info['cost'])):
                    if not np.array_equal(a, b, equal_nan=True):
                        bad += 1
                        print(f"MISMATCH rollout path={path} chunk={chunk} t={t} {name}")
            print(f"rollout path {path} chunk {chunk}: {100 * N} env-steps compared, dones in chunk {int(done.sum())}  ({time.time() - t0:.1f} s)", flush=True)
        E.close()
    print("TOTAL MISMATCHES:", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()

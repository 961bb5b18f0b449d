#!/usr/bin/env python3
"""Soak over configuration variants (MI355X box): the variants of test_variant_configs, several seeds, 150 steps of
Engine.step() + reset_done() with timeouts plus two 48-step fused rollouts, both kernel families and the two-kernel rollout,
HIP against the CPU restatement bit for bit.
Exits non-zero on any mismatch.

    python tests/soak_variants.py [ant|walker|point|swimmer] [seeds]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from helpers import task_config, SWIMMER, ANT, WALKER  # noqa: E402
from guardx_amd import Engine  # noqa: E402
from oracle import gxo  # noqa: E402

VARIANTS = [
    dict(hazards_num=3, lidar_num_bins=8),
    dict(hazards_num=12, lidar_num_bins=24, lidar_alias=False, hazards_keepout=0.25),
    dict(observe_vel=True, observe_acc=True),
    dict(observe_qpos=False, observe_ctrl=False, observe_goal_lidar=False),
    dict(lidar_max_dist=3.0, physics_steps_per_control_step=2, lidar_exp_gain=0.5),
    dict(hazards_num=20, goal_size=0.3, hazards_size=0.2, reward_distance=2.0, hazards_keepout=0.18,
         placements_extents=[-3, -3, 3, 3]),
    dict(pillars_num=8, observe_pillars=True, pillars_keepout=0.3, pillars_size=0.2, placements_extents=[-3, -3, 3, 3]),
    dict(robot_rot=0.7),
    dict(robot_rot=-2.4, hazards_num=5, goal_size=1.5),
]


def main():
    robot = sys.argv[1] if len(sys.argv) > 1 else "walker"
    seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    extra = {"point": {}, "swimmer": SWIMMER, "ant": ANT, "walker": WALKER}[robot]
    A = {"ant": 8, "walker": 10}.get(robot, 2)
    bad, t0, steps = 0, time.time(), 0
    for vi, v in enumerate(VARIANTS):
        for seed in range(seeds):
            for path in (1, 2, 3):       # 3: the same steps on the lane-group kernels, fused stretches on the two-kernel rollout
                N = 257
                cfg = task_config(N, seed=100 + seed, num_steps=40, **v, **extra)
                E = Engine(cfg, n_candidates=30000); E.set_path(path)
                O = gxo.OracleEngine(cfg, n_candidates=30000)
                if not np.array_equal(E.reset().cpu().numpy(), O.reset(check=False)):
                    bad += 1
                rng = np.random.default_rng(seed)
                for t in range(150):
                    act = (rng.uniform(-1, 1, (N, A)) * (3.0 if t % 37 == 5 else 1.0)).astype(np.float32)
                    og, rg, dg, ig = E.step(torch.from_numpy(act).cuda())
                    oo, ro, do, io = O.step(act)
                    ok = (np.array_equal(og.cpu().numpy(), oo, equal_nan=True) and np.array_equal(rg.cpu().numpy(), ro)
                          and np.array_equal(dg.cpu().numpy(), do) and np.array_equal(ig['cost'].cpu().numpy(), io['cost'], equal_nan=True))
                    if dg.any():
                        ok = ok and np.array_equal(E.reset_done().cpu().numpy(), O.reset_done(), equal_nan=True)
                    if not ok:
                        bad += 1
                        print(f"MISMATCH variant {vi} seed {seed} path {path} step {t}", flush=True)
                        break
                    steps += N
                # fused stretches (the two-kernel rollout where the variant allows it, the persistent kernels otherwise)
                for rep in range(2):
                    acts = rng.uniform(-1, 1, (48, N, A)).astype(np.float32)
                    obs, rew, cost, done = (x.cpu().numpy() for x in E.rollout(torch.from_numpy(acts).cuda()))
                    for t in range(48):
                        oo, ro, do, io = O.step(acts[t])
                        oo = O.reset_done()
                        if not (np.array_equal(obs[t], oo, equal_nan=True) and np.array_equal(rew[t], ro) and
                                np.array_equal(done[t], do) and np.array_equal(cost[t], io['cost'], equal_nan=True)):
                            bad += 1
                            print(f"MISMATCH variant {vi} seed {seed} path {path} fused rep {rep} t {t}", flush=True)
                            break
                    steps += 48 * N
                E.close()
        print(f"variant {vi}: done ({time.time() - t0:.0f} s, {steps} env-steps compared, mismatches so far {bad})", flush=True)
    print("TOTAL MISMATCHES:", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()

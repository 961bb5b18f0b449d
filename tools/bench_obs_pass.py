#!/usr/bin/env python3
"""Observation pass alone (gx_expand_tape on a resident tape) and the two-kernel gx_rollout, HIP-event timed.
GX_OBS_GRID_CAP=<workgroups> python tools/bench_obs_pass.py [--robot xmls/point.xml] [--shards 1]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--robot", default="xmls/point.xml")
ap.add_argument("--shards", type=int, default=1)
ap.add_argument("--n", type=int, default=60)
a = ap.parse_args()
dev = torch.device("cuda", 0)
env = bench.make_engine(bench.ENV_NUM, 0, 1, n_candidates=200_000, robot_base=a.robot)
env.set_prefetch(-1)
env.reset()
A = env.action_space.shape[0]
acts = bench.action_tape(bench.EP_LEN, bench.ENV_NUM, 1, dev, A)
sh, tok = env.rollout_tape(acts)
T, W = bench.EP_LEN, env.obs_flat_size + A + 3
n = sh.numel()
pad = (-n) % 4
buf = torch.zeros(a.shards * (n + pad), device=dev)
for s in range(a.shards):
    buf[s * (n + pad):s * (n + pad) + n] = sh
out = torch.empty(a.shards, T, bench.ENV_NUM, W, device=dev)


def timeit(fn, reps):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        ts.append((e0, e1))
    torch.cuda.synchronize()
    v = np.array([x.elapsed_time(y) for x, y in ts]) * 1e3
    return float(np.median(v)), float(v.min())


if a.shards == 1:
    m, b = timeit(lambda: env.expand_tape(sh, tok, T, out=out[0]), a.n)
else:
    m, b = timeit(lambda: env.expand_tapes(buf, n + pad, a.shards, tok, T, out), a.n)
mr, br = timeit(lambda: env.rollout(acts), a.n)
print(f"cap={os.environ.get('GX_OBS_GRID_CAP', 'none'):>6} robot={a.robot} shards={a.shards}: expand median {m:7.1f} us  min {b:7.1f} us "
      f"({m / a.shards:6.1f} per shard)   gx_rollout median {mr:7.1f} min {br:7.1f}")

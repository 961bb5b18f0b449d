#!/usr/bin/env python3
"""Epoch rate (reset + one 200-step rollout, env_num = 2000) of one task, R repetitions of E epochs: median / min / max.
The in-situ number (sampler prefetch beside the rollout), as bench.py's other_robots measures it -- for same-box A/B runs of
library variants (tools/build_variant.py):

    python tools/ab_epoch.py Goal_Swimmer_8Hazards [--reps 7] [--epochs 40] [--tag name]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from guardx_amd import Engine, configuration  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("name")
ap.add_argument("--reps", type=int, default=7)
ap.add_argument("--epochs", type=int, default=40)
ap.add_argument("--tag", default=os.environ.get("GX_LIB", "product"))
ap.add_argument("--alone", action="store_true", help="layout prefetch off: the rollout kernels alone on the chip")
args = ap.parse_args()
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
cfg = dict(configuration(args.name))
cfg.update(env_num=bench.ENV_NUM, _seed=0, num_steps=bench.EP_LEN, device_id=0)
env = Engine(cfg)
if args.alone:
    env.set_prefetch(-1)
tape = bench.action_tape(bench.EP_LEN, bench.ENV_NUM, 0, dev, env.action_space.shape[0])


def epoch():
    if not args.alone:
        env.reset(check=False)
    env.rollout(tape)


if args.alone:
    env.reset()
for _ in range(5):
    epoch()
rates = []
for _ in range(args.reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.epochs):
        epoch()
    torch.cuda.synchronize()
    rates.append(bench.ENV_NUM * bench.EP_LEN * args.epochs / (time.perf_counter() - t0))
env.check_layouts()
rates.sort()
print(json.dumps({"task": args.name, "tag": args.tag, "alone": args.alone, "median_M": round(rates[len(rates) // 2] / 1e6, 1),
                  "min_M": round(rates[0] / 1e6, 1), "max_M": round(rates[-1] / 1e6, 1),
                  "ms_per_epoch_median": round(bench.ENV_NUM * bench.EP_LEN / rates[len(rates) // 2] * 1e3, 4)}), flush=True)

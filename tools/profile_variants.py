#!/usr/bin/env python3
"""single-step launches of several kernel variants (for rocprofv3 --kernel-trace --stats)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from guardx_amd import Engine
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
for kw in (dict(emit_qacc=True), dict(emit_qacc=False)):
    cfg = dict(bench.TASK); cfg.update(env_num=2000, _seed=0, num_steps=200)
    env = Engine(cfg, n_candidates=100000, **kw)
    env.set_prefetch(-1)
    env.reset()
    act = bench.action_tape(1, 2000, 0, dev)[0]
    for _ in range(300):
        env.step(act)
    torch.cuda.synchronize()
    tape = bench.action_tape(5, 2000, 0, dev)
    for _ in range(100):
        env.rollout(tape)
    torch.cuda.synchronize()
    env.close()
print("done")

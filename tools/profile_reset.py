#!/usr/bin/env python3
"""reset() x3 with the inline sampler (for rocprofv3).  --task: a guardx_amd.configuration() name (default: the bench task)"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
ap = argparse.ArgumentParser()
ap.add_argument("--task", default=None)
a = ap.parse_args()
torch.cuda.set_device(0)
if a.task:
    from guardx_amd import Engine, configuration
    cfg = dict(configuration(a.task))
    cfg.update(env_num=2000, _seed=0, num_steps=200, device_id=0)
    env = Engine(cfg)
else:
    env = bench.make_engine(2000, 0, 1)
env.set_prefetch(-1)
for _ in range(3):
    env.reset()
torch.cuda.synchronize()
print("done", env.layout_size)

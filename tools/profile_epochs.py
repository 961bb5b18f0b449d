#!/usr/bin/env python3
"""A few bench epochs (reset + 200-step rollout, env_num=2000) of one task, for rocprofv3:
    rocprofv3 --kernel-trace --stats -- python3 tools/profile_epochs.py Ant_8Hazards_8Pillars_synthetic [epochs]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from guardx_amd import Engine, configuration
name = sys.argv[1] if len(sys.argv) > 1 else "Goal_Point_8Hazards"
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 10
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
cfg = dict(configuration(name))
cfg.update(env_num=bench.ENV_NUM, _seed=0, num_steps=bench.EP_LEN, device_id=0)
env = Engine(cfg)
tape = bench.action_tape(bench.EP_LEN, bench.ENV_NUM, 0, dev, env.action_space.shape[0])
for _ in range(epochs):
    env.reset(check=False)
    env.rollout(tape)
torch.cuda.synchronize()
env.check_layouts()
print("done", name, epochs)

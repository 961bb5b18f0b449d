#!/usr/bin/env python3
"""reset() x3 with the inline sampler (for rocprofv3)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
torch.cuda.set_device(0)
env = bench.make_engine(2000, 0, 1)
env.set_prefetch(-1)
for _ in range(3):
    env.reset()
torch.cuda.synchronize()
print("done")

#!/usr/bin/env python3
"""closed-loop policy rollout by hidden width: env-steps/s and us per control step (reset() + rollout_policy per epoch)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda", 0)
for h in (64, 128, 192, 256):
    r = bench.closed_loop_rate(dev, 20, h)
    print(f"hidden {h}: {r/1e6:7.1f} M env-steps/s   {bench.ENV_NUM/r*1e6:6.2f} us per control step (incl. reset() every 200)")

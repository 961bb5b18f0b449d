#!/usr/bin/env python3
"""Bandwidth-regime timing of the thread-per-env step kernel (2^22 envs by default)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
torch.cuda.set_device(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 22
print(json.dumps(bench.roofline_step(n, 30, torch.device("cuda", 0))))

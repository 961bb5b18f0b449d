#!/usr/bin/env python3
"""bench.reset_done_heavy of the tree this is run from, five times (same-box A/B of two trees, like tools/ab_api.py)"""
import json
import os
import sys
sys.path.insert(0, os.getcwd())
import torch  # noqa: E402
import bench  # noqa: E402
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
bench.precondition_clocks(dev)
r = sorted(bench.reset_done_heavy(dev)["env_steps_per_s"] for _ in range(5))
print(json.dumps({"tag": sys.argv[1] if len(sys.argv) > 1 else "", "median_M": round(r[2] / 1e6, 1), "min_M": round(r[0] / 1e6, 1), "max_M": round(r[-1] / 1e6, 1)}))

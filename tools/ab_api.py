#!/usr/bin/env python3
"""The learner-driven step()+reset_done() loop (bench.api_loop_rate) of the tree this is run from: median / best of 7.
For same-box A/B of two trees:  (cd _r04 && python ../tools/ab_api.py r04); python tools/ab_api.py new"""
import json
import os
import sys
sys.path.insert(0, os.getcwd())
import torch  # noqa: E402
import bench  # noqa: E402
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
env = bench.make_engine(bench.ENV_NUM, 0, 1)
env.set_prefetch(bench.EP_LEN)
tape = bench.action_tape(bench.EP_LEN, bench.ENV_NUM, 0, dev)
bench.api_loop_rate(env, tape, 1000)
rates = sorted(bench.api_loop_rate(env, tape, 2000) for _ in range(7))
print(json.dumps({"tag": sys.argv[1] if len(sys.argv) > 1 else "", "median_M": round(rates[3] / 1e6, 1), "best_M": round(rates[-1] / 1e6, 1),
                  "worst_M": round(rates[0] / 1e6, 1)}))

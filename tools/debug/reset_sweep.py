#!/usr/bin/env python3
"""inline reset() time against the number of layout candidates (residency / tail effects of the sampler kernels)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
torch.cuda.set_device(0)
for M in [int(x) for x in (sys.argv[1:] or "700000 800000 850000 900000 950000 1000000 1100000 1300000 2000000".split())]:
    env = bench.make_engine(2000, 0, 1, n_candidates=M)
    env.set_prefetch(-1)
    for _ in range(3):
        env.reset()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        env.reset(check=False)
    b.record(); torch.cuda.synchronize()
    print(M, "reset %.1f us  -> %.3f ns/candidate" % (a.elapsed_time(b) * 100, a.elapsed_time(b) * 1e5 / M), flush=True)
    del env

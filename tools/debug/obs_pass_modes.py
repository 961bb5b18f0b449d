"""gx_rollout unpacked vs packed vs tape + expand at env_num=2000, T=200 (HIP events, 30 launches each)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
torch.cuda.set_device(0); dev = torch.device("cuda", 0)
env = bench.make_engine(2000, 0, 1, n_candidates=200000)
env.set_prefetch(-1)
env.reset()
acts = bench.action_tape(200, 2000, 0, dev)
def timeit(fn, n=30):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1000 / n
print("rollout unpacked      %.1f us" % timeit(lambda: env.rollout(acts)))
print("rollout packed        %.1f us" % timeit(lambda: env.rollout(acts, packed=True)))
sh, tok = env.rollout_tape(acts)
print("rollout_tape          %.1f us" % timeit(lambda: env.rollout_tape(acts, out=sh)))
sh, tok = env.rollout_tape(acts, out=sh)
out = env.expand_tape(sh, tok, 200)
print("expand_tape           %.1f us" % timeit(lambda: env.expand_tape(sh, tok, 200, out=out)))

"""Shader-clock stamps of the lane-group kernel at T=1 (Engine.step) and T=200 (rollout): where a wave's time goes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from guardx_amd import _native
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
env = bench.make_engine(2000, 0, 1, n_candidates=100000)
env.reset()
lib = _native.load()
G = 500
st = torch.zeros(G, 8, dtype=torch.int64, device=dev)
act = bench.action_tape(1, 2000, 0, dev)[0]
for _ in range(20): env.step(act)
_native.check(lib.gx_debug_stamps(env._h, st.data_ptr()))
def report(tag):
    torch.cuda.synchronize()
    s = st.cpu().numpy().astype(np.int64)
    t0 = s[:, 0].min()
    d = np.diff(s[:, [0, 1, 2, 3, 4, 5, 6, 7]], axis=1)
    print(tag, "per-workgroup deltas in s_memtime ticks (median / p90):")
    for k, n in enumerate(["loads issued", "loads arrived", "-> step t* start", "t*: dynamics", "t*: lidar+exchange", "t*: reward/rows/reset", "-> state stored"]):
        print(f"   {n:22s} {np.median(d[:, k]):9.0f} {np.percentile(d[:, k], 90):9.0f}")
    print(f"   wave lifetime median {np.median(s[:, 7] - s[:, 0]):.0f}")
for rep in range(3):
    env.step(act); report(f"T=1 step #{rep}")
tape = bench.action_tape(200, 2000, 0, dev)
env.rollout(tape); report("T=200 rollout")
_native.check(lib.gx_debug_stamps(env._h, None))

"""Shader-clock stamps of the closed-loop policy rollout kernel (step t* of a 200-step launch)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from guardx_amd import _native, Engine
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
env = bench.make_engine(2000, 0, 1, n_candidates=100000)
env.reset()
lib = _native.load()
D = env.obs_flat_size
torch.manual_seed(0)
mk = lambda out: torch.nn.Sequential(torch.nn.Linear(D, 64), torch.nn.Tanh(), torch.nn.Linear(64, 64), torch.nn.Tanh(), torch.nn.Linear(64, out))
params = Engine.pack_actor_critic(mu_net=mk(2), v_net=mk(1), log_std=torch.full((2,), -0.5)).to(dev)
for impl in (2, 1):
    env.set_policy_impl(impl)
    env.rollout_policy(params, 200)
    G = 2000
    st = torch.zeros(G, 8, dtype=torch.int64, device=dev)
    _native.check(lib.gx_debug_stamps(env._h, st.data_ptr()))
    env.rollout_policy(params, 200)
    torch.cuda.synchronize()
    _native.check(lib.gx_debug_stamps(env._h, None))
    s = st.cpu().numpy().astype(np.int64)
    s = s[s[:, 0] != 0]
    d = np.diff(s, axis=1)
    print(f"impl {impl}: {len(s)} stamped workgroups; deltas in s_memtime ticks (median / p90)")
    for k, n in enumerate(["loads issued", "loads arrived", "-> step t* start", "t*: policy + dynamics", "t*: lidar+exchange", "t*: reward/rows/reset", "-> state stored"]):
        print(f"   {n:24s} {np.median(d[:, k]):9.0f} {np.percentile(d[:, k], 90):9.0f}")
    print(f"   lifetime median {np.median(s[:, 7] - s[:, 0]):.0f}  => per step {np.median(s[:, 7] - s[:, 0]) / 200:.0f} ticks")
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): env.rollout_policy(params, 200)
    b.record(); torch.cuda.synchronize()
    print(f"   rollout_policy(200): {a.elapsed_time(b) * 100:.1f} us per launch = {a.elapsed_time(b) * 100 / 200:.2f} us per step")

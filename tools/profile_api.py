#!/usr/bin/env python3
"""Host-side cost breakdown of Engine.step (MI355X box)."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from guardx_amd import _native

torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
env = bench.make_engine(2000, 0, 1, n_candidates=100000)
env.reset()
act = bench.action_tape(1, 2000, 0, dev)[0]
lib = _native.load()
n = 3000


def t(fn, label):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{label:44s} host {1e6*(t1-t0)/n:7.2f} us/iter   (+drain {1e6*(t2-t1)/n:6.2f})")


t(lambda: env.step(act), "env.step")
t(lambda: env.reset_done(), "env.reset_done")
t(lambda: torch.empty(2000, 43, device=dev), "torch.empty(N,D)")
t(lambda: [torch.empty(2000, device=dev) for _ in range(3)], "3x torch.empty(N)")
t(lambda: env._stream(), "_stream()")
t(lambda: env._as_action(act), "_as_action")
obs = torch.empty(2000, 43, device=dev); r = torch.empty(2000, device=dev); c = torch.empty(2000, device=dev)
d = torch.empty(2000, device=dev); q = torch.empty(2000, 3, device=dev)
st = env._stream()
t(lambda: lib.gx_step(env._h, act.data_ptr(), obs.data_ptr(), r.data_ptr(), c.data_ptr(), d.data_ptr(), q.data_ptr(), st),
  "raw gx_step ctypes call")
t(lambda: lib.gx_reset_done(env._h, obs.data_ptr(), obs.data_ptr(), st), "raw gx_reset_done ctypes call")
t(lambda: lib.gx_obs_dim(env._h), "trivial ctypes call")
t(lambda: act.data_ptr(), "data_ptr()")

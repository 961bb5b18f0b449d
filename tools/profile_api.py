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

# the pair an unmodified learner drives, by output ownership mode (round 4: slab sized by bytes)
from guardx_amd import Engine
tape = bench.action_tape(bench.EP_LEN, 2000, 0, dev)
for ring, slab in ((0, 256), (0, 32), (8, 256)):
    e2 = bench.make_engine(2000, 0, 1)
    e2._out_ring = ring
    e2._SLAB_STEPS = slab
    e2.set_prefetch(bench.EP_LEN)
    rates = sorted(bench.api_loop_rate(e2, tape, 2000) for _ in range(5))
    print(f"api loop out_ring={ring} slab cap {slab}: median {rates[2]/1e6:.1f} M  best {rates[-1]/1e6:.1f} M env-steps/s  "
          f"({2000/rates[2]*1e6:.2f} us per step()+reset_done() pair)   slab steps {e2._slab_steps()}")
    t(lambda: e2.step(act), "  step() alone, outputs dropped")
    keep = []
    t(lambda: keep.append(e2.step(act)[0]) or (len(keep) > 400 and keep.clear()), "  step() alone, outputs kept (400)")
    e2.close()

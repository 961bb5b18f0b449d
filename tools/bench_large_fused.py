#!/usr/bin/env python3
"""Bandwidth regime, K fused steps per launch: Engine.rollout at 2^22 envs (thread-per-env persistent kernel)
next to K single-step launches (step + reset_done kernels)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 22
K = int(sys.argv[2]) if len(sys.argv) > 2 else 32
env = bench._fresh_engine(n)
tape = bench.action_tape(K, n, 3, dev)


def timeit(fn, reps):
    fn(); fn(); fn(); torch.cuda.synchronize()      # the first calls pay hipMalloc of the outputs
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


t_f = timeit(lambda: env.rollout(tape), 5)


def loop():
    for t in range(K):
        env.step(tape[t]); env.reset_done()


t_l = timeit(loop, 3)
print(f"N={n} K={K}: fused rollout {t_f/K*1e6:.1f} us/step = {n*K/t_f/1e9:.2f} G env-steps/s ; "
      f"K x (step + reset_done) launches {t_l/K*1e6:.1f} us/step = {n*K/t_l/1e9:.2f} G env-steps/s")
print(f"algorithmic bytes: fused {(8+172+12) + (372-192)/K:.0f} B/env-step -> {((8+172+12) + 180/K)*n*K/t_f/1e12:.2f} TB/s ; "
      f"per-step kernels 372 B -> {372*n*K/t_l/1e12:.2f} TB/s")

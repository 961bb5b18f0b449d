#!/usr/bin/env python3
"""Micro-benchmarks of the pieces of one epoch (MI355X): fused rollout alone, reset alone
(sampler inline vs prefetched), and the overlap of the two."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402


def timeit(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    robot = sys.argv[2] if len(sys.argv) > 2 else None      # e.g. xmls/swimmer.xml
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    env = bench.make_engine(N, 0, 1, robot_base=robot)
    print("robot", env.robot_base, "obs", env.obs_flat_size)
    tape = bench.action_tape(200, N, 0, dev, env.action_space.shape[0])
    env.set_prefetch(-1)
    env.reset()
    t_roll = timeit(lambda: env.rollout(tape), 20)
    t_reset = timeit(lambda: env.reset(), 10)
    print(f"N={N} rollout(200) alone: {t_roll*1e3:.3f} ms = {t_roll/200*1e6:.2f} us/step ; reset (inline sampler): {t_reset*1e3:.3f} ms")

    def epoch():
        env.reset()
        env.rollout(tape)
    t_seq = timeit(epoch, 10)
    env.set_prefetch(200)
    epoch()
    t_ovl = timeit(epoch, 20)
    print(f"epoch sequential {t_seq*1e3:.3f} ms ; with side-stream prefetch {t_ovl*1e3:.3f} ms "
          f"-> {N*200/t_ovl/1e6:.1f} M env-steps/s")
    for T in (1, 10, 50):
        tp = tape[:T].contiguous()
        t = timeit(lambda: env.rollout(tp), 50)
        print(f"rollout(T={T}): {t*1e6:.1f} us total, {t/T*1e6:.2f} us/step")
    act = tape[0]
    t = timeit(lambda: env.step(act), 200)
    print(f"Engine.step python loop: {t*1e6:.2f} us/step")


if __name__ == "__main__" and not (len(sys.argv) > 3 and sys.argv[3] == "policy"):
    main()


def policy_bench():
    """closed-loop epochs: reset + rollout_policy(200) with the default (64,64) tanh actor-critic"""
    import torch
    from guardx_amd import Engine
    N = 2000
    dev = torch.device("cuda", 0)
    env = bench.make_engine(N, 0, 1)
    D = env.obs_flat_size
    torch.manual_seed(0)
    mk = lambda out: torch.nn.Sequential(torch.nn.Linear(D, 64), torch.nn.Tanh(), torch.nn.Linear(64, 64),  # noqa: E731
                                         torch.nn.Tanh(), torch.nn.Linear(64, out))
    params = Engine.pack_actor_critic(mu_net=mk(2), v_net=mk(1), log_std=torch.full((2,), -0.5)).to(dev)
    for impl in (1, 2):
        env.set_policy_impl(impl)
        env.set_prefetch(-1)
        env.reset()
        t_roll = timeit(lambda: env.rollout_policy(params, 200), 10)
        print(f"impl={impl} ({'VALU' if impl == 1 else 'MFMA'}): policy rollout(200) alone {t_roll*1e3:.3f} ms = "
              f"{t_roll/200*1e6:.2f} us/step")

    def epoch():
        env.reset()
        env.rollout_policy(params, 200)
    env.set_prefetch(200)
    epoch()
    t_ep = timeit(epoch, 20)
    print(f"policy rollout(200) alone: {t_roll*1e3:.3f} ms = {t_roll/200*1e6:.2f} us/step ; closed-loop epoch "
          f"(prefetch): {t_ep*1e3:.3f} ms -> {N*200/t_ep/1e6:.1f} M env-steps/s")


if __name__ == "__main__" and len(sys.argv) > 3 and sys.argv[3] == "policy":
    policy_bench()

#!/usr/bin/env python3
"""The bench epoch (reset + 200-step rollout, env_num = 2000) two ways per robot: Engine.rollout (dynamics pass, then the
observation pass, on the caller's stream) and guardx_amd.dist.TapeHandoff in a world of one (the observation pass of epoch
k runs on the hand-off's stream during epoch k + 1: packed rows one epoch late)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from guardx_amd import Engine
from guardx_amd.dist import TapeHandoff

dev = torch.device("cuda", 0)
bench.precondition_clocks(dev)
for robot in ("xmls/point.xml", "xmls/swimmer.xml", "xmls/ant.xml", "xmls/walker.xml"):
    res = {}
    for mode in ("rollout", "pipelined"):
        env = bench.make_engine(bench.ENV_NUM, 0, 1, robot_base=robot)
        env.set_prefetch(bench.EP_LEN)
        A = env.action_space.shape[0]
        tape = bench.action_tape(bench.EP_LEN, bench.ENV_NUM, 0, dev, A)
        h = TapeHandoff(env, bench.EP_LEN, sharded_sampler=False) if mode == "pipelined" else None

        def epoch():
            env.reset(check=False)
            if h is None:
                env.rollout(tape)
            else:
                h.step(tape)
        for _ in range(5):
            epoch()
        torch.cuda.synchronize()
        n = 60
        t0 = time.perf_counter()
        for _ in range(n):
            epoch()
        if h is not None:
            h.drain()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        env.check_layouts()
        env.close()
        res[mode] = dt
    print(f"{robot:18s} rollout {res['rollout']*1e3:.4f} ms = {bench.ENV_NUM*bench.EP_LEN/res['rollout']/1e6:6.1f} M   "
          f"pipelined {res['pipelined']*1e3:.4f} ms = {bench.ENV_NUM*bench.EP_LEN/res['pipelined']/1e6:6.1f} M")

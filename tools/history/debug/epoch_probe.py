"""Headline epoch (Point) and bench.other_robots, after clock preconditioning."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
bench.precondition_clocks(dev)
env = bench.make_engine(bench.ENV_NUM, 0, 1); env.set_prefetch(bench.EP_LEN)
tapes = [bench.action_tape(bench.EP_LEN, bench.ENV_NUM, k, dev) for k in range(4)]
bench.run_epochs(env, tapes, 60, None); torch.cuda.synchronize()
for _ in range(2):
    t0 = time.perf_counter(); bench.run_epochs(env, tapes, 200, None); torch.cuda.synchronize()
    print(f"Point epoch {(time.perf_counter() - t0) / 200 * 1e6:.1f} us", flush=True)
env.close()
for k, v in bench.other_robots(dev).items():
    print(k, round(v['env_steps_per_s'] / 1e6, 1), "M", v['ms_per_epoch'], "ms", flush=True)

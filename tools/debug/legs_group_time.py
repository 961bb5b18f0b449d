"""us per step of the lane-group rollout kernel at env_num=2000, T=200 (Ant, Walker)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
for xml, A in (("xmls/ant.xml", 8), ("xmls/walker.xml", 10)):
    env = bench.make_engine(2000, 0, 1, n_candidates=300000, robot_base=xml)
    env.set_prefetch(-1); env.reset()
    tape = bench.action_tape(200, 2000, 0, dev, A)
    env.rollout(tape); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): env.rollout(tape)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3 / 200
    print(f"{xml}: {dt*1e6:.2f} us/step", flush=True)
    env.close()

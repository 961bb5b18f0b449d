"""us per step of the lane-group rollout at env_num=2000, T=200 for the synthetic config 5 (Ant + 8 hazards + 8 pillars)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from guardx_amd import Engine, configuration
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
cfg = dict(configuration("Ant_8Hazards_8Pillars_synthetic"), env_num=2000, _seed=0, num_steps=200, device_id=0)
env = Engine(cfg, n_candidates=300000)
env.set_prefetch(-1); env.reset()
tape = bench.action_tape(200, 2000, 0, dev, 8)
env.rollout(tape); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3): env.rollout(tape)
torch.cuda.synchronize()
print(f"config 5: {(time.perf_counter() - t0) / 3 / 200 * 1e6:.2f} us/step")

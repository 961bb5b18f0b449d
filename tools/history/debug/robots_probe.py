"""Epoch rates of the other robots (bench.other_robots) and their rollout alone, per fused step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
bench.precondition_clocks(dev)
print(bench.other_robots(dev), flush=True)
from guardx_amd import Engine, configuration
for name in ("Goal_Swimmer_8Hazards", "Goal_Ant_8Hazards", "Goal_Walker_8Hazards"):
    for mode in (0, 2):
        env = Engine({**configuration(name), 'env_num': 2000, '_seed': 0, 'num_steps': 200}, n_candidates=200000)
        env.set_prefetch(-1); env.reset(); env.set_path(mode)
        A = env.action_space.shape[0]
        acts = torch.rand(200, 2000, A, device=dev) * 2 - 1
        for _ in range(3): env.rollout(acts)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): env.rollout(acts)
        torch.cuda.synchronize()
        print(f"{name} path_mode {mode}: {(time.perf_counter()-t0)/10/200*1e6:.2f} us per fused step", flush=True)
        env.close()

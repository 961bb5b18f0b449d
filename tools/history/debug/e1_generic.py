"""E1: mode-1 fused rollout on the generic lane-group kernel with scattered dones, against the checker.
usage: e1_generic.py <repo root> <robot>"""
import sys, os
ROOT = sys.argv[1]
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from helpers import task_config, WALKER, ANT
from guardx_amd import Engine
from oracle import gxo
def eq(a, b): return np.array_equal(a, b, equal_nan=True)
v = dict(hazards_num=12, lidar_num_bins=24, lidar_alias=False, hazards_keepout=0.25)
N = 130
robot = sys.argv[2]
extra, A = {"walker": (WALKER, 10), "ant": (ANT, 8), "point": ({}, 2)}[robot]
cfg = task_config(N, seed=9, num_steps=50, goal_size=2.5, **v, **extra)
E = Engine(cfg, n_candidates=30000); E.set_path(2)
O = gxo.OracleEngine(cfg, n_candidates=30000)
E.reset(); O.reset(check=False)
rng = np.random.default_rng(3)
acts = rng.uniform(-1, 1, (40, N, A)).astype(np.float32)
obs, rew, cost, done = E.rollout(torch.from_numpy(acts).cuda())
torch.cuda.synchronize()
first = None
for t in range(40):
    oo, ro, do, io = O.step(acts[t]); rd = O.reset_done()
    ok = eq(obs[t].cpu().numpy(), rd) and eq(rew[t].cpu().numpy(), ro) and eq(done[t].cpu().numpy(), do)
    if not ok and first is None:
        first = (t, int(do.sum()), np.nonzero(do)[0][:6])
print("E1", ROOT, robot, "OK" if first is None else f"first mismatch at t={first[0]} ndone={first[1]} {first[2]}", "total dones", int(done.sum().item()))

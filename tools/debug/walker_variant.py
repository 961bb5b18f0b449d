import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
from helpers import task_config, WALKER
from guardx_amd import Engine
from oracle import gxo
variants = [
    dict(hazards_num=3, lidar_num_bins=8),
    dict(hazards_num=12, lidar_num_bins=24, lidar_alias=False, hazards_keepout=0.25),
    dict(observe_vel=True, observe_acc=True),
    dict(observe_qpos=False, observe_ctrl=False, observe_goal_lidar=False),
    dict(lidar_max_dist=3.0, physics_steps_per_control_step=2, lidar_exp_gain=0.5),
    dict(hazards_num=20, goal_size=0.3, hazards_size=0.2, reward_distance=2.0, hazards_keepout=0.18, placements_extents=[-3, -3, 3, 3]),
]
for vi, v in enumerate(variants):
    N = 130
    cfg = task_config(N, seed=9, num_steps=50, **v, **WALKER)
    E = Engine(cfg, n_candidates=30000); E.set_path(2)
    O = gxo.OracleEngine(cfg, n_candidates=30000)
    E.reset(); O.reset(check=False)
    rng = np.random.default_rng(3)
    for t in range(60):
        act = rng.uniform(-1, 1, (N, 10)).astype(np.float32)
        og, rg, dg, ig = E.step(torch.from_numpy(act).cuda())
        oo, ro, do, io = O.step(act)
        rg = rg.cpu().numpy(); og = og.cpu().numpy()
        bad = ~((rg == ro) | (np.isnan(rg) & np.isnan(ro)))
        if bad.any():
            i = int(np.nonzero(bad)[0][0])
            print("variant", vi, v, "t", t, "env", i, "rew gpu", rg[i], "oracle", ro[i], "done", dg[i].item(), do[i])
            print(" obs finite gpu", np.isfinite(og[i]).all(), "oracle", np.isfinite(oo[i]).all())
            sg, so = E.get_state(), O.get_state()
            print(" qpos gpu", sg['qpos'][i], "\n qpos ora", so['qpos'][i], "\n steps", sg['steps'][i], so['steps'][i], "pose0", sg['pose0'][i], so['pose0'][i])
            print(" obs gpu", og[i][:12], "\n obs ora", oo[i][:12])
            sys.exit(0)
        if t % 7 == 6:
            a = E.reset_done().cpu().numpy(); b = O.reset_done()
            if not np.array_equal(a, b, equal_nan=True): print("reset_done obs mismatch variant", vi, "t", t); sys.exit(0)
print("no mismatch")

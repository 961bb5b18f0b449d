import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from helpers import task_config, WALKER, ANT
from guardx_amd import Engine
from oracle import gxo
np.set_printoptions(precision=5, linewidth=200)
def eq(a, b): return np.array_equal(a, b, equal_nan=True)
v = dict(hazards_num=12, lidar_num_bins=24, lidar_alias=False, hazards_keepout=0.25)
N = 130
for robot, extra, A in (("walker", WALKER, 10), ("ant", ANT, 8), ("point", {}, 2)):
    # E1: mode-1 fused rollout on the generic lane-group kernel, scattered dones (wide goal)
    cfg = task_config(N, seed=9, num_steps=50, goal_size=2.5, **v, **extra)
    E = Engine(cfg, n_candidates=30000); E.set_path(2)
    O = gxo.OracleEngine(cfg, n_candidates=30000)
    E.reset(); O.reset(check=False)
    rng = np.random.default_rng(3)
    acts = rng.uniform(-1, 1, (40, N, A)).astype(np.float32)
    obs, rew, cost, done = E.rollout(torch.from_numpy(acts).cuda())
    first = None
    for t in range(40):
        oo, ro, do, io = O.step(acts[t]); rd = O.reset_done()
        ok = eq(obs[t].cpu().numpy(), rd) and eq(rew[t].cpu().numpy(), ro) and eq(done[t].cpu().numpy(), do)
        if not ok and first is None:
            first = (t, int(do.sum()), np.nonzero(do)[0][:6])
    print("E1", robot, "rollout mode 1 generic kernel:", "OK" if first is None else f"first mismatch at t={first[0]} ndone={first[1]} {first[2]}", "total dones", int(done.sum().item()))
    # E2: step()+speculation, print pose0 of the first diverging envs
    cfg = task_config(N, seed=9, num_steps=50, **v, **extra)
    E = Engine(cfg, n_candidates=30000); E.set_path(2)
    O = gxo.OracleEngine(cfg, n_candidates=30000)
    E.reset(); O.reset(check=False)
    rng = np.random.default_rng(3)
    for t in range(46):
        act = rng.uniform(-1, 1, (N, A)).astype(np.float32)
        E.step(torch.from_numpy(act).cuda()); oo, ro, do, io = O.step(act)
        sg, so = E.get_state(), O.get_state()
        if not eq(sg['pose0'], so['pose0']):
            idx = np.nonzero(~(sg['pose0'] == so['pose0']).all(1))[0]
            print("E2", robot, "t", t, "done envs", np.nonzero(do)[0], "pose0 differs at", idx)
            for i in idx[:3]:
                print("   env", i, "gpu", sg['pose0'][i], "oracle", so['pose0'][i], "qpos[:3]", so['qpos'][i][:3])
            break
        if t % 7 == 6:
            E.reset_done(); O.reset_done()
    else:
        print("E2", robot, "no pose0 divergence in 46 steps")

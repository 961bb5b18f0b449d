import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from helpers import task_config, WALKER
from guardx_amd import Engine
from oracle import gxo
v = dict(hazards_num=12, lidar_num_bins=24, lidar_alias=False, hazards_keepout=0.25)
N = 130
cfg = task_config(N, seed=9, num_steps=50, **v, **WALKER)
peek = len(sys.argv) > 1
E = Engine(cfg, n_candidates=30000); E.set_path(2)
O = gxo.OracleEngine(cfg, n_candidates=30000)
E.reset(); O.reset(check=False)
rng = np.random.default_rng(3)
def eq(a, b): return np.array_equal(a, b, equal_nan=True)
for t in range(60):
    act = rng.uniform(-1, 1, (N, 10)).astype(np.float32)
    og, rg, dg, ig = E.step(torch.from_numpy(act).cuda())
    oo, ro, do, io = O.step(act)
    res = dict(obs=eq(og.cpu().numpy(), oo), rew=eq(rg.cpu().numpy(), ro), done=eq(dg.cpu().numpy(), do), cost=eq(ig['cost'].cpu().numpy(), io['cost']))
    msg = f"t {t} " + " ".join(f"{k}={'ok' if x else 'BAD'}" for k, x in res.items()) + f" ndone {int(do.sum())}"
    if peek:
        sg, so = E.get_state(), O.get_state()
        for k in ('qpos', 'qvel', 'pose0', 'objs', 'done0', 'steps'):
            if not eq(sg[k], so[k]):
                idx = np.nonzero(~np.isclose(sg[k].reshape(N, -1), so[k].reshape(N, -1), equal_nan=True).all(1))[0]
                msg += f" STATE {k} differs at envs {idx[:8]}"
    if not all(res.values()) or 'STATE' in msg:
        bad = np.nonzero(~((rg.cpu().numpy() == ro) | (np.isnan(rg.cpu().numpy()) & np.isnan(ro))))[0]
        msg += f" rew-bad envs {bad[:8]} done-at-those {do[bad[:8]]}"
    print(msg)
    if t % 7 == 6:
        a = E.reset_done().cpu().numpy(); b = O.reset_done()
        print("   reset_done obs", "ok" if eq(a, b) else "BAD", "done envs", np.nonzero(do)[0][:10])

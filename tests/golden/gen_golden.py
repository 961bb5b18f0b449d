#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the CPU restatement (oracle/).

These are SELF-CONSISTENCY vectors: the reference cannot run here,
so they pin the restatement (and, on the GPU, the HIP path) against regressions; they
are not reference outputs.

    python tests/golden/gen_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import task_config  # noqa: E402
from oracle import gxo  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def episode(name, cfg, n_candidates, T, seed_act):
    E = gxo.OracleEngine(cfg, n_candidates=n_candidates)
    N = cfg['env_num']
    rec = {'reset_obs': E.reset(check=False), 'layout_size': np.int64(E.layout_size),
           'pool_head': E.get_pool(8)}
    rng = np.random.RandomState(seed_act)
    acts = rng.uniform(-1, 1, (T, N, E.na)).astype(np.float32)
    obs, rew, done, cost, rdo, qacc = [], [], [], [], [], []
    for t in range(T):
        o, r, d, info = E.step(acts[t])
        obs.append(o); rew.append(r); done.append(d); cost.append(info['cost']); qacc.append(info['qacc'])
        rdo.append(E.reset_done())
    rec.update(actions=acts, obs=np.stack(obs), reward=np.stack(rew), done=np.stack(done),
               cost=np.stack(cost), qacc=np.stack(qacc), reset_done_obs=np.stack(rdo))
    st = E.get_state()
    rec.update({f"final_{k}": np.asarray(v) for k, v in st.items()})
    rec['reset2_obs'] = E.reset(check=False)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **rec)
    print(name, {k: getattr(v, 'shape', v) for k, v in rec.items() if k in ('obs', 'layout_size')},
          "dones", int(np.stack(done).sum()))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    # config 0 of BASELINE.json: Goal_Point_8Hazards, env_num=4
    episode("goal_point_8hazards_n4_seed0", task_config(4, seed=0, num_steps=200), 20000, 60, 0)
    # a denser case that exercises done / reset_done / timeout
    episode("goal_point_8hazards_n24_seed5", task_config(24, seed=5, num_steps=40, goal_size=1.2), 30000, 60, 1)
    # config 2 of BASELINE.json (articulated dynamics + joint-limit rows), small
    episode("goal_swimmer_8hazards_n12_seed2",
            task_config(12, seed=2, num_steps=40, goal_size=1.0, robot_base='xmls/swimmer.xml'), 30000, 60, 2)
    # Goal_Ant_8Hazards (11-DOF tree, joint limits, foot-floor contacts), small
    episode("goal_ant_8hazards_n12_seed4",
            task_config(12, seed=4, num_steps=40, goal_size=1.0, robot_base='xmls/ant.xml'), 30000, 60, 3)
    # Goal_Walker_8Hazards (13-DOF tree with 3-D leg chains), small
    episode("goal_walker_8hazards_n12_seed6",
            task_config(12, seed=6, num_steps=40, goal_size=1.0, robot_base='xmls/walker.xml'), 30000, 60, 4)

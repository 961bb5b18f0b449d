#!/usr/bin/env python3
"""Generate REFERENCE golden vectors: tests/golden/ref_<case>.npz from the reference Engine itself.

    python tests/golden/gen_reference_golden.py [--out DIR] [--cases name,name] [--steps T]

This is the command that flips "PARITY UNPINNED" (DESIGN.md section 0): it imports
`safe_rl_envs.envs.engine.Engine` from the reference checkout (/root/reference, or $GUARDX_REFERENCE), drives
reset() / step() / reset_done() with the same configurations, seeds and action tapes as gen_golden.py, and writes
what the reference returned -- plus the state it was in before every step, so that the parity tests can compare
single steps from the reference's own states (no accumulated drift) as well as the free-running trajectory.

It runs only where the reference runs: it needs `gym`, `jax` + `jaxlib`, `mujoco` (with `mujoco.mjx`), `xmltodict`
and `torch`.  None of jax / mujoco / gym is installed in the build container or on the GPU box (no network either),
so there it prints what is missing and exits with status 3 WITHOUT writing anything; nothing is installed, stubbed
or shimmed.  The vectors are data (inputs and outputs); no reference source travels with them.

The reference draws 1e6 layout candidates per reset() whatever the config says (engine.py:263), so the fixtures are
replayed with n_candidates = 1_000_000 (tests/test_golden.py: ref_* cases).

File schema (T steps, N envs, D = flat observation, nq/nv = the ROBOT's slice of the world model's qpos/qvel):
  config_json            the Engine config dict (json) the case was generated with
  layout_size            number of valid layouts of the first reset()           (engine.py:441)
  pool_head (8, K, 2)    first 8 valid layouts, objects in placement order goal, hazard0.., robot  (engine.py:533-544)
  reset_obs (N, D)       reset()                                                 (engine.py:454-467)
  actions (T, N, A)      the action tape
  obs (T, N, D), reward, done, cost (T, N), qacc (T, N, nv)     step()          (engine.py:469-495)
  reset_done_obs (T, N, D)                                       reset_done()    (engine.py:497-505), every step
  pre_qpos (T, N, nq), pre_qvel (T, N, nv)    robot state BEFORE step t (after the previous reset_done)
  pre_pose (T, N, 4)     (x, y, xmat[0,0], xmat[1,0]) of the robot body in `_data` before step t: the stale pose
  pre_objs (T, N, K-1, 2) goal and hazard positions before step t
  pre_done (T, N), pre_steps (T, N), pre_key (T, 2)    `_done` (0 before the first step), `_steps`, PRNG key data
  final_qpos, final_qvel, final_key         after the last reset_done()
  reset2_obs (N, D)      the next reset()
  versions               "jax x.y.z mujoco a.b.c ..." (the pins the vectors were made with)
"""
import argparse
import importlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import task_config  # noqa: E402

REFERENCE = os.environ.get("GUARDX_REFERENCE", "/root/reference")

# (case name, config, steps, action seed): the gen_golden.py cases, so that ref_<case> and <case> can be compared
CASES = [
    ("goal_point_8hazards_n4_seed0", task_config(4, seed=0, num_steps=200), 60, 0),
    ("goal_point_8hazards_n24_seed5", task_config(24, seed=5, num_steps=40, goal_size=1.2), 60, 1),
    ("goal_swimmer_8hazards_n12_seed2",
     task_config(12, seed=2, num_steps=40, goal_size=1.0, robot_base='xmls/swimmer.xml'), 60, 2),
    ("goal_ant_8hazards_n12_seed4", task_config(12, seed=4, num_steps=40, goal_size=1.0, robot_base='xmls/ant.xml'), 60, 3),
    ("goal_walker_8hazards_n12_seed6",
     task_config(12, seed=6, num_steps=40, goal_size=1.0, robot_base='xmls/walker.xml'), 60, 4),
    # robot_rot (engine.py:114,342-345 -> world.py:117): the root body turned about z; DESIGN.md section 9
    ("goal_point_8hazards_n8_rot07", task_config(8, seed=1, num_steps=40, goal_size=1.2, robot_rot=0.7), 60, 5),
    ("goal_ant_8hazards_n8_rot07",
     task_config(8, seed=1, num_steps=40, goal_size=1.0, robot_rot=0.7, robot_base='xmls/ant.xml'), 60, 6),
]


def import_reference():
    """The reference Engine class, or exit(3) naming what is missing (nothing is installed or stubbed)."""
    missing = []
    for mod in ("torch", "gym", "jax", "jaxlib", "mujoco", "mujoco.mjx", "xmltodict"):
        try:
            importlib.import_module(mod)
        except Exception as exc:  # noqa: BLE001 - ImportError, or a broken install
            missing.append(f"{mod} ({type(exc).__name__}: {exc})")
    pkg = os.path.join(REFERENCE, "safe_rl_envs")
    if not os.path.isdir(pkg):
        missing.append(f"reference checkout at {REFERENCE} (set GUARDX_REFERENCE)")
    if missing:
        print("gen_reference_golden.py: cannot run the reference here; missing:\n  " + "\n  ".join(missing) +
              "\nNothing was written.  Run this script on a machine where the reference's requirements are installed "
              "(safe_rl_envs/requirements.txt: mujoco, mujoco-mjx, jax[cpu]).", file=sys.stderr)
        sys.exit(3)
    sys.path.insert(0, pkg)
    from safe_rl_envs.envs.engine import Engine      # the reference's own class, unmodified
    return Engine


def to_np(x):
    """torch tensor / jax array -> numpy"""
    if hasattr(x, "detach"):
        return x.detach().cpu().numpy().copy()
    return np.asarray(x).copy()


def key_data(key):
    try:
        import jax
        return np.asarray(jax.random.key_data(key), np.uint32).reshape(-1)[-2:]
    except Exception:  # noqa: BLE001 - raw uint32[2] keys (or no jax: the stand-in run of tests/test_pin_machinery.py)
        return np.asarray(key, np.uint32).reshape(-1)[-2:]


def run_case(Engine, name, cfg, T, seed_act, out_dir):
    import torch
    env = Engine(dict(cfg))
    N = int(cfg['env_num'])
    nq, nv = int(env.robot.nq), int(env.robot.nv)
    A = int(env.action_space.shape[0])
    rb = env.body_name2xpos_id['robot']
    obj_bodies = [env.body_name2xpos_id['goal']] + list(env.body_name2xpos_id['hazards'])

    def state():
        d = env._data
        xpos, xmat = to_np(d.xpos), to_np(d.xmat).reshape(N, -1, 3, 3)
        pose = np.stack([xpos[:, rb, 0], xpos[:, rb, 1], xmat[:, rb, 0, 0], xmat[:, rb, 1, 0]], axis=1)
        done = np.zeros(N, np.float32) if env._done is None else to_np(env._done).astype(np.float32)
        return dict(qpos=to_np(d.qpos)[:, :nq], qvel=to_np(d.qvel)[:, :nv], pose=pose.astype(np.float32),
                    objs=xpos[:, obj_bodies, :2].astype(np.float32), done=done,
                    steps=to_np(env._steps).astype(np.float32), key=key_data(env.key))

    rec = {'config_json': np.array(json.dumps(cfg, sort_keys=True))}
    rec['reset_obs'] = to_np(env.reset())
    rec['layout_size'] = np.int64(env.layout_size)
    names = list(env.placements.keys())                       # goal, hazard0.., robot (engine.py:533-544)
    rec['pool_head'] = np.stack([to_np(env.layout[k])[:8] for k in names], axis=1).astype(np.float32)
    acts = np.random.RandomState(seed_act).uniform(-1, 1, (T, N, A)).astype(np.float32)
    keys = ('qpos', 'qvel', 'pose', 'objs', 'done', 'steps', 'key')
    pre = {k: [] for k in keys}
    out = {k: [] for k in ('obs', 'reward', 'done', 'cost', 'qacc', 'reset_done_obs')}
    for t in range(T):
        s = state()
        for k in keys:
            pre[k].append(s[k])
        o, r, d, info = env.step(torch.from_numpy(acts[t]))
        out['obs'].append(to_np(o)); out['reward'].append(to_np(r)); out['done'].append(to_np(d))
        out['cost'].append(to_np(info['cost']))
        out['qacc'].append(to_np(info['obs']['qacc'])[:, :nv] if 'qacc' in info.get('obs', {}) else np.zeros((N, nv), np.float32))
        out['reset_done_obs'].append(to_np(env.reset_done()))
    rec['actions'] = acts
    rec.update({k: np.stack(v) for k, v in out.items()})
    rec.update({'pre_' + k: np.stack(v) for k, v in pre.items()})
    s = state()
    rec.update(final_qpos=s['qpos'], final_qvel=s['qvel'], final_key=s['key'])
    rec['reset2_obs'] = to_np(env.reset())
    vers = []
    for mod in ("jax", "jaxlib", "mujoco", "numpy", "torch"):
        try:
            vers.append(f"{mod} {importlib.import_module(mod).__version__}")
        except Exception:  # noqa: BLE001
            pass
    rec['versions'] = np.array(" ".join(vers))
    path = os.path.join(out_dir, "ref_" + name + ".npz")
    np.savez_compressed(path, **rec)
    print(f"wrote {path}: layout_size {int(rec['layout_size'])}, dones {int(rec['done'].sum())}, {rec['versions']}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=HERE)
    ap.add_argument("--cases", default="")
    ap.add_argument("--steps", type=int, default=0)
    args = ap.parse_args()
    Engine = import_reference()
    want = set(filter(None, args.cases.split(",")))
    os.makedirs(args.out, exist_ok=True)
    for name, cfg, T, seed in CASES:
        if want and name not in want:
            continue
        run_case(Engine, name, cfg, args.steps or T, seed, args.out)


if __name__ == "__main__":
    main()

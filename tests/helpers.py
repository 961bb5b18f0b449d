"""Shared helpers for the parity tests."""
import numpy as np

GOAL_POINT_8HAZARDS = {   # safe_rl_libX/guard_utils/safe_rl_env_config.py:59-81
    'robot_base': 'xmls/point.xml', 'task': 'goal', 'goal_size': 0.5,
    'observe_goal_comp': True, 'observe_hazards': True,
    'constrain_hazards': True, 'constrain_indicator': False,
    'lidar_num_bins': 16, 'hazards_num': 8, 'hazards_size': 0.3,
}


def task_config(env_num, seed=0, num_steps=200, **over):
    cfg = dict(GOAL_POINT_8HAZARDS)
    cfg.update(env_num=env_num, _seed=seed, num_steps=num_steps)
    cfg.update(over)
    return cfg


SWIMMER = {'robot_base': 'xmls/swimmer.xml'}
ANT = {'robot_base': 'xmls/ant.xml'}
WALKER = {'robot_base': 'xmls/walker.xml'}
WALKER_LO = np.array([-25, -30, -100, -100, -45] * 2, np.float32) * np.float32(np.pi / 180)   # walker.xml joint ranges
WALKER_HI = np.array([5, 35, 10, 0, 20] * 2, np.float32) * np.float32(np.pi / 180)
ANT_SIGMA = np.array([1, -1, -1, 1], np.float32)     # sign of the ankle axes, ant.xml:27,44,61,77


def random_state(N, H, rng, spread=2.5, done_frac=0.1, near_frac=0.3, robot='point'):
    """A random but plausible engine state (env-major arrays, see gx_get_state)."""
    f = np.float32
    if robot == 'ant':
        return _random_state_ant(N, H, rng, spread, done_frac, near_frac)
    if robot == 'walker':
        return _random_state_walker(N, H, rng, spread, done_frac, near_frac)
    nq = 3 if robot == 'point' else 5
    qpos = np.empty((N, nq), f)
    qpos[:, :2] = rng.uniform(-spread, spread, (N, 2))
    qpos[:, 2] = rng.uniform(-40, 40, N)
    qvel = np.empty((N, nq), f)
    qvel[:, :2] = rng.uniform(-3, 3, (N, 2))
    qvel[:, 2] = rng.uniform(-30, 30, N)
    if robot == 'swimmer':
        qpos[:, 2] = rng.uniform(-8, 8, N)
        qpos[:, 3:] = rng.uniform(-1.7, 1.7, (N, 2))
        k = N // 3                      # a third of the envs sit beyond a joint limit (one or both)
        qpos[:k, 3] = rng.choice([-1, 1], k) * (1.7453293 + rng.uniform(1e-6, 0.3, k))
        qpos[k // 2:k, 4] = rng.choice([-1, 1], k - k // 2) * (1.7453293 + rng.uniform(1e-6, 0.3, k - k // 2))
        qvel[:, :2] = rng.uniform(-1, 1, (N, 2))
        qvel[:, 2:] = rng.uniform(-8, 8, (N, 3))
    th_prev = rng.uniform(-np.pi, np.pi, N)
    pose0 = np.empty((N, 4), f)
    pose0[:, :2] = qpos[:, :2] - rng.uniform(-0.05, 0.05, (N, 2)).astype(f)
    pose0[:, 2] = np.cos(th_prev)
    pose0[:, 3] = np.sin(th_prev)
    pose1 = (pose0[:, :2] - rng.uniform(-0.05, 0.05, (N, 2))).astype(f)
    objs = rng.uniform(-2, 2, (N, 1 + H, 2)).astype(f)
    # put some goals / hazards right next to the robot so done / cost fire
    near = rng.random(N) < near_frac
    objs[near, 0] = qpos[near, :2] + rng.uniform(-0.6, 0.6, (near.sum(), 2)).astype(f)
    nearh = rng.random(N) < near_frac
    objs[nearh, 1] = qpos[nearh, :2] + rng.uniform(-0.35, 0.35, (nearh.sum(), 2)).astype(f)
    done0 = (rng.random(N) < done_frac).astype(f)
    done1 = (rng.random(N) < done_frac).astype(f)
    steps = rng.integers(0, 250, N).astype(f)
    return dict(qpos=qpos, qvel=qvel, pose0=pose0, pose1=pose1, objs=objs,
                done0=done0, done1=done1, steps=steps,
                key=np.array([rng.integers(0, 2**32), rng.integers(0, 2**32)], np.uint32), hist=2)


def _random_state_ant(N, H, rng, spread, done_frac, near_frac):
    """qpos = (x, th, y, hip1, ankle1, ..., hip4, ankle4); a mix of in-range legs, legs beyond their
    joint limits and feet pressed into the floor (ankle beyond ~58 deg); headings away from the
    |th| = pi/2 singularity of the x / body-y slide pair."""
    f = np.float32
    qpos = np.zeros((N, 11), f); qvel = np.zeros((N, 11), f)
    qpos[:, 0] = rng.uniform(-spread, spread, N)
    qpos[:, 1] = rng.uniform(-1.1, 1.1, N)
    qpos[:, 2] = rng.uniform(-spread, spread, N)
    for leg in range(4):
        qpos[:, 3 + 2 * leg] = rng.uniform(-0.8, 0.8, N)
        qpos[:, 4 + 2 * leg] = ANT_SIGMA[leg] * rng.uniform(0.0, 1.5, N)
    rest = rng.random(N) < 0.1            # freshly reset robots: all joints at zero
    qpos[rest, 1] = 0.0; qpos[rest, 3:] = 0.0
    qvel[:, 0] = rng.uniform(-1, 1, N); qvel[:, 2] = rng.uniform(-1, 1, N)
    qvel[:, 1] = rng.uniform(-3, 3, N)
    qvel[:, 3:] = rng.uniform(-6, 6, (N, 8))
    qvel[rest] = 0.0
    th_prev = qpos[:, 1] + rng.uniform(-0.1, 0.1, N)
    pose0 = np.empty((N, 4), f)
    pose0[:, 0] = qpos[:, 0] - np.sin(qpos[:, 1]) * qpos[:, 2] - rng.uniform(-0.05, 0.05, N)
    pose0[:, 1] = np.cos(qpos[:, 1]) * qpos[:, 2] - rng.uniform(-0.05, 0.05, N)
    pose0[:, 2] = np.cos(th_prev); pose0[:, 3] = np.sin(th_prev)
    pose1 = (pose0[:, :2] - rng.uniform(-0.05, 0.05, (N, 2))).astype(f)
    objs = rng.uniform(-2, 2, (N, 1 + H, 2)).astype(f)
    near = rng.random(N) < near_frac
    objs[near, 0] = pose0[near, :2] + rng.uniform(-0.6, 0.6, (near.sum(), 2)).astype(f)
    nearh = rng.random(N) < near_frac
    objs[nearh, 1] = pose0[nearh, :2] + rng.uniform(-0.35, 0.35, (nearh.sum(), 2)).astype(f)
    done0 = (rng.random(N) < done_frac).astype(f)
    done1 = (rng.random(N) < done_frac).astype(f)
    steps = rng.integers(0, 250, N).astype(f)
    return dict(qpos=qpos, qvel=qvel, pose0=pose0, pose1=pose1, objs=objs,
                done0=done0, done1=done1, steps=steps,
                key=np.array([rng.integers(0, 2**32), rng.integers(0, 2**32)], np.uint32), hist=2)


def _random_state_walker(N, H, rng, spread, done_frac, near_frac):
    """qpos = (x, th, y, right leg 5, left leg 5): legs inside and beyond their joint ranges (feet pressed into
    the floor when the foot joint pitches down), headings away from the |th| = pi/2 singularity."""
    f = np.float32
    s = _random_state_ant(N, H, rng, spread, done_frac, near_frac)
    qpos = np.zeros((N, 13), f); qvel = np.zeros((N, 13), f)
    qpos[:, :3] = s['qpos'][:, :3]
    w = WALKER_HI - WALKER_LO
    qpos[:, 3:] = WALKER_LO - 0.15 * w + rng.uniform(0, 1.3, (N, 10)).astype(f) * w
    rest = rng.random(N) < 0.1
    qpos[rest, 1] = 0.0; qpos[rest, 3:] = 0.0
    qvel[:, :3] = s['qvel'][:, :3]
    qvel[:, 3:] = rng.uniform(-6, 6, (N, 10))
    qvel[rest] = 0.0
    s.update(qpos=qpos, qvel=qvel)
    return s


def assert_state_equal(a, b, fields=('qpos', 'qvel', 'pose0', 'objs', 'done0', 'steps')):
    for k in fields:
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    np.testing.assert_array_equal(a['key'], b['key'])
    assert a['hist'] == b['hist']

"""Closed-form checks of the CPU restatement against the formulas of the reference
(engine.py line numbers in each test)."""
import numpy as np
import pytest

from helpers import task_config, random_state

f32 = np.float32


def _engine(oracle, N=4, cand=6000, **kw):
    E = oracle.OracleEngine(task_config(N, **kw), n_candidates=cand)
    E.reset(check=False)
    return E


def _still_state(N, H=8, theta=0.0, robot=(0.0, 0.0)):
    """robot at rest at `robot`, heading theta, objects far away"""
    s = dict(
        qpos=np.tile(np.array([robot[0], robot[1], theta], f32), (N, 1)),
        qvel=np.zeros((N, 3), f32),
        pose0=np.tile(np.array([robot[0], robot[1], np.cos(theta), np.sin(theta)], f32), (N, 1)),
        pose1=np.tile(np.array(robot, f32), (N, 1)),
        objs=np.full((N, 1 + H, 2), 50.0, f32),
        done0=np.zeros(N, f32), done1=np.zeros(N, f32), steps=np.zeros(N, f32),
        key=np.array([0, 1], np.uint32), hist=2)
    return s


def test_obs_layout_and_dims(oracle):
    """flat obs = sorted keys: ctrl[0:3] goal_compass[3:5] goal_lidar[5:21] hazards_lidar[21:37]
    qpos[37:40] qvel[40:43]  (engine.py:386-409, 773-777)"""
    E = _engine(oracle)
    assert E.D == 43
    s = _still_state(4)
    s['qpos'][:, :] = [0.3, -0.2, 0.0]
    s['pose0'][:, :2] = [0.3, -0.2]
    s['objs'][:, 0] = [1.3, -0.2]      # goal 1 m straight ahead
    E.set_state(s)
    obs, r, d, info = E.step(np.tile(np.array([0.5, -0.25], f32), (4, 1)))
    np.testing.assert_allclose(obs[0, 0:3], [0.5, 0.0, -0.25], atol=1e-7)      # ctrl = (a0*c, a0*s, a1)
    np.testing.assert_allclose(obs[0, 3:5], [1.0, 0.0], atol=1e-6)             # compass, not normalised
    assert obs[0, 5] == pytest.approx(np.exp(-1.0), abs=2e-7)                  # goal lidar bin 0
    assert obs[0, 37] != 0.3 and abs(obs[0, 37] - 0.3) < 0.05                  # qpos is POST-integration
    assert obs[0, 40] > 0                                                      # qvel x after the push


@pytest.mark.parametrize("k", range(16))
def test_lidar_bin_centre_and_alias(oracle, k):
    """object in the middle of bin k: bin k = exp(-d), neighbours = 0.5*exp(-d) (engine.py:879-899)"""
    E = _engine(oracle)
    s = _still_state(4)
    ang = (k + 0.5) * 2 * np.pi / 16
    dist = 1.5
    s['objs'][:, 0] = [dist * np.cos(ang), dist * np.sin(ang)]
    E.set_state(s)
    obs, *_ = E.step(np.zeros((4, 2), f32))
    gl = obs[0, 5:21]
    want = np.zeros(16)
    want[k] = np.exp(-dist)
    want[(k + 1) % 16] = 0.5 * np.exp(-dist)
    want[(k - 1) % 16] = 0.5 * np.exp(-dist)
    np.testing.assert_allclose(gl, want, atol=2e-6)
    assert (obs[0, 21:37] < 1e-20).all()         # hazards 70 m away


def test_lidar_is_egocentric_and_max_over_objects(oracle):
    E = _engine(oracle)
    th = np.pi / 2
    s = _still_state(4, theta=th)
    # world +y is straight ahead for heading pi/2 -> ego angle ~0+ -> bin 0 (+ alias into 15)
    s['objs'][:, 1] = [-0.01, 1.0]
    s['objs'][:, 2] = [-0.02, 2.0]        # farther object in the same direction: occluded by max
    E.set_state(s)
    obs, *_ = E.step(np.zeros((4, 2), f32))
    hl = obs[0, 21:37]
    d = np.hypot(0.01, 1.0)
    assert hl[0] == pytest.approx(np.exp(-d), abs=1e-6)
    assert hl[2:15].max() < 1e-6
    alias = 0.01 / (2 * np.pi / 16)
    assert hl[15] == pytest.approx((1 - alias) * np.exp(-d), abs=2e-5)
    assert hl[1] == pytest.approx(max(alias * np.exp(-d), 2 * alias * np.exp(-np.hypot(0.02, 2.0))), abs=2e-5)


def test_lidar_angle_rounding_to_2pi_edge(oracle):
    """angle % 2pi can round to exactly f32(2pi): bin == 16, the scatter into obs[16] is dropped
    and the reading survives only through the (1-alias) write into bin 15."""
    E = _engine(oracle)
    s = _still_state(4, theta=np.pi / 2)   # cos(f32(pi/2)) = -4.4e-8 -> ego angle = -4.4e-8
    s['objs'][:, 1] = [0.0, 1.0]
    E.set_state(s)
    obs, *_ = E.step(np.zeros((4, 2), f32))
    hl = obs[0, 21:37]
    assert hl[0] == 0 and hl[1] == 0
    assert hl[15] == pytest.approx(np.exp(-1.0), abs=1e-6)


def test_lidar_uses_stale_pose_not_new_qpos(oracle):
    """lidar/compass/reward/cost read xpos/xmat of the forward() at the START of the step
    (SURVEY fact 8), i.e. the qpos passed in, not the integrated one."""
    E = _engine(oracle)
    s = _still_state(4)
    s['qvel'][:, 0] = 5.0                 # moving fast along +x
    s['objs'][:, 0] = [1.0, 0.0]
    E.set_state(s)
    obs, *_ = E.step(np.zeros((4, 2), f32))
    assert obs[0, 37] > 0.05                                   # qpos moved ...
    np.testing.assert_allclose(obs[0, 3:5], [1.0, 0.0], atol=1e-6)   # ... compass still from x=0


def test_action_rotated_by_prestep_heading(oracle):
    """convert_action uses data.xmat BEFORE the step (engine.py:672-685): heading of pose0,
    not the current qpos angle."""
    E = _engine(oracle)
    s = _still_state(4, theta=0.0)
    s['pose0'][:, 2:] = [0.0, 1.0]        # stale heading = +y while theta = 0
    E.set_state(s)
    obs, *_ = E.step(np.tile(np.array([1.0, 0.0], f32), (4, 1)))
    np.testing.assert_allclose(obs[0, 0:3], [0.0, 1.0, 0.0], atol=1e-7)
    assert abs(obs[0, 40]) < 1e-3 and obs[0, 41] > 0.05       # pushed along world y (force-limited: .3*.05 N)


def test_cost_dense_sum(oracle):
    """cost = sum_h (size - min(dist_h, size)), size = 0.3 (engine.py:804-811)"""
    E = _engine(oracle)
    s = _still_state(4)
    s['objs'][:, 1] = [0.1, 0.0]
    s['objs'][:, 2] = [0.0, -0.25]
    s['objs'][:, 3] = [0.3, 0.0]          # exactly on the rim: contributes 0
    s['objs'][:, 4] = [0.31, 0.0]
    E.set_state(s)
    _, _, _, info = E.step(np.zeros((4, 2), f32))
    assert info['cost'][0] == pytest.approx(0.2 + 0.05, abs=1e-6)
    s['objs'][:, 1:3] = 50.0
    E.set_state(s)
    _, _, _, info = E.step(np.zeros((4, 2), f32))
    assert info['cost'][0] < 1e-7 and info['cost'][0] >= 0


def test_reward_and_done(oracle):
    """reward = (last_dist - dist)*reward_distance; done = dist < goal_size;
    |delta|>1 -> done, reward 0 (engine.py:787-802)"""
    E = _engine(oracle)
    s = _still_state(4)
    s['objs'][:, 0] = [2.0, 0.0]
    s['pose0'][:, :2] = [-0.25, 0.0]      # last_data.xpos: robot was 0.25 further away
    E.set_state(s)
    _, r, d, _ = E.step(np.zeros((4, 2), f32))
    assert r[0] == pytest.approx(0.25, abs=1e-6) and d[0] == 0
    # inside the goal radius -> done
    s['objs'][:, 0] = [0.49, 0.0]
    s['pose0'][:, :2] = [0.0, 0.0]
    E.set_state(s)
    _, r, d, _ = E.step(np.zeros((4, 2), f32))
    assert d[0] == 1 and abs(r[0]) < 1e-6
    s['objs'][:, 0] = [0.5, 0.0]          # exactly goal_size: strict '<'
    E.set_state(s)
    assert E.step(np.zeros((4, 2), f32))[2][0] == 0
    # teleport guard
    s['objs'][:, 0] = [3.0, 0.0]
    s['pose0'][:, :2] = [-1.5, 0.0]
    E.set_state(s)
    _, r, d, _ = E.step(np.zeros((4, 2), f32))
    assert d[0] == 1 and r[0] == 0


def test_reward_zero_after_done_and_on_first_step(oracle):
    E = _engine(oracle)
    s = _still_state(4)
    s['objs'][:, 0] = [2.0, 0.0]
    s['pose0'][:, :2] = [-0.25, 0.0]
    s['done0'][:] = [1, 0, 1, 0]          # becomes _last_done in update_data
    E.set_state(s)
    _, r, _, _ = E.step(np.zeros((4, 2), f32))
    np.testing.assert_allclose(r, [0, 0.25, 0, 0.25], atol=1e-6)
    s['hist'] = 0                          # _last_done is None on the very first step (engine.py:793-796)
    s['done0'][:] = 0
    E.set_state(s)
    assert (E.step(np.zeros((4, 2), f32))[1] == 0).all()


def test_timeout_and_step_counter(oracle):
    """done |= steps > num_steps (strict); steps = done ? 0 : steps+1 (engine.py:492-493)"""
    E = _engine(oracle, num_steps=10)
    s = _still_state(4)
    s['steps'][:] = [9, 10, 11, 3]
    E.set_state(s)
    _, _, d, _ = E.step(np.zeros((4, 2), f32))
    np.testing.assert_array_equal(d, [0, 0, 1, 0])
    np.testing.assert_array_equal(E.get_state()['steps'], [10, 11, 0, 4])


def test_nan_guard(oracle):
    E = _engine(oracle)
    s = _still_state(4)
    s['objs'][:, 0] = [2.0, 0.0]
    s['pose0'][:, :2] = [-0.25, 0.0]
    E.set_state(s)
    a = np.zeros((4, 2), f32)
    a[1, 0] = np.nan
    a[2, 1] = np.inf
    obs, r, d, _ = E.step(a)
    np.testing.assert_array_equal(d, [0, 1, 1, 0])
    assert r[1] == 0 and r[2] == 0 and r[0] > 0
    assert np.isnan(obs[1]).any() and not np.isfinite(obs[2]).all()


def _bare_engine(oracle, N=4):
    E = oracle.OracleEngine(task_config(N), n_candidates=6000, point_actuators='bare')
    E.reset(check=False)
    return E


M_PT = 0.005188790204786391        # point.xml:5,19-20 (sphere + box, density 1)
I_C = 2.842182748581224e-05 - 1e-4 ** 2 / M_PT


def test_point_actuators_inherit_the_class_defaults(oracle):
    """point.xml:7-8,37-39 [derived, DESIGN section 0]: qfrc = gear*clip(clip(ctrl,+-1) - gear*qvel, +-.05),
    data.qacc = M^-1 f has no implicit damping."""
    E = _engine(oracle)
    # (a) force saturation from rest: any |ctrl| > .05 gives the same first-step acceleration .3*.05/m
    for a0 in (0.06, 0.5, 1.0, 7.0):
        E.set_state(_still_state(4))
        _, _, _, info = E.step(np.tile(np.array([a0, 0.0], f32), (4, 1)))
        assert info['qacc'][0, 0] == pytest.approx(0.3 * 0.05 / M_PT, rel=1e-5)
    # (b) below saturation the force is gear*ctrl: ctrl = .04 -> qacc = .3*.04/m ; hinge: .3*.04/I_c
    E.set_state(_still_state(4))
    _, _, _, info = E.step(np.tile(np.array([0.04, 0.04], f32), (4, 1)))
    assert info['qacc'][0, 0] == pytest.approx(0.3 * 0.04 / M_PT, rel=1e-5)
    assert info['qacc'][0, 2] == pytest.approx(0.3 * 0.04 / I_C, rel=1e-5)
    # (c) velocity-servo bias: moving at v with ctrl = gear*v the actuator force vanishes and only the
    #     joint damping acts: qacc = -d*v/m
    s = _still_state(4)
    s['qvel'][:, 0] = 0.1
    E.set_state(s)
    _, _, _, info = E.step(np.tile(np.array([0.03, 0.0], f32), (4, 1)))
    assert info['qacc'][0, 0] == pytest.approx(-0.01 * 0.1 / M_PT, rel=1e-4)
    #     and with ctrl = 0 the servo brakes: force = clip(-.3*v, +-.05) -> qacc = (-.3*.03 - .01*.1)/m
    E.set_state(s)
    _, _, _, info = E.step(np.zeros((4, 2), f32))
    assert info['qacc'][0, 0] == pytest.approx((-0.3 * 0.03 - 0.01 * 0.1) / M_PT, rel=1e-4)
    # (d) ctrl is clamped for the force only; the observation carries the raw converted ctrl
    E.set_state(_still_state(4))
    obs, *_ = E.step(np.tile(np.array([7.0, -9.0], f32), (4, 1)))
    np.testing.assert_array_equal(obs[0, 0:3], [7.0, 0.0, -9.0])


def test_point_terminal_speed(oracle):
    """terminal speed = gear*forcerange/damping = .3*.05/.01 = 1.5 m/s (3 rad/s on the hinge:
    .3*.05/.005); per-step decay of a coasting robot m/(m+h d) needs ctrl = gear*v (servo off)."""
    E = _engine(oracle)
    E.set_state(_still_state(4))
    for _ in range(400):
        obs, *_ = E.step(np.tile(np.array([1.0, 0.0], f32), (4, 1)))
    assert obs[0, 40] == pytest.approx(1.5, rel=2e-3)
    E.set_state(_still_state(4))
    for _ in range(400):
        obs, *_ = E.step(np.tile(np.array([0.0, 1.0], f32), (4, 1)))
    assert obs[0, 42] == pytest.approx(3.0, rel=2e-3)
    s = _still_state(4)
    s['qvel'][:, 0] = 0.1
    E.set_state(s)
    obs, *_ = E.step(np.tile(np.array([0.03, 0.0], f32), (4, 1)))
    assert obs[0, 40] == pytest.approx(0.1 * M_PT / (M_PT + 0.02 * 0.01), rel=1e-5)


def test_point_stays_in_the_arena_under_random_actions(oracle):
    """plausibility: under U(-1,1) actions the robot must not leave the 4 m arena
    as a matter of course; with the inherited limits the speed never exceeds 1.5*sqrt(2)."""
    N = 500
    E = oracle.OracleEngine(task_config(N, num_steps=200), n_candidates=40000)
    E.reset()
    rng = np.random.default_rng(0)
    outside = total = 0
    vmax = 0.0
    for _ in range(200):
        obs, _, d, _ = E.step(rng.uniform(-1, 1, (N, 2)).astype(f32))
        outside += int((np.abs(obs[:, 37:39]) > 2.0).any(axis=1).sum())
        total += N
        vmax = max(vmax, float(np.linalg.norm(obs[:, 40:42], axis=1).max()))
        if d.any():
            E.reset_done()
    assert outside / total < 0.03
    assert vmax <= 1.5 * np.sqrt(2) + 1e-3


def test_point_bare_model_is_still_selectable(oracle):
    """the round-1 reading (no class defaults): terminal velocity gear*ctrl/damping = 30 m/s, first-step
    angular acceleration .3/I_c."""
    E = _bare_engine(oracle)
    E.set_state(_still_state(4))
    for _ in range(400):
        obs, *_ = E.step(np.tile(np.array([1.0, 0.0], f32), (4, 1)))
    assert obs[0, 40] == pytest.approx(30.0, rel=2e-3)
    s = _still_state(4)
    s['qvel'][:, 0] = 1.0
    E.set_state(s)
    obs, *_ = E.step(np.zeros((4, 2), f32))
    assert obs[0, 40] == pytest.approx(M_PT / (M_PT + 0.02 * 0.01), rel=1e-5)
    E.set_state(_still_state(4))
    _, _, _, info = E.step(np.tile(np.array([0.0, 1.0], f32), (4, 1)))
    assert info['qacc'][0, 2] == pytest.approx(0.3 / I_C, rel=1e-5)


def test_reset_layout_constraints_and_key_use(oracle):
    E = oracle.OracleEngine(task_config(64, seed=3), n_candidates=30000)
    o1 = E.reset()
    st = E.get_state()
    objs, robot = st['objs'], st['qpos'][:, :2]
    assert (np.abs(objs[:, 0]) <= 1.5 + 1e-6).all() and (np.abs(objs[:, 1:]) <= 1.6 + 1e-6).all()
    assert (np.abs(robot) <= 1.6 + 1e-6).all()
    assert (np.linalg.norm(robot - objs[:, 0], axis=1) >= 3.0).all()          # engine.py:570-571
    d_gh = np.linalg.norm(objs[:, 1:] - objs[:, :1], axis=2)
    assert (d_gh >= 0.9 - 1e-6).all()                                          # 0.5 + 0.4
    for i in range(1, 9):
        for j in range(i + 1, 9):
            assert (np.linalg.norm(objs[:, i] - objs[:, j], axis=1) >= 0.8 - 1e-6).all()
    assert (np.linalg.norm(objs[:, 1:] - robot[:, None], axis=2) >= 0.8 - 1e-6).all()
    assert (st['qvel'] == 0).all() and (st['qpos'][:, 2] == 0).all() and (st['steps'] == 0).all()
    np.testing.assert_array_equal(st['pose0'], np.c_[robot, np.ones(64), np.zeros(64)].astype(f32))
    assert (o1[:, 0:3] == 0).all() and (o1[:, 40:43] == 0).all()
    np.testing.assert_array_equal(o1[:, 37:39], robot)
    # reset() does not advance the key: same layouts again; a step() does
    np.testing.assert_array_equal(E.reset(), o1)
    E.step(np.zeros((64, 2), f32))
    assert not np.array_equal(E.reset(), o1)


def test_reset_done_semantics(oracle):
    """only done rows change; stale pose kept; next reward of a reset env is 0"""
    N = 32
    E = oracle.OracleEngine(task_config(N, seed=1), n_candidates=30000)
    E.reset()
    rng = np.random.default_rng(0)
    s = random_state(N, 8, rng, done_frac=0.0)
    s['done0'][::4] = 1
    s['steps'][::4] = 0
    E.set_state(s)
    # _obs of the engine is whatever the last step wrote; make one by stepping from a copy
    before = E.get_state()
    obs = E.reset_done()
    after = E.get_state()
    dn = before['done0'] > 0
    assert dn.sum() == 8
    np.testing.assert_array_equal(after['qpos'][~dn], before['qpos'][~dn])
    np.testing.assert_array_equal(after['objs'][~dn], before['objs'][~dn])
    np.testing.assert_array_equal(after['pose0'], before['pose0'])            # stale xpos/xmat (:731)
    assert (after['qvel'][dn] == 0).all() and (after['qpos'][dn, 2] == 0).all()
    assert (np.linalg.norm(after['qpos'][dn, :2] - after['objs'][dn, 0], axis=1) >= 3.0).all()
    assert (obs[dn, 0:3] == 0).all() and (obs[dn, 40:43] == 0).all()
    np.testing.assert_array_equal(obs[dn, 37:39], after['qpos'][dn, :2])
    # idempotent: the key did not move, _done did not change
    np.testing.assert_array_equal(E.reset_done(), obs)
    _, r, _, _ = E.step(rng.uniform(-1, 1, (N, 2)).astype(f32))
    assert (r[dn] == 0).all()


def test_sharded_oracle_equals_unsharded(oracle):
    N, W = 24, 3
    full = oracle.OracleEngine(task_config(N * W, seed=2), n_candidates=40000)
    of = full.reset()
    for r in range(W):
        sh = oracle.OracleEngine(task_config(N, seed=2), n_candidates=40000, env_total=N * W, env_offset=r * N)
        np.testing.assert_array_equal(sh.reset(), of[r * N:(r + 1) * N])


def test_layout_assert(oracle):
    E = oracle.OracleEngine(task_config(500, seed=0), n_candidates=2000)
    with pytest.raises(AssertionError):
        E.reset()


def test_pillars_extension_closed_forms(oracle):
    """synthetic config-5 objects (no reference counterpart): own lidar row sorted between hazards_lidar and
    qpos, dense cost with pillars_size after the hazard terms, own keepout in the sampler"""
    N = 4
    cfg = task_config(N, pillars_num=3, observe_pillars=True, placements_extents=[-3, -3, 3, 3])
    E = oracle.OracleEngine(cfg, n_candidates=8000)
    E.reset(check=False)
    assert E.D == 43 + 16
    s = _still_state(N, H=8 + 3)
    s['objs'][:, 9] = [1.0, 0.0]            # pillar 0 one metre ahead  -> bin 0 of pillars_lidar
    s['objs'][:, 10] = [0.0, 0.1]           # pillar 1 at 0.1 m: inside pillars_size = 0.2
    s['objs'][:, 1] = [0.25, 0.0]           # hazard 0 at 0.25 m: inside hazards_size = 0.3
    E.set_state(s)
    obs, r, d, info = E.step(np.zeros((N, 2), f32))
    pl = obs[0, 37:53]
    assert pl[0] == pytest.approx(np.exp(-1.0), abs=2e-7) and pl[4] == pytest.approx(np.exp(-0.1), abs=2e-7)
    assert (obs[0, 53:56] == 0).all() and obs.shape[1] == 59          # qpos follows the pillars lidar
    assert info['cost'][0] == pytest.approx((0.3 - 0.25) + (0.2 - 0.1), abs=1e-6)
    # layout: pillar keepouts .3 (pillar-pillar .6, pillar-hazard .7, pillar-goal .8, pillar-robot .7)
    E = oracle.OracleEngine(task_config(32, seed=3, pillars_num=8, observe_pillars=True,
                                        placements_extents=[-3, -3, 3, 3]), n_candidates=40000)
    E.reset()
    st = E.get_state()
    objs, robot = st['objs'], st['qpos'][:, :2]
    P = objs[:, 9:]
    for i in range(8):
        for j in range(i + 1, 8):
            assert (np.linalg.norm(P[:, i] - P[:, j], axis=1) >= 0.6 - 1e-6).all()
    assert (np.linalg.norm(P[:, :, None] - objs[:, None, 1:9], axis=3) >= 0.7 - 1e-6).all()
    assert (np.linalg.norm(P - objs[:, :1], axis=2) >= 0.8 - 1e-6).all()
    assert (np.linalg.norm(P - robot[:, None], axis=2) >= 0.7 - 1e-6).all()


@pytest.mark.parametrize("robot", ["point", "swimmer", "ant", "walker"])
def test_robot_rot_turns_the_world_pose_not_the_joint_dynamics(oracle, robot):
    """robot_rot (engine.py:114,342-345) turns the robot's root body about z (world.py:117): the joint-space
    dynamics of the robots whose ctrl is the action do not notice (same qpos / qvel trajectory as robot_rot = 0 until a
    done differs), the world pose is the unrotated one turned by the angle, layout2qpos still writes the layout's xy
    into the slide joints (engine.py:635-638), and the compass is the goal vector in the turned body frame."""
    from helpers import SWIMMER, ANT, WALKER
    extra = {"point": {}, "swimmer": SWIMMER, "ant": ANT, "walker": WALKER}[robot]
    A = {"ant": 8, "walker": 10}.get(robot, 2)
    N, th = 16, 0.9
    O0 = oracle.OracleEngine(task_config(N, seed=3, num_steps=100, **extra), n_candidates=20000)
    O1 = oracle.OracleEngine(task_config(N, seed=3, num_steps=100, robot_rot=th, **extra), n_candidates=20000)
    o0, o1 = O0.reset(), O1.reset()
    s0, s1 = O0.get_state(), O1.get_state()
    np.testing.assert_array_equal(s0['qpos'], s1['qpos'])            # same layouts, same joint coordinates
    c, s = np.cos(th), np.sin(th)
    R = np.array([[c, -s], [s, c]])

    def check_pose(p0, p1):
        np.testing.assert_allclose(p1[:, :2], p0[:, :2] @ R.T, atol=2e-6)
        np.testing.assert_allclose(p1[:, 2], c * p0[:, 2] - s * p0[:, 3], atol=1e-6)
        np.testing.assert_allclose(p1[:, 3], s * p0[:, 2] + c * p0[:, 3], atol=1e-6)
    check_pose(s0['pose0'], s1['pose0'])
    # compass of the rotated engine = goal vector in the turned body frame (obs_compass engine.py:834-844)
    D = O1.D
    off = {"point": 3, "swimmer": 2, "ant": 8, "walker": 10}[robot]
    goal = s1['objs'][:, 0, :]
    d = goal - s1['pose0'][:, :2]
    comp = np.stack([d[:, 0] * s1['pose0'][:, 2] + d[:, 1] * s1['pose0'][:, 3],
                     -d[:, 0] * s1['pose0'][:, 3] + d[:, 1] * s1['pose0'][:, 2]], axis=1)
    np.testing.assert_allclose(o1[:, off:off + 2], comp, atol=1e-5)
    assert not np.allclose(o0, o1)
    rng = np.random.default_rng(0)
    for t in range(6):
        a = rng.uniform(-1, 1, (N, A)).astype(np.float32)
        _, _, d0, _ = O0.step(a)
        _, _, d1, _ = O1.step(a)
        if d0.any() or d1.any():
            break
        s0, s1 = O0.get_state(), O1.get_state()
        if robot != "point":                                          # Point: ctrl = xmat . action (engine.py:672-685)
            np.testing.assert_array_equal(s0['qpos'], s1['qpos'])
            np.testing.assert_array_equal(s0['qvel'], s1['qvel'])
            check_pose(s0['pose0'], s1['pose0'])
    assert t >= 2


def test_point_solve_single_step_error_against_float64(oracle):
    """Independent bound on the arithmetic of the Point's 3x3 solve (round 3 replaced the division by the fp32
    Schur complement Io - (b*b + d*d)/m with a multiplication by the reciprocal of its exact value -- b^2 + d^2 is the
    model constant (m xc)^2 -- in checker and kernels together).  Reference here: the SAME forces (fp32, the checker's
    own sincos) solved in float64 with a dense 3x3 solve.  On 40 000 random states, servo-saturated and not:
      * the checker's step (get_state after one step) IS the fp32 reciprocal form restated below, bit for bit;
      * its single-step error against float64 is a few ulp of the result and not larger than the error of the
        round-2 division form on the same states (both restated here in numpy fp32).
    The 60-step trajectories of the two forms nevertheless separate (2e-4 in obs): the hinge's velocity servo is
    unstable at h = 0.02 and amplifies ANY last-bit difference, which is why the 1e-5 bar is
    stated per step from a common state (tests/test_golden.py replays the reference step by step)."""
    f = np.float32
    N = 40_000
    rng = np.random.default_rng(17)
    cfg = task_config(N, seed=1)
    E = oracle.OracleEngine(cfg, n_candidates=3000)
    E.reset(check=False)
    s = random_state(N, 8, rng, done_frac=0.0)
    s['qvel'][: N // 2] *= f(0.05)                    # half of the states in the unsaturated servo regime
    s['hist'] = 2
    E.set_state(s)
    act = rng.uniform(-1.5, 1.5, (N, 2)).astype(f)
    q, v = s['qpos'].copy(), s['qvel'].copy()
    _, _, _, info = E.step(act)
    st = E.get_state()

    # the forces exactly as the checker forms them (fp32, its sincos)
    sh, ch, _, _ = oracle.math_probe(f(0.5) * q[:, 2], np.zeros(N, f))
    c, sn = ch * ch - sh * sh, f(2.0) * (ch * sh)
    a0 = s['pose0'][:, 2] * act[:, 0], s['pose0'][:, 3] * act[:, 0]
    ctrl = np.stack([a0[0], a0[1], act[:, 1]], 1).astype(f)
    MXC, DXY, DT, G, H = f(0.0001), f(0.01), f(0.005), f(0.3), f(0.02)
    m64, io64 = 0.005188790204786391, 2.842182748581224e-05
    b, d = -(MXC * sn), MXC * c
    w2 = v[:, 2] * v[:, 2]

    def actu(u, vel):
        u = np.clip(u, f(-1), f(1))
        return G * np.clip(u - f(1.0) * (G * vel), f(-0.05), f(0.05))
    fx = (-(DXY * v[:, 0]) - (-(d * w2))) + actu(ctrl[:, 0], v[:, 0])
    fy = (-(DXY * v[:, 1]) - (b * w2)) + actu(ctrl[:, 1], v[:, 1])
    ft = (-(DT * v[:, 2]) - f(0.0)) + actu(ctrl[:, 2], v[:, 2])
    t = b * fx + d * fy

    def solve32(ia, num_d3, recip):
        y3 = ft - t * ia
        q3 = y3 * recip if recip is not None else y3 / (num_d3 - (b * b + d * d) * ia)
        return np.stack([(fx - b * q3) * ia, (fy - d * q3) * ia, q3], 1)
    iaA = f(1.0 / (m64 + 0.02 * 0.01))
    idA = f(1.0 / ((io64 + 0.02 * 0.005) - (1.0e-4 * 1.0e-4) / (m64 + 0.02 * 0.01)))
    qa_rec = solve32(iaA, None, idA)
    qa_div = solve32(iaA, f(io64 + 0.02 * 0.005), None)
    # float64 dense solve of (M + h D) qa = f with the same fp32 forces and fp32 (b, d)
    Mi = np.zeros((N, 3, 3))
    Mi[:, 0, 0] = Mi[:, 1, 1] = m64 + 0.02 * 0.01
    Mi[:, 2, 2] = io64 + 0.02 * 0.005
    Mi[:, 0, 2] = Mi[:, 2, 0] = b.astype(np.float64)
    Mi[:, 1, 2] = Mi[:, 2, 1] = d.astype(np.float64)
    rhs = np.stack([fx, fy, ft], 1).astype(np.float64)
    qa64 = np.linalg.solve(Mi, rhs[..., None])[..., 0]

    v_rec = (v + H * qa_rec).astype(f)
    np.testing.assert_array_equal(st['qvel'], v_rec)                       # the checker IS the reciprocal form
    np.testing.assert_array_equal(st['qpos'], (q + H * v_rec).astype(f))
    scale = np.abs(qa64) + 1e-3 * np.abs(qa64).max(axis=1, keepdims=True)
    e_rec = np.abs(qa_rec - qa64) / scale
    e_div = np.abs(qa_div - qa64) / scale
    assert e_rec.max() < 4e-6 and e_div.max() < 4e-6, (e_rec.max(), e_div.max())
    assert np.sqrt((e_rec ** 2).mean()) <= 1.25 * np.sqrt((e_div ** 2).mean()) + 1e-9
    # one step from a common state: far inside the 1e-5 bar, in either form
    v64 = v.astype(np.float64) + 0.02 * qa64
    assert np.abs(v_rec - v64).max() < 2e-6 * max(1.0, np.abs(v64).max())
    assert np.abs((v + H * qa_div).astype(f) - v64).max() < 2e-6 * max(1.0, np.abs(v64).max())

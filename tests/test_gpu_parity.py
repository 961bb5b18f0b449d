"""HIP path vs the CPU restatement (oracle/) on identical inputs -- through the C ABI.

Bar: bit-exact done / cost>0 masks, fp32 obs / reward within 1e-5.
Because kernel and checker evaluate the same fp32 operation sequence, these
tests actually demand exact equality of every output and of the state.
"""
import numpy as np
import pytest

from helpers import task_config, random_state, assert_state_equal, SWIMMER, ANT, WALKER

pytestmark = pytest.mark.gpu

TOL = 1e-5  # the north_star tolerance for obs / reward


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    return torch


# split: two-kernel rollouts (dynamics tape + observation pass), lane-group steps; split-alone: the same with the layout
# prefetch off, which selects the dynamics pass's alone-on-the-chip form where a robot has one (Swimmer: a quad per env;
# Ant / Walker: one or two envs per wave instead of four)
PATHS = {"thread": 1, "group": 2, "split": 3, "split-alone": 3}


def _engines(cfg, oracle, n_candidates=20000, path=None, **kw):
    from guardx_amd import Engine
    E = Engine(cfg, n_candidates=n_candidates, **kw)
    assert E.action_space.shape[0] == E._lib.gx_act_dim(E._h)
    if path is not None:
        E.set_path(PATHS[path])
        if path == "split-alone":
            E.set_prefetch(-1)
    O = oracle.OracleEngine(cfg, n_candidates=n_candidates, env_total=E._cfg.env_total,
                            env_offset=E._cfg.env_offset, point_actuators=kw.get('point_actuators', 'mjcf'))
    return E, O


def _cmp_step(out_g, out_o):
    obs_g, r_g, d_g, info_g = out_g
    obs_o, r_o, d_o, info_o = out_o
    obs_g, r_g, d_g = obs_g.cpu().numpy(), r_g.cpu().numpy(), d_g.cpu().numpy()
    c_g = info_g['cost'].cpu().numpy()
    # masks: bit exact
    np.testing.assert_array_equal(d_g, d_o)
    np.testing.assert_array_equal(c_g > 0, info_o['cost'] > 0)
    # values: stated tolerance ...
    np.testing.assert_allclose(obs_g, obs_o, rtol=0, atol=TOL, equal_nan=True)
    np.testing.assert_allclose(r_g, r_o, rtol=0, atol=TOL)
    np.testing.assert_allclose(c_g, info_o['cost'], rtol=0, atol=TOL, equal_nan=True)
    # ... and in fact exact, by construction
    np.testing.assert_array_equal(obs_g, obs_o)
    np.testing.assert_array_equal(r_g, r_o)
    np.testing.assert_array_equal(c_g, info_o['cost'])
    if 'qacc' in info_g['obs']:
        np.testing.assert_array_equal(info_g['obs']['qacc'].cpu().numpy(), info_o['qacc'])


def test_device_math_bitexact(torch_cuda, oracle):
    torch = torch_cuda
    import ctypes as C
    from guardx_amd import _native
    lib = _native.load()
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.uniform(-300, 300, 1 << 20), rng.uniform(-8, 8, 1 << 20),
                        rng.uniform(-100, 1, 1 << 18), rng.normal(0, 1e-3, 1 << 18),
                        [0.0, -0.0, np.inf, -np.inf, np.nan, 1e9, -1e9, 1e-45, -87.5, -86.9, 88.5]]
                       ).astype(np.float32)
    y = rng.uniform(-5, 5, x.size).astype(np.float32)
    y[::97] = 0.0
    y[1::197] = -0.0
    x[5::211] = 0.0
    so, co, ao, eo = oracle.math_probe(x, y)
    xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    outs = [torch.empty_like(xd) for _ in range(4)]
    _native.check(lib.gx_math_probe(x.size, xd.data_ptr(), yd.data_ptr(), *[o.data_ptr() for o in outs],
                                    C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    for name, g, o in zip(("sin", "cos", "atan2", "exp"), outs, (so, co, ao, eo)):
        np.testing.assert_array_equal(g.cpu().numpy(), o, err_msg=name)


def test_device_split_matches_oracle(torch_cuda, oracle):
    torch = torch_cuda
    import ctypes as C
    from guardx_amd import _native
    lib = _native.load()
    for n, key in ((2, (0, 0)), (7, (1, 2)), (1000, (123, 456)), (100001, (0xdeadbeef, 42))):
        out = torch.empty(2 * n, dtype=torch.int32, device='cuda')
        k = (C.c_uint32 * 2)(*key)
        _native.check(lib.gx_split_probe(k, n, out.data_ptr(),
                                         C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        got = out.cpu().numpy().view(np.uint32).reshape(n, 2)
        np.testing.assert_array_equal(got, oracle.split(key, n))


@pytest.mark.parametrize("path", ["thread", "group"])
@pytest.mark.parametrize("N", [1, 4, 63, 64, 65, 2000, 5000])
def test_step_parity_random_states(torch_cuda, oracle, N, path):
    torch = torch_cuda
    E, O = _engines(task_config(N, seed=3), oracle, path=path)
    assert float(E.action_space.low[0]) == -1.0 and float(E.action_space.high[1]) == 1.0   # ctrllimited, engine.py:291-297
    rng = np.random.default_rng(N)
    for trial in range(3):
        s = random_state(N, 8, rng)
        s['hist'] = [2, 1, 0][trial]
        # actions beyond the ctrl range (clamped for the force only) and, for a third of the envs, small
        # speeds and actions so that the actuator force is NOT saturated at +-.05 (velocity-servo regime)
        act = rng.uniform(-1.6, 1.6, (N, 2)).astype(np.float32)
        s['qvel'][::3] *= np.float32(0.03)
        act[::3] *= np.float32(0.03)
        E.set_state(s)
        O.set_state(s)
        out_g = E.step(torch.from_numpy(act).cuda())
        out_o = O.step(act)
        _cmp_step(out_g, out_o)
        assert_state_equal(E.get_state(), O.get_state())


@pytest.mark.parametrize("path", ["thread", "group", "split"])
def test_point_bare_actuator_model_parity(torch_cuda, oracle, path):
    """the round-1 reading of point.xml's actuators (no class defaults) stays selectable and bit exact"""
    torch = torch_cuda
    N, T = 300, 40
    E, O = _engines(task_config(N, seed=4, num_steps=30, goal_size=0.9), oracle, path=path, point_actuators='bare')
    assert np.isinf(E.action_space.low).all() and np.isinf(E.action_space.high).all()
    np.testing.assert_array_equal(E.reset().cpu().numpy(), O.reset())
    rng = np.random.default_rng(2)
    for t in range(10):
        act = rng.uniform(-1.6, 1.6, (N, 2)).astype(np.float32)
        _cmp_step(E.step(torch.from_numpy(act).cuda()), O.step(act))
        np.testing.assert_array_equal(E.reset_done().cpu().numpy(), O.reset_done())
    acts = rng.uniform(-1, 1, (T, N, 2)).astype(np.float32)
    obs, rew, cost, done = E.rollout(torch.from_numpy(acts).cuda())
    for t in range(T):
        o, r, d, info = O.step(acts[t])
        np.testing.assert_array_equal(obs[t].cpu().numpy(), O.reset_done())
        np.testing.assert_array_equal(rew[t].cpu().numpy(), r)
        np.testing.assert_array_equal(done[t].cpu().numpy(), d)
    assert_state_equal(E.get_state(), O.get_state())
    # and it is a different model: the default engine disagrees after one unclamped push
    E2, _ = _engines(task_config(N, seed=4, num_steps=30, goal_size=0.9), oracle, path=path)
    E2.reset()
    E.reset()
    a = torch.ones(N, 2, device='cuda')
    assert not torch.equal(E.step(a)[0], E2.step(a)[0])


@pytest.mark.parametrize("path", ["thread", "group"])
def test_step_parity_nan_inf_actions(torch_cuda, oracle, path):
    torch = torch_cuda
    N = 256
    E, O = _engines(task_config(N, seed=5), oracle, path=path)
    rng = np.random.default_rng(0)
    s = random_state(N, 8, rng, done_frac=0.0)
    E.set_state(s); O.set_state(s)
    act = rng.uniform(-1, 1, (N, 2)).astype(np.float32)
    act[::7, 0] = np.nan
    act[3::11, 1] = np.inf
    act[5::13, 0] = -np.inf
    out_g = E.step(torch.from_numpy(act).cuda())
    out_o = O.step(act)
    d = out_g[2].cpu().numpy()
    np.testing.assert_array_equal(d, out_o[2])
    np.testing.assert_array_equal(out_g[1].cpu().numpy(), out_o[1])
    np.testing.assert_array_equal(out_g[0].cpu().numpy(), out_o[0])   # NaNs compare equal here
    assert d[::7].all() and (out_g[1].cpu().numpy()[::7] == 0).all()  # guard engine.py:696-699


@pytest.mark.parametrize("path", ["thread", "group", "split"])
def test_rollout_nan_guard_parity(torch_cuda, oracle, path):
    """NaN / Inf actions inside a fused rollout: the guard (engine.py:696-699) ends the episode, reset_done
    re-initialises the env, the NaN row is replaced by the reset observation -- same on every path"""
    torch = torch_cuda
    N, T = 150, 24
    E, O = _engines(task_config(N, seed=6, num_steps=50), oracle, n_candidates=30000, path=path)
    np.testing.assert_array_equal(E.reset().cpu().numpy(), O.reset())
    rng = np.random.default_rng(9)
    acts = rng.uniform(-1, 1, (T, N, 2)).astype(np.float32)
    acts[3, ::7, 0] = np.nan
    acts[5, 1::11, 1] = np.inf
    acts[9, 2::13, 0] = -np.inf
    acts[9, 4::17, 0] = 3e38            # finite but huge: the exact evaluation decides
    obs, rew, cost, done = E.rollout(torch.from_numpy(acts).cuda())
    for t in range(T):
        o, r, d, info = O.step(acts[t])
        np.testing.assert_array_equal(obs[t].cpu().numpy(), O.reset_done())
        np.testing.assert_array_equal(rew[t].cpu().numpy(), r)
        np.testing.assert_array_equal(done[t].cpu().numpy(), d)
        np.testing.assert_array_equal(cost[t].cpu().numpy(), info['cost'])
    assert done[3, ::7].all().item() and done.sum().item() >= 20
    assert_state_equal(E.get_state(), O.get_state())


@pytest.mark.parametrize("N,cand", [(4, 20000), (2000, 200000)])
def test_reset_parity(torch_cuda, oracle, N, cand):
    E, O = _engines(task_config(N, seed=0), oracle, n_candidates=cand)
    obs_g = E.reset().cpu().numpy()
    obs_o = O.reset()
    assert E.layout_size == O.layout_size
    np.testing.assert_array_equal(E.get_pool(512), O.get_pool(512))
    np.testing.assert_array_equal(obs_g, obs_o)
    assert_state_equal(E.get_state(), O.get_state())


@pytest.mark.parametrize("path", ["thread", "group", "split"])
def test_rollout_parity_with_reset_done(torch_cuda, oracle, path):
    """200-step random-policy episode with reset_done() whenever any env is done."""
    torch = torch_cuda
    N, T = 500, 200
    # the force-limited Point (1.5 m/s) rarely covers the >= 3 m to a 0.5 m goal under random actions: a wide
    # goal makes envs finish at scattered times, the 120-step timeout (engine.py:492) ends the rest together
    E, O = _engines(task_config(N, seed=11, num_steps=120, goal_size=2.7), oracle, n_candidates=60000, path=path)
    og, oo = E.reset(), O.reset()
    np.testing.assert_array_equal(og.cpu().numpy(), oo)
    rng = np.random.RandomState(0)
    n_done = 0
    for t in range(T):
        act = rng.uniform(-1, 1, (N, 2)).astype(np.float32)   # myTest.py:28-31 style
        out_g = E.step(torch.from_numpy(act).cuda())
        out_o = O.step(act)
        _cmp_step(out_g, out_o)
        if out_o[2].any():
            n_done += int(out_o[2].sum())
            rg, ro = E.reset_done().cpu().numpy(), O.reset_done()
            np.testing.assert_array_equal(rg, ro)
    assert n_done > 0, "the episode never exercised reset_done"
    assert_state_equal(E.get_state(), O.get_state())
    # second epoch: reset() with the advanced key, stale _done carried over (SURVEY 3.2 note)
    np.testing.assert_array_equal(E.reset().cpu().numpy(), O.reset())
    act = rng.uniform(-1, 1, (N, 2)).astype(np.float32)
    _cmp_step(E.step(torch.from_numpy(act).cuda()), O.step(act))


@pytest.mark.parametrize("path", ["thread", "group", "split"])
def test_fused_rollout_equals_stepwise(torch_cuda, oracle, path):
    torch = torch_cuda
    N, T = 301, 64
    cfg = task_config(N, seed=2, num_steps=40, goal_size=0.9)
    E, O = _engines(cfg, oracle, n_candidates=40000, path=path)
    E.reset(); O.reset()
    rng = np.random.default_rng(5)
    acts = rng.uniform(-1, 1, (T, N, 2)).astype(np.float32)
    obs, rew, cost, done = E.rollout(torch.from_numpy(acts).cuda())
    assert done.sum().item() > 0
    for t in range(T):
        o, r, d, info = O.step(acts[t])
        o = O.reset_done()
        np.testing.assert_array_equal(obs[t].cpu().numpy(), o)
        np.testing.assert_array_equal(rew[t].cpu().numpy(), r)
        np.testing.assert_array_equal(done[t].cpu().numpy(), d)
        np.testing.assert_array_equal(cost[t].cpu().numpy(), info['cost'])
    assert_state_equal(E.get_state(), O.get_state())


@pytest.mark.parametrize("robot", ["point", "swimmer", "ant", "walker"])
@pytest.mark.parametrize("path", ["thread", "group", "split"])
def test_variant_configs(torch_cuda, oracle, path, robot):
    torch = torch_cuda
    extra = {"point": {}, "swimmer": SWIMMER, "ant": ANT, "walker": WALKER}[robot]
    A = {"ant": 8, "walker": 10}.get(robot, 2)
    variants = [
        dict(hazards_num=3, lidar_num_bins=8),
        dict(hazards_num=12, lidar_num_bins=24, lidar_alias=False, hazards_keepout=0.25),
        dict(observe_vel=True, observe_acc=True),
        dict(observe_qpos=False, observe_ctrl=False, observe_goal_lidar=False),
        dict(lidar_max_dist=3.0, physics_steps_per_control_step=2, lidar_exp_gain=0.5),
        dict(hazards_num=20, goal_size=0.3, hazards_size=0.2, reward_distance=2.0,
             hazards_keepout=0.18, placements_extents=[-3, -3, 3, 3]),
        dict(robot_rot=0.7),                                      # engine.py:114,342-345 -> world.py:117
        dict(robot_rot=-2.4, observe_vel=True, hazards_num=5, goal_size=1.5),
    ]
    for v in variants:
        N = 130
        cfg = task_config(N, seed=9, num_steps=50, **v, **extra)
        E, O = _engines(cfg, oracle, n_candidates=30000, path=path)
        assert E.obs_flat_size == O.D
        np.testing.assert_array_equal(E.reset().cpu().numpy(), O.reset(check=False))
        rng = np.random.default_rng(3)
        for t in range(60):   # crosses the num_steps timeout (engine.py:492)
            act = rng.uniform(-1, 1, (N, A)).astype(np.float32)
            out_g, out_o = E.step(torch.from_numpy(act).cuda()), O.step(act)
            _cmp_step(out_g, out_o)
            if t % 7 == 6:
                np.testing.assert_array_equal(E.reset_done().cpu().numpy(), O.reset_done())
        # and a fused stretch on top (group path: persistent kernel with in-kernel reset_done)
        TF = 40 if 'robot_rot' in v else 9
        acts = rng.uniform(-1, 1, (TF, N, A)).astype(np.float32)
        obs, rew, cost, done = E.rollout(torch.from_numpy(acts).cuda())
        for t in range(TF):
            o, r, d, info = O.step(acts[t])
            np.testing.assert_array_equal(obs[t].cpu().numpy(), O.reset_done())
            np.testing.assert_array_equal(rew[t].cpu().numpy(), r)
            np.testing.assert_array_equal(done[t].cpu().numpy(), d)
            np.testing.assert_array_equal(cost[t].cpu().numpy(), info['cost'])
        assert_state_equal(E.get_state(), O.get_state(),
                           fields=('qpos', 'qvel', 'pose0', 'pose1', 'objs', 'done0', 'done1', 'steps')
                           if v.get('observe_vel') else ('qpos', 'qvel', 'pose0', 'objs', 'done0', 'steps'))


@pytest.mark.parametrize("robot", ["point", "ant"])
def test_step_speculates_reset_done_without_installing_it(torch_cuda, oracle, robot):
    """Engine.step() evaluates reset_done() in the same launch (gx_step_rd) but must not re-initialise
    anything unless reset_done() is called (the *_one_episode learners never call it): every mix of
    step / step+reset_done / get_state / rollout must follow the checker, on fresh and ring buffers."""
    torch = torch_cuda
    extra = ANT if robot == "ant" else {}
    A = 8 if robot == "ant" else 2
    N = 257
    for ring in (8, 0):
        cfg = task_config(N, seed=17, num_steps=12, goal_size=2.6, **extra)
        E, O = _engines(cfg, oracle, n_candidates=30000, out_ring=ring)
        np.testing.assert_array_equal(E.reset().cpu().numpy(), O.reset())
        rng = np.random.default_rng(4)
        called = skipped = 0
        for t in range(70):
            act = rng.uniform(-1, 1, (N, A)).astype(np.float32)
            out_g, out_o = E.step(torch.from_numpy(act).cuda()), O.step(act)
            _cmp_step(out_g, out_o)
            mode = t % 5
            if mode in (0, 1):                       # reset_done right after the step (trpo.py:547)
                rg = E.reset_done()
                np.testing.assert_array_equal(rg.cpu().numpy(), O.reset_done())
                if mode == 1:                        # idempotent (same key, _done unchanged)
                    np.testing.assert_array_equal(E.reset_done().cpu().numpy(), O.reset_done())
                called += int(out_o[2].sum())
            elif mode == 2:                          # state read-back installs a requested reset first
                np.testing.assert_array_equal(E.reset_done().cpu().numpy(), O.reset_done())
                assert_state_equal(E.get_state(), O.get_state())
            else:                                    # no reset_done: finished envs stay where they are
                skipped += int(out_o[2].sum())
                if mode == 4:
                    assert_state_equal(E.get_state(), O.get_state())
        assert called > 0 and skipped > 0
        acts = rng.uniform(-1, 1, (5, N, A)).astype(np.float32)   # a rollout right after a requested reset_done
        o, r, d, info = E.step(torch.from_numpy(acts[0]).cuda()); O.step(acts[0])
        np.testing.assert_array_equal(E.reset_done().cpu().numpy(), O.reset_done())
        obs, rew, cost, done = E.rollout(torch.from_numpy(acts).cuda())
        for t in range(5):
            oo, ro, do, io = O.step(acts[t])
            np.testing.assert_array_equal(obs[t].cpu().numpy(), O.reset_done())
            np.testing.assert_array_equal(rew[t].cpu().numpy(), ro)
        assert_state_equal(E.get_state(), O.get_state())
        # reset() right after a requested reset_done: the request is moot
        E.step(torch.from_numpy(acts[1]).cuda()); O.step(acts[1])
        E.reset_done(); O.reset_done()
        np.testing.assert_array_equal(E.reset().cpu().numpy(), O.reset())
        assert_state_equal(E.get_state(), O.get_state())


def test_step_outputs_are_never_overwritten_by_default(torch_cuda):
    """Ownership (engine.py:495 returns fresh buffers): with the default out_ring=0 every tensor
    step() / reset_done() handed out keeps its values however many calls follow (across slab boundaries: here 32
    calls per slab); the opt-in ring of k sets overwrites a tensor exactly k calls later."""
    torch = torch_cuda
    from guardx_amd import Engine
    N = 130
    gen = torch.Generator(device='cuda').manual_seed(3)
    acts = torch.rand(80, N, 2, device='cuda', generator=gen) * 2 - 1
    env = Engine(task_config(N, seed=2, num_steps=9, goal_size=2.8), n_candidates=30000)
    assert env._slab_steps() == 256                              # small outputs: the step cap binds, not the 64 MB
    env._SLAB_STEPS = 32
    env.reset()
    kept, copies = [], []
    for t in range(80):
        o, r, d, info = env.step(acts[t])
        rd = env.reset_done()
        outs = (o, r, d, info['cost'], info['obs']['qacc'], rd)
        kept.append(outs)
        copies.append(tuple(x.clone() for x in outs))
    torch.cuda.synchronize()
    ptrs = {x.data_ptr() for outs in kept for x in outs}
    assert len(ptrs) == 80 * 6                                   # no two live outputs share memory
    for outs, cps in zip(kept, copies):
        for a, b in zip(outs, cps):
            assert torch.equal(a, b)
    env.close()
    ring = Engine(task_config(N, seed=2, num_steps=9, goal_size=2.8), n_candidates=30000, out_ring=8)
    ring.reset()
    first = ring.step(acts[0])[0]
    snap = first.clone()
    for t in range(1, 8):
        assert ring.step(acts[t])[0].data_ptr() != first.data_ptr()
    assert torch.equal(first, snap)                              # untouched for k - 1 further calls ...
    again = ring.step(acts[8])[0]
    assert again.data_ptr() == first.data_ptr() and not torch.equal(first, snap)   # ... and reused by call k
    ring.close()


def test_step_output_slab_is_sized_by_bytes(torch_cuda):
    """the slab step() carves its outputs from holds as many output sets as fit 64 MB (at most 256, at
    least one) -- at 2^20 Point envs one 394 MB set per call, not 32 of them in one 12.6 GB allocation."""
    torch = torch_cuda
    from guardx_amd import Engine
    small = Engine(task_config(2000, seed=1), n_candidates=30000)
    assert small._slab_steps() == (64 << 20) // (4 * small._slab_floats()) == 89
    small.close()
    N = 1 << 20
    env = Engine(task_config(N, seed=1), n_candidates=30000)
    assert env._slab_steps() == 1
    env.reset(check=False)                                       # (pool smaller than env_num: drawn with replacement)
    act = torch.zeros(N, 2, device='cuda')
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    out = env.step(act)
    torch.cuda.synchronize()
    grown = torch.cuda.max_memory_allocated() - base
    assert grown <= 1.25 * 4 * env._slab_floats(), grown         # one output set (+ allocator rounding)
    assert out[0].shape == (N, env.obs_flat_size)
    env.close()


def test_step_reset_done_above_the_group_limit(torch_cuda, oracle):
    """env_num > 16384 runs the thread-per-env kernels: step() does not speculate, reset_done() launches"""
    torch = torch_cuda
    N = 16384 + 640
    E, O = _engines(task_config(N, seed=3, num_steps=6, goal_size=2.7), oracle, n_candidates=120000)
    np.testing.assert_array_equal(E.reset(check=False).cpu().numpy(), O.reset(check=False))
    rng = np.random.default_rng(0)
    for t in range(9):
        act = rng.uniform(-1, 1, (N, 2)).astype(np.float32)
        _cmp_step(E.step(torch.from_numpy(act).cuda()), O.step(act))
        assert E._rd_obs is None
        if t % 2:
            np.testing.assert_array_equal(E.reset_done().cpu().numpy(), O.reset_done())
    assert_state_equal(E.get_state(), O.get_state())


@pytest.mark.parametrize("path", ["thread", "group", "split"])
@pytest.mark.parametrize("robot", ["point", "walker"])
def test_packed_rollout_equals_plain(torch_cuda, oracle, path, robot):
    """gx_rollout_packed: rows (obs | action | reward, cost, done) written by the kernel == gx_rollout's arrays"""
    torch = torch_cuda
    from guardx_amd import Engine
    extra = WALKER if robot == "walker" else {}
    A = 10 if robot == "walker" else 2
    N, T = 203, 31
    cfg = task_config(N, seed=5, num_steps=10, goal_size=2.6, **extra)
    acts = torch.from_numpy(np.random.default_rng(1).uniform(-1, 1, (T, N, A)).astype(np.float32)).cuda()
    outs = []
    for packed in (False, True):
        E = Engine(cfg, n_candidates=30000)
        E.set_path(PATHS[path])
        E.reset()
        outs.append(E.rollout(acts, packed=packed))
        st = E.get_state()
    (o0, r0, c0, d0), (o1, r1, c1, d1, pk) = outs
    D = o0.shape[-1]
    assert pk.shape == (T, N, D + A + 3) and d0.sum().item() > 0
    for a, b in ((o0, o1), (r0, r1), (c0, c1), (d0, d1)):
        assert torch.equal(a, b)
    assert torch.equal(pk[..., D:D + A], acts)
    assert torch.equal(pk[..., :D], o0) and torch.equal(pk[..., D + A + 2], d0)


def test_prefetch_horizon_follows_the_learner(torch_cuda, oracle):
    """the prefetch predicts the interval between the last two reset() calls, not num_steps"""
    torch = torch_cuda
    N = 64
    E, O = _engines(task_config(N, seed=2, num_steps=1000), oracle, n_candidates=30000)   # DEFAULT num_steps
    tape = torch.zeros(25, N, 2, device='cuda')
    for ep in range(5):                                  # the learner resets every max_ep_len = 25 steps
        np.testing.assert_array_equal(E.reset().cpu().numpy(), O.reset())
        E.rollout(tape)
        for t in range(25):
            O.step(np.zeros((N, 2), np.float32)); O.reset_done()
    hits, misses, horizon = E.prefetch_stats()
    assert horizon == 25 and hits == 3 and misses == 1   # epoch 1 mispredicts (1000), 2.. hit


# ---------------------------------------------------------------------------
# BASELINE config 5 (synthetic, no reference counterpart): Ant + 8 hazards + 8 pillars
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("path", ["thread", "group", "split"])
@pytest.mark.parametrize("robot", ["ant", "point"])
def test_pillars_config5_parity(torch_cuda, oracle, path, robot):
    torch = torch_cuda
    from guardx_amd import configuration
    cfg = dict(configuration("Ant_8Hazards_8Pillars_synthetic"))
    A = 8
    if robot == "point":
        cfg['robot_base'] = 'xmls/point.xml'; A = 2
    N, T = 260, 50
    cfg.update(env_num=N, _seed=12, num_steps=35, goal_size=2.4)
    E, O = _engines(cfg, oracle, n_candidates=60000, path=path)
    assert E.obs_flat_size == O.D == (64 if robot == "ant" else 43) + 16
    assert 'pillars_lidar' in E.obs_space_dict
    np.testing.assert_array_equal(E.reset().cpu().numpy(), O.reset())
    assert E.layout_size == O.layout_size
    np.testing.assert_array_equal(E.get_pool(256), O.get_pool(256))
    rng = np.random.default_rng(5)
    s = O.get_state()                      # some robots right next to a pillar / a hazard: both cost terms fire
    base = s['qpos'][:, [0, 2]] if robot == "ant" else s['qpos'][:, :2]
    s['objs'][::3, 9] = base[::3] + rng.uniform(-0.15, 0.15, (len(base[::3]), 2)).astype(np.float32)
    s['objs'][1::3, 1] = base[1::3] + rng.uniform(-0.2, 0.2, (len(base[1::3]), 2)).astype(np.float32)
    E.set_state(s); O.set_state(s)
    pcost = 0
    for t in range(20):                       # step()/reset_done() API
        act = rng.uniform(-1, 1, (N, A)).astype(np.float32)
        out_g, out_o = E.step(torch.from_numpy(act).cuda()), O.step(act)
        _cmp_step(out_g, out_o)
        pcost += int((out_o[3]['cost'] > 0).sum())
        np.testing.assert_array_equal(E.reset_done().cpu().numpy(), O.reset_done())
    assert pcost > 0
    acts = rng.uniform(-1, 1, (T, N, A)).astype(np.float32)   # fused rollout, scattered dones + timeouts
    obs, rew, cost, done = E.rollout(torch.from_numpy(acts).cuda())
    for t in range(T):
        o, r, d, info = O.step(acts[t])
        np.testing.assert_array_equal(obs[t].cpu().numpy(), O.reset_done())
        np.testing.assert_array_equal(rew[t].cpu().numpy(), r)
        np.testing.assert_array_equal(done[t].cpu().numpy(), d)
        np.testing.assert_array_equal(cost[t].cpu().numpy(), info['cost'])
    assert done.sum().item() > 0
    assert_state_equal(E.get_state(), O.get_state())
    np.testing.assert_array_equal(E.reset().cpu().numpy(), O.reset())


def _bench_epochs(torch, oracle, cfg, A, epochs, N=2000, T=200, M=1_000_000, num_steps=None, prefetch=None):
    """bench.py's epoch at its own sizes: env_num=2000, n_candidates=1e6 (engine.py:263), 200-step epochs of
    reset(check=False) + rollout with the default layout-pool prefetch, back to back, against the checker:
    pool head, layout_size, every observation / reward / cost / done row, final state, prefetch hits.
    Returns the number of done events seen."""
    cfg = dict(cfg)
    cfg.update(env_num=N, _seed=0, num_steps=T if num_steps is None else num_steps)
    E, O = _engines(cfg, oracle, n_candidates=M)
    if prefetch is not None:
        E.set_prefetch(prefetch)                      # bench.py: env.set_prefetch(EP_LEN)
    rng = np.random.default_rng(0)
    ndone = 0
    for ep in range(epochs):
        og, oo = E.reset(check=False), O.reset()
        np.testing.assert_array_equal(og.cpu().numpy(), oo)
        if ep != 1:
            np.testing.assert_array_equal(E.get_pool(4096), O.get_pool(4096))
        acts = rng.uniform(-1, 1, (T, N, A)).astype(np.float32)
        obs, rew, cost, done = E.rollout(torch.from_numpy(acts).cuda())
        obs, rew, cost, done = obs.cpu().numpy(), rew.cpu().numpy(), cost.cpu().numpy(), done.cpu().numpy()
        for t in range(T):
            o, r, d, info = O.step(acts[t])
            if d.any():
                o = O.reset_done()
            np.testing.assert_array_equal(obs[t], o, err_msg=f"obs epoch {ep} step {t}")
            np.testing.assert_array_equal(rew[t], r, err_msg=f"reward epoch {ep} step {t}")
            np.testing.assert_array_equal(done[t], d, err_msg=f"done epoch {ep} step {t}")
            np.testing.assert_array_equal(cost[t], info['cost'], err_msg=f"cost epoch {ep} step {t}")
            ndone += int(d.sum())
        assert E.check_layouts() == O.layout_size > N
    assert_state_equal(E.get_state(), O.get_state())
    hits, misses, horizon = E.prefetch_stats()
    assert (hits, misses, horizon) == (epochs - 1, 0, T)       # every epoch after the first took the prefetched pool
    E.close()
    return ndone


def test_bench_workload_three_epochs(torch_cuda, oracle):
    """EXACTLY bench.py's headline workload (BASELINE config 2), three epochs back to back."""
    _bench_epochs(torch_cuda, oracle, task_config(2000), 2, epochs=3)


@pytest.mark.parametrize("name", ["Goal_Swimmer_8Hazards", "Goal_Ant_8Hazards", "Goal_Walker_8Hazards",
                                  "Ant_8Hazards_8Pillars_synthetic"])
def test_bench_workload_epochs_other_robots(torch_cuda, oracle, name):
    """bench.py's `other_robots` epochs at the sizes it times them (env_num=2000, 1e6 layout candidates, 200 steps,
    prefetch hits): BASELINE config 3 (Swimmer), Ant, Walker, and config 5's synthetic stand-in (Ant + 8 hazards + 8
    pillars: the 18-object sampler and the two-objects-per-lane kernel instance)."""
    from guardx_amd import configuration
    cfg = configuration(name)
    A = {'xmls/swimmer.xml': 2, 'xmls/ant.xml': 8, 'xmls/walker.xml': 10}[cfg['robot_base']]
    _bench_epochs(torch_cuda, oracle, cfg, A, epochs=2)


@pytest.mark.parametrize("name", ["Goal_Point_8Hazards", "Goal_Swimmer_8Hazards", "Ant_8Hazards_8Pillars_synthetic"])
def test_bench_workload_reset_done_heavy(torch_cuda, oracle, name):
    """bench.py's `reset_done_heavy` extra at the bench's sizes: the same epoch with episodes shorter than the
    epoch (num_steps = 60: the timeout of engine.py:492 ends every episode on its 62nd step) and a goal wide enough
    that envs also finish at scattered times -- every env runs the reset_done branch (layout draw, re-placement,
    re-initialised observation row) about three times per epoch inside the rollout."""
    from guardx_amd import configuration
    cfg = dict(configuration(name))
    cfg['goal_size'] = 2.9
    A = {'xmls/point.xml': 2, 'xmls/swimmer.xml': 2, 'xmls/ant.xml': 8}[cfg['robot_base']]
    ndone = _bench_epochs(torch_cuda, oracle, cfg, A, epochs=2, num_steps=60, prefetch=200)
    assert ndone >= 2 * 3 * 2000


def test_config4_shape_eight_shards_of_2000(torch_cuda, oracle):
    """BASELINE config 4 on one GPU: eight Engine(env_num=2000, shard=(r, 8)) reproduce the rows of one
    16 000-env engine (1e6 candidates) through reset, a fused rollout with resets and the next reset; the
    16 000-env engine itself is checked against the checker."""
    torch = torch_cuda
    from guardx_amd import Engine
    N, W, T, M = 2000, 8, 40, 1_000_000
    cfg_full = task_config(N * W, seed=3, num_steps=25, goal_size=2.95)
    full, O = _engines(cfg_full, oracle, n_candidates=M)
    o_full = full.reset()
    np.testing.assert_array_equal(o_full.cpu().numpy(), O.reset())
    assert full.layout_size == O.layout_size > N * W
    acts = torch.from_numpy(np.random.default_rng(1).uniform(-1, 1, (T, N * W, 2)).astype(np.float32)).cuda()
    obs_f, rew_f, cost_f, done_f = full.rollout(acts)
    assert done_f.sum().item() >= N * W                     # every env times out once (plus the odd goal): reset_done on every shard
    a_np = acts.cpu().numpy()
    for t in range(T):
        o, r, d, info = O.step(a_np[t])
        if t in (0, 12, 26, 27, T - 1):                     # spot-check the big engine against the checker
            np.testing.assert_array_equal(obs_f[t].cpu().numpy(), O.reset_done())
            np.testing.assert_array_equal(done_f[t].cpu().numpy(), d)
        else:
            O.reset_done()
    o_full2 = full.reset()
    np.testing.assert_array_equal(o_full2.cpu().numpy(), O.reset())
    for r in range(W):
        sh = Engine(task_config(N, seed=3, num_steps=25, goal_size=2.95), n_candidates=M, shard=(r, W))
        sl = slice(r * N, (r + 1) * N)
        assert torch.equal(sh.reset(), o_full[sl])
        obs, rew, cost, done = sh.rollout(acts[:, sl].contiguous())
        assert torch.equal(obs, obs_f[:, sl]) and torch.equal(rew, rew_f[:, sl])
        assert torch.equal(cost, cost_f[:, sl]) and torch.equal(done, done_f[:, sl])
        assert torch.equal(sh.reset(), o_full2[sl])
        sh.close()


def test_sharded_equals_unsharded(torch_cuda, oracle):
    """rank r of a world-of-4 engine reproduces rows [r*N, (r+1)*N) of one 4N-env engine."""
    torch = torch_cuda
    from guardx_amd import Engine
    N, W = 96, 4
    full = Engine(task_config(N * W, seed=4), n_candidates=50000)
    obs_full = full.reset().cpu().numpy()
    rng = np.random.default_rng(0)
    act = rng.uniform(-1, 1, (N * W, 2)).astype(np.float32)
    o_full, r_full, d_full, _ = full.step(torch.from_numpy(act).cuda())
    full_rd = full.reset_done().cpu().numpy()
    for r in range(W):
        sh = Engine(task_config(N, seed=4), n_candidates=50000, shard=(r, W))
        np.testing.assert_array_equal(sh.reset().cpu().numpy(), obs_full[r * N:(r + 1) * N])
        o, rew, d, _ = sh.step(torch.from_numpy(act[r * N:(r + 1) * N]).cuda())
        np.testing.assert_array_equal(o.cpu().numpy(), o_full[r * N:(r + 1) * N].cpu().numpy())
        np.testing.assert_array_equal(sh.reset_done().cpu().numpy(), full_rd[r * N:(r + 1) * N])


@pytest.mark.parametrize("prefetch", [None, -1, 7])
def test_layout_prefetch_hit_and_miss_give_identical_resets(torch_cuda, oracle, prefetch):
    """The pool of the next reset() is sampled ahead on a side stream for a PREDICTED key; a
    mispredicted number of steps must fall back to inline sampling with identical results."""
    torch = torch_cuda
    N = 64
    E, O = _engines(task_config(N, seed=21, num_steps=10), oracle, n_candidates=30000)
    if prefetch is not None:
        E.set_prefetch(prefetch)
    rng = np.random.default_rng(1)
    for nsteps in (0, 3, 10, 10, 7, 1, 10):      # default prediction is num_steps = 10
        np.testing.assert_array_equal(E.reset().cpu().numpy(), O.reset())
        assert E.layout_size == O.layout_size
        np.testing.assert_array_equal(E.get_pool(64), O.get_pool(64))
        for _ in range(nsteps):
            act = rng.uniform(-1, 1, (N, 2)).astype(np.float32)
            _cmp_step(E.step(torch.from_numpy(act).cuda()), O.step(act))
            np.testing.assert_array_equal(E.reset_done().cpu().numpy(), O.reset_done())
    assert_state_equal(E.get_state(), O.get_state())


def test_engine_api_contract(torch_cuda):
    """Shapes / dtypes / devices the unmodified learners rely on (trpo.py:449-547)."""
    torch = torch_cuda
    from guardx_amd import Engine
    import pickle
    N = 100
    env = Engine(task_config(N, seed=0, num_steps=20), n_candidates=30000)
    assert env.observation_space.shape == (43,) and env.action_space.shape == (2,)
    o = env.reset()
    assert o.shape == (N, 43) and o.dtype == torch.float32 and o.device.type == 'cuda'
    act = torch.randn(N, 2, device='cuda')
    o2, r, d, info = env.step(act)
    assert o2.data_ptr() != o.data_ptr()                      # fresh buffers every call
    assert r.shape == d.shape == info['cost'].shape == (N,)
    assert all(t.dtype == torch.float32 for t in (o2, r, d, info['cost']))
    assert set(torch.unique(d).tolist()) <= {0.0, 1.0}
    assert set(info['obs']) == {'ctrl', 'goal_compass', 'goal_lidar', 'hazards_lidar', 'qpos', 'qvel', 'qacc'}
    torch.testing.assert_close(info['obs']['goal_lidar'], o2[:, 5:21], rtol=0, atol=0)
    torch.testing.assert_close(info['obs']['qvel'], o2[:, 40:43], rtol=0, atol=0)
    keep = o2.clone()
    o3 = env.reset_done()
    torch.testing.assert_close(o2, keep, rtol=0, atol=0)      # self._obs is not modified (engine.py:501)
    nd = d == 0
    torch.testing.assert_close(o3[nd], o2[nd], rtol=0, atol=0)
    # learner-style in-place edit of a returned tensor must not corrupt the env (trpo.py:529)
    o3[o3.isnan()] = 0
    env.step(act)
    # the pickle is the config (EzPickle, engine.py:214)
    env2 = pickle.loads(pickle.dumps(env))
    assert env2.env_num == N and env2.num_steps == 20
    torch.testing.assert_close(env2.reset(), Engine(task_config(N, seed=0, num_steps=20), n_candidates=30000).reset(),
                               rtol=0, atol=0)
    with pytest.raises(ValueError):
        env.step(torch.zeros(N, 3, device='cuda'))


def test_learner_loop_contract(torch_cuda, oracle):
    """The TRPO collection loop (trpo.py:466-547) driven verbatim against the Engine, with the
    oracle in lock-step: host-side bookkeeping (ep_ret / ep_cost / done handling) matches."""
    torch = torch_cuda
    N, T = 200, 60
    E, O = _engines(task_config(N, seed=8, num_steps=T, goal_size=2.8), oracle, n_candidates=40000)   # wide goal: the 1.5 m/s Point finishes some episodes
    o, oo = E.reset(), O.reset()
    gen = torch.Generator(device='cuda').manual_seed(0)
    ep_ret = np.zeros(N); ep_cost = np.zeros(N); ep_ret_o = np.zeros(N); ep_cost_o = np.zeros(N)
    finished = 0
    for t in range(T):
        act = torch.rand(N, 2, device='cuda', generator=gen) * 2 - 1          # stand-in for ac.step(o)
        next_o, r, d, info = E.step(act)
        no_o, r_o, d_o, info_o = O.step(act.cpu().numpy())
        assert 'cost' in info.keys()
        ep_ret += r.cpu().numpy().squeeze(); ep_cost += info['cost'].cpu().numpy().squeeze()
        ep_ret_o += r_o; ep_cost_o += info_o['cost']
        o = next_o
        timeout = (t + 1) == T
        terminal = d.cpu().numpy().any() > 0 or timeout
        if terminal and not timeout:
            done = d.cpu().numpy()
            np.testing.assert_array_equal(done, d_o)
            finished += int(done.sum())
            ep_ret[np.where(done == 1)] = 0; ep_ret_o[np.where(d_o == 1)] = 0
            o = E.reset_done()
            np.testing.assert_array_equal(o.cpu().numpy(), O.reset_done())
    assert finished > 0
    np.testing.assert_array_equal(ep_ret, ep_ret_o)
    np.testing.assert_array_equal(ep_cost, ep_cost_o)


# ---------------------------------------------------------------------------
# Swimmer (BASELINE config 3): articulated dynamics + joint-limit rows
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("path", ["thread", "group"])
@pytest.mark.parametrize("N", [5, 64, 2000])
def test_swimmer_step_parity_random_states(torch_cuda, oracle, N, path):
    torch = torch_cuda
    E, O = _engines(task_config(N, seed=3, **SWIMMER), oracle, path=path)
    assert E.obs_flat_size == O.D == 46 and E.action_space.shape == (2,)
    assert float(E.action_space.low[0]) == -1.0 and float(E.action_space.high[1]) == 1.0
    rng = np.random.default_rng(N)
    for trial in range(3):
        s = random_state(N, 8, rng, robot='swimmer')
        s['hist'] = [2, 1, 0][trial]
        E.set_state(s)
        O.set_state(s)
        act = rng.uniform(-1.4, 1.4, (N, 2)).astype(np.float32)      # beyond ctrlrange: clipped for the force
        out_g = E.step(torch.from_numpy(act).cuda())
        out_o = O.step(act)
        _cmp_step(out_g, out_o)
        assert_state_equal(E.get_state(), O.get_state())


@pytest.mark.parametrize("path", ["thread", "group", "split", "split-alone"])
def test_swimmer_rollout_parity(torch_cuda, oracle, path):
    torch = torch_cuda
    N, T = 300, 120
    E, O = _engines(task_config(N, seed=6, num_steps=80, goal_size=1.0, **SWIMMER), oracle,
                    n_candidates=60000, path=path)
    np.testing.assert_array_equal(E.reset().cpu().numpy(), O.reset())
    rng = np.random.RandomState(1)
    for t in range(40):                       # step()/reset_done() API
        act = rng.uniform(-1, 1, (N, 2)).astype(np.float32)
        _cmp_step(E.step(torch.from_numpy(act).cuda()), O.step(act))
        np.testing.assert_array_equal(E.reset_done().cpu().numpy(), O.reset_done())
    acts = rng.uniform(-1, 1, (T, N, 2)).astype(np.float32)   # fused rollout
    obs, rew, cost, done = E.rollout(torch.from_numpy(acts).cuda())
    hit = 0
    for t in range(T):
        o, r, d, info = O.step(acts[t])
        hit += int((np.abs(O.get_state()['qpos'][:, 3:]) > 1.7453293).sum())
        np.testing.assert_array_equal(obs[t].cpu().numpy(), O.reset_done())
        np.testing.assert_array_equal(rew[t].cpu().numpy(), r)
        np.testing.assert_array_equal(done[t].cpu().numpy(), d)
        np.testing.assert_array_equal(cost[t].cpu().numpy(), info['cost'])
    assert hit > 0 and done.sum().item() > 0
    assert_state_equal(E.get_state(), O.get_state())
    np.testing.assert_array_equal(E.reset().cpu().numpy(), O.reset())


# ---------------------------------------------------------------------------
# Ant (Goal_Ant_8Hazards): 11-DOF tree, joint-limit rows, foot-floor contacts with friction pyramids
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("path", ["thread", "group"])
@pytest.mark.parametrize("N", [5, 64, 2000])
def test_ant_step_parity_random_states(torch_cuda, oracle, N, path):
    torch = torch_cuda
    E, O = _engines(task_config(N, seed=3, **ANT), oracle, path=path)
    assert E.obs_flat_size == O.D == 64 and E.action_space.shape == (8,)
    assert float(E.action_space.low[0]) == -1.0 and float(E.action_space.high[7]) == 1.0
    rng = np.random.default_rng(N)
    for trial in range(3):
        s = random_state(N, 8, rng, robot='ant')
        s['hist'] = [2, 1, 0][trial]
        E.set_state(s)
        O.set_state(s)
        act = rng.uniform(-1.4, 1.4, (N, 8)).astype(np.float32)      # beyond ctrlrange: clipped for the force
        out_g = E.step(torch.from_numpy(act).cuda())
        out_o = O.step(act)
        _cmp_step(out_g, out_o)
        assert_state_equal(E.get_state(), O.get_state())


@pytest.mark.parametrize("path", ["thread", "group", "split", "split-alone"])
def test_ant_rollout_parity(torch_cuda, oracle, path):
    torch = torch_cuda
    N, T = 300, 100
    E, O = _engines(task_config(N, seed=6, num_steps=60, goal_size=1.0, **ANT), oracle,
                    n_candidates=60000, path=path)
    np.testing.assert_array_equal(E.reset().cpu().numpy(), O.reset())
    rng = np.random.RandomState(1)
    for t in range(30):                       # step()/reset_done() API
        act = rng.uniform(-1, 1, (N, 8)).astype(np.float32)
        _cmp_step(E.step(torch.from_numpy(act).cuda()), O.step(act))
        np.testing.assert_array_equal(E.reset_done().cpu().numpy(), O.reset_done())
    acts = rng.uniform(-1, 1, (T, N, 8)).astype(np.float32)   # fused rollout
    obs, rew, cost, done = E.rollout(torch.from_numpy(acts).cuda())
    floor = 0
    for t in range(T):
        o, r, d, info = O.step(acts[t])
        floor += int((np.abs(O.get_state()['qpos'][:, 4::2]) > 1.02).sum())   # ankle past ~58 deg: foot on the floor
        np.testing.assert_array_equal(obs[t].cpu().numpy(), O.reset_done())
        np.testing.assert_array_equal(rew[t].cpu().numpy(), r)
        np.testing.assert_array_equal(done[t].cpu().numpy(), d)
        np.testing.assert_array_equal(cost[t].cpu().numpy(), info['cost'])
    assert floor > 0 and done.sum().item() > 0
    assert_state_equal(E.get_state(), O.get_state())
    np.testing.assert_array_equal(E.reset().cpu().numpy(), O.reset())


# ---------------------------------------------------------------------------
# Walker (Goal_Walker_8Hazards): 13-DOF tree with 3-D leg chains, joint limits, springs, gravity, foot contacts
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("path", ["thread", "group"])
@pytest.mark.parametrize("N", [5, 64, 2000])
def test_walker_step_parity_random_states(torch_cuda, oracle, N, path):
    torch = torch_cuda
    E, O = _engines(task_config(N, seed=3, **WALKER), oracle, path=path)
    assert E.obs_flat_size == O.D == 70 and E.action_space.shape == (10,)
    rng = np.random.default_rng(N)
    for trial in range(3):
        s = random_state(N, 8, rng, robot='walker')
        s['hist'] = [2, 1, 0][trial]
        E.set_state(s)
        O.set_state(s)
        act = rng.uniform(-1.4, 1.4, (N, 10)).astype(np.float32)
        out_g = E.step(torch.from_numpy(act).cuda())
        out_o = O.step(act)
        _cmp_step(out_g, out_o)
        assert_state_equal(E.get_state(), O.get_state())


@pytest.mark.parametrize("path", ["thread", "group", "split", "split-alone"])
def test_walker_rollout_parity(torch_cuda, oracle, path):
    torch = torch_cuda
    N, T = 300, 100
    E, O = _engines(task_config(N, seed=6, num_steps=60, goal_size=1.0, **WALKER), oracle,
                    n_candidates=60000, path=path)
    np.testing.assert_array_equal(E.reset().cpu().numpy(), O.reset())
    rng = np.random.RandomState(1)
    for t in range(30):                       # step()/reset_done() API
        act = rng.uniform(-1, 1, (N, 10)).astype(np.float32)
        _cmp_step(E.step(torch.from_numpy(act).cuda()), O.step(act))
        np.testing.assert_array_equal(E.reset_done().cpu().numpy(), O.reset_done())
    acts = rng.uniform(-1, 1, (T, N, 10)).astype(np.float32)   # fused rollout
    obs, rew, cost, done = E.rollout(torch.from_numpy(acts).cuda())
    for t in range(T):
        o, r, d, info = O.step(acts[t])
        np.testing.assert_array_equal(obs[t].cpu().numpy(), O.reset_done())
        np.testing.assert_array_equal(rew[t].cpu().numpy(), r)
        np.testing.assert_array_equal(done[t].cpu().numpy(), d)
        np.testing.assert_array_equal(cost[t].cpu().numpy(), info['cost'])
    assert done.sum().item() > 0
    assert_state_equal(E.get_state(), O.get_state())
    np.testing.assert_array_equal(E.reset().cpu().numpy(), O.reset())


def test_fixed_locations_and_placement_rectangles(torch_cuda, oracle):
    """*_locations / single-rectangle *_placements (engine.py:507-531, 600-612)"""
    torch = torch_cuda
    from guardx_amd import Engine
    N = 96
    cfg = task_config(N, seed=13, goal_locations=[(1.2, 1.1)], hazards_locations=[(0.0, 0.0), (-0.5, 0.9)],
                      robot_placements=[(-2, -2, -1, 2)], hazards_placements=[(-2, -2, 2, 0.5)])
    E, O = _engines(cfg, oracle, n_candidates=40000)
    og, oo = E.reset().cpu().numpy(), O.reset()
    np.testing.assert_array_equal(og, oo)
    np.testing.assert_array_equal(E.get_pool(256), O.get_pool(256))
    st = E.get_state()
    np.testing.assert_allclose(st['objs'][:, 0], np.tile([1.2, 1.1], (N, 1)), atol=1e-6)
    np.testing.assert_allclose(st['objs'][:, 1], 0.0, atol=1e-6)
    assert (st['qpos'][:, 0] <= -1.4 + 1e-6).all() and (st['objs'][:, 3:, 1] <= 0.1 + 1e-6).all()
    rng = np.random.default_rng(0)
    for t in range(12):
        act = rng.uniform(-1, 1, (N, 2)).astype(np.float32)
        _cmp_step(E.step(torch.from_numpy(act).cuda()), O.step(act))
        np.testing.assert_array_equal(E.reset_done().cpu().numpy(), O.reset_done())
    with pytest.raises(AttributeError):      # multi-rectangle placements: self.rs is undefined (engine.py:616)
        Engine(task_config(4, goal_placements=[(-2, -2, 0, 0), (0, 0, 2, 2)]), n_candidates=1000)


def test_deferred_layout_check(torch_cuda):
    """reset(check=False) + check_layouts(): same assert as engine.py:444, one sync for many resets"""
    torch = torch_cuda
    from guardx_amd import Engine, ResamplingError
    env = Engine(task_config(50, seed=0, num_steps=5), n_candidates=20000)
    tape = torch.zeros(5, 50, 2, device='cuda')
    for _ in range(3):
        env.reset(check=False)
        env.rollout(tape)
    assert env.check_layouts() > 50
    bad = Engine(task_config(500, seed=0, num_steps=5), n_candidates=2000)   # pool smaller than env_num
    bad.reset(check=False)
    with pytest.raises(ResamplingError):
        bad.check_layouts()
    with pytest.raises(ResamplingError):
        bad.reset()


@pytest.mark.parametrize("robot", ["point", "swimmer", "ant", "walker"])
def test_tape_handoff_two_ranks_equal_the_packed_rollout(torch_cuda, robot):
    """The multi-GPU hand-off on one GPU: two shard engines ("ranks") step with rollout_tape(), exchange the shard
    buffers, and each expands BOTH tapes with expand_tape() on another stream one epoch later (what the all-gather
    overlap needs: the pool ring keeps the layouts alive).  Every expansion equals the packed rows of one unsharded
    engine's rollout(packed=True) bit for bit -- reset_done events, timeouts and the NaN guard included; a tape
    whose pool has been resampled is refused."""
    torch = torch_cuda
    from guardx_amd import Engine
    N, W, T, M, EPOCHS = 192, 2, 50, 60000, 4
    kw = dict(seed=9, num_steps=17, goal_size=2.6)
    kw.update({"swimmer": SWIMMER, "ant": ANT, "walker": WALKER}.get(robot, {}))
    full = Engine(task_config(N * W, **kw), n_candidates=M)
    ranks = [Engine(task_config(N, **kw), n_candidates=M, shard=(r, W)) for r in range(W)]
    A = full.action_space.shape[0]
    rng = np.random.default_rng(5)
    side = torch.cuda.Stream()
    pending = None                                             # (epoch, shards, tokens) waiting to be expanded
    ref = {}
    done_total = 0

    def expand_and_check(ep, shards, tokens):
        want = ref.pop(ep)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            outs = [[e.expand_tape(shards[src], tokens[src], T) for src in range(W)] for e in ranks]
        torch.cuda.current_stream().wait_stream(side)
        for by in range(W):
            for src in range(W):
                w = want[:, src * N:(src + 1) * N].contiguous()
                assert torch.equal(outs[by][src].view(torch.int32), w.view(torch.int32)), (ep, by, src)   # bits: NaN rows too

    for ep in range(EPOCHS):
        acts = rng.uniform(-1, 1, (T, N * W, A)).astype(np.float32)
        if ep == 2:
            acts[7, 3::11] = np.nan                            # the NaN guard fires in both shards
        acts = torch.from_numpy(acts).cuda()
        o_full = full.reset()
        for r, e in enumerate(ranks):
            assert torch.equal(e.reset(), o_full[r * N:(r + 1) * N])
        # the previous epoch's tapes are expanded only now, AFTER the next reset (the pool ring keeps their layouts)
        if pending is not None:
            expand_and_check(*pending)
        *_, pk = full.rollout(acts, packed=True)
        ref[ep] = pk
        done_total += int(pk[..., -1].sum().item())
        shards, tokens = [], []
        for r, e in enumerate(ranks):
            sh, tok = e.rollout_tape(acts[:, r * N:(r + 1) * N].contiguous())
            shards.append(sh); tokens.append(tok)
        pending = (ep, shards, tokens)
    stale = pending
    expand_and_check(*pending)
    assert done_total > 2 * N * W                              # timeouts every 18 steps plus goals: reset_done on every env
    # two more resets: the pool of the last tapes is resampled, the token is refused
    for e in ranks:
        e.reset(); e.reset()
    with pytest.raises(RuntimeError, match="resampled"):
        ranks[0].expand_tape(stale[1][0], stale[2][0], T)
    # the engines are still in step with the unsharded one
    full.reset(); full.reset()
    o_full = full.reset()
    for r, e in enumerate(ranks):
        assert torch.equal(e.reset(), o_full[r * N:(r + 1) * N])
    full.close()
    for e in ranks:
        e.close()


def test_tape_handoff_object_pipeline(torch_cuda):
    """guardx_amd.dist.TapeHandoff as bench.py drives it (world of one: no collective): reset() + step(actions) per
    epoch, the expansion of epoch k enqueued during epoch k+1 on the hand-off's own stream, default prefetch.  After
    drain() `rollout` holds the last epoch's packed rows = a twin engine's rollout(packed=True)."""
    torch = torch_cuda
    from guardx_amd import Engine
    from guardx_amd.dist import TapeHandoff
    N, T, M = 500, 60, 200_000
    cfg = task_config(N, seed=2, num_steps=T, goal_size=2.9)
    a, b = Engine(cfg, n_candidates=M), Engine(cfg, n_candidates=M)
    h = TapeHandoff(a, T)
    rng = np.random.default_rng(11)
    dones = 0
    for ep in range(5):
        acts = torch.from_numpy(rng.uniform(-1, 1, (T, N, 2)).astype(np.float32)).cuda()
        oa, ob = a.reset(check=False), b.reset(check=False)
        assert torch.equal(oa, ob)
        h.step(acts)
        *_, pk = b.rollout(acts, packed=True)
        if ep >= 1:
            torch.cuda.current_stream().wait_stream(h.stream)
            assert torch.equal(h.rollout[0], prev)                 # epoch ep-1, expanded during epoch ep
        prev = pk
        dones += int(pk[..., -1].sum().item())
    h.drain()
    assert torch.equal(h.rollout[0], prev)
    assert dones > 0                                               # goals were reached: reset_done rows in the tapes
    a.check_layouts(); b.check_layouts()
    assert a.prefetch_stats()[0] == 4                               # every later reset took the prefetched pool
    a.close(); b.close()


def test_sampler_grid_stride_iterations(torch_cuda, oracle, monkeypatch):
    """sample_phase1 / sample_phase2 are grid-stride kernels whose grids are capped (3072 / 2048 workgroups); at the
    production size every workgroup handles about one batch.  With the cap forced down to 6 workgroups the same
    kernels take ~30 (phase 1) and ~8 (phase 2) iterations per workgroup: pool, layout_size and reset observations
    still equal the checker's, with pillars too (18 objects: the one-wave-per-workgroup LDS layout of phase 2)."""
    torch = torch_cuda
    monkeypatch.setenv("GX_SAMPLE_GRID_CAP", "6")
    for extra in ({}, {'pillars_num': 8, 'observe_pillars': True, 'pillars_keepout': 0.3, 'pillars_size': 0.2,
                       'placements_extents': [-3, -3, 3, 3]},
                  {'pillars_num': 24, 'observe_pillars': True, 'pillars_keepout': 0.15, 'pillars_size': 0.1,
                   'placements_extents': [-4, -4, 4, 4]}):
        cfg = task_config(300, seed=21)
        cfg.update(extra)
        E, O = _engines(cfg, oracle, n_candidates=60000)
        E.set_prefetch(-1)
        for _ in range(2):
            np.testing.assert_array_equal(E.reset().cpu().numpy(), O.reset())
            assert E.layout_size == O.layout_size
            n = min(E.layout_size, 3000)
            np.testing.assert_array_equal(E.get_pool(n), O.get_pool(n))
        E.close()


@pytest.mark.parametrize("fused", ["0", "1"])
@pytest.mark.parametrize("name", ["Goal_Point_8Hazards", "Ant_8Hazards_8Pillars_synthetic"])
def test_both_forms_of_the_layout_sampler_give_the_checkers_pool(torch_cuda, oracle, monkeypatch, name, fused):
    """The layout sampler has two forms (gx_kernels.hip): three phases (phase 1 walks the key chain to the robot's tries
    and rejects the candidates that cannot end 3.0 from the goal before any hazard is placed) and the fused form for
    sparse arenas (one walk: objects first, the robot's tries at the end).  gx_create picks by geometry; GX_SAMPLE_FUSED
    forces either.  Both give the checker's pool -- size, rows, order -- in the reference's dense arena and in the sparse
    synthetic one, over consecutive resets (prefetch hits), at a candidate count that takes several grid-stride rounds."""
    torch = torch_cuda
    from guardx_amd import configuration
    monkeypatch.setenv("GX_SAMPLE_FUSED", fused)
    monkeypatch.setenv("GX_SAMPLE_GRID_CAP", "64")
    cfg = dict(configuration(name))
    cfg.update(env_num=96, _seed=11, num_steps=50)
    E, O = _engines(cfg, oracle, n_candidates=150_000)
    for ep in range(3):
        np.testing.assert_array_equal(E.reset().cpu().numpy(), O.reset())
        assert E.layout_size == O.layout_size > 96
        n = min(E.layout_size, 4000)
        np.testing.assert_array_equal(E.get_pool(n), O.get_pool(n))
    E.close()


def test_bench_two_ranks_on_one_gpu_rehearsal(torch_cuda):
    """bench.py's N > 1 path end to end on this box: `--gpus 2` with no launcher starts two rank processes itself;
    both share GPU 0 and talk over gloo (RCCL refuses two ranks on one device), shards staged through host memory.
    What is checked is the plumbing the 8-GPU run depends on -- spawn, rendezvous, barrier, max-over-ranks timing, the
    tape hand-off ring with the layout sampler sharded over the two ranks (each samples half of the candidates of the
    reset after next, the rows ride on the tape all-gather, also over gloo), every leg the line reports at N > 1
    (`value` as the median of --reps repetitions, stepping_only, unsharded_sampler, local_expand, preconditioned), the deferred layout check of every leg, ONE
    JSON line from rank 0 -- not the rate (gloo through host memory is two orders of magnitude slower than xGMI)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GX_BENCH_FORCE_DEVICE="0", GX_DIST_BACKEND="gloo", GX_BENCH_SPAWN_TIMEOUT="500")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                          "--reps", "3", "--no-extras", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["scaling"] == "weak"
    assert line["env_steps_timed"] == 2 * 2000 * 200 * 3 and line["value"] > 0
    so = line["stepping_only"]
    assert so["handoff"] == "tape" and so["value"] > line["value"]
    assert so["handoff_bytes_received_per_rank_per_epoch"] < so["packed_rows_bytes_per_rank_per_epoch"] / 2
    assert "roofline" in line and line["preconditioned"]["value"] > 0
    reps = line["repetitions"]
    assert reps["n"] == 3 and reps["min"] <= line["value"] <= reps["max"] and line["value"] in reps["values"]
    assert line["config"]["layout_sampler"].startswith("sharded")
    assert line["legs"]["unsharded_sampler"]["value"] > 0 and line["legs"]["local_expand"]["value"] > 0


def test_tape_handoff_is_refused_where_it_does_not_apply(torch_cuda):
    """gx_rollout_tape / gx_expand_tape: configurations with the pose history in the observation are
    GX_ERR_UNSUPPORTED (every robot is supported since round 3: the Ant / Walker reset_done rows come from the pool's
    fake-step table), a rollout before reset() is GX_ERR_STATE, a shard of the wrong size is caught by the Python
    mirror."""
    torch = torch_cuda
    from guardx_amd import Engine
    from guardx_amd._native import GxError, GX_ERR_UNSUPPORTED, GX_ERR_STATE
    acts = torch.zeros(4, 32, 8, device="cuda")
    for cfg in (task_config(32, seed=1, observe_vel=True), task_config(32, seed=1, observe_vel=True, **ANT)):
        e = Engine(cfg, n_candidates=20000)
        e.reset()
        with pytest.raises(GxError) as ei:
            e.tape_floats(4)
        assert ei.value.status == GX_ERR_UNSUPPORTED
        e.close()
    e = Engine(task_config(32, seed=1), n_candidates=20000)
    with pytest.raises(GxError) as ei:
        e.rollout_tape(acts[..., :2].contiguous())
    assert ei.value.status == GX_ERR_STATE
    e.reset()
    sh, tok = e.rollout_tape(acts[..., :2].contiguous())
    with pytest.raises(AssertionError):
        e.expand_tape(sh[:-4], tok, 4)
    out = e.expand_tape(sh, tok, 4)
    assert out.shape == (4, 32, 48) and torch.isfinite(out).all()
    e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("robot,n_shards", [("point", 2), ("point", 8), ("ant", 3)])
def test_sharded_layout_sampler_equals_the_unsharded_reset(torch_cuda, oracle, robot, n_shards):
    """gx_sample_shard / gx_reset_from_shards (optional second collective, default off): every shard samples its range
    of the 1e6-candidate list, the exports are concatenated as an all-gather would, and the installed pool -- layout_size,
    rows, reset observation, the steps and reset_done draws that follow, the next reset -- equals the unsharded engine's and
    the checker's."""
    torch = torch_cuda
    from guardx_amd import Engine, ResamplingError
    extra = {"point": {}, "ant": ANT}[robot]
    A = {"ant": 8}.get(robot, 2)
    N, M = 96, 200_000
    cfg = task_config(N, seed=4, num_steps=30, goal_size=1.5, **extra)
    full, O = _engines(cfg, oracle, n_candidates=M)
    sh = Engine(cfg, n_candidates=M)
    with pytest.raises(RuntimeError, match="prefetch"):
        sh.sample_shard(0, n_shards)                       # refused while the layout prefetch is on
    sh.set_prefetch(-1)
    rng = np.random.default_rng(2)
    for ep in range(3):
        o_full = full.reset()
        if ep == 0:                                        # (the checker is not stepped below)
            np.testing.assert_array_equal(o_full.cpu().numpy(), O.reset())
        parts = [sh.sample_shard(r, n_shards) for r in range(n_shards)]   # one engine plays every rank in turn
        rows_all = torch.stack([p[0].clone() for p in parts]); counts = torch.cat([p[1].clone() for p in parts])
        assert int(counts.sum().item()) == full.layout_size and int(counts.min().item()) > 0
        o_sh = sh.reset_from_shards(rows_all, counts)
        assert sh.layout_size == full.layout_size
        assert torch.equal(o_sh, o_full)
        np.testing.assert_array_equal(sh.get_pool(64), full.get_pool(64))
        for t in range(35):                                # crosses the timeout: reset_done draws from the installed pool
            a = torch.from_numpy(rng.uniform(-1, 1, (N, A)).astype(np.float32)).cuda()
            og, rg, dg, ig = sh.step(a)
            of, rf, df, i_f = full.step(a)
            assert torch.equal(og, of) and torch.equal(dg, df) and torch.equal(ig['cost'], i_f['cost'])
            assert torch.equal(sh.reset_done(), full.reset_done())
        acts = torch.from_numpy(rng.uniform(-1, 1, (20, N, A)).astype(np.float32)).cuda()
        for x, y in zip(sh.rollout(acts), full.rollout(acts)):
            assert torch.equal(x, y)
    # an export that does not fit is reported, not installed
    tiny = torch.zeros(n_shards, sh.shard_capacity(n_shards), sh.n_layout_objects, 2, device='cuda')
    big = torch.full((n_shards,), sh.shard_capacity(n_shards) + 1, dtype=torch.int32, device='cuda')
    with pytest.raises(ResamplingError, match="exported more"):
        sh.reset_from_shards(tiny, big)
    full.close(); sh.close()


@pytest.mark.parametrize("robot,W", [("point", 2), ("point", 8), ("ant", 3)])
def test_piggybacked_shard_sampler_equals_the_unsharded_engine(torch_cuda, oracle, robot, W):
    """The default multi-GPU path on one GPU: W shard engines ("ranks") each sample THEIR candidates of the reset after
    next beside the dynamics pass (sample_shard_ahead), the export blocks ride in the tail of the tape shards, the
    "all-gather" (a concatenation here) delivers them one epoch later, and install_shards assembles the pool the reset
    after that takes like a prefetch hit.  Every reset observation, layout_size, pool rows, the expanded rows of every
    tape on every rank, reset_done draws included, equal ONE unsharded engine's (which samples all candidates itself)
    and -- first reset -- the checker's.  No reset after the first samples inline."""
    torch = torch_cuda
    from guardx_amd import Engine
    extra = {"point": {}, "ant": ANT}[robot]
    N, T, M, EPOCHS = 64, 40, 120_000, 6
    kw = dict(seed=6, num_steps=T, goal_size=2.6, **extra)
    full, O = _engines(task_config(N * W, **kw), oracle, n_candidates=M)
    ranks = [Engine(task_config(N, **kw), n_candidates=M, shard=(r, W)) for r in range(W)]
    A = full.action_space.shape[0]
    rng = np.random.default_rng(3)
    side = torch.cuda.Stream()
    n_tape = sum(ranks[0].tape_floats(T))
    off = n_tape + (-n_tape) % 4
    cap = n_blk = None
    bufs, pend = [], None            # per epoch: the W "send" buffers; (buffers, tokens, tickets, reference rows)
    for ep in range(EPOCHS):
        o_full = full.reset()
        if ep == 0:
            np.testing.assert_array_equal(o_full.cpu().numpy(), O.reset())
        for r, e in enumerate(ranks):
            assert torch.equal(e.reset(), o_full[r * N:(r + 1) * N]), (ep, r)
            assert e.layout_size == full.layout_size
            np.testing.assert_array_equal(e.get_pool(48), full.get_pool(48))
        if cap is None:
            cap = 2 * (-(-full.layout_size // W)) + 256
            n_blk = ranks[0].shard_block_floats(cap)
            for e in ranks:
                e.set_layout_source('shards')
        acts = torch.from_numpy(rng.uniform(-1, 1, (T, N * W, A)).astype(np.float32)).cuda()
        *_, pk = full.rollout(acts, packed=True)
        cur = [torch.zeros(off + n_blk, device='cuda') for _ in range(W)]
        tokens, tickets = [], []
        for r, e in enumerate(ranks):
            tickets.append(e.sample_shard_ahead(r, W, cur[r][off:], cap))
            _, tok = e.rollout_tape(acts[:, r * N:(r + 1) * N].contiguous(), out=cur[r][:n_tape])
            e.shard_join()
            tokens.append(tok)
        assert len(set(tickets)) == 1                        # the same call sequence on every rank
        if pend is not None:                                 # the previous epoch's collective has "arrived"
            pbuf, ptok, ptick, pref = pend
            gathered = torch.cat(pbuf)                       # what all_gather_into_tensor leaves on every rank
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                outs = []
                for by, e in enumerate(ranks):
                    e.install_shards(ptick, gathered[off:], off + n_blk, W, cap)
                    outs.append([e.expand_tape(gathered[s * (off + n_blk):s * (off + n_blk) + n_tape], ptok[s], T)
                                 for s in range(W)])
                assert len(set(ptok)) == 1                   # same pool ring position and generation on every rank
                fused = ranks[ep % W].expand_tapes(gathered, off + n_blk, W, ptok[0], T,
                                                   torch.empty(W, T, N, pref.shape[-1], device='cuda'))
            torch.cuda.current_stream().wait_stream(side)
            for by in range(W):
                for s in range(W):
                    want = pref[:, s * N:(s + 1) * N].contiguous()
                    assert torch.equal(outs[by][s].view(torch.int32), want.view(torch.int32)), (ep, by, s)
            for s in range(W):                               # all shards in ONE launch: the same rows
                assert torch.equal(fused[s].view(torch.int32), outs[0][s].view(torch.int32)), (ep, s)
        pend = (cur, tokens, tickets[0], pk)
    for e in ranks:
        hits, misses, horizon = e.prefetch_stats()
        # reset 0 sampled inline, reset 1 took the pool the engine's own prefetch had started before the switch to
        # 'shards', every later one an installed pool
        assert horizon == T and hits == EPOCHS - 1 and misses == 0, (hits, misses)
        e.check_layouts()
    # a block of another key / a foreign shard is refused by the reset that would take the pool
    e = ranks[0]
    e.reset()                                                # reset EPOCHS: the pool installed in the last epoch
    e.rollout_tape(acts[:, :N].contiguous())                 # T steps: the key of reset EPOCHS + 1, which the last blocks are for
    bad = torch.cat(pend[0]).clone()
    bad[off + 1] = 12345.0                                   # key word of shard 0's block
    e.install_shards(pend[2], bad[off:], off + n_blk, W, cap)
    from guardx_amd import ResamplingError
    with pytest.raises(ResamplingError):
        e.reset()
    from guardx_amd._native import GxError, GX_ERR_STATE
    with pytest.raises(GxError) as ei:
        e.install_shards(pend[2] + 100, bad[off:], off + n_blk, W, cap)
    assert ei.value.status == GX_ERR_STATE
    full.close()
    for e in ranks:
        e.close()


@pytest.mark.parametrize("expand", ["all", "local"])
def test_tape_handoff_with_piggybacked_sampler_pipeline(torch_cuda, expand):
    """guardx_amd.dist.TapeHandoff as bench.py drives it at N > 1, in a world of one (the whole candidate list is this
    rank's shard; no collective): the blocks sampled two resets ahead are installed one epoch later, a drain() in the
    middle (bench.py drains after its warm-up epochs) defers the install it cannot make yet, and every reset from the
    third on is a hit.  Rows, observations and pool sizes equal a twin engine's that samples for itself."""
    torch = torch_cuda
    from guardx_amd import Engine
    from guardx_amd.dist import TapeHandoff
    N, T, M = 300, 50, 150_000
    cfg = task_config(N, seed=8, num_steps=T, goal_size=2.9)
    a, b = Engine(cfg, n_candidates=M), Engine(cfg, n_candidates=M)
    assert torch.equal(a.reset(), b.reset())
    h = TapeHandoff(a, T, sharded_sampler=True, expand=expand)
    assert h.sharded and h.cap < M
    rng = np.random.default_rng(12)
    prev = None
    for ep in range(8):
        acts = torch.from_numpy(rng.uniform(-1, 1, (T, N, 2)).astype(np.float32)).cuda()
        if ep:
            assert torch.equal(a.reset(check=False), b.reset(check=False)), ep
        h.step(acts)
        *_, pk = b.rollout(acts, packed=True)
        if ep >= 1:
            torch.cuda.current_stream().wait_stream(h.stream)
            assert torch.equal(h.rollout[0], prev)
        prev = pk
        if ep == 3:
            h.drain()
            assert torch.equal(h.rollout[0], prev) and h.deferred is not None
    h.drain()
    assert torch.equal(h.rollout[0], prev)
    if expand == "local":
        assert torch.equal(h.expand_rank(0), prev)
    a.check_layouts(); b.check_layouts()
    hits, misses, _ = a.prefetch_stats()
    # reset 1 took the engine's own prefetch (started before the switch to 'shards'), reset 2 sampled inline (no pool
    # announced: neither hit nor miss), resets 3..7 took installed pools
    assert hits == 6 and misses == 0 and h.blocks_installed == 6, (hits, misses, h.blocks_installed)
    h.close()
    assert torch.equal(a.reset(), b.reset())                 # back on its own sampler, still in step
    assert torch.equal(a.reset(), b.reset())
    a.close(); b.close()


# ---------------------------------------------------------------------------
# streams and devices (SURVEY 8b "Threading / streams"): every gx_* call runs on torch's CURRENT stream of the engine's
# device; the prefetch sampler runs on the engine's own side stream, ordered by events
# ---------------------------------------------------------------------------
def test_step_reset_done_rollout_on_a_side_stream(torch_cuda, oracle):
    """reset / step / reset_done / rollout issued under torch.cuda.stream(side) -- with the actions produced on that
    stream and the results consumed on it, no device-wide synchronisation in between -- equal the checker's, across three
    epochs with the pool prefetch hitting (the sampler on the engine's own stream is ordered behind / in front of work
    on a stream that is NOT the default one)."""
    torch = torch_cuda
    N, T = 256, 40
    E, O = _engines(task_config(N, seed=31, num_steps=T, goal_size=2.7), oracle, n_candidates=60000)
    side = torch.cuda.Stream()
    rng = np.random.default_rng(7)
    with torch.cuda.stream(side):
        for ep in range(3):
            np.testing.assert_array_equal(E.reset().cpu().numpy(), O.reset())
            assert E.layout_size == O.layout_size
            steps = []
            for t in range(12):
                a = rng.uniform(-1, 1, (N, 2)).astype(np.float32)
                ad = torch.from_numpy(a).to('cuda', non_blocking=True) * 1.0       # produced on `side`
                steps.append((E.step(ad), E.reset_done(), a))
            for out, rd, a in steps:                                               # consumed on `side`, in order
                _cmp_step(out, O.step(a))
                np.testing.assert_array_equal(rd.cpu().numpy(), O.reset_done())
            acts = rng.uniform(-1, 1, (T - 12, N, 2)).astype(np.float32)
            og, rg, cg, dg = E.rollout(torch.from_numpy(acts).to('cuda', non_blocking=True))
            for t in range(T - 12):
                oo, ro, do, io = O.step(acts[t])
                o2 = O.reset_done()
                np.testing.assert_array_equal(og[t].cpu().numpy(), o2)
                np.testing.assert_array_equal(dg[t].cpu().numpy(), do)
                np.testing.assert_array_equal(rg[t].cpu().numpy(), ro)
                np.testing.assert_array_equal(cg[t].cpu().numpy(), io['cost'])
    hits, misses, _ = E.prefetch_stats()
    assert hits == 2 and misses == 0
    E.close()


def test_key_staging_ring_wraps_without_reusing_a_slot_in_flight(torch_cuda, oracle, monkeypatch):
    """The per-step reset_done keys of a fused rollout are staged in a ring of pinned slots that the kernels read while
    they run (gx_api.hip:stage_rollout_keys): a lap of the ring ends with the host waiting for the streams it handed
    slots to.  With a ring of THREE slots (GX_KEY_RING, read when the ring is allocated) and rollouts issued back to back
    on two streams without any synchronisation, eleven rollouts -- three and a half laps, every kind of launch that takes
    a slot -- still draw the checker's layouts at every reset_done (goal_size 2.9: an env finishes every few steps), and a
    longer rollout than the ring was sized for regrows it."""
    torch = torch_cuda
    monkeypatch.setenv("GX_KEY_RING", "3")
    N, T = 192, 24
    cfg = task_config(N, seed=77, num_steps=1000, goal_size=2.9)
    E, O = _engines(cfg, oracle, n_candidates=60000)
    np.testing.assert_array_equal(E.reset().cpu().numpy(), O.reset())
    rng = np.random.default_rng(3)
    side = torch.cuda.Stream()
    pending = []

    def check():
        for (og, rg, cg, dg), acts in pending:
            for t in range(acts.shape[0]):
                oo, ro, do, io = O.step(acts[t])
                o2 = O.reset_done()
                np.testing.assert_array_equal(og[t].cpu().numpy(), o2)
                np.testing.assert_array_equal(dg[t].cpu().numpy(), do)
                np.testing.assert_array_equal(rg[t].cpu().numpy(), ro)
        pending.clear()

    for k in range(11):
        acts = rng.uniform(-1, 1, (T, N, 2)).astype(np.float32)
        if k % 3 == 2:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                pending.append((E.rollout(torch.from_numpy(acts).to('cuda', non_blocking=True)), acts))
            torch.cuda.current_stream().wait_stream(side)
        else:
            pending.append((E.rollout(torch.from_numpy(acts).to('cuda', non_blocking=True)), acts))
    check()
    acts = rng.uniform(-1, 1, (300, N, 2)).astype(np.float32)         # longer than a slot (256 keys): the ring regrows
    pending.append((E.rollout(torch.from_numpy(acts).cuda()), acts))
    acts = rng.uniform(-1, 1, (T, N, 2)).astype(np.float32)
    pending.append((E.rollout(torch.from_numpy(acts).cuda()), acts))
    check()
    E.close()


def test_two_engines_interleaved_on_two_streams(torch_cuda, oracle):
    """Two engines (different seeds) in one process, each driven on its own stream, calls interleaved on the host, three
    epochs with prefetch hits: per engine the pool ring (in use | being prefetched | referenced by a tape in flight), its
    side stream and the pool_ready / pool_free / expand events are its own -- results equal each engine's checker, the
    tape of epoch k expanded during epoch k + 1 on the OTHER engine's stream included."""
    torch = torch_cuda
    from guardx_amd import Engine
    N, T, M = 192, 30, 50000
    cfgs = [task_config(N, seed=41, num_steps=T, goal_size=2.6), task_config(N, seed=42, num_steps=T, goal_size=2.6, **SWIMMER)]
    pairs = [_engines(c, oracle, n_candidates=M) for c in cfgs]
    twins = [Engine(c, n_candidates=M) for c in cfgs]            # for the tape expansion reference (packed rows)
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    rng = np.random.default_rng(9)
    pending = [None, None]
    for ep in range(3):
        acts = [torch.from_numpy(rng.uniform(-1, 1, (T, N, 2)).astype(np.float32)).cuda() for _ in range(2)]
        torch.cuda.synchronize()
        obs = [None, None]
        for k in (0, 1):                                          # reset both, then roll both: launches interleave
            with torch.cuda.stream(streams[k]):
                obs[k] = pairs[k][0].reset(check=False)
        outs = [None, None]
        for k in (1, 0):
            with torch.cuda.stream(streams[k]):
                if pending[k] is not None:                        # epoch ep-1's tape, expanded on the OTHER stream
                    with torch.cuda.stream(streams[1 - k]):
                        streams[1 - k].wait_stream(streams[k])
                        sh, tok, want = pending[k]
                        got = pairs[k][0].expand_tape(sh, tok, T)
                        pending[k] = (got, want)
                sh, tok = pairs[k][0].rollout_tape(acts[k])
                outs[k] = (sh, tok)
        torch.cuda.synchronize()
        for k in (0, 1):
            E, O = pairs[k]
            np.testing.assert_array_equal(obs[k].cpu().numpy(), O.reset())
            assert torch.equal(twins[k].reset(check=False), obs[k])
            *_, pk = twins[k].rollout(acts[k], packed=True)
            if pending[k] is not None:
                got, want = pending[k]
                assert torch.equal(got.view(torch.int32), want.view(torch.int32)), (ep, k)
            pending[k] = (outs[k][0], outs[k][1], pk)
            a = acts[k].cpu().numpy()
            for t in range(T):
                _, _, d, _ = O.step(a[t]); O.reset_done()
            np.testing.assert_array_equal(pk[-1, :, -1].cpu().numpy(), d)         # last done column of the packed rows
            st, so = E.get_state(), O.get_state()
            assert_state_equal(st, so)
    for k in (0, 1):
        E = pairs[k][0]
        E.check_layouts()
        assert E.prefetch_stats()[:2] == (2, 0)
        E.close(); twins[k].close()


def test_engine_on_a_second_device_while_the_first_is_current(torch_cuda, oracle):
    """Engine(device_id=1) created and driven while cuda:0 is the current device (every gx_* call switches to the handle's
    device and back): results equal the checker's; tensors live on cuda:1.  Skipped on a one-GPU box."""
    torch = torch_cuda
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two HIP devices")
    from guardx_amd import Engine
    torch.cuda.set_device(0)
    N = 128
    cfg = task_config(N, seed=5, num_steps=30, goal_size=2.5, device_id=1)
    E = Engine(cfg, n_candidates=30000)
    O = oracle.OracleEngine(cfg, n_candidates=30000)
    assert torch.cuda.current_device() == 0
    o = E.reset()
    assert o.device == torch.device('cuda', 1) and torch.cuda.current_device() == 0
    np.testing.assert_array_equal(o.cpu().numpy(), O.reset())
    rng = np.random.default_rng(1)
    for t in range(35):
        a = rng.uniform(-1, 1, (N, 2)).astype(np.float32)
        _cmp_step(E.step(torch.from_numpy(a).to('cuda:1')), O.step(a))
        np.testing.assert_array_equal(E.reset_done().cpu().numpy(), O.reset_done())
    assert torch.cuda.current_device() == 0
    E.close()


def test_config4_default_path_eight_ranks_full_size(torch_cuda, oracle):
    """BASELINE config 4 at full size through the DEFAULT multi-GPU path, on one GPU: eight Engine(env_num=2000,
    shard=(r, 8)), each behind its own guardx_amd.dist.TapeHandoff (sharded sampler over 1e6 candidates -- 125 000 per
    rank --, tape hand-off, one launch expanding all eight tapes), with an in-process stand-in for the ONE collective per
    epoch (the eight send buffers concatenated, delivered an epoch later, as all_gather_into_tensor leaves them).  Every
    reset observation, the pool and its size, and the expanded rows of all 16 000 envs on every rank equal ONE
    16 000-env engine that samples all candidates itself; that engine's first reset equals the checker's.  From the
    fourth reset on no rank samples inline."""
    torch = torch_cuda
    from guardx_amd import Engine
    from guardx_amd.dist import TapeHandoff
    N, W, T, M, EPOCHS = 2000, 8, 30, 1_000_000, 6
    kw = dict(seed=5, num_steps=T, goal_size=2.9)
    full, O = _engines(task_config(N * W, **kw), oracle, n_candidates=M)
    ranks = [Engine(task_config(N, **kw), n_candidates=M, shard=(r, W)) for r in range(W)]
    o_full = full.reset()
    np.testing.assert_array_equal(o_full.cpu().numpy(), O.reset())
    for r, e in enumerate(ranks):
        assert torch.equal(e.reset(), o_full[r * N:(r + 1) * N]) and e.layout_size == full.layout_size > N * W
        e.set_prefetch(T)

    class Wire:
        """what the eight ranks' all_gather_into_tensor calls of one epoch leave on every rank"""
        def __init__(self):
            self.bufs = {}
        def put(self, ep, r, buf):
            self.bufs.setdefault(ep, [None] * W)[r] = buf
        def gathered(self, ep):
            return torch.cat(self.bufs.pop(ep))
    wire = Wire()

    class InProcess(TapeHandoff):
        def __init__(self, env, rank):
            super().__init__(env, T, sharded_sampler=True, expand="all", _play=(rank, W))
            self.ep = 0
        def _gather(self, i, buf):
            wire.put(self.ep, self.rank, buf)
            ep, self.ep = self.ep, self.ep + 1
            me = self

            class Work:                     # "wait" = the collective of that epoch is complete: every rank has put
                def wait(self_inner):
                    me.recv[i] = gathered[ep]
            return Work()
    hs = [InProcess(e, r) for r, e in enumerate(ranks)]
    assert len({h.cap for h in hs}) == 1 and hs[0].n % 4 == 0
    gathered = {}
    rng = np.random.default_rng(8)
    prev = None
    sizes = [full.layout_size]
    for ep in range(EPOCHS):
        acts = torch.from_numpy(rng.uniform(-1, 1, (T, N * W, 2)).astype(np.float32)).cuda()
        if ep:
            o_full = full.reset()
            sizes.append(full.layout_size)
            for r, e in enumerate(ranks):
                assert torch.equal(e.reset(check=False), o_full[r * N:(r + 1) * N]), (ep, r)
        *_, pk = full.rollout(acts, packed=True)
        # every rank steps: its tape + the block sampled for a later reset go "on the wire", and -- as in the real
        # pipeline -- it expands / installs the PREVIOUS epoch's gathered buffer, whose collective is complete by now
        for r, h in enumerate(hs):
            h.step(acts[:, r * N:(r + 1) * N].contiguous())
        gathered[ep] = wire.gathered(ep)
        if prev is not None:
            for r, h in enumerate(hs):
                torch.cuda.current_stream().wait_stream(h.stream)
                for s in range(W):
                    assert torch.equal(h.rollout[s].view(torch.int32),
                                       prev[:, s * N:(s + 1) * N].contiguous().view(torch.int32)), (ep, r, s)
        prev = pk
    torch.cuda.synchronize()
    for r, e in enumerate(ranks):
        hits, misses, _ = e.prefetch_stats()
        # reset 1: the engine's own prefetch (started before the hand-off took over); reset 2: inline (no pool announced);
        # resets 3 .. EPOCHS - 1: installed pools
        assert hits == EPOCHS - 2 and misses == 0, (r, hits, misses)
        assert e.check_layouts() == min(sizes)                  # the smallest pool of all its resets = the big engine's
        np.testing.assert_array_equal(e.get_pool(32), full.get_pool(32))
    for h in hs:
        h.close()
    full.close()
    for e in ranks:
        e.close()


@pytest.mark.parametrize("robot,N,T", [("point", 37, 9), ("swimmer", 51, 7), ("ant", 13, 5), ("walker", 9, 3)])
def test_tape_rows_odd_sizes(torch_cuda, robot, N, T):
    """Tape rows are 9 / 13 / 32 / 38 floats (4-byte aligned, not 16): the tape is rounded up to a multiple of 4 floats so
    that the layout snapshot behind it stays 16-byte aligned.  rollout_tape + expand_tape
    (own stream, one and three "ranks" in one launch) equal rollout(packed=True) bit for bit at such sizes."""
    torch = torch_cuda
    from guardx_amd import Engine
    extra = {"point": {}, "swimmer": SWIMMER, "ant": ANT, "walker": WALKER}[robot]
    cfg = task_config(N, seed=11, num_steps=1, goal_size=2.7, **extra)      # (the timeout fires on step num_steps + 2)
    a, b = Engine(cfg, n_candidates=30000), Engine(cfg, n_candidates=30000)
    A = a.action_space.shape[0]
    assert (N * T) % 2 == 1
    tape, lay, ent = a.tape_floats(T)
    width = {"point": 9, "swimmer": 13, "ant": 32, "walker": 38}[robot]
    assert tape == (N * T * width + 3) // 4 * 4 and tape % 4 == 0
    rng = np.random.default_rng(3)
    for ep in range(2):
        assert torch.equal(a.reset(), b.reset())
        acts = torch.from_numpy(rng.uniform(-1, 1, (T, N, A)).astype(np.float32)).cuda()
        sh, tok = a.rollout_tape(acts)
        *_, pk = b.rollout(acts, packed=True)
        assert pk[..., -1].sum().item() > 0                      # timeouts: reset_done rows in the tape
        assert torch.equal(a.expand_tape(sh, tok, T).view(torch.int32), pk.view(torch.int32))
        n = sh.numel()
        three = torch.cat([sh, sh, sh])
        out = a.expand_tapes(three, n, 3, tok, T, torch.empty(3, T, N, pk.shape[-1], device='cuda'))
        for s in range(3):
            assert torch.equal(out[s].view(torch.int32), pk.view(torch.int32)), s
    a.close(); b.close()


@pytest.mark.parametrize("path", ["thread", "group", "split"])
def test_lidar_angles_at_and_below_the_fast_division_bound(torch_cuda, oracle, path):
    """lidar_terms divides by bin_size without a division (div_bin16, proved exact for 2^-100 <= |x| <= 2 pi by
    tests/test_div_bin_size.py) and keeps the true division for any wave that holds an angle of exactly 0 or below the
    bound.  Objects dead ahead of a robot with heading 0 (angle +0), a hair to the left (angles 1e-38, 1e-33 -- denormal
    and tiny quotients) and just above the bound (2^-99) sit in some envs of a wave, ordinary layouts in the others:
    observations, cost and reward equal the checker's bit for bit on every kernel family."""
    torch = torch_cuda
    N = 200
    E, O = _engines(task_config(N, seed=3), oracle, n_candidates=30000, path=path)
    E.reset(); O.reset()
    rng = np.random.default_rng(17)
    s = random_state(N, 8, rng)
    special = np.arange(0, N, 3)
    s['qpos'][special] = np.array([0.5, 0.0, 0.0], np.float32)          # heading 0: cos = 1, sin = 0 exactly
    s['qvel'][special] = 0.0
    s['pose0'][special] = np.array([0.5, 0.0, 1.0, 0.0], np.float32)
    for n, e in enumerate(special):
        dy = [0.0, 1e-38, 1e-33, 2.0 ** -99, 2.0 ** -101][n % 5]
        s['objs'][e, 0] = (1.5, dy)                                     # goal
        s['objs'][e, 1] = (2.5, dy)                                     # hazards 0, 1 (one in front of the other)
        s['objs'][e, 2] = (0.75, 0.0)
    E.set_state(s); O.set_state(s)
    act = np.zeros((N, 2), np.float32)
    _cmp_step(E.step(torch.from_numpy(act).cuda()), O.step(act))
    E.set_state(s); O.set_state(s)
    acts = np.zeros((8, N, 2), np.float32)
    obs, rew, cost, done = E.rollout(torch.from_numpy(acts).cuda())
    for t in range(8):
        o, r, d, info = O.step(acts[t])
        np.testing.assert_array_equal(obs[t].cpu().numpy(), O.reset_done())
        np.testing.assert_array_equal(rew[t].cpu().numpy(), r)
        np.testing.assert_array_equal(cost[t].cpu().numpy(), info['cost'])
    assert_state_equal(E.get_state(), O.get_state())

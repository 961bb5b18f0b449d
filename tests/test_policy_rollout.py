"""Closed-loop fused rollout with the policy evaluated inside the kernel:
`ac.step(o)` of MLPActorCritic((64,64), tanh) -- trpo_core.py:110-173."""
import numpy as np
import pytest

from helpers import task_config, assert_state_equal, SWIMMER, ANT, WALKER


def _torch_ac(D, A, seed=0, hidden=64):
    import torch
    torch.manual_seed(seed)
    h = hidden
    mk = lambda out: torch.nn.Sequential(torch.nn.Linear(D, h), torch.nn.Tanh(), torch.nn.Linear(h, h),   # noqa: E731
                                         torch.nn.Tanh(), torch.nn.Linear(h, out), torch.nn.Identity())
    mu_net, v_net = mk(A), mk(1)
    for net in (mu_net, v_net):           # livelier than the default init so tanh is exercised
        for m in net:
            if isinstance(m, torch.nn.Linear):
                torch.nn.init.normal_(m.weight, std=0.5 * (64 / h) ** 0.5 if m.in_features == h else 0.5)
                torch.nn.init.normal_(m.bias, std=0.3)
    log_std = torch.tensor([-0.5, -0.3, -0.7, -0.1, -0.9, -0.4, -0.6, -0.2, -0.8, -0.35][:A])
    return mu_net, v_net, log_std


def test_log_tanh_accuracy(oracle):
    x = np.concatenate([np.arange(1, 2 ** 24 + 1, 4099) * 2.0 ** -24, np.linspace(0.5, 30, 10001)]).astype(np.float32)
    lg, _ = oracle.math_probe2(x)
    t = np.log(x.astype(np.float64))
    assert np.abs(lg - t).max() < 5e-7 and (np.abs(lg - t) / np.maximum(np.abs(t), 1e-3)).max() < 3e-7
    y = np.linspace(-12, 12, 200001).astype(np.float32)
    _, th = oracle.math_probe2(y)
    assert np.abs(th - np.tanh(y.astype(np.float64))).max() < 2.5e-7
    assert th[0] == -1 and th[-1] == 1 and oracle.math_probe2(np.array([0.0], np.float32))[1][0] == 0


@pytest.mark.parametrize("hidden", [64, 128, 256])
def test_oracle_policy_matches_torch_and_noise_is_standard_normal(oracle, hidden):
    """oracle MLP / logp against torch (fp32, different summation order -> 1e-5), noise statistics -- at the reference's
    default width and at the wider networks its command line offers (trpo.py:606-607 --hid)"""
    import torch
    from guardx_amd import Engine
    N, T = 256, 40
    cfg = task_config(N, seed=1, num_steps=T)
    O = oracle.OracleEngine(cfg, n_candidates=30000)
    o0 = O.reset()
    mu_net, v_net, log_std = _torch_ac(43, 2, hidden=hidden)
    params = Engine.pack_actor_critic(mu_net=mu_net, v_net=v_net, log_std=log_std).numpy()
    assert params.size == Engine._policy_floats(43, 2, hidden)
    out = O.rollout_policy(params, T, o0, noise_seed=(7, 9), hidden=hidden)
    obs = torch.from_numpy(out['obs'])
    with torch.no_grad():
        mu_t, v_t = mu_net(obs), v_net(obs).squeeze(-1)
        pi = torch.distributions.Normal(mu_t, torch.exp(log_std))
        logp_t = pi.log_prob(torch.from_numpy(out['act'])).sum(-1)
    np.testing.assert_allclose(out['mu'], mu_t.numpy(), rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(out['val'], v_t.numpy(), rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(out['logp'], logp_t.numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(out['logstd'], log_std.numpy(), atol=1e-6)
    z = (out['act'] - out['mu']) / np.exp(log_std.numpy())
    assert abs(z.mean()) < 0.02 and abs(z.std() - 1) < 0.02 and abs(np.corrcoef(z[..., 0].ravel(), z[..., 1].ravel())[0, 1]) < 0.03
    assert out['done'].sum() >= 0 and np.isfinite(out['obs']).all()


@pytest.mark.gpu
def test_device_log_tanh_bitexact(oracle):
    import ctypes as C
    import torch
    from guardx_amd import _native
    lib = _native.load()
    x = np.concatenate([np.arange(1, 2 ** 24 + 1, 17) * 2.0 ** -24, np.linspace(-20, 20, 400001), [np.nan, 0.0, -0.0]]
                       ).astype(np.float32)
    xd = torch.from_numpy(x).cuda()
    lg, th = torch.empty_like(xd), torch.empty_like(xd)
    _native.check(lib.gx_math_probe2(x.size, xd.data_ptr(), lg.data_ptr(), th.data_ptr(),
                                     C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    lo, to = oracle.math_probe2(x)
    pos = x > 0
    np.testing.assert_array_equal(lg.cpu().numpy()[pos], lo[pos])
    np.testing.assert_array_equal(th.cpu().numpy(), to)


@pytest.mark.gpu
@pytest.mark.parametrize("impl", ["valu", "mfma"])
@pytest.mark.parametrize("robot", ["point", "swimmer", "ant", "walker"])
def test_policy_rollout_parity(oracle, robot, impl):
    import torch
    from guardx_amd import Engine
    N, T = 203, 50
    extra = {"point": {}, "swimmer": SWIMMER, "ant": ANT, "walker": WALKER}[robot]
    cfg = task_config(N, seed=3, num_steps=30, goal_size=0.9, **extra)
    E = Engine(cfg, n_candidates=40000)
    E.set_policy_impl({"valu": 1, "mfma": 2}[impl])
    O = oracle.OracleEngine(cfg, n_candidates=40000)
    og, oo = E.reset(), O.reset()
    np.testing.assert_array_equal(og.cpu().numpy(), oo)
    D, A = E.obs_flat_size, E.action_space.shape[0]
    mu_net, v_net, log_std = _torch_ac(D, A, seed=5)
    params = Engine.pack_actor_critic(mu_net=mu_net, v_net=v_net, log_std=log_std)
    g = E.rollout_policy(params.cuda(), T, noise_seed=(11, 13))
    o = O.rollout_policy(params.numpy(), T, oo, noise_seed=(11, 13))
    assert o['done'].sum() > 0
    for k in ('obs', 'act', 'mu', 'logp', 'val', 'rew', 'cost', 'done', 'obs_last', 'val_last', 'logstd'):
        np.testing.assert_array_equal(g[k].cpu().numpy(), o[k], err_msg=k)
    assert_state_equal(E.get_state(), O.get_state())
    # a second call continues the noise stream (t0 advances) and the env
    g2 = E.rollout_policy(params.cuda(), 7, noise_seed=(11, 13))
    o2 = O.rollout_policy(params.numpy(), 7, o['obs_last'], noise_seed=(11, 13), t0=T)
    for k in ('obs', 'act', 'logp', 'val', 'rew', 'done'):
        np.testing.assert_array_equal(g2[k].cpu().numpy(), o2[k], err_msg=k)
    # and the ordinary API still lines up afterwards
    act = np.zeros((N, A), np.float32)
    og3, _, dg3, _ = E.step(torch.from_numpy(act).cuda())
    oo3, _, do3, _ = O.step(act)
    np.testing.assert_array_equal(og3.cpu().numpy(), oo3)
    np.testing.assert_array_equal(dg3.cpu().numpy(), do3)
    # ... including reset_done() requested right before the next policy rollout: the re-initialisation the
    # step speculated is installed by the policy kernel itself on load (gx_step_rd / gx_reset_done_commit)
    rd_g, rd_o = E.reset_done(), O.reset_done()
    np.testing.assert_array_equal(rd_g.cpu().numpy(), rd_o)
    g3 = E.rollout_policy(params.cuda(), 5, obs0=rd_g, noise_seed=(11, 13))
    o3 = O.rollout_policy(params.numpy(), 5, rd_o, noise_seed=(11, 13), t0=T + 7)
    for k in ('obs', 'act', 'rew', 'done', 'obs_last'):
        np.testing.assert_array_equal(g3[k].cpu().numpy(), o3[k], err_msg=k)
    assert_state_equal(E.get_state(), O.get_state())


@pytest.mark.gpu
@pytest.mark.parametrize("robot,hidden,impl", [("point", 128, "mfma"), ("point", 128, "valu"), ("point", 128, "stepwise"),
                                               ("swimmer", 128, "mfma"), ("point", 256, "mfma"), ("point", 256, "stepwise"),
                                               ("swimmer", 192, "mfma"), ("point", 192, "mfma"), ("swimmer", 256, "mfma"),
                                               ("swimmer", 192, "stepwise"), ("ant", 128, "mfma"), ("walker", 256, "valu"),
                                               ("walker", 256, "mfma"), ("point", 64, "mfma"), ("ant", 64, "mfma"),
                                               ("point-narrow", 128, "mfma"), ("point-narrow", 256, "mfma")])
def test_policy_rollout_other_widths_parity(oracle, robot, hidden, impl):
    """hidden_sizes (h, h) beyond 64 (trpo.py:606-607 --hid).  On the light robots ("mfma" = the default) ONE launch:
    h = 128 with the hidden-layer weights resident in registers (group_rollout_kernel<.., 3>), h = 192 / 256 with the
    weights streamed from their L2-resident transposed copy (group_rollout_kernel<.., 192 / 256>, gx_policy.h); everything
    else (Ant, Walker, other observation widths): the step-wise form (two launches per control step, gx_policy_step.hip; hidden layers as v_mfma_f32_16x16x4_f32
    tiles -- "stepwise" = gx_set_policy_impl(2) forces it at 128 -- or as fmaf chains with gx_set_policy_impl(1)).  Each
    equals the checker bit for bit -- every output, the state afterwards, a second call that continues the noise stream;
    at h = 64 (gx_set_policy_impl(3)) the step-wise form also equals the FUSED kernel's outputs."""
    import torch
    from guardx_amd import Engine
    N, T = 203, 40
    # "point-narrow": an observation width other than the default task's (8 lidar bins: D = 27) -- the one-launch forms
    # of the wide networks have the first layer's k-steps compiled in, so this takes the step-wise form
    extra = {"point": {}, "point-narrow": {"lidar_num_bins": 8}, "swimmer": SWIMMER, "ant": ANT, "walker": WALKER}[robot]
    cfg = task_config(N, seed=3, num_steps=25, goal_size=0.9, **extra)
    E = Engine(cfg, n_candidates=40000)
    if hidden == 64:
        E.set_policy_impl(3)
    elif impl == "valu":
        E.set_policy_impl(1)
    elif impl == "stepwise":
        E.set_policy_impl(2)
    O = oracle.OracleEngine(cfg, n_candidates=40000)
    og, oo = E.reset(), O.reset()
    np.testing.assert_array_equal(og.cpu().numpy(), oo)
    D, A = E.obs_flat_size, E.action_space.shape[0]
    mu_net, v_net, log_std = _torch_ac(D, A, seed=5, hidden=hidden)
    params = Engine.pack_actor_critic(mu_net=mu_net, v_net=v_net, log_std=log_std)
    g = E.rollout_policy(params.cuda(), T, noise_seed=(11, 13))
    o = O.rollout_policy(params.numpy(), T, oo, noise_seed=(11, 13), hidden=hidden)
    assert o['done'].sum() > 0
    keys = ('obs', 'act', 'mu', 'logp', 'val', 'rew', 'cost', 'done', 'obs_last', 'val_last', 'logstd')
    for k in keys:
        np.testing.assert_array_equal(g[k].cpu().numpy(), o[k], err_msg=k)
    assert_state_equal(E.get_state(), O.get_state())
    g2 = E.rollout_policy(params.cuda(), 7, noise_seed=(11, 13))
    o2 = O.rollout_policy(params.numpy(), 7, o['obs_last'], noise_seed=(11, 13), t0=T, hidden=hidden)
    for k in keys:
        np.testing.assert_array_equal(g2[k].cpu().numpy(), o2[k], err_msg=k)
    # the ordinary API lines up afterwards, reset_done speculated by step() included
    act = np.zeros((N, A), np.float32)
    og3, _, dg3, _ = E.step(torch.from_numpy(act).cuda())
    oo3, _, do3, _ = O.step(act)
    np.testing.assert_array_equal(og3.cpu().numpy(), oo3)
    rd_g, rd_o = E.reset_done(), O.reset_done()
    np.testing.assert_array_equal(rd_g.cpu().numpy(), rd_o)
    g3 = E.rollout_policy(params.cuda(), 5, obs0=rd_g, noise_seed=(11, 13))
    o3 = O.rollout_policy(params.numpy(), 5, rd_o, noise_seed=(11, 13), t0=T + 7, hidden=hidden)
    for k in ('obs', 'act', 'rew', 'done', 'obs_last'):
        np.testing.assert_array_equal(g3[k].cpu().numpy(), o3[k], err_msg=k)
    assert_state_equal(E.get_state(), O.get_state())
    if hidden == 64:                       # the fused kernel on a twin engine: same bits
        F = Engine(cfg, n_candidates=40000)
        F.reset()
        f = F.rollout_policy(params.cuda(), T, noise_seed=(11, 13))
        for k in keys:
            assert torch.equal(f[k], g[k]), k
        F.close()
    E.close()


@pytest.mark.gpu
def test_policy_rollout_width_errors():
    import torch
    from guardx_amd import Engine
    E = Engine(task_config(16, seed=1), n_candidates=20000)
    E.reset()
    with pytest.raises(ValueError, match="expected one of"):
        E.rollout_policy(torch.zeros(1234), 3)
    mk = lambda h1, h2: torch.nn.Sequential(torch.nn.Linear(43, h1), torch.nn.Tanh(), torch.nn.Linear(h1, h2),   # noqa: E731
                                            torch.nn.Tanh(), torch.nn.Linear(h2, 2))
    with pytest.raises(NotImplementedError):
        Engine.pack_actor_critic(mu_net=mk(64, 32), v_net=mk(64, 32), log_std=torch.zeros(2))
    with pytest.raises(NotImplementedError):
        Engine.pack_actor_critic(mu_net=mk(96, 96), v_net=mk(96, 96), log_std=torch.zeros(2))
    E.close()

"""Device rollout buffer + GAE against the numpy/scipy restatement of
TRPOBufferX (oracle/trpo_buffer_np.py)."""
import numpy as np
import pytest

from oracle.trpo_buffer_np import TRPOBufferNP, discount_cumsum


def test_discount_cumsum_closed_form():
    x = np.array([1.0, 2.0, 3.0])
    np.testing.assert_allclose(discount_cumsum(x, 0.5), [1 + 1 + 0.75, 2 + 1.5, 3])


def test_oracle_buffer_paths():
    """GAE of a 2-env, 4-step buffer where env 0 finishes after step 2 (closed form)."""
    B = TRPOBufferNP(2, 4, 3, 2, gamma=0.5, lam=1.0)
    rng = np.random.default_rng(0)
    for t in range(4):
        B.store(rng.normal(size=(2, 3)), rng.normal(size=(2, 2)), np.array([1.0, 2.0]), np.array([0.5, 0.25]),
                np.zeros(2), np.zeros((2, 2)), np.zeros((2, 2)))
        if t == 1:
            B.finish_path(np.array([0.0, 9.0]), np.array([1, 0]))
    B.finish_path(np.zeros(2), np.ones(2))
    # env 0: two paths of length 2, no bootstrap: ret = [1 + .5, 1]
    np.testing.assert_allclose(B.ret_buf[0], [1.5, 1.0, 1.5, 1.0])
    # env 1: one path of length 4: ret = 2 * (1 + .5 + .25 + .125)...
    np.testing.assert_allclose(B.ret_buf[1], [3.75, 3.5, 3.0, 2.0])
    d = B.get()
    assert d['obs'].shape == (8, 3) and abs(d['adv'].reshape(2, 4).mean(1)).max() < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("N,T,p_done", [(7, 13, 0.0), (200, 200, 0.02), (2000, 50, 0.2)])
def test_device_buffer_matches_trpo_buffer(N, T, p_done):
    import torch
    from guardx_amd.rollout_buffer import DeviceRolloutBuffer
    D, A = 43, 2
    rng = np.random.default_rng(N)
    G = DeviceRolloutBuffer(N, T, (D,), (A,), gamma=0.99, lam=0.95, device='cuda')
    O = TRPOBufferNP(N, T, D, A, gamma=0.99, lam=0.95)
    dev = lambda x: torch.from_numpy(np.ascontiguousarray(x, np.float32)).cuda()   # noqa: E731
    for t in range(T):
        obs, act = rng.normal(size=(N, D)).astype(np.float32), rng.normal(size=(N, A)).astype(np.float32)
        rew, val, logp = (rng.normal(size=N).astype(np.float32) for _ in range(3))
        mu, ls = rng.normal(size=(N, A)).astype(np.float32), rng.normal(size=(N, A)).astype(np.float32)
        G.store(dev(obs), dev(act), dev(rew), dev(val), dev(logp), dev(mu), dev(ls))
        O.store(obs, act, rew, val, logp, mu, ls)
        timeout = t + 1 == T
        done = (rng.random(N) < p_done).astype(np.float32)
        if timeout:                                   # trpo.py:506-515: every path ends, no bootstrap
            G.finish_path(torch.zeros(N, device='cuda'), torch.ones(N, device='cuda'))
            O.finish_path(np.zeros(N, np.float32), np.ones(N))
        elif done.any():                              # trpo.py:523-545: bootstrap, zero for the done envs
            v = rng.normal(size=N).astype(np.float32)
            v[done == 1] = 0
            G.finish_path(dev(v), dev(done))
            O.finish_path(v, done)
    if p_done > 0:   # the reference's batched branch (no env finished mid-epoch) leaves path_start_idx at 0
        np.testing.assert_array_equal(G.path_start_idx.cpu().numpy(), O.path_start_idx)
    np.testing.assert_allclose(G.adv_buf.cpu().numpy(), O.adv_buf, rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(G.ret_buf.cpu().numpy(), O.ret_buf, rtol=1e-6, atol=1e-6)
    dg, do = G.get(), O.get()
    for k in ('obs', 'act', 'logp', 'mu', 'logstd', 'ret'):
        np.testing.assert_allclose(dg[k].cpu().numpy(), do[k], rtol=1e-6, atol=1e-6, err_msg=k)
    np.testing.assert_allclose(dg['adv'].cpu().numpy(), do['adv'], rtol=2e-5, atol=2e-5)   # fp32 mean/std order
    assert G.ptr == 0 and int(G.path_start_idx.abs().sum()) == 0


@pytest.mark.gpu
def test_device_cost_buffer_matches_cpo_semantics():
    """CPOBufferX = TRPO buffer + (cost, cost_val) channel through the same GAE; cost advantage
    centred, not scaled (cpo.py:22-175)."""
    import torch
    from guardx_amd.rollout_buffer import DeviceCostRolloutBuffer
    N, T, D, A = 64, 40, 43, 2
    rng = np.random.default_rng(0)
    G = DeviceCostRolloutBuffer(N, T, (D,), (A,), device='cuda')
    Or, Oc = TRPOBufferNP(N, T, D, A), TRPOBufferNP(N, T, D, A)     # reward channel, cost channel
    dev = lambda x: torch.from_numpy(np.ascontiguousarray(x, np.float32)).cuda()   # noqa: E731
    for t in range(T):
        obs, act = rng.normal(size=(N, D)).astype(np.float32), rng.normal(size=(N, A)).astype(np.float32)
        rew, val, logp, cost, cval = (rng.normal(size=N).astype(np.float32) for _ in range(5))
        mu, ls = rng.normal(size=(N, A)).astype(np.float32), rng.normal(size=(N, A)).astype(np.float32)
        G.store(dev(obs), dev(act), dev(rew), dev(val), dev(logp), dev(cost), dev(cval), dev(mu), dev(ls))
        Or.store(obs, act, rew, val, logp, mu, ls)
        Oc.store(obs, act, cost, cval, logp, mu, ls)
        done = (rng.random(N) < 0.1).astype(np.float32)
        if t + 1 == T:
            G.finish_path(torch.zeros(N, device='cuda'), torch.zeros(N, device='cuda'), torch.ones(N, device='cuda'))
            Or.finish_path(np.zeros(N), np.ones(N)); Oc.finish_path(np.zeros(N), np.ones(N))
        elif done.any():
            v, cv = rng.normal(size=N).astype(np.float32), rng.normal(size=N).astype(np.float32)
            v[done == 1] = 0; cv[done == 1] = 0
            G.finish_path(dev(v), dev(cv), dev(done))
            Or.finish_path(v, done); Oc.finish_path(cv, done)
    np.testing.assert_allclose(G.adc_buf.cpu().numpy(), Oc.adv_buf, rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(G.cost_ret_buf.cpu().numpy(), Oc.ret_buf, rtol=1e-6, atol=1e-6)
    adc_raw = Oc.adv_buf.copy()
    d = G.get()
    ro = Or.get()
    np.testing.assert_allclose(d['adv'].cpu().numpy(), ro['adv'], rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(d['adc'].cpu().numpy().reshape(N, T), adc_raw - adc_raw.mean(1, keepdims=True),
                               rtol=2e-5, atol=2e-5)
    assert set(d) == {'obs', 'act', 'ret', 'adv', 'cost_ret', 'adc', 'logp', 'mu', 'logstd'}


@pytest.mark.gpu
def test_gae_rollout_equals_stepwise_buffer():
    """gx_gae_rollout on (T, N) arrays == store() + finish_path() at every done step + closing
    finish_path(), i.e. the TRPO collection loop (trpo.py:466-547)."""
    import torch
    from guardx_amd.rollout_buffer import gae_rollout
    N, T, D, A = 300, 70, 5, 2
    rng = np.random.default_rng(3)
    rew, val = rng.normal(size=(T, N)).astype(np.float32), rng.normal(size=(T, N)).astype(np.float32)
    done = (rng.random((T, N)) < 0.05).astype(np.float32)
    O = TRPOBufferNP(N, T, D, A)
    z = np.zeros
    for t in range(T):
        O.store(z((N, D)), z((N, A)), rew[t], val[t], z(N), z((N, A)), z((N, A)))
        if t + 1 == T:
            O.finish_path(z(N, np.float32), np.ones(N))
        elif done[t].any():
            O.finish_path(z(N, np.float32), done[t])       # v = 0 for the done envs (trpo.py:530-531)
    adv, ret = gae_rollout(torch.from_numpy(rew).cuda(), torch.from_numpy(val).cuda(), torch.from_numpy(done).cuda())
    np.testing.assert_allclose(adv.cpu().numpy().T, O.adv_buf, rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(ret.cpu().numpy().T, O.ret_buf, rtol=1e-6, atol=1e-6)

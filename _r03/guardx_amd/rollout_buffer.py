"""Device-resident rollout buffer with GAE-lambda: the `TRPOBufferX` of
safe_rl_libX/trpo/trpo.py:24-146 with the same attribute names and methods
(`store`, `finish_path`, `get`), but `finish_path` and the advantage normalisation
run as HIP kernels on the env's stream -- no `.cpu()`, no per-env Python/scipy loop
when some environments finish mid-epoch (trpo.py:101-119), no host round trip in `get`
(trpo.py:131-135).  SURVEY.md row f1.
"""
import ctypes as C

import torch

from . import _native


class DeviceRolloutBuffer:
    def __init__(self, env_num, max_ep_len, obs_dim, act_dim, gamma=0.99, lam=0.95, device=None):
        obs_dim = int(obs_dim[0]) if hasattr(obs_dim, '__len__') else int(obs_dim)
        act_dim = int(act_dim[0]) if hasattr(act_dim, '__len__') else int(act_dim)
        self.device = torch.device(device if device is not None else 'cuda')
        z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=self.device)   # noqa: E731
        N, T = int(env_num), int(max_ep_len)
        self.obs_buf, self.act_buf = z(N, T, obs_dim), z(N, T, act_dim)
        self.adv_buf, self.rew_buf, self.ret_buf = z(N, T), z(N, T), z(N, T)
        self.val_buf, self.logp_buf = z(N, T), z(N, T)
        self.mu_buf, self.logstd_buf = z(N, T, act_dim), z(N, T, act_dim)
        self.gamma, self.lam = float(gamma), float(lam)
        self.ptr = 0                                                     # same for every env (trpo.py:54)
        self.path_start_idx = torch.zeros(N, dtype=torch.int32, device=self.device)
        self.max_ep_len, self.env_num = T, N
        self.obs_dim, self.act_dim = obs_dim, act_dim
        self._lib = _native.load()

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    @staticmethod
    def _f32(t, shape):
        t = t.detach()
        if t.dtype != torch.float32 or not t.is_contiguous():
            t = t.to(torch.float32).contiguous()
        assert tuple(t.shape) == shape, (tuple(t.shape), shape)
        return t

    def store(self, obs, act, rew, val, logp, mu, logstd):
        """trpo.py:49-64: one step of every env into column `ptr`."""
        assert self.ptr < self.max_ep_len
        N, D, A = self.env_num, self.obs_dim, self.act_dim
        obs, act = self._f32(obs, (N, D)), self._f32(act, (N, A))
        mu, logstd = self._f32(mu, (N, A)), self._f32(logstd, (N, A))
        rew, val, logp = (self._f32(x.reshape(N), (N,)) for x in (rew, val, logp))
        _native.check(self._lib.gx_buffer_store(
            N, self.max_ep_len, self.ptr, D, A, obs.data_ptr(), act.data_ptr(), rew.data_ptr(),
            val.data_ptr(), logp.data_ptr(), mu.data_ptr(), logstd.data_ptr(),
            self.obs_buf.data_ptr(), self.act_buf.data_ptr(), self.rew_buf.data_ptr(),
            self.val_buf.data_ptr(), self.logp_buf.data_ptr(), self.mu_buf.data_ptr(),
            self.logstd_buf.data_ptr(), self._stream()))
        self.ptr += 1

    def finish_path(self, last_val=None, done=None):
        """trpo.py:66-119.  `done` (env_num,) selects the envs whose current path ends (== 1);
        None / all-ones closes every path.  `last_val` (env_num,) bootstraps the tail."""
        N = self.env_num
        if last_val is None:
            last_val = torch.zeros(N, device=self.device)
        last_val = self._f32(torch.as_tensor(last_val, device=self.device).reshape(N), (N,))
        dptr = None
        if done is not None:
            done = self._f32(torch.as_tensor(done, device=self.device).reshape(N), (N,))
            dptr = done.data_ptr()
        _native.check(self._lib.gx_gae_finish_path(
            N, self.max_ep_len, self.ptr, self.rew_buf.data_ptr(), self.val_buf.data_ptr(),
            last_val.data_ptr(), dptr, self.path_start_idx.data_ptr(), self.gamma, self.lam,
            self.adv_buf.data_ptr(), self.ret_buf.data_ptr(), 1, self._stream()))

    def get(self):
        """trpo.py:121-146: per-env normalised advantages, flattened views of every field."""
        assert self.ptr == self.max_ep_len
        self.ptr = 0
        self.path_start_idx.zero_()
        N, T = self.env_num, self.max_ep_len
        _native.check(self._lib.gx_adv_normalize(N, T, self.adv_buf.data_ptr(), 1, self._stream()))
        return dict(obs=self.obs_buf.view(N * T, -1), act=self.act_buf.view(N * T, -1),
                    ret=self.ret_buf.view(N * T), adv=self.adv_buf.view(N * T),
                    logp=self.logp_buf.view(N * T), mu=self.mu_buf.view(N * T, -1),
                    logstd=self.logstd_buf.view(N * T, -1))


class DeviceCostRolloutBuffer(DeviceRolloutBuffer):
    """`CPOBufferX` (safe_rl_libX/cpo/cpo.py:22-175): the TRPO buffer plus a cost channel
    (`cost_buf`, `cost_val_buf` -> `adc_buf`, `cost_ret_buf`); the cost advantage is centred but
    not scaled in `get()` (cpo.py:158-162)."""

    def __init__(self, env_num, max_ep_len, obs_dim, act_dim, gamma=0.99, lam=0.95, device=None):
        super().__init__(env_num, max_ep_len, obs_dim, act_dim, gamma, lam, device)
        z = lambda: torch.zeros(self.env_num, self.max_ep_len, dtype=torch.float32, device=self.device)  # noqa: E731
        self.cost_buf, self.cost_ret_buf, self.cost_val_buf, self.adc_buf = z(), z(), z(), z()

    def store(self, obs, act, rew, val, logp, cost, cost_val, mu, logstd):   # cpo.py:51-69
        p = self.ptr
        super().store(obs, act, rew, val, logp, mu, logstd)
        self.cost_buf[:, p] = cost.reshape(self.env_num)
        self.cost_val_buf[:, p] = cost_val.reshape(self.env_num)

    def finish_path(self, last_val=None, last_cost_val=None, done=None):     # cpo.py:71-140
        N = self.env_num
        zeros = lambda: torch.zeros(N, device=self.device)   # noqa: E731
        last_val = self._f32(torch.as_tensor(zeros() if last_val is None else last_val, device=self.device).reshape(N), (N,))
        last_cv = self._f32(torch.as_tensor(zeros() if last_cost_val is None else last_cost_val,
                                            device=self.device).reshape(N), (N,))
        dptr = None
        if done is not None:
            done = self._f32(torch.as_tensor(done, device=self.device).reshape(N), (N,))
            dptr = done.data_ptr()
        for rew, val, lv, adv, ret, advance in (
                (self.cost_buf, self.cost_val_buf, last_cv, self.adc_buf, self.cost_ret_buf, 0),
                (self.rew_buf, self.val_buf, last_val, self.adv_buf, self.ret_buf, 1)):
            _native.check(self._lib.gx_gae_finish_path(
                N, self.max_ep_len, self.ptr, rew.data_ptr(), val.data_ptr(), lv.data_ptr(), dptr,
                self.path_start_idx.data_ptr(), self.gamma, self.lam, adv.data_ptr(), ret.data_ptr(),
                advance, self._stream()))

    def get(self):                                                           # cpo.py:142-175
        data = super().get()
        N, T = self.env_num, self.max_ep_len
        _native.check(self._lib.gx_adv_normalize(N, T, self.adc_buf.data_ptr(), 0, self._stream()))
        data['cost_ret'] = self.cost_ret_buf.view(N * T)
        data['adc'] = self.adc_buf.view(N * T)
        return data


def gae_rollout(rew, val, done, last_val=None, gamma=0.99, lam=0.95):
    """GAE-lambda advantages and rewards-to-go for a whole fused rollout (time-major (T, N) tensors from
    Engine.rollout / rollout_policy): equivalent to TRPOBufferX.store + finish_path at every done step +
    the closing finish_path (trpo.py:466-547), in one kernel launch.  Returns (adv, ret), both (T, N)."""
    T, N = rew.shape
    dev = rew.device
    f = lambda x: x.to(torch.float32).contiguous()   # noqa: E731
    rew, val, done = f(rew), f(val), f(done)
    last_val = torch.zeros(N, device=dev) if last_val is None else f(last_val.reshape(N))
    adv, ret = torch.empty_like(rew), torch.empty_like(rew)
    lib = _native.load()
    _native.check(lib.gx_gae_rollout(N, T, rew.data_ptr(), val.data_ptr(), done.data_ptr(), last_val.data_ptr(),
                                     float(gamma), float(lam), adv.data_ptr(), ret.data_ptr(),
                                     C.c_void_p(torch._C._cuda_getCurrentRawStream(dev.index))))
    return adv, ret

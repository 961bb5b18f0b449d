"""CPU restatement of `TRPOBufferX` (safe_rl_libX/trpo/trpo.py:24-146) and of
`discount_cumsum` / `batch_discount_cumsum` (trpo_core.py:42-70) in numpy + scipy.

TEST INFRASTRUCTURE ONLY (checker of guardx_amd.rollout_buffer.DeviceRolloutBuffer).
PARITY UNPINNED like the rest of oracle/: the reference learner imports `gym` and `mpi4py`,
neither of which is installable here; this file follows its source lines.
"""
import numpy as np
import scipy.signal


def discount_cumsum(x, discount):
    """trpo_core.py:42-58"""
    return scipy.signal.lfilter([1], [1, float(-discount)], x[::-1], axis=0)[::-1]


class TRPOBufferNP:
    def __init__(self, env_num, max_ep_len, obs_dim, act_dim, gamma=0.99, lam=0.95):
        f = np.float32
        self.obs_buf = np.zeros((env_num, max_ep_len, obs_dim), f)
        self.act_buf = np.zeros((env_num, max_ep_len, act_dim), f)
        self.adv_buf = np.zeros((env_num, max_ep_len), f)
        self.rew_buf = np.zeros((env_num, max_ep_len), f)
        self.ret_buf = np.zeros((env_num, max_ep_len), f)
        self.val_buf = np.zeros((env_num, max_ep_len), f)
        self.logp_buf = np.zeros((env_num, max_ep_len), f)
        self.mu_buf = np.zeros((env_num, max_ep_len, act_dim), f)
        self.logstd_buf = np.zeros((env_num, max_ep_len, act_dim), f)
        self.gamma, self.lam = gamma, lam
        self.ptr = np.zeros(env_num, dtype=np.int16)
        self.path_start_idx = np.zeros(env_num, dtype=np.int16)
        self.max_ep_len, self.env_num = max_ep_len, env_num

    def store(self, obs, act, rew, val, logp, mu, logstd):      # trpo.py:49-64
        assert len(set(self.ptr)) == 1 and self.ptr[0] < self.max_ep_len
        p = self.ptr[0]
        self.obs_buf[:, p, :] = obs; self.act_buf[:, p, :] = act
        self.rew_buf[:, p] = rew; self.val_buf[:, p] = val; self.logp_buf[:, p] = logp
        self.mu_buf[:, p, :] = mu; self.logstd_buf[:, p, :] = logstd
        self.ptr += 1

    def finish_path(self, last_val, done):                      # trpo.py:66-119
        last_val = np.asarray(last_val, np.float32).reshape(-1)
        if np.all(self.path_start_idx == 0) and np.all(self.ptr == self.max_ep_len):
            lv = last_val[:, None]
            rews = np.hstack((self.rew_buf, lv))
            vals = np.hstack((self.val_buf, lv))
            deltas = rews[:, :-1] + np.float32(self.gamma) * vals[:, 1:] - vals[:, :-1]
            self.adv_buf = np.asarray([discount_cumsum(r, self.gamma * self.lam) for r in deltas]).astype(np.float32)
            self.ret_buf = np.asarray([discount_cumsum(r, self.gamma) for r in rews])[:, :-1].astype(np.float32)
        else:
            for e in np.where(np.asarray(done) == 1)[0]:
                sl = slice(self.path_start_idx[e], self.ptr[e])
                rews = np.append(self.rew_buf[e, sl], last_val[e])
                vals = np.append(self.val_buf[e, sl], last_val[e])
                deltas = rews[:-1] + self.gamma * vals[1:] - vals[:-1]
                self.adv_buf[e, sl] = discount_cumsum(deltas, self.gamma * self.lam).astype(np.float32)
                self.ret_buf[e, sl] = discount_cumsum(rews, self.gamma)[:-1].astype(np.float32)
                self.path_start_idx[e] = self.ptr[e]

    def get(self):                                              # trpo.py:121-146
        assert len(set(self.ptr)) == 1 and self.ptr[0] == self.max_ep_len
        self.ptr[:] = 0
        self.path_start_idx[:] = 0

        def norm(x):                                            # mpi_statistics_scalar, 1 process
            x = np.array(x, dtype=np.float32)
            mean = np.sum(x) / len(x)
            std = np.sqrt(np.sum((x - mean) ** 2) / len(x))
            return (x - mean) / std
        self.adv_buf = np.asarray([norm(r) for r in self.adv_buf])
        N, T = self.env_num, self.max_ep_len
        return dict(obs=self.obs_buf.reshape(N * T, -1), act=self.act_buf.reshape(N * T, -1),
                    ret=self.ret_buf.reshape(N * T), adv=self.adv_buf.reshape(N * T),
                    logp=self.logp_buf.reshape(N * T), mu=self.mu_buf.reshape(N * T, -1),
                    logstd=self.logstd_buf.reshape(N * T, -1))

#!/usr/bin/env python3
"""End-to-end use of the fused path: PPO-clip on Goal_Point_8Hazards where the whole collection phase
of an epoch -- ac.step, env.step, reset_done for 2000 envs x 200 steps -- is ONE kernel launch
(Engine.rollout_policy), GAE is one more (gae_rollout), and torch only does the policy update.

The learners of safe_rl_libX drive `env.step()` themselves and stay drop-in on `guardx_amd.Engine`;
this script shows what a learner gains when it hands the collection loop to the device instead.
It also serves as a sanity check of the environment semantics: the return must go up.

    python examples/train_ppo_fused.py [--epochs 30] [--env-num 2000] [--hid 64]

--hid: hidden width of actor and critic (trpo.py:606; 64, 128, 192 or 256 -- at 64 the collection phase is one launch,
wider networks take two launches per control step, guardx_amd/csrc/gx_policy_step.hip).
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402

from guardx_amd import Engine, configuration  # noqa: E402
from guardx_amd.rollout_buffer import gae_rollout  # noqa: E402


def mlp(sizes):          # trpo_core.py:30-35
    layers = []
    for j in range(len(sizes) - 1):
        layers += [nn.Linear(sizes[j], sizes[j + 1]), nn.Tanh() if j < len(sizes) - 2 else nn.Identity()]
    return nn.Sequential(*layers)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=30)
    ap.add_argument("--env-num", type=int, default=2000)
    ap.add_argument("--task", default="Goal_Point_8Hazards")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--hid", type=int, default=64, choices=Engine.POLICY_HIDDEN)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.manual_seed(args.seed)
    T = 200
    cfg = dict(configuration(args.task), env_num=args.env_num, _seed=args.seed, num_steps=T)
    env = Engine(cfg)
    D, A = env.obs_flat_size, env.action_space.shape[0]
    mu_net, v_net = mlp([D, args.hid, args.hid, A]).to(dev), mlp([D, args.hid, args.hid, 1]).to(dev)
    log_std = nn.Parameter(torch.full((A,), -0.5, device=dev))
    pi_opt = torch.optim.Adam(list(mu_net.parameters()) + [log_std], lr=3e-4)
    v_opt = torch.optim.Adam(v_net.parameters(), lr=1e-3)
    print(f"{'epoch':>5} {'EpRet':>9} {'EpCost':>9} {'EpLen':>7} {'goals/env':>9} {'collect ms':>10} {'update ms':>10}")
    for epoch in range(args.epochs):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        env.reset(check=False)
        params = Engine.pack_actor_critic(mu_net=mu_net, v_net=v_net, log_std=log_std)
        out = env.rollout_policy(params, T, noise_seed=(args.seed, epoch))
        adv, ret = gae_rollout(out['rew'], out['val'], out['done'])
        torch.cuda.synchronize(); t1 = time.perf_counter()
        # episode statistics (an episode ends at done or at the end of the tape)
        done = out['done']
        n_ep = done.sum() + (done[-1] == 0).sum()
        ep_ret = out['rew'].sum() / n_ep
        ep_cost = out['cost'].sum() / n_ep
        ep_len = done.numel() / n_ep
        # PPO-clip update
        obs, act = out['obs'].reshape(-1, D), out['act'].reshape(-1, A)
        logp_old, adv_f, ret_f = out['logp'].reshape(-1), adv.reshape(-1), ret.reshape(-1)
        adv_f = (adv_f - adv_f.mean()) / (adv_f.std() + 1e-8)
        n = obs.shape[0]
        for it in range(8):
            idx = torch.randint(0, n, (65536,), device=dev)
            dist = torch.distributions.Normal(mu_net(obs[idx]), torch.exp(log_std))
            logp = dist.log_prob(act[idx]).sum(-1)
            ratio = torch.exp(logp - logp_old[idx])
            loss_pi = -torch.min(ratio * adv_f[idx], torch.clamp(ratio, 0.8, 1.2) * adv_f[idx]).mean()
            pi_opt.zero_grad(); loss_pi.backward(); pi_opt.step()
            loss_v = ((v_net(obs[idx]).squeeze(-1) - ret_f[idx]) ** 2).mean()
            v_opt.zero_grad(); loss_v.backward(); v_opt.step()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"{epoch:5d} {ep_ret.item():9.4f} {ep_cost.item():9.4f} {ep_len.item():7.1f} "
              f"{(done.sum() / args.env_num).item():9.3f} {(t1 - t0) * 1e3:10.2f} {(t2 - t1) * 1e3:10.2f}")
    env.check_layouts()


if __name__ == "__main__":
    main()

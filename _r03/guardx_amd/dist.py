"""Multi-GPU glue: one process per GPU, environments sharded with no data-path
collective; the only exchange is the once-per-epoch rollout hand-off to the
learner -- one all-gather (RCCL over xGMI on GPUs, gloo on CPU) per epoch
(SURVEY.md section 8e): either of the packed per-rank rollout shard
(all_gather_rollout), or -- 2.4x fewer bytes for the Point -- of the dynamics
tape, which every rank then expands into the packed rows itself (TapeHandoff).
The reference has no counterpart: it runs on a single device (engine.py:100,
trpo.py:21)."""
import contextlib
import os

import torch
import torch.distributed as dist

ROLLOUT_FIELDS = ("obs", "act", "rew", "cost", "done")


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # GX_DIST_BACKEND=gloo: rehearse the multi-rank path on a box with fewer GPUs than ranks
            backend = os.environ.get("GX_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def pack_rollout(obs, act, rew, cost, done):
    """(T,N,D) (T,N,A) (T,N) (T,N) (T,N) -> one (T,N,D+A+3) tensor: one big collective
    instead of five small ones."""
    return torch.cat([obs, act, rew.unsqueeze(-1), cost.unsqueeze(-1), done.unsqueeze(-1)], dim=-1)


def unpack_rollout(packed, obs_dim, act_dim):
    o = packed[..., :obs_dim]
    a = packed[..., obs_dim:obs_dim + act_dim]
    r, c, d = (packed[..., obs_dim + act_dim + k] for k in range(3))
    return dict(obs=o, act=a, rew=r, cost=c, done=d)


def all_gather_rollout(packed, out=None):
    """All-gather the per-rank packed shard -> (world, T, N, W).  World size 1: a view."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return packed.unsqueeze(0)
    world = dist.get_world_size()
    shape = tuple(packed.shape)
    if out is None:
        out = torch.empty((world,) + shape, dtype=packed.dtype, device=packed.device)
    # concatenated-along-dim-0 form: accepted by both RCCL and gloo
    dist.all_gather_into_tensor(out.view((world * shape[0],) + shape[1:]), packed.contiguous())
    return out


class TapeHandoff:
    """Once-per-epoch rollout hand-off by dynamics tape.

        h = TapeHandoff(env, T)                    # after init_process_group
        per epoch:  env.reset(); h.step(actions)   # = env.rollout_tape + async all-gather + expansion of the
                                                   #   PREVIOUS epoch's gathered tapes on a side stream
        h.drain(); h.rollout                       # (world, T, N, obs+act+3): the last expanded epoch

    The rank that steps writes 48 B per env-step (Point: qpos, qvel, action, done, two layout-row indices) instead of
    the 192-B packed row; ONE
    all_gather_into_tensor per epoch moves the shards as they are; every rank runs the observation pass
    (Engine.expand_tape) over all `world` tapes and so holds the same rows rollout(packed=True) + an all-gather of
    the packed shards would have given it, bit for bit.  The all-gather of epoch k overlaps epoch k+1 entirely:
    its expansion is enqueued during epoch k+1 behind a stream-level wait for the collective, and the engine orders
    the layout sampler that recycles epoch k's pool behind that expansion (three pools), so a slow link slows the
    epochs down instead of corrupting anything.  On the gloo rehearsal backend the shard goes through host memory."""

    def __init__(self, env, T, depth=3):
        self.env, self.T, self.depth = env, int(T), depth
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.n = sum(env.tape_floats(self.T))
        self.host = dist.is_initialized() and dist.get_backend() != "nccl"
        dev = env.device
        self.send = [torch.empty(self.n, dtype=torch.float32, device=dev) for _ in range(depth)]
        self.recv = [torch.empty(self.world * self.n, dtype=torch.float32, device="cpu" if self.host else dev,
                                 pin_memory=self.host and dev.type == "cuda") for _ in range(depth)]
        W = env.obs_flat_size + env.action_space.shape[0] + 3
        self.out = [torch.empty(self.world, self.T, env.env_num, W, dtype=torch.float32, device=dev) for _ in range(2)]
        # the expansion runs on its own stream (a CPU stand-in engine, as in the gloo unit test, has none)
        self.stream = torch.cuda.Stream(device=dev) if dev.type == "cuda" else None
        self.pending = None            # (work, slot, token) of the epoch whose tapes are in flight
        self.k = 0
        self.rollout = None            # the most recently expanded epoch (valid after drain())
        self.bytes_received = 0

    def step(self, actions):
        i = self.k % self.depth
        shard, token = self.env.rollout_tape(actions, out=self.send[i])
        self._expand_pending()         # epoch k-1: its collective has had a whole epoch
        if self.world == 1:
            work, self.recv[i] = None, shard
        else:
            src = shard.to("cpu") if self.host else shard
            work = dist.all_gather_into_tensor(self.recv[i], src, async_op=True)
            self.bytes_received += (self.world - 1) * self.n * 4
        self.pending = (work, i, token)
        self.k += 1

    def _expand_pending(self):
        if self.pending is None:
            return
        work, i, token = self.pending
        self.pending = None
        out = self.out[self.k % 2]
        # no wait for the caller's stream: the buffers are this object's own, and their reuse three epochs later is
        # ordered behind this expansion by the engine (sampler of the recycled pool -> reset_apply -> rollout_tape)
        if work is None and self.stream is not None:   # single rank: the shard comes straight from the caller's stream
            self.stream.wait_stream(torch.cuda.current_stream())
        with (torch.cuda.stream(self.stream) if self.stream is not None else contextlib.nullcontext()):
            if work is not None:
                work.wait()            # RCCL: this stream waits for the collective; gloo: the host does
            recv = self.recv[i]
            if self.host:
                recv = recv.to(self.env.device, non_blocking=True)
            for s in range(self.world):
                self.env.expand_tape(recv[s * self.n:(s + 1) * self.n], token, self.T, out=out[s])
        self.rollout = out

    def drain(self):
        self._expand_pending()
        if self.stream is not None:
            torch.cuda.current_stream().wait_stream(self.stream)


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        if dist.get_backend() == "nccl":   # name the device: RCCL otherwise guesses it from the rank
            dist.barrier(device_ids=[torch.cuda.current_device()])
        else:
            dist.barrier()


def max_over_ranks(value, device):
    if dist.is_initialized() and dist.get_backend() != "nccl":
        device = "cpu"                      # gloo rehearsal
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


class ShardedReset:
    """OPTIONAL: reset() with the reference's 1e6-candidate layout sampler split over the ranks.

        sr = ShardedReset(env)         # after init_process_group; env.set_prefetch(-1) is done here
        obs = sr.reset()               # instead of env.reset()

    The candidates are independent (candidate c draws from split(key, 1e6)[c]), so rank r samples candidates
    [r 1e6 / W, (r + 1) 1e6 / W) alone (Engine.sample_shard), the ranks all-gather their valid layouts (a few MB: ~2 % of
    the candidates are valid) and every rank installs the concatenation -- shard after shard, i.e. candidate order -- as
    its pool (Engine.reset_from_shards): layout_size, pool rows, the observation and every later randint draw are those of
    the unsharded reset(), bit for bit, and the sampler's 0.5 ms of vector-ALU work is done once per node instead of once
    per GPU.  It is a SECOND collective (north_star allows one, the rollout hand-off), so nothing uses it unless asked:
    bench.py with GX_SHARD_SAMPLER=1.  The sampler then runs on the caller's stream in front of the epoch (no prefetch
    overlap): per epoch 1/W of the sampler + one small all-gather + the install."""

    def __init__(self, env):
        self.env = env
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.host = dist.is_initialized() and dist.get_backend() != "nccl"
        env.set_prefetch(-1)
        cap, dev = env.shard_capacity(self.world), env.device
        K = int(env.n_layout_objects)
        self.rows = torch.empty(cap, K, 2, dtype=torch.float32, device=dev)
        self.count = torch.empty(1, dtype=torch.int32, device=dev)
        self.rows_all = torch.empty(self.world, cap, K, 2, dtype=torch.float32, device=dev)
        self.counts = torch.empty(self.world, dtype=torch.int32, device=dev)
        self.bytes_received = 0

    def reset(self, check=True):
        env, W = self.env, self.world
        env.sample_shard(self.rank, W, self.rows, self.count)
        if W == 1:
            self.rows_all[0].copy_(self.rows); self.counts.copy_(self.count)
        elif self.host:   # gloo rehearsal: through host memory
            ra = torch.empty(self.rows_all.shape, dtype=torch.float32)
            ca = torch.empty(W, dtype=torch.int32)
            dist.all_gather_into_tensor(ra.view(-1), self.rows.cpu().view(-1))
            dist.all_gather_into_tensor(ca, self.count.cpu())
            self.rows_all.copy_(ra); self.counts.copy_(ca)
        else:
            dist.all_gather_into_tensor(self.rows_all.view(-1), self.rows.view(-1))
            dist.all_gather_into_tensor(self.counts, self.count)
        self.bytes_received += (W - 1) * (self.rows.numel() * 4 + 4)
        return env.reset_from_shards(self.rows_all, self.counts, check=check)

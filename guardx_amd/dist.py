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
import time

import torch
import torch.distributed as dist

_GX_ERR_STATE = 5   # include/guardx.h:37 (= guardx_amd._native.GX_ERR_STATE; not imported: a CPU stand-in engine has no library)

ROLLOUT_FIELDS = ("obs", "act", "rew", "cost", "done")


def forced_dist():
    """GX_FORCE_DIST=1: a world of ONE rank still forms a process group and issues every collective of the N > 1 path
    (RCCL on a GPU) instead of short-circuiting it -- the whole multi-GPU code path on a one-GPU box
    (tests/test_rccl_one_rank.py, `GX_FORCE_DIST=1 python bench.py --gpus 1`)."""
    return os.environ.get("GX_FORCE_DIST", "0") == "1"


def _collective():
    """do the collectives of this process run? (a group exists and has company, or is forced)"""
    return dist.is_initialized() and (dist.get_world_size() > 1 or forced_dist())


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or forced_dist()) and not dist.is_initialized():
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
    if not _collective():
        return packed.unsqueeze(0)
    world = dist.get_world_size()
    shape = tuple(packed.shape)
    if out is None:
        out = torch.empty((world,) + shape, dtype=packed.dtype, device=packed.device)
    # concatenated-along-dim-0 form: accepted by both RCCL and gloo
    dist.all_gather_into_tensor(out.view((world * shape[0],) + shape[1:]), packed.contiguous())
    return out


_HANDOFF_STREAMS = {}


def _handoff_stream(dev):
    key = (dev.type, dev.index)
    if key not in _HANDOFF_STREAMS:
        _HANDOFF_STREAMS[key] = torch.cuda.Stream(device=dev)
    return _HANDOFF_STREAMS[key]


class TapeHandoff:
    """Once-per-epoch rollout hand-off by dynamics tape -- the ONE collective of the multi-GPU path.

        h = TapeHandoff(env, T)                    # after init_process_group
        per epoch:  env.reset(); h.step(actions)   # = env.rollout_tape + async all-gather + expansion of the
                                                   #   PREVIOUS epoch's gathered tapes on a side stream
        h.drain(); h.rollout                       # (world, T, N, obs+act+3): the last expanded epoch

    The rank that steps writes 36 B per env-step (Point: qpos, qvel, action, one word for done + the layout row in effect + the layout row a reset_done installed) instead of
    the 192-B packed row; ONE
    all_gather_into_tensor per epoch moves the shards as they are; every rank runs the observation pass
    (Engine.expand_tape) over all `world` tapes and so holds the same rows rollout(packed=True) + an all-gather of
    the packed shards would have given it, bit for bit.  The all-gather of epoch k overlaps epoch k+1 entirely:
    its expansion is enqueued during epoch k+1 behind a stream-level wait for the collective, and the engine orders
    the layout sampler that recycles epoch k's pool behind that expansion (three pools), so a slow link slows the
    epochs down instead of corrupting anything.  On the gloo rehearsal backend the shard goes through host memory.

    sharded_sampler (default: on at world > 1): the reference's 1e6-candidate layout sampler
    (engine.py:433-444; candidate c draws from split(key, 1e6)[c], so the candidates are independent) is split over the
    ranks WITHOUT a second collective.  The key of a later reset is known now (this key advanced by one split per
    step(), engine.py:431), so once epoch k's tape is on its way rank r samples candidates [r 1e6 / W, (r + 1) 1e6 / W)
    of reset(k + 3) on the engine's side stream, into the tail of the buffer that will carry epoch k + 1's tape; the
    all-gather of epoch k + 1 delivers every rank's block, and while epoch k + 2 runs the blocks are installed -- shard
    after shard = candidate order -- as the pool reset(k + 3) takes like a prefetch hit: layout_size, pool rows,
    observations and every later randint draw are the unsharded reset()'s, bit for bit.  A reset without an installed
    pool (the first three, a changed episode length) samples all candidates inline.

    expand: "all" (default) -- every rank expands every rank's tape (the hand-off contract above); "local" -- only its
    own; expand_rank(s) then expands rank s's tape of the last gathered epoch on demand (before the next step()).

    force_collective (default: GX_FORCE_DIST=1): a world of one issues the collective, shards the sampler "over" its one
    rank and expands through the all-ranks launch, i.e. runs the N > 1 path as it is instead of the short cuts."""

    def __init__(self, env, T, depth=3, sharded_sampler=None, expand="all", _play=None, force_collective=None):
        assert expand in ("all", "local")
        self.env, self.T, self.depth, self.expand = env, int(T), depth, expand
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        if _play is not None:          # (rank, world) played without a process group: tools/rehearse_rank.py, which
            self.rank, self.world = _play   # overrides _gather()
        if force_collective is None:
            force_collective = forced_dist() and dist.is_initialized()
        self.collective = self.world > 1 or bool(force_collective)
        self.n_tape = sum(env.tape_floats(self.T))
        self.host = dist.is_initialized() and dist.get_backend() != "nccl"
        dev = env.device
        if sharded_sampler is None:    # a world of one has nobody to share the sampler with
            sharded_sampler = hasattr(env, "sample_shard_ahead") and self.collective
        self.sharded = bool(sharded_sampler)
        self.sharded_used = self.sharded   # (close() clears `sharded`)
        self.cap = self.n_block = 0
        if self.sharded:
            # rows one block holds: twice this rank's expected share of the valid layouts (binomial: the share's spread
            # is a few per cent) + slack; an overflow is reported by the reset that would take the pool, never installed
            size = getattr(env, "layout_size", None)
            if not size:
                raise RuntimeError("TapeHandoff(sharded_sampler=True): call env.reset() once first (the size of its "
                                   "layout pool sizes the export blocks)")
            self.cap = int(min(-(-int(env._cfg.n_candidates) // self.world), 2 * (-(-int(size) // self.world)) + 1024))
            self.n_block = env.shard_block_floats(self.cap)
            env.set_layout_source('shards')
        pad = (-self.n_tape) % 4 if self.sharded else 0    # the block starts 16-byte aligned
        self.off_block = self.n_tape + pad
        self.n = self.off_block + self.n_block             # floats per rank in the collective
        self.send = [torch.zeros(self.n, dtype=torch.float32, device=dev) for _ in range(depth)]
        self.recv = [torch.empty(self.world * self.n, dtype=torch.float32, device="cpu" if self.host else dev,
                                 pin_memory=self.host and dev.type == "cuda") for _ in range(depth)]
        W = env.obs_flat_size + env.action_space.shape[0] + 3
        self.out = [torch.empty(self.world, self.T, env.env_num, W, dtype=torch.float32, device=dev) for _ in range(2)]
        # the expansion runs on its own stream (a CPU stand-in engine, as in the gloo unit test, has none): ONE per device
        # and process, shared by successive hand-offs (bench.py makes one per leg) -- HIP multiplexes a process's
        # streams onto a few hardware queues in creation order, and every extra stream is a chance of an alias
        if dev.type != "cuda":
            self.stream = None
        elif hasattr(env, "aux_stream") and os.environ.get("GX_HANDOFF_STREAM", "aux") == "aux":
            self.stream = env.aux_stream()     # one per device and process; checked against the default stream's queue
        else:
            self.stream = _handoff_stream(dev)
        self.queue_probe = []          # per probe round: (spin ms, collective ms, shared a queue?)
        if self.collective and not self.host and self.stream is not None and dist.is_initialized() \
                and os.environ.get("GX_HANDOFF_QUEUE_PROBE", "0") == "1":
            self._probe_collective_queue()
        self.pending = None            # (work, slot, token, ticket of the block it carries) of the epoch in flight
        self.last = None               # (gathered buffer on the device, token) of the last expanded epoch (expand_rank)
        self.k = 0
        self.rollout = None            # the most recently expanded epoch (valid after drain())
        self.bytes_received = 0
        self.blocks_installed = 0
        self.deferred = None           # (ticket, gathered buffer) of a block drain() could not install yet
        self.works = [None] * depth    # the collective that last read send[i]
        self.next_ticket = None        # ticket of the block being sampled into the next send buffer's tail
        self.shard_skips = 0           # epochs whose block was not sampled (the engine refused: see step())
        self.closed = False

    PROBE_ROUNDS = 4
    PROBE_SPIN_CYCLES = 3_000_000      # torch.cuda._sleep: ~1.2-1.5 ms

    def _probe_collective_queue(self):
        """Does the hand-off's stream share a hardware queue with the stream the collectives run on?  HIP maps a process's
        streams of one priority onto a few hardware queues; RCCL's stream is torch's, made at the first collective.  On a
        shared queue the epoch still computes the same thing, but collective k + 1 queues behind expansion k, and when the
        link is the bound it idles for an expansion per epoch.  The test: a spinner (~1.3 ms) goes onto the hand-off's
        stream, a four-float all-gather is issued and awaited -- if the host sees it complete only after the spinner, the
        two share a queue and the engine is asked for another stream (gx_aux_stream_renew; the old one stays alive, so the
        next lands elsewhere).  PROBE_ROUNDS rounds on EVERY rank, whatever each finds: the rounds are collectives.
        OFF by default (GX_HANDOFF_QUEUE_PROBE=1 runs it): on the one GPU this project can measure on, a one-rank group, the
        two streams DO share a queue (the probe finds it in round 2 and the third stream runs beside the collective:
        0.06-0.11 ms against a 1.26 ms spinner) -- and bench.py --gpus 1 is slower afterwards, 590-596 M env-steps/s against
        645-654 M (profiles/r05_ab_handoff_queue_probe.log): with a collective that is one local copy the shared queue costs
        nothing and the stream the hand-off moves to shares its queue with something else.  Whether it pays when the link
        is the bound needs the 8-GPU node."""
        dev = self.env.device
        x = torch.zeros(4, dtype=torch.float32, device=dev)
        y = torch.zeros(4 * self.world, dtype=torch.float32, device=dev)
        dist.all_gather_into_tensor(y, x)                  # the first collective builds the communicator and its stream
        torch.cuda.synchronize(dev)
        cur = torch.cuda.current_stream(dev)
        for rnd in range(self.PROBE_ROUNDS):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(self.stream):
                e0.record()
                torch.cuda._sleep(self.PROBE_SPIN_CYCLES)
                e1.record()
            t0 = time.perf_counter()
            work = dist.all_gather_into_tensor(y, x, async_op=True)
            work.wait()                                    # the current stream waits for the collective's stream ...
            cur.synchronize()                              # ... and the host for the current stream: not for the spinner
            ms = (time.perf_counter() - t0) * 1e3
            torch.cuda.synchronize(dev)
            spin = e0.elapsed_time(e1)
            shared = ms > 0.7 * spin
            self.queue_probe.append((round(spin, 3), round(ms, 3), bool(shared)))
            if shared and rnd + 1 < self.PROBE_ROUNDS and hasattr(self.env, "aux_stream"):
                self.stream = self.env.aux_stream(renew=True)

    def close(self):
        """Give the layout sampling back to the engine (its own prefetch) and order the CURRENT stream behind everything
        this object still has in flight on streams torch's allocator does not know of -- the collectives that read
        send[] / write recv[], the expansion on the hand-off's stream, the shard sampler the last step() queued into a
        send buffer's tail on the engine's side stream -- so that the buffers may go back to the allocator."""
        if self.closed:
            return
        self.closed = True
        self.drain()
        for w in self.works:
            if w is not None:
                w.wait()
        self.works = [None] * self.depth
        if self.sharded_used and hasattr(self.env, "shard_join") and getattr(self.env, "_h", True) is not None:
            self.env.shard_join()      # the current stream waits for shard_done
        self.next_ticket = self.deferred = None
        if self.sharded:
            self.env.set_layout_source('own')
            self.sharded = False

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001 - interpreter shutdown / the engine is gone: nothing left to order
            pass

    def step(self, actions):
        i = self.k % self.depth
        buf = self.send[i]
        ticket, self.next_ticket = self.next_ticket, None   # the block sampled into this buffer's tail during the last step
        if ticket is not None:
            self.env.shard_join()      # the collective below must see the block (it was finished long ago)
        if self.sharded:
            # The block that will travel with the NEXT epoch's tape: this rank's candidates of the reset three resets
            # after the last one.  Launched FIRST, behind the reset that is already queued: the sampler then runs beside this
            # epoch's dynamics pass and has until the next epoch's collective -- a whole epoch.  (Rounds 4-5 launched it
            # after this epoch's collective was issued; queued behind the caller's stream it could only start when the
            # dynamics pass had ENDED and had to be finished one dynamics pass later: at W = 2, where a rank's share of
            # the sampler is as long as its dynamics pass, the next collective waited for it.)
            j = (self.k + 1) % self.depth
            if self.works[j] is not None:
                self.works[j].wait()   # the collective that last read send[j] (two epochs ago)
            try:
                self.next_ticket = self.env.sample_shard_ahead(self.rank, self.world, self.send[j][self.off_block:],
                                                               self.cap, resets_ahead=3)
            except RuntimeError as exc:
                # The engine has no horizon to sample for (prefetch switched off, or more steps since the last reset
                # than three intervals cover): a state every rank shares, so every rank skips this block alike and the
                # reset it was meant for samples inline.  Raising here would leave the other ranks waiting in this
                # epoch's collective.
                if getattr(exc, "status", None) != _GX_ERR_STATE:
                    raise
                self.next_ticket = None
                self.shard_skips += 1
        shard, token = self.env.rollout_tape(actions, out=buf[:self.n_tape])
        self._expand_pending()         # epoch k-1: its collective has had a whole epoch
        work = self._gather(i, buf)
        self.works[i] = work
        self.pending = (work, i, token, ticket)
        self.k += 1

    def _gather(self, i, buf):
        """the ONE collective of the epoch: every rank's [tape | layouts | entry records | shard block] into recv[i]"""
        if not self.collective:
            self.recv[i] = buf
            return None
        src = buf.to("cpu") if self.host else buf
        work = dist.all_gather_into_tensor(self.recv[i], src, async_op=True)
        self.bytes_received += (self.world - 1) * self.n * 4
        return work

    def _install(self, ticket, recv):
        self.env.install_shards(ticket, recv[self.off_block:], self.n, self.world, self.cap)
        self.blocks_installed += 1

    def _expand_pending(self, install=True):
        if self.pending is None:
            if install and self.deferred is not None:      # the block a drain() left behind: its reset is the next one
                with (torch.cuda.stream(self.stream) if self.stream is not None else contextlib.nullcontext()):
                    self._install(*self.deferred)
                self.deferred = None
            return
        work, i, token, ticket = self.pending
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
            if ticket is not None and self.sharded:
                if install:            # first: the next reset() waits for this pool, not for the expansions
                    self._install(ticket, recv)
                else:                  # drain(): the pool slot still holds the NEXT reset's layouts; install after it
                    self.deferred = (ticket, recv)
            if self.expand == "all" and self.collective and hasattr(self.env, "expand_tapes"):
                self.env.expand_tapes(recv, self.n, self.world, token, self.T, out)   # every rank's tape, one launch
            else:
                for s in (range(self.world) if self.expand == "all" else (self.rank,)):
                    self.env.expand_tape(recv[s * self.n:s * self.n + self.n_tape], token, self.T, out=out[s])
        self.rollout = out
        self.last = (recv, token)

    def expand_rank(self, s):
        """expand="local": the packed rows of rank s for the epoch `rollout` holds, expanded now on the hand-off's
        stream (valid until the next step(): the layout pool the tape names is recycled two resets after its rollout)."""
        recv, token = self.last
        with (torch.cuda.stream(self.stream) if self.stream is not None else contextlib.nullcontext()):
            self.env.expand_tape(recv[s * self.n:s * self.n + self.n_tape], token, self.T, out=self.rollout[s])
        return self.rollout[s]

    def drain(self):
        if getattr(self.env, "_h", True) is None:      # the engine is closed (it synchronised the device): nothing in flight
            self.pending = None
            return
        self._expand_pending(install=False)
        if self.stream is not None:
            torch.cuda.current_stream().wait_stream(self.stream)


def barrier():
    if _collective():
        if dist.get_backend() == "nccl":   # name the device: RCCL otherwise guesses it from the rank
            dist.barrier(device_ids=[torch.cuda.current_device()])
        else:
            dist.barrier()


def max_over_ranks(value, device):
    if dist.is_initialized() and dist.get_backend() != "nccl":
        device = "cpu"                      # gloo rehearsal
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if _collective():
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
    a caller without a rollout hand-off (with one, TapeHandoff shards the sampler on the hand-off's own collective).  The
    sampler then runs on the caller's stream in front of the epoch (no prefetch
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
        if W == 1 and not _collective():
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

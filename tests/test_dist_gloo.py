"""N>1 path on CPU: two gloo ranks each own a contiguous env shard (no data-path
collective); the per-epoch rollout hand-off is one all-gather of the packed shard.
Checks that the gathered global rollout equals the unsharded rollout."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, N, T, q):
    try:
        _worker_body(rank, world, port, N, T, q)
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put(("err", traceback.format_exc()))
        raise


def _worker_body(rank, world, port, N, T, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from guardx_amd import dist as gxd
    from helpers import task_config
    from oracle import gxo
    r, _, w = gxd.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    E = gxo.OracleEngine(task_config(N, seed=3, num_steps=T), n_candidates=30000,
                         env_total=N * world, env_offset=N * rank)
    E.reset()
    acts = np.random.RandomState(0).uniform(-1, 1, (T, N * world, 2)).astype(np.float32)[:, rank * N:(rank + 1) * N]
    obs, rew, cost, done = [], [], [], []
    for t in range(T):
        o, rr, d, info = E.step(acts[t])
        o = E.reset_done()
        obs.append(o); rew.append(rr); cost.append(info['cost']); done.append(d)
    packed = gxd.pack_rollout(*[torch.from_numpy(np.stack(x)) for x in (obs, acts, rew, cost, done)])
    full = gxd.all_gather_rollout(packed)            # (world, T, N, D+2+3)
    assert gxd.max_over_ranks(rank, torch.device("cpu")) == world - 1
    # bench.py's asynchronous hand-off ring (three gathered buffers in flight) on the same shards
    sys.path.insert(0, ROOT)
    import bench
    ring = bench.RolloutHandoff(world)
    shards = [packed * float(k + 1) for k in range(5)]          # five "epochs"
    for sh in shards:
        ring.submit(sh)
    ring.drain()
    assert ring.bytes == 5 * packed.numel() * 4 * world
    for k in (2, 3, 4):                                         # the three buffers still held
        got = ring.slots[k % ring.depth][2]
        assert torch.equal(got, full * float(k + 1))
    gxd.barrier()
    if rank == 0:
        q.put(("ok", full.numpy()))
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_rollout_allgather_equals_unsharded():
    from helpers import task_config
    from oracle import gxo
    from guardx_amd import dist as gxd
    world, N, T = 2, 16, 12
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, N, T, q)) for r in range(world)]
    for p in procs:
        p.start()
    tag, full = q.get(timeout=240)
    assert tag == "ok", full
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # unsharded run
    E = gxo.OracleEngine(task_config(N * world, seed=3, num_steps=T), n_candidates=30000)
    E.reset()
    acts = np.random.RandomState(0).uniform(-1, 1, (T, N * world, 2)).astype(np.float32)
    for t in range(T):
        o, r, d, info = E.step(acts[t])
        o = E.reset_done()
        got = np.concatenate([full[k, t] for k in range(world)], axis=0)   # (N*world, 48)
        parts = gxd.unpack_rollout(torch.from_numpy(got), 43, 2)
        np.testing.assert_array_equal(parts['obs'].numpy(), o)
        np.testing.assert_array_equal(parts['act'].numpy(), acts[t])
        np.testing.assert_array_equal(parts['rew'].numpy(), r)
        np.testing.assert_array_equal(parts['cost'].numpy(), info['cost'])
        np.testing.assert_array_equal(parts['done'].numpy(), d)


class _StubEngine:
    """CPU stand-in with the Engine surface TapeHandoff uses: the "tape" of (rank, epoch) is a ramp, the expansion an
    affine map of it that depends on the token, so slots, ordering and shard indexing all show in the result."""

    def __init__(self, rank, N, D):
        self.rank, self.env_num, self.obs_flat_size = rank, N, D
        self.device = torch.device("cpu")
        self.action_space = type("Box", (), {"shape": (2,)})()
        self.epoch = 0
        self.expanded = []

    def tape_floats(self, T):
        return T * self.env_num * 4, 8, T * self.env_num * 2

    def rollout_tape(self, actions, out=None):
        n = sum(self.tape_floats(actions.shape[0]))
        out.copy_(torch.arange(n, dtype=torch.float32) + 1000.0 * self.rank + 10000.0 * self.epoch)
        self.epoch += 1
        return out, 77 + self.epoch

    def expand_tape(self, shard, token, T, out=None):
        W = self.obs_flat_size + 2 + 3
        self.expanded.append(int(token))
        out.copy_((shard[:T * self.env_num * 4].reshape(T, self.env_num, 4).sum(-1, keepdim=True) + float(token)).expand(T, self.env_num, W))
        return out


def _tape_worker(rank, world, port, q):
    try:
        os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                          MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        from guardx_amd import dist as gxd
        gxd.init_from_env("gloo")
        T, N, D = 3, 5, 6
        env = _StubEngine(rank, N, D)
        h = gxd.TapeHandoff(env, T)
        n = sum(env.tape_floats(T))
        acts = torch.zeros(T, N, 2)
        for ep in range(5):
            h.step(acts)
            if ep >= 1:            # the previous epoch has been expanded by now
                for src in range(world):
                    ramp = torch.arange(n, dtype=torch.float32) + 1000.0 * src + 10000.0 * (ep - 1)
                    want = ramp[:T * N * 4].reshape(T, N, 4).sum(-1, keepdim=True) + float(77 + ep)
                    assert torch.equal(h.rollout[src], want.expand(T, N, D + 5)), (rank, ep, src)
        h.drain()
        assert env.expanded == [78] * world + [79] * world + [80] * world + [81] * world + [82] * world
        assert h.bytes_received == 5 * (world - 1) * n * 4
        gxd.barrier()
        if rank == 0:
            q.put(("ok", None))
        torch.distributed.destroy_process_group()
    except Exception as exc:  # noqa: BLE001
        q.put(("fail", f"rank {rank}: {type(exc).__name__}: {exc}"))
        raise


@pytest.mark.timeout(300)
def test_two_rank_tape_handoff_ring_over_gloo():
    """guardx_amd.dist.TapeHandoff over 2 gloo ranks with a stand-in engine: every rank ends up with every rank's
    expanded rows, epoch by epoch, one epoch behind the stepping (the GPU engine's own tape / expand parity is
    tests/test_gpu_parity.py::test_tape_handoff_*)."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_tape_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    tag, msg = q.get(timeout=240)
    assert tag == "ok", msg
    for p in procs:
        p.join(60)
        assert p.exitcode == 0


def test_single_process_passthrough():
    from guardx_amd import dist as gxd
    x = torch.arange(24, dtype=torch.float32).reshape(2, 3, 4)
    assert gxd.all_gather_rollout(x).shape == (1, 2, 3, 4)
    p = gxd.pack_rollout(torch.zeros(2, 3, 5), torch.ones(2, 3, 2), *[torch.full((2, 3), float(k)) for k in (2, 3, 4)])
    assert p.shape == (2, 3, 10)
    u = gxd.unpack_rollout(p, 5, 2)
    assert (u['act'] == 1).all() and (u['rew'] == 2).all() and (u['cost'] == 3).all() and (u['done'] == 4).all()


class _StubShardEngine:
    """CPU stand-in with the Engine surface ShardedReset uses: shard r "samples" r + 2 rows whose entries name the
    shard, the reset and the row; reset_from_shards records what it was handed."""

    def __init__(self, rank):
        self.rank = rank
        self.device = torch.device("cpu")
        self.n_layout_objects = 10
        self.prefetch = None
        self.resets = 0
        self.installed = []

    def set_prefetch(self, steps):
        self.prefetch = steps

    def shard_capacity(self, n_shards):
        return 6

    def sample_shard(self, shard, n_shards, rows, count):
        assert self.prefetch == -1 and shard == self.rank
        rows.zero_()
        for k in range(shard + 2):
            rows[k] = 100.0 * shard + 10.0 * self.resets + k
        count[0] = shard + 2
        return rows, count

    def reset_from_shards(self, rows_all, counts, check=True):
        self.installed.append((rows_all.clone(), counts.clone()))
        self.resets += 1
        return torch.full((3,), float(self.resets))


def _shard_worker(rank, world, port, q):
    try:
        os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                          MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        from guardx_amd import dist as gxd
        gxd.init_from_env("gloo")
        env = _StubShardEngine(rank)
        sr = gxd.ShardedReset(env)
        for ep in range(3):
            obs = sr.reset()
            assert obs[0].item() == ep + 1
            rows_all, counts = env.installed[-1]
            assert counts.tolist() == [s + 2 for s in range(world)]
            for s in range(world):                      # shard after shard = candidate order, on every rank
                for k in range(s + 2):
                    assert (rows_all[s, k] == 100.0 * s + 10.0 * ep + k).all(), (rank, ep, s, k)
                assert (rows_all[s, s + 2:] == 0).all()
        assert sr.bytes_received == 3 * (world - 1) * (6 * 10 * 2 * 4 + 4)
        gxd.barrier()
        if rank == 0:
            q.put(("ok", None))
        torch.distributed.destroy_process_group()
    except Exception as exc:  # noqa: BLE001
        q.put(("fail", f"rank {rank}: {type(exc).__name__}: {exc}"))
        raise


@pytest.mark.timeout(300)
def test_two_rank_sharded_reset_over_gloo():
    """guardx_amd.dist.ShardedReset over 2 gloo ranks with a stand-in engine: every rank installs every rank's export,
    in rank (= candidate) order, reset after reset (the GPU engine's own equality with the unsharded sampler is
    tests/test_gpu_parity.py::test_sharded_layout_sampler_equals_the_unsharded_reset)."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_shard_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    tag, msg = q.get(timeout=240)
    assert tag == "ok", msg
    for p in procs:
        p.join(60)
        assert p.exitcode == 0


class _StubShardAheadEngine(_StubEngine):
    """_StubEngine + the surface of the piggy-backed sharded sampler: the "block" a rank samples at call c names
    (rank, c); install_shards records what arrived with which ticket and whether a reset separates consecutive
    installs (an install must not clobber the pool the NEXT reset still has to take)."""

    def __init__(self, rank, N, D):
        super().__init__(rank, N, D)
        self.layout_size = 40
        self._cfg = type("Cfg", (), {"n_candidates": 1000})()
        self.source = 'own'
        self.calls = 0
        self.joined = 0
        self.installs = []
        self.resets = 0
        self.installs_since_reset = 0

    def set_layout_source(self, source):
        self.source = source

    def shard_block_floats(self, cap):
        return 4 + cap * 10 * 2

    def sample_shard_ahead(self, shard, n_shards, block, cap, resets_ahead=2):
        assert self.source == 'shards' and shard == self.rank and resets_ahead == 3
        assert block.numel() == self.shard_block_floats(cap)
        self.calls += 1
        block.fill_(100.0 * shard + self.calls)
        return self.calls

    def shard_join(self):
        self.joined += 1

    def reset(self):
        self.resets += 1
        self.installs_since_reset = 0

    def install_shards(self, ticket, blocks, stride, n_shards, cap):
        self.installs_since_reset += 1
        assert self.installs_since_reset == 1, "two installs between resets: the second overwrites a pool not yet taken"
        nb = self.shard_block_floats(cap)
        self.installs.append((ticket, [float(blocks[s * stride]) for s in range(n_shards)],
                              all(bool((blocks[s * stride:s * stride + nb] == blocks[s * stride]).all()) for s in range(n_shards))))


def _shard_ahead_worker(rank, world, port, q):
    try:
        os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                          MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        from guardx_amd import dist as gxd
        gxd.init_from_env("gloo")
        T, N, D = 3, 5, 6
        env = _StubShardAheadEngine(rank, N, D)
        h = gxd.TapeHandoff(env, T, expand="local")
        assert h.sharded and env.source == 'shards'
        assert h.cap == 500             # min(M / W, 2 L / W + 1024)
        assert h.n % 4 == 0 and h.off_block % 4 == 0
        acts = torch.zeros(T, N, 2)
        n_tape = sum(env.tape_floats(T))
        for ep in range(7):
            env.reset()
            h.step(acts)
            if ep >= 1 and ep != 4:     # the previous epoch's OWN tape has been expanded (expand="local")
                ramp = torch.arange(n_tape, dtype=torch.float32) + 1000.0 * rank + 10000.0 * (ep - 1)
                want = ramp[:T * N * 4].reshape(T, N, 4).sum(-1, keepdim=True) + float(77 + ep)
                assert torch.equal(h.rollout[rank], want.expand(T, N, D + 5)), (rank, ep)
            if ep == 3:
                h.drain()               # bench.py drains between its warm-up and its timed epochs
                assert h.deferred is not None
                other = 1 - rank        # on demand: the other rank's tape of the epoch just drained
                ramp = torch.arange(n_tape, dtype=torch.float32) + 1000.0 * other + 10000.0 * 3
                want = ramp[:T * N * 4].reshape(T, N, 4).sum(-1, keepdim=True) + float(77 + 4)
                assert torch.equal(h.expand_rank(other), want.expand(T, N, D + 5))
        h.drain()
        # the block of call c (sampled right after epoch c - 1's tape went out) travels with epoch c's tape and is installed
        # during epoch c + 1 -- one install per reset, never two -- with every rank's block in rank order; the drain after
        # epoch 3 defers block 3 to epoch 4's step, the final drain leaves block 6 uninstalled
        assert env.joined == 6 and env.calls == 7
        assert [t for t, _, _ in env.installs] == [1, 2, 3, 4, 5]
        for t, firsts, uniform in env.installs:
            assert firsts == [100.0 * s + t for s in range(world)] and uniform, (rank, t, firsts)
        assert h.bytes_received == 7 * (world - 1) * h.n * 4
        h.close()
        assert env.source == 'own'
        gxd.barrier()
        if rank == 0:
            q.put(("ok", None))
        torch.distributed.destroy_process_group()
    except Exception as exc:  # noqa: BLE001
        import traceback
        q.put(("fail", f"rank {rank}: {traceback.format_exc()}"))
        raise


@pytest.mark.timeout(300)
def test_two_rank_tape_handoff_with_piggybacked_shard_blocks_over_gloo():
    """The default N > 1 hand-off over 2 gloo ranks with a stand-in engine: each rank's export block of a later reset
    (sampled as soon as the previous tape is on its way) travels in the tail of its tape shard, ONE
    all_gather_into_tensor per epoch; one epoch later every rank
    installs both blocks (rank order = candidate order) under the ticket of that epoch -- before the expansions, exactly
    one install between consecutive resets, a drain() deferring the install it may not make yet; expand="local" expands
    the own tape only and any other on demand.  (The GPU engine's equality with the unsharded sampler:
    tests/test_gpu_parity.py::test_piggybacked_shard_sampler_equals_the_unsharded_engine.)"""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_shard_ahead_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    tag, msg = q.get(timeout=240)
    assert tag == "ok", msg
    for p in procs:
        p.join(60)
        assert p.exitcode == 0


def _forced_one_rank_worker(port, q):
    try:
        os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                          GX_FORCE_DIST="1")
        from guardx_amd import dist as gxd
        r, _, w = gxd.init_from_env("gloo")
        assert (r, w) == (0, 1) and torch.distributed.is_initialized() and gxd._collective()
        T, N, D = 3, 5, 6
        env = _StubShardAheadEngine(0, N, D)
        h = gxd.TapeHandoff(env, T)                     # forced: collective + sharded sampler + host staging (gloo)
        assert h.collective and h.sharded and h.host and h.world == 1 and h.cap == 1000
        acts = torch.zeros(T, N, 2)
        n_tape = sum(env.tape_floats(T))
        for ep in range(5):
            env.reset()
            h.step(acts)
            assert h.pending[0] is not None              # a real Work object: the all-gather was issued
            if ep >= 1:
                ramp = torch.arange(n_tape, dtype=torch.float32) + 10000.0 * (ep - 1)
                want = ramp[:T * N * 4].reshape(T, N, 4).sum(-1, keepdim=True) + float(77 + ep)
                assert torch.equal(h.rollout[0], want.expand(T, N, D + 5)), ep
        h.drain()
        assert [t for t, _, _ in env.installs] == [1, 2, 3] and env.calls == 5
        h.close()
        assert env.source == 'own' and h.next_ticket is None and env.joined == 5   # 4 in step() + close()
        x = torch.arange(6, dtype=torch.float32).reshape(1, 2, 3)
        g = gxd.all_gather_rollout(x)
        assert g.data_ptr() != x.data_ptr() and torch.equal(g[0], x)
        assert gxd.max_over_ranks(2.5, torch.device("cpu")) == 2.5
        gxd.barrier()
        senv = _StubShardEngine(0)
        sr = gxd.ShardedReset(senv)
        sr.reset()
        assert senv.installed[-1][1].tolist() == [2]
        q.put(("ok", None))
        torch.distributed.destroy_process_group()
    except Exception:  # noqa: BLE001
        import traceback
        q.put(("fail", traceback.format_exc()))
        raise


@pytest.mark.timeout(300)
def test_forced_collective_in_a_world_of_one_over_gloo():
    """GX_FORCE_DIST=1: a one-rank group runs the N > 1 path as it is (collective issued, sampler "sharded" over the one
    rank, close() joining the shard sampler) -- the CPU twin of tests/test_rccl_one_rank.py"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_forced_one_rank_worker, args=(_free_port(), q))
    p.start()
    tag, msg = q.get(timeout=240)
    assert tag == "ok", msg
    p.join(60)
    assert p.exitcode == 0

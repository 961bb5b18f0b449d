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


def test_single_process_passthrough():
    from guardx_amd import dist as gxd
    x = torch.arange(24, dtype=torch.float32).reshape(2, 3, 4)
    assert gxd.all_gather_rollout(x).shape == (1, 2, 3, 4)
    p = gxd.pack_rollout(torch.zeros(2, 3, 5), torch.ones(2, 3, 2), *[torch.full((2, 3), float(k)) for k in (2, 3, 4)])
    assert p.shape == (2, 3, 10)
    u = gxd.unpack_rollout(p, 5, 2)
    assert (u['act'] == 1).all() and (u['rew'] == 2).all() and (u['cost'] == 3).all() and (u['done'] == 4).all()

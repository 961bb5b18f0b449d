"""Child process of tests/test_rccl_one_rank.py: the N > 1 code path of guardx_amd.dist over a REAL RCCL process group of
one rank on one GPU (GX_FORCE_DIST=1: nothing short-circuits at world size 1).

    python tests/rccl_one_rank_child.py <report.json>

Started fresh (it has made no HIP call before init_process_group), it runs, over backend "nccl" (= RCCL on ROCm):
  * init_from_env(), barrier() (device_ids form), max_over_ranks() (all_reduce MAX on the device),
  * all_gather_rollout() and bench.RolloutHandoff (the packed-rows hand-off, async all_gather_into_tensor),
  * TapeHandoff with the collective forced: device-resident receive rings (host == False), work.wait() issued inside
    torch.cuda.stream(handoff stream), the sharded sampler "over" its one rank riding in the tape's tail
    (sample_shard_ahead -> shard_join -> all-gather -> install_shards), one launch expanding all tapes; eight epochs,
    rows bit-equal to a twin engine's rollout(packed=True) and to the CPU checker's step loop,
  * ShardedReset on its non-host branch (two all_gather_into_tensor calls on the device) against a plain reset().
The report lists what ran; NCCL_DEBUG=INFO lines (stdout/stderr of this process) are the parent's evidence that RCCL itself
was entered."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402


def main(report_path):
    assert os.environ.get("GX_FORCE_DIST") == "1" and os.environ.get("WORLD_SIZE") == "1"
    import torch
    import torch.distributed as dist
    from guardx_amd import Engine, dist as gxd
    from helpers import task_config
    from oracle import gxo

    report = {"calls": []}

    def ran(what):
        report["calls"].append(what)

    rank, local, world = gxd.init_from_env()
    assert (rank, world) == (0, 1) and dist.is_initialized() and dist.get_backend() == "nccl"
    ran("init_process_group(nccl, world_size=1)")
    dev = torch.device("cuda", local)
    gxd.barrier()
    ran("barrier(device_ids)")
    assert gxd.max_over_ranks(3.25, dev) == 3.25
    ran("all_reduce(MAX) [max_over_ranks]")

    # ---- packed-rows hand-off: synchronous and the async ring bench.py uses with GX_HANDOFF=packed ----
    x = torch.arange(2 * 3 * 5, dtype=torch.float32, device=dev).reshape(2, 3, 5)
    g = gxd.all_gather_rollout(x)
    assert g.shape == (1, 2, 3, 5) and g.data_ptr() != x.data_ptr() and torch.equal(g[0], x)
    ran("all_gather_into_tensor [all_gather_rollout]")
    import bench
    ring = bench.RolloutHandoff(1)
    assert not ring.host
    for k in range(5):
        ring.submit(x * float(k + 1))
    ring.drain()
    torch.cuda.synchronize()
    for k in (2, 3, 4):
        assert torch.equal(ring.slots[k % ring.depth][2][0], x * float(k + 1))
    ran("all_gather_into_tensor(async_op=True) x5 [bench.RolloutHandoff]")

    # ---- the default hand-off: tape + sharded sampler, collective forced ----
    N, T, M, EPOCHS = 256, 30, 120_000, 8
    cfg = task_config(N, seed=14, num_steps=T, goal_size=2.9)
    a, b = Engine(cfg, n_candidates=M), Engine(cfg, n_candidates=M)
    O = gxo.OracleEngine(cfg, n_candidates=M)
    o0 = a.reset()
    assert torch.equal(o0, b.reset())
    np.testing.assert_array_equal(o0.cpu().numpy(), O.reset())
    a.set_prefetch(T); b.set_prefetch(T)
    os.environ["GX_HANDOFF_QUEUE_PROBE"] = "1"      # (off by default: see TapeHandoff._probe_collective_queue)
    h = gxd.TapeHandoff(a, T)
    del os.environ["GX_HANDOFF_QUEUE_PROBE"]
    assert h.collective and h.sharded and not h.host and h.world == 1
    # the probe of the collective's hardware queue ran its rounds; the last stream it settled on runs beside the collective
    assert len(h.queue_probe) == h.PROBE_ROUNDS and not h.queue_probe[-1][2], h.queue_probe
    report["queue_probe"] = h.queue_probe
    assert all(r.device.type == "cuda" for r in h.recv) and h.recv[0].data_ptr() != h.send[0].data_ptr()
    rng = np.random.default_rng(21)
    prev = None
    D = a.obs_flat_size
    for ep in range(EPOCHS):
        acts_np = rng.uniform(-1, 1, (T, N, 2)).astype(np.float32)
        acts = torch.from_numpy(acts_np).to(dev)
        if ep:
            oa = a.reset(check=False)
            assert torch.equal(oa, b.reset(check=False)), ep
            np.testing.assert_array_equal(oa.cpu().numpy(), O.reset())
        h.step(acts)
        *_, pk = b.rollout(acts, packed=True)
        pkn = pk.cpu().numpy()
        for t in range(T):                                  # the twin's rows are the checker's
            _, r, d, info = O.step(acts_np[t])
            o = O.reset_done()
            np.testing.assert_array_equal(pkn[t, :, :D], o)
            np.testing.assert_array_equal(pkn[t, :, D + 2], r)
            np.testing.assert_array_equal(pkn[t, :, D + 3], info["cost"])
            np.testing.assert_array_equal(pkn[t, :, D + 4], d)
        if ep >= 1:                                         # epoch ep-1 came back through RCCL and was expanded
            torch.cuda.current_stream().wait_stream(h.stream)
            assert torch.equal(h.rollout[0].view(torch.int32), prev.view(torch.int32)), ep
        prev = pk
        if ep == 3:
            h.drain()
            assert torch.equal(h.rollout[0], prev) and h.deferred is not None
    h.drain()
    assert torch.equal(h.rollout[0].view(torch.int32), prev.view(torch.int32))
    a.check_layouts(); b.check_layouts()
    hits, misses, _ = a.prefetch_stats()
    assert hits == EPOCHS - 2 and misses == 0 and h.blocks_installed == EPOCHS - 2, (hits, misses, h.blocks_installed)
    assert h.bytes_received == 0 and h.shard_skips == 0     # (world - 1) shards come from elsewhere: none
    report["tape_handoff"] = {"epochs": EPOCHS, "N": N, "T": T, "candidates": M, "blocks_installed": h.blocks_installed,
                              "prefetch_hits": hits, "floats_per_rank": h.n, "receive_ring_on_device": True}
    ran(f"all_gather_into_tensor(async_op=True) x{EPOCHS} [TapeHandoff._gather, device receive ring]")
    ran("work.wait() inside torch.cuda.stream(handoff stream) [TapeHandoff._expand_pending]")
    ran("sample_shard_ahead -> shard_join -> all-gather -> install_shards [sharded sampler on the collective]")
    h.close()
    assert torch.equal(a.reset(), b.reset())
    np.testing.assert_array_equal(b._obs.cpu().numpy(), O.reset())

    # ---- the engine refuses to sample a block (no horizon): every rank skips it alike, nothing raises mid-epoch ----
    h2 = gxd.TapeHandoff(a, T)
    a.set_prefetch(-1); b.set_prefetch(-1)
    for ep in range(3):
        acts = torch.from_numpy(rng.uniform(-1, 1, (T, N, 2)).astype(np.float32)).to(dev)
        if ep:
            assert torch.equal(a.reset(check=False), b.reset(check=False))
        h2.step(acts)
        *_, pk = b.rollout(acts, packed=True)
    h2.drain()
    assert torch.equal(h2.rollout[0].view(torch.int32), pk.view(torch.int32))
    assert h2.shard_skips == 3 and h2.blocks_installed == 0
    h2.close()
    del h2
    a.set_prefetch(T); b.set_prefetch(T)
    assert torch.equal(a.reset(), b.reset())
    ran("TapeHandoff.step with the engine refusing the block (GX_ERR_STATE): skipped on every rank, no raise")

    # ---- ShardedReset, non-host branch ----
    c, d = Engine(cfg, n_candidates=M), Engine(cfg, n_candidates=M)      # fresh: both start from PRNGKey(seed)
    sr = gxd.ShardedReset(c)
    assert not sr.host
    d.set_prefetch(-1)
    O2 = gxo.OracleEngine(cfg, n_candidates=M)
    for ep in range(3):
        oc = sr.reset()
        assert torch.equal(oc, d.reset()), ep
        np.testing.assert_array_equal(oc.cpu().numpy(), O2.reset())
        acts_np = rng.uniform(-1, 1, (5, N, 2)).astype(np.float32)
        acts = torch.from_numpy(acts_np).to(dev)
        for t in range(5):
            c.step(acts[t]); d.step(acts[t]); O2.step(acts_np[t])
            oc = c.reset_done()
            assert torch.equal(oc, d.reset_done())
            np.testing.assert_array_equal(oc.cpu().numpy(), O2.reset_done())
    ran("all_gather_into_tensor x2 per reset [ShardedReset, device branch] x3")

    gxd.barrier()
    torch.cuda.synchronize()
    for e in (a, b, c, d):
        e.close()
    dist.destroy_process_group()
    ran("destroy_process_group")
    report["backend"] = "nccl"
    report["torch"] = torch.__version__
    report["device"] = torch.cuda.get_device_name(local)
    try:
        report["nccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception:  # noqa: BLE001
        report["nccl_version"] = None
    with open(report_path, "w") as f:
        json.dump(report, f, indent=1)
    print("RCCL_ONE_RANK_OK", flush=True)


if __name__ == "__main__":
    main(sys.argv[1])

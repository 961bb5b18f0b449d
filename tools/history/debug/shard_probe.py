"""Per-rank cost of the sharded reset: sample 1/W of the candidates + install (the all-gather replaced by local copies)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
bench.precondition_clocks(dev)
env = bench.make_engine(2000, 0, 1); env.set_prefetch(-1)
acts = bench.action_tape(200, 2000, 0, dev)
for W in (1, 2, 4, 8):
    cap, K = env.shard_capacity(W), env.n_layout_objects
    rows_all = torch.empty(W, cap, K, 2, device=dev); counts = torch.empty(W, dtype=torch.int32, device=dev)
    # fill every shard once (one engine plays all ranks), then time rank 0's share + install + rollout
    for r in range(W):
        rows, c = env.sample_shard(r, W); rows_all[r].copy_(rows); counts[r:r + 1].copy_(c)
    env.reset_from_shards(rows_all, counts)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50):
        rows, c = env.sample_shard(0, W); rows_all[0].copy_(rows); counts[0:1].copy_(c)
        env.reset_from_shards(rows_all, counts, check=False)
        env.rollout(acts)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50 * 1e6
    env.check_layouts()
    print(f"W={W}: shard sample + install + reset_apply + rollout = {dt:.1f} us per epoch; export {cap * K * 8 / 1e6:.2f} MB per rank, layout_size {env.layout_size}", flush=True)

import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
print(bench.precondition_clocks(dev)); env = bench.make_engine(2000, 0, 1); env.set_prefetch(200)
tapes = [bench.action_tape(200, 2000, k, dev) for k in range(4)]
bench.run_epochs(env, tapes, 30, None); torch.cuda.synchronize()
print("after epochs: api_loop_rate", bench.api_loop_rate(env, tapes[0], 2000) / 1e6, "M", env.prefetch_stats(), flush=True)
r = bench.roofline_rollout(2000, 200, 30, dev)
print("after roofline_rollout: api_loop_rate", bench.api_loop_rate(env, tapes[0], 2000) / 1e6, "M", flush=True)
r = bench.roofline_step(1 << 22, 30, dev)
print("after roofline_step 4M: api_loop_rate", bench.api_loop_rate(env, tapes[0], 2000) / 1e6, "M", flush=True)
r = bench.large_batch_fused(1 << 22, 32, dev)
print("after large_batch_fused: api_loop_rate", bench.api_loop_rate(env, tapes[0], 2000) / 1e6, "M", flush=True)
print("again: api_loop_rate", bench.api_loop_rate(env, tapes[0], 2000) / 1e6, "M", flush=True)
torch.cuda.empty_cache()
print("after empty_cache: api_loop_rate", bench.api_loop_rate(env, tapes[0], 2000) / 1e6, "M", flush=True)

import os, sys, time
sys.path.insert(0, os.getcwd())
import torch, bench
torch.cuda.set_device(0)
extra = [torch.cuda.Stream() for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 0)]   # other streams of the process
envs = []
for k in range(24):
    e = bench.make_engine(256, 0, 1, n_candidates=20000)
    e.aux_stream()
    envs.append(e)
    if k % 3 == 2:
        envs.pop(0).close()
print("created 24 engines,", len(extra), "extra torch streams")

import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
env = bench.make_engine(2000, 0, 1, n_candidates=100000)
env.reset()
act = bench.action_tape(1, 2000, 0, dev)[0]
for _ in range(50): env.step(act)
torch.cuda.synchronize()
n = 3000
t0 = time.perf_counter()
for _ in range(n):
    env.step(act); env.reset_done()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"HIP_FORCE_DEV_KERNARG={os.environ.get('HIP_FORCE_DEV_KERNARG')} GX_NO_SPECULATE={os.environ.get('GX_NO_SPECULATE')}: "
      f"step+reset_done {dt*1e6:.2f} us -> {2000/dt/1e6:.1f} M env-steps/s")

"""epoch = reset() [host-synchronised layout check] + rollout, as bench.py's extras time it, against the headline loop"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
torch.cuda.set_device(0); dev = torch.device("cuda", 0)
def run(robot=None, check=True, pf=None, n=30):
    env = bench.make_engine(2000, 0, 1, robot_base=robot)
    if pf is not None: env.set_prefetch(pf)
    A = env.action_space.shape[0]
    tape = bench.action_tape(200, 2000, 3, dev, act_dim=A)
    def epoch():
        env.reset(check=check); env.rollout(tape)
    for _ in range(4): epoch()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): epoch()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    env.close()
    return dt * 1e6
print("point   reset(check=True)  default prefetch: %.1f us" % run())
print("point   reset(check=True)  prefetch 200    : %.1f us" % run(pf=200))
print("point   reset(check=False) prefetch 200    : %.1f us" % run(check=False, pf=200))
print("swimmer reset(check=True)  default prefetch: %.1f us" % run('xmls/swimmer.xml'))
print("swimmer reset(check=False) default prefetch: %.1f us" % run('xmls/swimmer.xml', check=False))

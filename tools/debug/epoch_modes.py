import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
def timeit(fn, n):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
tape = bench.action_tape(200, 2000, 0, dev)
for mode, name in ((2, "lane-group"), (3, "split")):
    env = bench.make_engine(2000, 0, 1)
    env.set_path(mode); env.set_prefetch(200)
    def epoch():
        env.reset(check=False); env.rollout(tape)
    for _ in range(5): epoch()
    t = timeit(epoch, 50)
    act = tape[0]
    def steps():
        env.reset(check=False)
        for k in range(200):
            env.step(act); env.reset_done()
    steps(); ts = timeit(steps, 5)
    print(f"GX_SIDE_PRIORITY={os.environ.get('GX_SIDE_PRIORITY')} {name:10s}: epoch {t*1e3:.3f} ms -> {400000/t/1e6:.1f} M/s ; api loop epoch {ts*1e3:.3f} ms -> {400000/ts/1e6:.1f} M/s")
    env.close()

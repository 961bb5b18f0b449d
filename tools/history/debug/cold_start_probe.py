"""Per-epoch device time of the first epochs after engine creation / after an idle gap."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
env = bench.make_engine(bench.ENV_NUM, 0, 1)
env.set_prefetch(bench.EP_LEN)
tapes = [bench.action_tape(bench.EP_LEN, bench.ENV_NUM, k, dev) for k in range(4)]
torch.cuda.synchronize()


def series(label, n=60):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for k in range(n):
        bench.run_epochs(env, tapes, 1, None)
        ev[k + 1].record()
    torch.cuda.synchronize()
    d = [ev[k].elapsed_time(ev[k + 1]) for k in range(n)]
    print(f"{label}: " + " ".join(f"{x:.3f}" for x in d[:24]) + f" ... mean last 20 {sum(d[-20:])/20:.4f}", flush=True)


series("first")
series("immediately again")
time.sleep(0.05); series("after 50 ms idle")
time.sleep(1.0); series("after 1 s idle")
time.sleep(0.005); series("after 5 ms idle")
series("long", 400)

a = torch.randn(4096, 4096, device=dev); b = torch.randn(4096, 4096, device=dev)
for ms in (10, 30, 100):
    time.sleep(1.0)
    t0 = time.perf_counter()
    n = 0
    while (time.perf_counter() - t0) * 1e3 < ms:
        for _ in range(4):
            c = a @ b
        n += 4
        torch.cuda.synchronize()
    series(f"after 1 s idle + {ms} ms of fp32 matmul ({n} mm)")
x = torch.randn(1 << 24, device=dev)
time.sleep(1.0)
t0 = time.perf_counter()
while (time.perf_counter() - t0) * 1e3 < 50:
    for _ in range(8):
        y = torch.sin(x) * torch.cos(x) + torch.exp(-x * x)
    torch.cuda.synchronize()
series("after 1 s idle + 50 ms of elementwise sin/cos/exp")

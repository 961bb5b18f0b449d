"""Per-epoch device time of the driver-style region (5 warm-up + 20 timed epochs) right after bench.precondition_clocks."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
for rep in range(3):
    env = bench.make_engine(bench.ENV_NUM, 0, 1); env.set_prefetch(bench.EP_LEN)
    tapes = [bench.action_tape(bench.EP_LEN, bench.ENV_NUM, k, dev) for k in range(4)]
    torch.cuda.synchronize(); time.sleep(0.5)
    pc = bench.precondition_clocks(dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(26)]
    t0 = time.perf_counter()
    ev[0].record()
    for k in range(25):
        bench.run_epochs(env, tapes, 1, None)   # includes one check_layouts() sync per call: NOT what bench does
        ev[k + 1].record()
    torch.cuda.synchronize()
    d = [ev[k].elapsed_time(ev[k + 1]) for k in range(25)]
    print(f"rep {rep} ({pc['ms']} ms pre): " + " ".join(f"{x:.3f}" for x in d), flush=True)
    env.close()

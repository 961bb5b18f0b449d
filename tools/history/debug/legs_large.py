import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from guardx_amd import ResamplingError
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
for xml, A in (("xmls/ant.xml", 8), ("xmls/walker.xml", 10)):
    for N in [int(x) for x in (os.environ.get("GX_LEGS_N") or "2000,4096,8192").split(",")]:
        for mode, name in ((1, "thread"), (2, "group")):
            env = bench.make_engine(N, 0, 1, n_candidates=300000, robot_base=xml)
            env.set_path(mode); env.set_prefetch(-1)
            try: env.reset()
            except ResamplingError: pass
            T = 16
            tape = bench.action_tape(T, N, 0, dev, A)
            env.rollout(tape); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3): env.rollout(tape)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 3 / T
            print(f"{xml} N={N} {name}: {dt*1e6:.1f} us/step -> {N/dt/1e6:.1f} M env-steps/s")
            env.close()

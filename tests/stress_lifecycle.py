#!/usr/bin/env python3
"""Lifecycle stress (MI355X box): create / use / destroy many engines and run a long epoch loop, watching the
device memory the HIP library holds (hipMemGetInfo) for leaks and the run for hangs.

    python tests/stress_lifecycle.py
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from helpers import task_config, SWIMMER, ANT, WALKER  # noqa: E402
from guardx_amd import Engine  # noqa: E402


def free_mb():
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info()[0] / 2**20


def main():
    torch.cuda.set_device(0)
    torch.zeros(1, device="cuda")
    base = free_mb()
    marks = []
    for k in range(24):
        extra = [{}, SWIMMER, ANT, WALKER][k % 4]
        E = Engine(task_config(512, seed=k, **extra), n_candidates=200000)
        A = E.action_space.shape[0]
        E.reset()
        acts = torch.rand(20, 512, A, device="cuda") * 2 - 1
        E.rollout(acts)
        E.step(acts[0]); E.reset_done()
        E.close()
        del E, acts
        torch.cuda.empty_cache()
        if k % 8 == 7:
            marks.append(free_mb())
            print(f"after {k + 1} engines: free memory delta {marks[-1] - base:+.1f} MiB", flush=True)
    leak = marks[0] - marks[-1]      # growth after the runtime's one-time pools are warm
    E = Engine(task_config(2000, seed=0), n_candidates=1000000)
    tape = torch.rand(200, 2000, 2, device="cuda") * 2 - 1
    t0 = time.time()
    for ep in range(3000):
        E.reset(check=False)
        E.rollout(tape)
        if ep % 1000 == 999:
            torch.cuda.synchronize()
            E.check_layouts()
            print(f"epoch {ep + 1}: {time.time() - t0:.1f} s", flush=True)
    E.close()
    print(f"growth between the 8th and the 24th engine lifecycle: {leak:.1f} MiB")
    sys.exit(0 if leak < 64 else 1)


if __name__ == "__main__":
    main()

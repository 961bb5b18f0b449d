#!/usr/bin/env python3
"""Launch the step kernel `--launches` times on `--env-num` envs (for rocprofv3).

    rocprofv3 --kernel-trace --stats -- python3 tools/profile_step.py --env-num 4194304
    rocprofv3 --pmc FETCH_SIZE -- python3 tools/profile_step.py ...
    rocprofv3 --pmc WRITE_SIZE -- python3 tools/profile_step.py ...
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--env-num", type=int, default=1 << 22)
    ap.add_argument("--launches", type=int, default=20)
    ap.add_argument("--mode", default="step", choices=["step", "rollout"])
    ap.add_argument("--robot", default=None, help="e.g. xmls/ant.xml (default: point)")
    ap.add_argument("--repeat", type=int, default=1,
                    help="rollout mode: this many gx_rollout calls after bench.precondition_clocks (a single call from an idle "
                         "GPU runs inside the firmware's clock ramp: 100-119 us for the same kernel)")
    a = ap.parse_args()
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    from guardx_amd import ResamplingError
    env = bench.make_engine(a.env_num, 0, 1, n_candidates=200_000 if a.env_num > 100_000 else 1_000_000,
                            robot_base=a.robot)
    A = env.action_space.shape[0]
    try:
        env.reset()
    except ResamplingError:
        pass
    if a.mode == "step":
        act = bench.action_tape(1, a.env_num, 3, dev, A)[0]
        for _ in range(a.launches):
            env.step(act)
    else:
        acts = bench.action_tape(a.launches, a.env_num, 3, dev, A)
        if a.repeat > 1:
            bench.precondition_clocks(dev)   # (also lets the prefetch sampler that reset() started finish first)
        for _ in range(a.repeat):
            env.rollout(acts)
    torch.cuda.synchronize()
    print("done", a.env_num, a.launches)


if __name__ == "__main__":
    main()

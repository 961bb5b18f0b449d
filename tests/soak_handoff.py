#!/usr/bin/env python3
"""Soak of the default N > 1 epoch (MI355X box): W ranks -- Engine(shard=(r, W)) behind guardx_amd.dist.TapeHandoff with
the sharded layout sampler and every rank expanding every rank's tape -- in ONE process on one GPU, the collective played
by a "link" stream that copies the send buffers into every receive buffer (tests/test_slow_link.py's harness with a short
sleep), for many epochs and every robot: every expanded row of every rank, every reset observation and the installed
layout pools against ONE engine of W x N envs, bit for bit.  Exits non-zero on any mismatch.

    python tests/soak_handoff.py [point|swimmer|ant|walker] [W] [epochs] [N] [T]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import test_slow_link as tsl  # noqa: E402
from helpers import SWIMMER, ANT, WALKER  # noqa: E402


def main():
    robot = sys.argv[1] if len(sys.argv) > 1 else "point"
    W = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    epochs = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    N = int(sys.argv[4]) if len(sys.argv) > 4 else 256
    T = int(sys.argv[5]) if len(sys.argv) > 5 else 40
    extra = {"point": {}, "swimmer": SWIMMER, "ant": ANT, "walker": WALKER}[robot]
    M = max(120_000, 80 * N * W)      # the pool has to hold more valid layouts (~1.9 % of the candidates) than W x N envs
    t0 = time.time()
    full, ranks, hs, deliver = tsl._harness(torch, W, N, T, M, seed=41, sleep_cycles=2000, extra=extra)
    A = full.action_space.shape[0]
    rng = np.random.default_rng(9)
    bad = 0
    prev = None
    for ep in range(epochs):
        acts = torch.from_numpy(rng.uniform(-1, 1, (T, N * W, A)).astype(np.float32)).cuda()
        if ep:
            o_full = full.reset(check=False)
            for r, e in enumerate(ranks):
                if not torch.equal(e.reset(check=False), o_full[r * N:(r + 1) * N]):
                    bad += 1
                    print(f"MISMATCH reset epoch {ep} rank {r}", flush=True)
        *_, pk = full.rollout(acts, packed=True)
        for r, h in enumerate(hs):
            h.step(acts[:, r * N:(r + 1) * N].contiguous())
        deliver(ep)
        if prev is not None:                       # epoch ep - 1 has been expanded on every rank's hand-off stream
            for r, h in enumerate(hs):
                torch.cuda.current_stream().wait_stream(h.stream)
                for s in range(W):
                    if not torch.equal(h.rollout[s].view(torch.int32), prev[:, s * N:(s + 1) * N].contiguous().view(torch.int32)):
                        bad += 1
                        print(f"MISMATCH rows epoch {ep - 1} rank {r} shard {s}", flush=True)
        prev = pk
        if ep % 10 == 9:
            print(f"{robot} W={W}: {ep + 1} epochs, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
    for h in hs:
        h.drain()
    torch.cuda.synchronize()
    for r, h in enumerate(hs):
        for s in range(W):
            if not torch.equal(h.rollout[s].view(torch.int32), prev[:, s * N:(s + 1) * N].contiguous().view(torch.int32)):
                bad += 1
                print(f"MISMATCH rows last epoch rank {r} shard {s}", flush=True)
    for r, e in enumerate(ranks):
        hits, misses, _ = e.prefetch_stats()
        if hits != epochs - 2 or misses != 0 or hs[r].blocks_installed != epochs - 2 or hs[r].shard_skips:
            bad += 1
            print(f"MISMATCH pipeline rank {r}: hits {hits} misses {misses} blocks {hs[r].blocks_installed} skips {hs[r].shard_skips}", flush=True)
        e.check_layouts()
        if not np.array_equal(e.get_pool(64), full.get_pool(64)):
            bad += 1
            print(f"MISMATCH pool rank {r}", flush=True)
    for h in hs:
        h.close()
    print(f"soak_handoff {robot} W={W} N={N} T={T} epochs={epochs}: {W * W * epochs} expanded rollouts of {T * N} rows compared")
    print(f"TOTAL MISMATCHES: {bad}")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""One GPU plays rank 0 of W in the default multi-GPU epoch -- everything a rank does except the xGMI transfer.

Per epoch, exactly as bench.py --gpus W drives it (guardx_amd.dist.TapeHandoff):
  reset()                       takes the pool installed from the W export blocks (a prefetch hit)
  sample_shard_ahead(0, W)      1/W of the 1e6 layout candidates of the reset after next, side stream
  rollout_tape()                the serial dynamics pass of this rank's 2000 envs
  "all-gather"                  here: device copies on a stream of their own -- this rank's buffer into slot 0 and, for
                                the other W - 1 ranks, a copy of the own tape (the same amount of expansion work) plus
                                the export block that rank WOULD have sent, sampled beforehand by a twin engine replaying
                                the same key schedule (the keys are data independent, engine.py:431) -- so the installed
                                pools are the true ones and every reset's layout check holds
  install_shards + W (or 1) observation passes on the hand-off stream, one epoch later

What it measures is the GPU time of one rank's epoch at world size W; what it cannot measure is the link.  The model
printed with it: epoch(W) = max(measured epoch, bytes received / bus bandwidth), weak-scaling efficiency =
epoch(1 GPU, own sampler) / epoch(W).

    python tools/rehearse_rank.py [--world 8] [--epochs 40] [--robot xmls/point.xml] [--json out.json]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import bench  # noqa: E402
from guardx_amd import Engine  # noqa: E402
from guardx_amd.dist import TapeHandoff  # noqa: E402


class _CopyWork:
    def __init__(self, stream):
        self.ev = torch.cuda.Event()
        self.ev.record(stream)

    def wait(self):
        torch.cuda.current_stream().wait_event(self.ev)


class RehearsalHandoff(TapeHandoff):
    """TapeHandoff playing rank 0 of `world` on one GPU; `blocks[epoch][s - 1]` = the export block rank s sends in `epoch`."""

    def __init__(self, env, T, world, blocks, expand):
        super().__init__(env, T, sharded_sampler=True, expand=expand, _play=(0, world))
        self.blocks, self.epoch = blocks, 0
        self.comm = torch.cuda.Stream(device=env.device)

    def _gather(self, i, buf):
        # three copy launches per epoch, whatever the world size (a real collective is ONE call: the stand-in must not make
        # the rank's epoch host-bound -- with one copy per shard and block, 16 launches at W = 8, it did on slow hosts)
        self.comm.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.comm):
            recv = self.recv[i].view(self.world, self.n)
            recv[0].copy_(buf, non_blocking=True)
            if self.world > 1:
                recv[1:, :self.n_tape].copy_(buf[:self.n_tape].unsqueeze(0).expand(self.world - 1, self.n_tape), non_blocking=True)
                blk = self.blocks[self.epoch]
                if blk is not None:
                    recv[1:, self.off_block:].copy_(blk, non_blocking=True)
            work = _CopyWork(self.comm)
        self.epoch += 1
        self.bytes_received += (self.world - 1) * self.n * 4
        return work


def make(world, robot):
    cfg = dict(bench.TASK, robot_base=robot)
    cfg.update(env_num=bench.ENV_NUM, _seed=0, num_steps=bench.EP_LEN, device_id=torch.cuda.current_device())
    e = Engine(cfg, shard=(0, world) if world > 1 else None)
    e.set_prefetch(bench.EP_LEN)
    return e


def other_ranks_blocks(world, robot, epochs, tapes, cap):
    """the blocks ranks 1..W-1 send in epochs 0..epochs-1: a twin of rank 0's engine replays its key schedule"""
    twin = make(world, robot)
    twin.reset()
    twin.set_layout_source('shards')
    nb = twin.shard_block_floats(cap)
    out = [None]                      # epoch 0 carries no block (TapeHandoff samples the first one after its first tape)
    for ep in range(epochs):
        if ep:
            twin.reset(check=False)
        twin.rollout_tape(tapes[ep % len(tapes)])
        row = torch.zeros(max(world - 1, 1), nb, device=twin.device)   # row s - 1: the block of rank s
        for s in range(1, world):     # what rank s samples at the end of its step(ep): travels with the tape of ep + 1
            twin.sample_shard_ahead(s, world, row[s - 1], cap, resets_ahead=3)
        out.append(row if world > 1 else None)
    twin.shard_join()
    torch.cuda.synchronize()
    twin.check_layouts()
    twin.close()
    return out


def rehearse(world, robot, epochs, warmup, expand, device):
    env = make(world, robot)
    A = env.action_space.shape[0]
    tapes = [bench.action_tape(bench.EP_LEN, bench.ENV_NUM, k, device, A) for k in range(4)]
    env.reset()
    probe = TapeHandoff(env, bench.EP_LEN, sharded_sampler=True, _play=(0, world))   # sizes the blocks
    cap = probe.cap
    probe.close()
    del probe
    blocks = other_ranks_blocks(world, robot, warmup + epochs, tapes, cap)
    h = RehearsalHandoff(env, bench.EP_LEN, world, blocks, expand)
    assert h.cap == cap

    def run(n, first):
        for ep in range(n):
            if ep or not first:
                env.reset(check=False)
            h.step(tapes[(h.epoch) % len(tapes)])
        h.drain()
    run(warmup, True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(epochs, False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / epochs
    env.check_layouts()                      # every pool was the true one
    hits, misses, _ = env.prefetch_stats()
    out = {"ms_per_epoch": round(dt * 1e3, 4), "env_steps_per_s_this_rank": round(bench.ENV_NUM * bench.EP_LEN / dt, 1),
           "prefetch_hits": hits, "prefetch_misses": misses, "blocks_installed": h.blocks_installed,
           "bytes_received_per_epoch": (world - 1) * h.n * 4, "block_rows_cap": cap, "layout_size": env.layout_size}
    h.close()
    env.close()
    return out


def single(robot, epochs, warmup, device):
    env = make(1, robot)
    A = env.action_space.shape[0]
    tapes = [bench.action_tape(bench.EP_LEN, bench.ENV_NUM, k, device, A) for k in range(4)]
    bench.run_epochs(env, tapes, warmup, None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    bench.run_epochs(env, tapes, epochs, None)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / epochs
    env.close()
    return {"ms_per_epoch": round(dt * 1e3, 4), "env_steps_per_s": round(bench.ENV_NUM * bench.EP_LEN / dt, 1)}


def model(one, rank, world, gbps=(310.0, 200.0)):
    out = {}
    for bw in gbps:
        t_link = rank["bytes_received_per_epoch"] / (bw * 1e9) * 1e3
        t = max(rank["ms_per_epoch"], t_link)
        out[f"at_{int(bw)}GBps"] = {"allgather_ms": round(t_link, 4), "epoch_ms": round(t, 4),
                                    "weak_scaling_efficiency": round(one["ms_per_epoch"] / t, 4),
                                    "env_steps_per_s_all_ranks": round(world * bench.ENV_NUM * bench.EP_LEN / (t * 1e-3), 1)}
    # the bandwidths one SCALE run can be read against: below `link_bound_below_GBps` the all-gather, not the GPU, sets
    # the rank's epoch; at `break_even_GBps` the sharded epoch equals the one-GPU epoch (weak-scaling efficiency 1.0)
    nbytes = rank["bytes_received_per_epoch"]
    out["link_bound_below_GBps"] = round(nbytes / (rank["ms_per_epoch"] * 1e-3) / 1e9, 1)
    out["break_even_GBps"] = round(nbytes / (one["ms_per_epoch"] * 1e-3) / 1e9, 1)
    out["efficiency_if_gpu_bound"] = round(one["ms_per_epoch"] / rank["ms_per_epoch"], 4)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--epochs", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--robot", default="xmls/point.xml")
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    device = torch.device("cuda", torch.cuda.current_device())
    bench.precondition_clocks(device)
    one = single(args.robot, args.epochs, args.warmup, device)
    from guardx_amd import _native
    res = {"what": f"one GPU playing rank 0 of {args.world} (tools/rehearse_rank.py): GPU time of a rank's epoch, no link",
           "library_build": _native.load().gx_build_id().decode(),
           "robot": args.robot, "world": args.world, "epochs": args.epochs,
           "one_gpu_own_sampler": one}
    for expand in ("all", "local"):
        r = rehearse(args.world, args.robot, args.epochs, args.warmup, expand, device)
        r["model"] = model(one, r, args.world)
        res["expand_" + expand] = r
    print(json.dumps(res, indent=1))
    if args.json:
        with open(args.json, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()

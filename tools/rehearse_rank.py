#!/usr/bin/env python3
"""One GPU plays rank 0 of W in the default multi-GPU epoch -- everything a rank does except the xGMI transfer.

Per epoch, exactly as bench.py --gpus W drives it (guardx_amd.dist.TapeHandoff):
  reset()                       takes the pool installed from the W export blocks (a prefetch hit)
  sample_shard_ahead(0, W)      1/W of the 1e6 layout candidates of the reset after next, side stream
  rollout_tape()                the serial dynamics pass of this rank's 2000 envs
  "all-gather"                  here: this rank's buffer is copied into slot 0 of a receive buffer whose other W - 1
                                slots were STAGED before the timed region: a copy of the own tape of that epoch (the
                                same amount of expansion work; written by a twin engine replaying the same key
                                schedule and actions -- the keys are data independent, engine.py:431) plus the export
                                block that rank WOULD have sent, sampled by the same twin -- so the installed pools
                                are the true ones and every reset's layout check holds.  The expansion is ordered
                                behind this epoch's dynamics pass exactly as behind a collective.
                                (GX_REHEARSAL_COMM=copy: the round-4 stand-in, W - 1 device copies per epoch on a
                                stream of normal priority; =copy-low: the same at the least priority.  Both distort
                                what they stand in for: the Ant's 380 MB blit of ~190 000 workgroups queues ahead of
                                reset_apply's 32 waves -- 15 us alone, 150-230 us beside it -- or, at low priority,
                                arrives late and holds up the block's install.  A real all-gather is a few
                                persistent workgroups per peer and writes arriving over xGMI.)
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


STAND_IN = os.environ.get("GX_REHEARSAL_COMM", "staged")      # "staged" | "copy" | "copy-low": see the docstring
assert STAND_IN in ("staged", "copy", "copy-low"), STAND_IN


class _CopyWork:
    def __init__(self, stream):
        self.ev = torch.cuda.Event()
        self.ev.record(stream)

    def wait(self):
        torch.cuda.current_stream().wait_event(self.ev)


def low_priority_stream(device):
    """A stream of the least urgent priority the device offers.  The stand-in for the collective is a device copy of up to
    W - 1 tapes (the Ant: 380 MB, a blit of ~190 000 workgroups); a real all-gather is a few persistent workgroups per peer
    plus writes arriving over xGMI.  On a stream of normal priority the blit's workgroups queue ahead of reset_apply's 32
    waves (15 us alone, 150-230 us beside it: gpurun_out/trace_ant8_kernel_trace.csv): an artefact of the stand-in, not of
    the rank's epoch.  At the lowest priority it takes the wave slots the epoch leaves free, as the sampler's streams do."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    lo, hi = ctypes.c_int(0), ctypes.c_int(0)
    if hip.hipDeviceGetStreamPriorityRange(ctypes.byref(lo), ctypes.byref(hi)) != 0:
        raise RuntimeError("hipDeviceGetStreamPriorityRange failed")
    s = ctypes.c_void_p()
    if hip.hipStreamCreateWithPriority(ctypes.byref(s), ctypes.c_uint(1), lo) != 0:      # 1 = hipStreamNonBlocking
        raise RuntimeError("hipStreamCreateWithPriority failed")
    return torch.cuda.ExternalStream(s.value, device=device)


class RehearsalHandoff(TapeHandoff):
    """TapeHandoff playing rank 0 of `world` on one GPU; `blocks[epoch][s - 1]` = the export block rank s sends in `epoch`."""

    def __init__(self, env, T, world, blocks, expand, staged=None):
        super().__init__(env, T, sharded_sampler=True, expand=expand, _play=(0, world))
        self.blocks, self.staged, self.epoch = blocks, staged, 0
        self.comm = (low_priority_stream(env.device) if STAND_IN == "copy-low" else torch.cuda.Stream(device=env.device))

    def _gather(self, i, buf):
        # at most three copy launches per epoch, whatever the world size (a real collective is ONE call: the stand-in must
        # not make the rank's epoch host-bound -- with one copy per shard and block, 16 launches at W = 8, it did)
        if self.staged is not None:
            self.recv[i] = self.staged[self.epoch].view(-1)    # slots 1.. W-1 were written before the timed region
        self.comm.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.comm):
            recv = self.recv[i].view(self.world, self.n)
            recv[0].copy_(buf, non_blocking=True)
            if self.world > 1 and self.staged is None:
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


def other_ranks_blocks(world, robot, epochs, tapes, cap, stage=None):
    """the blocks ranks 1..W-1 send in epochs 0..epochs-1: a twin of rank 0's engine replays its key schedule.
    stage = (n, n_tape, off_block) of the hand-off: also returns, per epoch, the receive buffer [W][n] with the slots of
    ranks 1..W-1 filled in (the twin's tape of that epoch -- bit for bit rank 0's -- and their blocks)."""
    twin = make(world, robot)
    twin.reset()
    twin.set_layout_source('shards')
    nb = twin.shard_block_floats(cap)
    out = [None]                      # epoch 0 carries no block (TapeHandoff samples the first one after its first tape)
    staged = None
    if stage is not None and world > 1:
        n, n_tape, off_block = stage
        staged = [torch.zeros(world, n, device=twin.device) for _ in range(epochs)]
    for ep in range(epochs):
        if ep:
            twin.reset(check=False)
        if staged is not None:
            twin.rollout_tape(tapes[ep % len(tapes)], out=staged[ep][1, :n_tape])
            if world > 2:
                staged[ep][2:, :n_tape].copy_(staged[ep][1, :n_tape].unsqueeze(0).expand(world - 2, n_tape))
        else:
            twin.rollout_tape(tapes[ep % len(tapes)])
        row = torch.zeros(max(world - 1, 1), nb, device=twin.device)   # row s - 1: the block of rank s
        for s in range(1, world):     # what rank s samples at the end of its step(ep): travels with the tape of ep + 1
            twin.sample_shard_ahead(s, world, row[s - 1], cap, resets_ahead=3)
        out.append(row if world > 1 else None)
    twin.shard_join()
    torch.cuda.synchronize()
    if staged is not None:            # (only now: the blocks were written on the twin's side stream)
        for ep in range(epochs):
            if out[ep] is not None:
                staged[ep][1:, off_block:].copy_(out[ep])
        torch.cuda.synchronize()
    twin.check_layouts()
    twin.close()
    return out, staged


def rehearse(world, robot, epochs, warmup, expand, device):
    env = make(world, robot)
    A = env.action_space.shape[0]
    tapes = [bench.action_tape(bench.EP_LEN, bench.ENV_NUM, k, device, A) for k in range(4)]
    env.reset()
    probe = TapeHandoff(env, bench.EP_LEN, sharded_sampler=True, _play=(0, world))   # sizes the blocks
    cap, stage = probe.cap, (probe.n, probe.n_tape, probe.off_block)
    probe.close()
    del probe
    blocks, staged = other_ranks_blocks(world, robot, warmup + epochs, tapes, cap, stage if STAND_IN == "staged" else None)
    h = RehearsalHandoff(env, bench.EP_LEN, world, blocks, expand, staged)
    assert h.cap == cap and (h.n, h.n_tape, h.off_block) == stage

    host = {"reset": 0.0, "step": 0.0, "n": 0}      # host time inside the calls (they only enqueue): is the host ahead?

    def run(n, first):
        for ep in range(n):
            t_a = time.perf_counter()
            if ep or not first:
                env.reset(check=False)
            t_b = time.perf_counter()
            h.step(tapes[(h.epoch) % len(tapes)])
            t_c = time.perf_counter()
            host["reset"] += t_b - t_a; host["step"] += t_c - t_b; host["n"] += 1
        h.drain()
    run(warmup, True)
    torch.cuda.synchronize()
    host.update(reset=0.0, step=0.0, n=0)
    t0 = time.perf_counter()
    run(epochs, False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / epochs
    env.check_layouts()                      # every pool was the true one
    hits, misses, _ = env.prefetch_stats()
    out = {"ms_per_epoch": round(dt * 1e3, 4), "env_steps_per_s_this_rank": round(bench.ENV_NUM * bench.EP_LEN / dt, 1),
           "prefetch_hits": hits, "prefetch_misses": misses, "blocks_installed": h.blocks_installed,
           "bytes_received_per_epoch": (world - 1) * h.n * 4, "block_rows_cap": cap, "layout_size": env.layout_size,
           "host_ms_per_epoch": {"reset": round(host["reset"] / host["n"] * 1e3, 4), "step": round(host["step"] / host["n"] * 1e3, 4)}}
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
           "robot": args.robot, "world": args.world, "epochs": args.epochs, "link_stand_in": STAND_IN,
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

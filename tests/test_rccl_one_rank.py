"""The multi-GPU path's RCCL branches executed for real on ONE GPU: a fresh child process forms a
one-rank "nccl" process group with GX_FORCE_DIST=1, so that nothing short-circuits at world size 1, and drives
TapeHandoff / ShardedReset / barrier / max_over_ranks / the packed hand-off over it (tests/rccl_one_rank_child.py).
The parent only spawns, reads the report and counts RCCL's own log lines -- it never joins a process group itself."""
import json
import os
import re
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _dist_env(**extra):
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(_free_port()), GX_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0",
               NCCL_DEBUG="INFO", NCCL_DEBUG_SUBSYS="INIT,COLL")
    env.pop("GX_DIST_BACKEND", None)
    env.update(extra)
    return env


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_dist_path_over_a_one_rank_rccl_group(tmp_path):
    """every `nccl` branch of guardx_amd/dist.py, with the collective forced at world size 1; rows equal to a twin
    engine's rollout(packed=True) and to the CPU checker (asserted inside the child)"""
    report = tmp_path / "report.json"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_one_rank_child.py"), str(report)],
                         env=_dist_env(), capture_output=True, text=True, timeout=800)
    log = out.stdout + out.stderr
    keep = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(keep):                      # evidence for profiles/ (the GPU box merges gpurun_out/ back)
        with open(os.path.join(keep, "rccl_one_rank.log"), "w") as f:
            f.write(log[-200000:])
    assert out.returncode == 0, log[-4000:]
    assert "RCCL_ONE_RANK_OK" in out.stdout
    rep = json.loads(report.read_text())
    if os.path.isdir(keep):
        with open(os.path.join(keep, "rccl_one_rank_report.json"), "w") as f:
            json.dump(rep, f, indent=1)
    assert rep["backend"] == "nccl" and rep["tape_handoff"]["blocks_installed"] == rep["tape_handoff"]["epochs"] - 2
    # RCCL itself logged its initialisation and the collectives it was handed
    assert re.search(r"NCCL INFO.*(Init COMPLETE|ncclCommInitRank|comm 0x)", log), log[-3000:]
    n_ag = len(re.findall(r"NCCL INFO AllGather", log))
    n_ar = len(re.findall(r"NCCL INFO AllReduce", log))
    # 1 (all_gather_rollout) + 5 (packed ring) + 8 + 3 (two tape hand-offs) + 3 x 2 (ShardedReset)
    assert n_ag >= 23, (n_ag, log[-3000:])
    assert n_ar >= 1, (n_ar, log[-3000:])


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_bench_one_gpu_over_a_forced_one_rank_rccl_group():
    """`GX_FORCE_DIST=1 python bench.py --gpus 1`: the N > 1 bench path (process group, barrier + max-over-ranks timing,
    tape hand-off with the sharded sampler, all four legs) over RCCL with one rank; one JSON line"""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "2",
                          "--no-extras", "--no-cpu-baseline"], env=_dist_env(NCCL_DEBUG="WARN"), capture_output=True,
                         text=True, timeout=800)
    assert out.returncode == 0, (out.stdout + out.stderr)[-4000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    line = json.loads(lines[0])
    keep = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(keep):
        with open(os.path.join(keep, "bench_force_dist_one_rank.json"), "w") as f:
            f.write(lines[0] + "\n")
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["forced_dist"]["backend"] == "nccl"
    assert line["config"]["layout_sampler"].startswith("sharded")
    assert line["stepping_only"]["value"] > 0
    assert line["legs"]["unsharded_sampler"]["value"] > 0 and line["legs"]["local_expand"]["value"] > 0

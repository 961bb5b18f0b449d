"""Multi-GPU glue: one process per GPU, environments sharded with no data-path
collective; the only exchange is the once-per-epoch rollout hand-off to the
learner -- one all-gather (RCCL over xGMI on GPUs, gloo on CPU) of the packed
per-rank rollout shard.  The reference has no counterpart:
it runs on a single device (engine.py:100, trpo.py:21)."""
import os

import torch
import torch.distributed as dist

ROLLOUT_FIELDS = ("obs", "act", "rew", "cost", "done")


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # GX_DIST_BACKEND=gloo: rehearse the multi-rank path on a box with fewer GPUs than ranks
            backend = os.environ.get("GX_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def pack_rollout(obs, act, rew, cost, done):
    """(T,N,D) (T,N,A) (T,N) (T,N) (T,N) -> one (T,N,D+A+3) tensor: one big collective
    instead of five small ones."""
    return torch.cat([obs, act, rew.unsqueeze(-1), cost.unsqueeze(-1), done.unsqueeze(-1)], dim=-1)


def unpack_rollout(packed, obs_dim, act_dim):
    o = packed[..., :obs_dim]
    a = packed[..., obs_dim:obs_dim + act_dim]
    r, c, d = (packed[..., obs_dim + act_dim + k] for k in range(3))
    return dict(obs=o, act=a, rew=r, cost=c, done=d)


def all_gather_rollout(packed, out=None):
    """All-gather the per-rank packed shard -> (world, T, N, W).  World size 1: a view."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return packed.unsqueeze(0)
    world = dist.get_world_size()
    shape = tuple(packed.shape)
    if out is None:
        out = torch.empty((world,) + shape, dtype=packed.dtype, device=packed.device)
    # concatenated-along-dim-0 form: accepted by both RCCL and gloo
    dist.all_gather_into_tensor(out.view((world * shape[0],) + shape[1:]), packed.contiguous())
    return out


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        if dist.get_backend() == "nccl":   # name the device: RCCL otherwise guesses it from the rank
            dist.barrier(device_ids=[torch.cuda.current_device()])
        else:
            dist.barrier()


def max_over_ranks(value, device):
    if dist.is_initialized() and dist.get_backend() != "nccl":
        device = "cpu"                      # gloo rehearsal
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())

"""Data-parallel rollout plumbing: contiguous env shards, a counter-based action stream that does not depend
on the number of GPUs, and the single collective of the path -- one gather of episode returns per rollout batch
(RCCL over xGMI on MI355X; gloo in the CPU tests).

The reference has no parallelism at all (n_parallel=1, rllab/envs/trpo_cassie.py:48); environments are independent,
so the path shards with NO data-path collective inside a step.
"""
import os

import torch
import torch.distributed as dist

MASK64 = (1 << 64) - 1


def shard_bounds(n_total, rank, world):
    """GPU g owns the contiguous global env range [g*n/G, (g+1)*n/G)."""
    lo = (n_total * rank) // world
    hi = (n_total * (rank + 1)) // world
    return lo, hi


def _mix64(x):
    """splitmix64 finaliser on int64 tensors (wrap-around arithmetic)."""
    x = (x ^ (x >> 30) & 0x3FFFFFFFF) * -4658895280553007687  # 0xBF58476D1CE4E5B9
    x = (x ^ (x >> 27) & 0x1FFFFFFFFF) * -7723592293110705685  # 0x94D049BB133111EB
    return x ^ (x >> 31) & 0x1FFFFFFFF


def counter_uniform(seed, env_ids, step, dim, device=None, dtype=torch.float64):
    """U[0,1) keyed by (seed, GLOBAL env id, step, component): the same env sees the same action stream
    whichever rank owns it, so results are independent of the GPU count."""
    env_ids = torch.as_tensor(env_ids, dtype=torch.int64, device=device)
    comp = torch.arange(dim, dtype=torch.int64, device=env_ids.device)
    key = (env_ids[:, None] * 1000003 + comp[None, :]) * 2654435761 + int(step) * 40503 + int(seed) * 7919
    h = _mix64(_mix64(key) + 0x632BE59BD9B4E019)
    mant = (h >> 11) & ((1 << 53) - 1)
    return mant.to(dtype) * (1.0 / (1 << 53))


def counter_normal(seed, env_ids, step, dim, device=None, dtype=torch.float64):
    """N(0,1) keyed like counter_uniform (Box-Muller on two independent counter streams): the exploration noise of the TRPO
    sampler, so that a policy rollout -- like the random-action stream -- does not depend on how the envs are sharded."""
    u1 = counter_uniform(seed, env_ids, 2 * int(step), dim, device=device)
    u2 = counter_uniform(seed, env_ids, 2 * int(step) + 1, dim, device=device)
    r = torch.sqrt(-2.0 * torch.log1p(-u1))   # 1 - u1 in (0, 1]
    return (r * torch.cos(6.283185307179586 * u2)).to(dtype)


def random_actions(seed, env_ids, step, low, high, device=None):
    u = counter_uniform(seed, env_ids, step, len(low), device=device)
    low = torch.as_tensor(low, dtype=torch.float64, device=u.device)
    high = torch.as_tensor(high, dtype=torch.float64, device=u.device)
    return low + (high - low) * u


def local_device(local_rank):
    """HIP device of a local rank: the rank itself, unless CASSIE_DEVICE_MAP ("0,0,1,...") overrides it (test hook: several ranks
    on one GPU to exercise the N > 1 launch path on a 1-GPU box)."""
    m = os.environ.get("CASSIE_DEVICE_MAP", "")
    if m:
        ids = [int(x) for x in m.split(",")]
        return ids[local_rank % len(ids)]
    return local_rank


def init_distributed(backend=None):
    """One process per GPU; rendezvous from the torchrun environment (MASTER_ADDR/PORT, RANK, WORLD_SIZE)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # CASSIE_FORCE_PROCESS_GROUP=1: build the process group even for a single rank, so that a 1-GPU box exercises the RCCL
    # communicator and the gather exactly as an N-rank run does (tests/test_gpu_bench.py)
    force = os.environ.get("CASSIE_FORCE_PROCESS_GROUP", "") == "1" and "MASTER_ADDR" in os.environ
    if (world > 1 or force) and not dist.is_initialized():
        # One node, rendezvous on loopback: keep the backends' socket bootstrap on `lo` too.  Left to themselves gloo and RCCL look
        # their interfaces up through the host name, which a container may not be able to resolve -- measured on one MI355X box:
        # 322 s for the process group of a single rank, 5 s on another (this was the driver's 643 s GPU suite of round 3).
        if os.environ.get("MASTER_ADDR", "") in ("127.0.0.1", "localhost", "::1"):
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
            os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
            # c10d's store looks up the host name of every client it accepts (for a log line); where the container's resolver cannot reach a
            # name server each of those lookups waits out the resolver's time-out (glibc default: 5 s x 2 attempts per server)
            os.environ.setdefault("RES_OPTIONS", "timeout:1 attempts:1")
        if backend is None:
            backend = os.environ.get("CASSIE_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")  # "nccl" is RCCL on ROCm
        if backend == "nccl":
            torch.cuda.set_device(local_device(local_rank))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def gather_returns(local_returns):
    """The one collective of the path: all ranks receive the concatenated per-env episode returns
    (N/G float64 per rank; 32 KiB per rank at 4096 envs/GPU -> latency-bound on xGMI)."""
    if not dist.is_initialized():
        return local_returns.clone()
    world = dist.get_world_size()
    if dist.get_backend() == "gloo" and local_returns.is_cuda:  # test hook (CASSIE_BACKEND=gloo): stage through the host
        parts = [torch.empty(local_returns.numel(), dtype=local_returns.dtype) for _ in range(world)]
        dist.all_gather(parts, local_returns.detach().cpu().contiguous())
        return torch.cat(parts).to(local_returns.device)
    out = torch.empty(world * local_returns.numel(), dtype=local_returns.dtype, device=local_returns.device)
    dist.all_gather_into_tensor(out, local_returns.contiguous())
    return out


def max_over_ranks(value, device=None):
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=None if dist.get_backend() == "gloo" else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def min_over_ranks(value, device=None):
    """all_reduce(MIN): e.g. an `ok` flag every rank must agree on before a stage made of collectives is entered."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=None if dist.get_backend() == "gloo" else device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return float(t.item())


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()

"""N>1 path on CPU: world_size-2 gloo run of the sharding, the GPU-count-independent action stream and the single
return gather (cassierl_amd/rollout.py).  The env kernels are not involved (no GPU here)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cassierl_amd import rollout as R


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, n_total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    r, _, w = R.init_distributed("gloo")
    lo, hi = R.shard_bounds(n_total, r, w)
    ids = torch.arange(lo, hi)
    low, high = np.radians([-50, -164, -140] * 2), np.radians([80, -37, -30] * 2)
    returns = torch.zeros(hi - lo, dtype=torch.float64)
    for step in range(3):
        a = R.random_actions(1, ids, step, low, high)
        returns += a.sum(dim=1)  # stand-in for the per-env reward of that step
    allr = R.gather_returns(returns)
    t = R.max_over_ranks(10.0 + r)
    R.barrier()
    if r == 0:
        q.put((allr.numpy(), t))
    dist.destroy_process_group()


def test_shard_gather_world2():
    n_total = 64
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_total, q)) for r in range(2)]
    for p in procs:
        p.start()
    allr, tmax = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # single-process reference: the same global env ids give the same stream regardless of the world size
    low, high = np.radians([-50, -164, -140] * 2), np.radians([80, -37, -30] * 2)
    ref = torch.zeros(n_total, dtype=torch.float64)
    for step in range(3):
        ref += R.random_actions(1, torch.arange(n_total), step, low, high).sum(dim=1)
    assert np.array_equal(allr, ref.numpy())
    assert tmax == 11.0


def test_shard_bounds_cover_exactly():
    for n in (1, 7, 4096, 65536, 524288):
        for w in (1, 2, 4, 8):
            spans = [R.shard_bounds(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))


def test_counter_uniform_is_uniform_and_keyed():
    u = R.counter_uniform(1, torch.arange(20000), 0, 6).numpy()
    assert 0.0 <= u.min() and u.max() < 1.0
    assert abs(u.mean() - 0.5) < 5e-3 and abs(u.var() - 1 / 12) < 5e-3
    assert abs(np.corrcoef(u[:, 0], u[:, 1])[0, 1]) < 0.03 and abs(np.corrcoef(u[:-1, 0], u[1:, 0])[0, 1]) < 0.03
    v = R.counter_uniform(1, torch.arange(20000), 1, 6).numpy()
    assert not np.array_equal(u, v)
    assert np.array_equal(R.counter_uniform(1, torch.arange(100, 200), 0, 6).numpy(), u[100:200])

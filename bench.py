#!/usr/bin/env python3
"""Headline benchmark: env-steps/sec of the batched Cassie2d rollout (BASELINE.json metric).

Workload at N=1 = BASELINE.json configs[1]: 4096 parallel Cassie2d envs (rllab/envs/cassie2d.py semantics: PD control
mode, n=10 substeps per Env.step, reward / termination / auto-reset), random-policy rollout, one MI355X.
One bench "step" = one vectorised Env.step over the whole batch = 4096 env-steps (40960 physics substeps).
Weak scaling: every rank owns 4096 envs (global ids rank*4096 ..), no collective inside a step, ONE RCCL gather of the
per-env returns at the end of the rollout batch (inside the timed region).

Inputs (actions of a uniform random policy over the PD action box, counter-based stream keyed by the global env id)
are generated on the device before the timed region: the timed region starts with everything resident in HBM.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ENVS_PER_GPU = 4096
ALGO_BYTES_PER_ENV_STEP = 697 + 208  # state+action in, state+obs+reward+done out, + persisted warm-start vector
FLOP_PER_SUBSTEP = 60e3              # SURVEY.md 8(d) estimate (3-D formulation); the planar kernel needs fewer
HBM_PEAK_GBPS = 8000.0               # MI355X_MICROARCH.md: 8 TB/s spec
FP64_VALU_PEAK_TFLOPS = 78.6         # 256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz


def cpu_baseline(traj, cores, budget_s=12.0):
    """Oracle (C restatement, OpenMP over envs) on the host cores of this box, bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as O
    from cassierl_amd import rollout as R
    import torch
    n = 16 * cores
    envs = [O.OracleEnv("walk", "PD", traj=traj) for _ in range(n)]
    for e in envs:
        e.reset()
    low, high = np.radians([-50, -164, -140] * 2), np.radians([80, -37, -30] * 2)
    ids = torch.arange(n)
    steps, t_used = 0, 0.0
    O.envs_step(envs, R.random_actions(1, ids, 0, low, high).numpy(), 10, True, cores)  # warm-up
    while t_used < budget_s and steps < 400:
        a = R.random_actions(1, ids, steps + 1, low, high).numpy()
        t0 = time.perf_counter()
        O.envs_step(envs, a, 10, True, cores)
        t_used += time.perf_counter() - t0
        steps += 1
    return dict(value=n * steps / t_used, unit="env-steps/s", cores=cores, kind="port",
                sample="%d envs x %d Env.steps (walk env, PD, random policy) in %.1f s, OpenMP over envs" % (n, steps, t_used))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--envs-per-gpu", type=int, default=ENVS_PER_GPU)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    from cassierl_amd import rollout as R
    from cassierl_amd.vec_env import CassieVecEnv

    rank, local_rank, world = R.init_distributed()
    assert world == max(1, args.gpus) or world == 1, "launch with torch.distributed.run --nproc-per-node N for N>1"
    dev = local_rank if world > 1 else 0
    torch.cuda.set_device(dev)
    n_local = args.envs_per_gpu
    lo, hi = R.shard_bounds(n_local * world, rank, world)
    traj_npz = np.load(os.path.join(ROOT, "tests", "golden", "traj2d.npz"))
    traj = dict(time=traj_npz["time"], qpos=traj_npz["qpos"])

    env = CassieVecEnv(n_local, kind="walk", control_mode="PD", n_substeps=10, auto_reset=True, device=dev)
    env.set_trajectory(traj["time"], traj["qpos"])
    env.use_torch_stream()
    out = env.alloc()
    ids = torch.arange(lo, hi, device="cuda:%d" % dev)
    low, high = env.action_space.low, env.action_space.high
    total = args.warmup + args.steps
    actions = [R.random_actions(1, ids, t, low, high) for t in range(total)]  # resident in HBM before timing
    returns = torch.zeros(n_local, dtype=torch.float64, device="cuda:%d" % dev)
    env.reset(out)
    for t in range(args.warmup):
        env.step(actions[t], out)
    torch.cuda.synchronize()
    R.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(args.warmup, total):
        _, rew, _ = env.step(actions[t], out)
        returns += rew
    all_returns = R.gather_returns(returns)  # the single collective of the rollout batch (RCCL over xGMI)
    torch.cuda.synchronize()
    R.barrier()
    torch.cuda.synchronize()
    elapsed = R.max_over_ranks(time.perf_counter() - t0, device="cuda:%d" % dev)

    # kernel-only time of the dominant kernel, HIP events on the stream it is launched on
    kernel_ms = env.time_steps(actions[-1], args.steps, out)
    q, v = env.get_state_host()
    finite = bool(np.isfinite(q).all() and np.isfinite(v).all())

    if rank == 0:
        n_total = n_local * world
        value = n_total * args.steps / elapsed
        achieved_gbps = ALGO_BYTES_PER_ENV_STEP * n_local / (kernel_ms * 1e-3) / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": "env-steps/sec (whole node) for Cassie2d batched rollout", "value": value, "unit": "env-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "configs[1]: %d parallel Cassie2d envs per GPU, random-policy rollout, PD mode, 10 substeps/step, "
                                   "walk env reward/done/auto-reset" % n_local,
                       "envs_per_gpu": n_local, "envs_total": n_total, "substeps_per_env_step": 10, "parallelism": "env-shards x%d" % world,
                       "collective": "one all_gather of per-env returns per rollout batch"},
            "roofline": {"bound": "hbm", "achieved": achieved_gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved_gbps / HBM_PEAK_GBPS, "traffic": traffic,
                         "kernel": "cassie::g16::env_step_g16_kernel<0> (+ clean-up pass cassie::env_step_kernel<0,3,32>, ~1 % of the time)", "kernel_ms": kernel_ms, "algo_bytes_per_env_step": ALGO_BYTES_PER_ENV_STEP,
                         "note": "path is FP64-VALU/latency bound, not HBM bound (SURVEY.md 8d); see fp64_valu"},
            "fp64_valu": {"achieved_tflops_est": FLOP_PER_SUBSTEP * 10 * n_local / (kernel_ms * 1e-3) / 1e12, "peak_tflops": FP64_VALU_PEAK_TFLOPS,
                          "flop_model": "60 kflop/substep estimate of SURVEY.md 8(d)"},
            "physics_substeps_per_s": value * 10, "returns_checksum": float(all_returns.sum().item()), "finite": finite,
        }
        if not args.no_cpu_baseline:
            cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            line["cpu_baseline"] = cpu_baseline(traj, cores)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line))
    env.close()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

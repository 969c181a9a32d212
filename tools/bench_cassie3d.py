#!/usr/bin/env python3
"""Throughput of BASELINE.json configs[4] (not the bench line, not a test): 16 384 Cassie3d envs, torque mode,
actions ~ U(-ctrlrange, +ctrlrange) redrawn every env-step (counter-based RNG), 10 substeps per env-step.
Every `--episode` env-steps all environments are put back on the standing pose (there is no reference Cassie3d
environment, hence no reference termination rule)."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from cassierl_amd import rollout as R  # noqa: E402
from cassierl_amd.vec_env3d import Cassie3dVec, CTRL_RANGE  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=16384)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--episode", type=int, default=40)
    ap.add_argument("--cpu-baseline", action="store_true", help="also time oracle/liboracle3d.so on the host cores (OpenMP over envs)")
    a = ap.parse_args()
    env = Cassie3dVec(a.envs)
    ids = torch.arange(a.envs, device="cuda")
    acts = [R.random_actions(1, ids, t, -CTRL_RANGE, CTRL_RANGE) for t in range(8)]
    for t in range(5):
        env.step(acts[t % 8], 10)
    env.reset()
    env.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    tot_ms = 0.0
    for t0 in range(0, a.steps, a.episode):
        k = min(a.episode, a.steps - t0)
        for t in range(k):  # timed with HIP events on the library's own stream
            tot_ms += env.time_steps(acts[(t0 + t) % 8], 10, 1)
        env.reset()
    s = env.get_state_host()
    cnt = env.counters()
    out = dict(leg_handover_frac=cnt.get("leg_handover_frac"), general_frac=cnt.get("general_frac"), config="configs[4]: Cassie3d torque-mode random rollout", n_envs=a.envs, steps=a.steps, ms_per_step=tot_ms / a.steps,
               env_steps_per_s=a.envs * a.steps / (tot_ms * 1e-3), physics_substeps_per_s=10 * a.envs * a.steps / (tot_ms * 1e-3),
               overflowed_envs=int((s[:, 74] != 0).sum()), finite=bool(np.isfinite(s).all()))
    if a.cpu_baseline:
        import ctypes as ct
        import time
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle_py as O
        cores = len(os.sched_getaffinity(0))
        n_cpu, steps_cpu = 4 * cores, 3
        os_ = [O.Oracle3D() for _ in range(n_cpu)]
        arr = (ct.c_void_p * n_cpu)(*[o.h for o in os_])
        u = np.ascontiguousarray(acts[0][:n_cpu].cpu().numpy())
        L = O.lib3d()
        L.orc_batch_step_torque(arr, n_cpu, u.ctypes.data_as(O.dp), 10, cores)  # warm-up
        t0 = time.perf_counter()
        for t in range(steps_cpu):
            u = np.ascontiguousarray(acts[(t + 1) % 8][:n_cpu].cpu().numpy())
            L.orc_batch_step_torque(arr, n_cpu, u.ctypes.data_as(O.dp), 10, cores)
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = dict(value=n_cpu * steps_cpu / dt, unit="env-steps/s", cores=cores, kind="port",
                                   sample="%d envs x %d env-steps (10 substeps each) in %.1f s, OpenMP over envs" % (n_cpu, steps_cpu, dt))
    print(json.dumps(out))


if __name__ == "__main__":
    main()

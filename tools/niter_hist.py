#!/usr/bin/env python3
"""Distribution of the per-environment PGS iteration count (record field ES_NITER, summed over the ten substeps of an Env.step) in the
falling-robot workloads, and what it means for a wavefront that waits for the slowest of its 32 environments (not a test)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from cassierl_amd import rollout as R, vec_env as VE
from cassierl_amd.vec_env import CassieVecEnv
n = 65536
ids = torch.arange(n, device="cuda:0")
for name, mode, seed, reset in (("stand_torque_random", "Torque", 3, True), ("fallen", "Torque", 3, False), ("stand_pd_random", "PD", 2, True)):
    box = VE.action_space(mode)
    env = CassieVecEnv(n, kind="stand", control_mode=mode, n_substeps=10, auto_reset=reset, device=0)
    out = env.alloc(); env.reset(out)
    prev = None
    for t in range(190):
        env.step(R.random_actions(seed, ids, t, box.low, box.high), out)
        if t == 188: prev = env.get_full_state_host()[:, 85].copy()
    it = env.get_full_state_host()[:, 85]
    w = it.reshape(-1, 32)
    order = np.argsort(prev, kind="stable")
    ws = it[order].reshape(-1, 32)
    oracle = np.sort(it).reshape(-1, 32)
    print(json.dumps(dict(workload=name, mean=float(it.mean()), p50=float(np.percentile(it, 50)), p90=float(np.percentile(it, 90)), p99=float(np.percentile(it, 99)), max=float(it.max()),
                          wave_max_mean_now=float(w.max(1).mean()), wave_max_mean_sorted_by_previous_step=float(ws.max(1).mean()), wave_max_mean_sorted_exact=float(oracle.max(1).mean()),
                          corr_prev=float(np.corrcoef(prev, it)[0, 1]))))
    env.close()

#!/usr/bin/env python3
"""PGS sweeps per substep actually needed per environment vs per wavefront of 32 (not a test).  usage: python tools/niter_stats.py [PD|Torque]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cassierl_amd.vec_env import CassieVecEnv
from cassierl_amd import rollout as R
from cassierl_amd.trajectory import default_gait
mode = sys.argv[1] if len(sys.argv) > 1 else "PD"
n = 65536
g = default_gait()
env = CassieVecEnv(n, kind="walk" if mode == "PD" else "stand", control_mode=mode, n_substeps=1, auto_reset=False)
env.set_trajectory(g.time, g.qpos)
out = env.alloc(); env.reset(out)
ids = torch.arange(n, device="cuda")
lo, hi = env.action_space.low, env.action_space.high
hist = np.zeros(51)
wave = []
for t in range(30):
    if t % 10 == 0 and mode == "PD": env.reset(out)   # the headline workload: every env terminates and resets each Env.step
    a = R.random_actions(1, ids, t // 10, lo, hi)   # same action for 10 single-substep steps ~ one Env.step of 10 substeps
    env.step(a, out)
    s = env.get_full_state_host()
    it = s[:, 85].astype(int)
    hist += np.bincount(np.clip(it, 0, 50), minlength=51)
    wave.append(it.reshape(-1, 32).max(axis=1).mean())
    if t % 10 == 9: print("step", t, "mean niter", it.mean(), "wave-of-32 max mean", wave[-1], "p50/p90/p99", np.percentile(it, [50, 90, 99]))
print("overall mean per env %.2f, mean over waves of the max %.2f" % ((hist * np.arange(51)).sum() / hist.sum(), np.mean(wave)))
print("hist", hist.astype(int).tolist())

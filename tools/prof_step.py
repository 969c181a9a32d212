#!/usr/bin/env python3
"""Minimal driver for profiling the dominant kernel (not a test): N Env.steps of the bench workload."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cassierl_amd.vec_env import CassieVecEnv
from cassierl_amd import rollout as R
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
mode = sys.argv[3] if len(sys.argv) > 3 else "PD"
from cassierl_amd.trajectory import default_gait
g = default_gait()
d = dict(time=g.time, qpos=g.qpos)
env = CassieVecEnv(n, kind="walk" if mode == "PD" else "stand", control_mode=mode, n_substeps=10, auto_reset=True)
env.set_trajectory(d["time"], d["qpos"])
out = env.alloc(); env.reset(out)
ids = torch.arange(n, device="cuda")
lo, hi = env.action_space.low, env.action_space.high
if mode == "OSC":  # moderate task-space accelerations (the Box is +-20)
    lo, hi = np.array([-3, -3, -1, 0, -1, 0, -3.0]), np.array([3, 3, 1, 1, 1, 1, 3.0])
for t in range(steps):
    env.step(R.random_actions(1, ids, t, lo, hi), out)
env.synchronize()
print("done", float(out["reward"].sum()))

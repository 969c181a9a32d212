#!/usr/bin/env python3
"""Soak run (not a test): long random rollouts at scale in every mode; reports non-finite states and frozen Cassie3d envs."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cassierl_amd import rollout as R
from cassierl_amd.vec_env import CassieVecEnv
from cassierl_amd.vec_env3d import Cassie3dVec, CTRL_RANGE

d = np.load(os.path.join(ROOT, "tests", "golden", "traj2d.npz"))
for kind, mode, n, steps in (("walk", "PD", 65536, 300), ("stand", "Torque", 65536, 300), ("stand", "OSC", 16384, 200), ("stand", "Jacobian", 16384, 200)):
    env = CassieVecEnv(n, kind=kind, control_mode=mode, n_substeps=10, auto_reset=True)
    env.set_trajectory(d["time"], d["qpos"])
    out = env.alloc(); env.reset(out)
    ids = torch.arange(n, device="cuda")
    lo, hi = env.action_space.low, env.action_space.high
    if mode == "Jacobian":
        lo, hi = np.array([-80.0, -50.0, -40.0] * 2), np.array([80.0, 400.0, 40.0] * 2)
    t0 = time.perf_counter(); ndone = 0
    for t in range(steps):
        _, _, done = env.step(R.random_actions(3, ids, t, lo, hi), out)
        ndone += int(done.sum())
    env.synchronize()
    q, v = env.get_state_host()
    print("%-5s %-8s n=%d steps=%d  %.1f s  episodes=%d  nonfinite envs=%d  |v|max=%.1f" %
          (kind, mode, n, steps, time.perf_counter() - t0, ndone, int((~np.isfinite(q).all(1) | ~np.isfinite(v).all(1)).sum()), np.nanmax(np.abs(v))), flush=True)
    env.close()
e3 = Cassie3dVec(16384)
ids = torch.arange(16384, device="cuda")
t0 = time.perf_counter()
for t in range(300):
    e3.step(R.random_actions(5, ids, t, -CTRL_RANGE, CTRL_RANGE), 10)
    if t % 100 == 99:
        e3.reset()
e3.synchronize()
s = e3.get_state_host()
print("cassie3d torque n=16384 steps=300  %.1f s  nonfinite envs=%d  frozen (>64 rows)=%d" % (time.perf_counter() - t0, int((~np.isfinite(s).all(1)).sum()), int((s[:, 74] != 0).sum())))

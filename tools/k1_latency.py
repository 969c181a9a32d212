#!/usr/bin/env python3
"""Latency of the wave-per-environment kernel on fallen robots (not a test): N envs (few: one wavefront per SIMD at most), ms per
Env.step of ten substeps.   usage: python tools/k1_latency.py [n_envs]   (CASSIE2D_LIB selects the build)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cassierl_amd.vec_env import CassieVecEnv, action_space, WAVE_PER_ENV
from cassierl_amd import rollout as R
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
env = CassieVecEnv(n, kind="stand", control_mode="Torque", n_substeps=10, auto_reset=False, flags=WAVE_PER_ENV)
out = env.alloc(); env.reset(out)
ids = torch.arange(n, device="cuda")
box = action_space("Torque")
for t in range(230):
    env.step(R.random_actions(3, ids, t, box.low, box.high), out)
env.synchronize()
acts = [R.random_actions(3, ids, 230 + t, box.low, box.high) for t in range(20)]
torch.cuda.synchronize(); t0 = time.perf_counter()
for a in acts: env.step(a, out)
env.synchronize(); torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / 20 * 1e3
q, v = env.get_state_host()
print("K1LAT n=%d ms_per_step=%.3f us_per_substep=%.1f mean_z=%.3f" % (n, ms, ms * 100, q[:, 1].mean()))

#!/usr/bin/env python3
"""Phase breakdown of the lane-per-leg Cassie3d kernel on configs[4] (not a test).  Needs a profiling build of tu_3d
(profiles/tools/ab_build_units.sh phase3d "tu_3d" -DCASSIE3D_PHASE_TIMING) pointed to by CASSIE2D_LIB; prints shader cycles per phase and wavefront per
10-substep step (summed over the launches of the step)."""
import ctypes as ct
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from cassierl_amd import rollout as R  # noqa: E402
from cassierl_amd.vec_env3d import Cassie3dVec, CTRL_RANGE  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
env = Cassie3dVec(n)
ids = torch.arange(n, device="cuda")
acts = [R.random_actions(5, ids, t, -CTRL_RANGE, CTRL_RANGE) for t in range(60)]
buf = (ct.c_ulonglong * 16)()
f = env.L.Cassie3dDebugPhaseCycles
f.argtypes = [ct.POINTER(ct.c_ulonglong)]
for t in range(30):
    env.step(acts[t], 10)
assert f(buf) == 0
steps = 30
for t in range(30, 60):
    env.step(acts[t], 10)
assert f(buf) == 0
v = np.array(list(buf), dtype=np.float64)
waves = (n + 15) // 16
names = {0: "kinematics, mass matrix, bias", 1: "smooth force, factorisation, M^-1 tau", 2: "raw rows: joint limits, collision, connect",
         5: "connect rows finished (impedance, z, u~, diagonal)", 6: "limit rows finished", 7: "contact rows finished (+ 3 x 3 blocks, cone warm start)",
         3: "cost of the warm start (kept only if negative)", 8: "sweeps: connect steps", 9: "sweeps: limit steps",
         10: "sweeps: contact steps + stopping test", 4: "qacc, implicit damping, integration"}
tot = v.sum()
print("Cassie3d, torque, %d envs, robots falling (bench row): %.0f cycles per wavefront per 10-substep step in the substep function" % (n, tot / waves / steps))
for k in (0, 1, 2, 5, 6, 7, 3, 8, 9, 10, 4):
    print("  %2d %-66s %10.0f cycles  %6.2f %%" % (k, names[k], v[k] / waves / steps, 100.0 * v[k] / tot))

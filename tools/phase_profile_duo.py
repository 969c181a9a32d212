#!/usr/bin/env python3
"""Phase breakdown of the 64-environments-per-wavefront kernel on the bench workload (not a test).  Needs a profiling build of tu_duo
(profiles/tools/ab_build_units.sh phase "tu_duo" -ffp-contract=on -DCASSIE_PHASE_TIMING) pointed to by CASSIE2D_LIB; prints shader cycles per
phase and wavefront per Env.step.  The clocks wait for outstanding memory operations at every mark (a store's latency lands in its phase)."""
import ctypes as ct
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from cassierl_amd import rollout as R  # noqa: E402
from cassierl_amd.trajectory import default_gait  # noqa: E402
from cassierl_amd.vec_env import CassieVecEnv  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
kind = sys.argv[2] if len(sys.argv) > 2 else "walk"      # walk | stand
mode = sys.argv[3] if len(sys.argv) > 3 else "PD"        # PD | Torque | OSC (OSC: ten one-substep MODE 2 launches of this kernel per Env.step)
g = default_gait()
env = CassieVecEnv(n, kind=kind, control_mode=mode, n_substeps=10, auto_reset=True)
env.set_trajectory(g.time, g.qpos)
out = env.alloc(); env.reset(out)
ids = torch.arange(n, device="cuda")
buf = (ct.c_ulonglong * 16)()
env.L.CassieVecPhaseCycles.argtypes = [ct.c_void_p, ct.POINTER(ct.c_ulonglong)]
lo, hi = env.action_space.low, env.action_space.high
if mode == "OSC":
    lo, hi = np.array([-2.0, -2.0, -2.0, 0.0, -2.0, 0.0, -2.0]), np.full(7, 2.0)
for t in range(10):
    env.step(R.random_actions(1, ids, t, lo, hi), out)
env._chk(env.L.CassieVecPhaseCycles(env.h, buf))
steps = 20
for t in range(steps):
    env.step(R.random_actions(1, ids, 10 + t, lo, hi), out)
env._chk(env.L.CassieVecPhaseCycles(env.h, buf))
v = np.array(list(buf), dtype=np.float64)
names = {0: "glue: load, bookkeeping, outputs, op-space state, write-back", 15: "state in (workspace -> registers), before the set-up", 1: "kinematics (FK, sincos)",
         2: "subtree sums, mass-matrix blocks, bias", 3: "active set, connect anchors", 4: "motor commands, block factorisation, M^-1 tau",
         5: "constraint rows slot by slot", 6: "warm start", 10: "rows / factorisation out (registers -> workspace)", 11: "rows of both groups in",
         7: "transpose + joint sweeps + forces back", 12: "forces out", 13: "state / factorisation / forces in, before the finish",
         8: "generalised force from the rows' geometry", 9: "M^-1 g, implicit damping, integration", 14: "state out"}
waves = n / 64
tot = v.sum()
print("%s env, %s, %d envs: %.0f cycles per wavefront per Env.step (%.3f ms at 2.4 GHz)%s" % (kind, mode, n, tot / waves / steps, tot / waves / steps / 2.4e6,
      "; per one-substep launch: %.0f cycles = %.1f us" % (tot / waves / steps / 10, tot / waves / steps / 10 / 2.4e3) if mode == "OSC" else ""))
for k in (0, 15, 1, 2, 3, 4, 5, 6, 10, 11, 7, 12, 13, 8, 9, 14):
    print("  %2d %-64s %9.0f cycles  %6.2f %%" % (k, names[k], v[k] / waves / steps, 100 * v[k] / tot))
hand = v[[15, 10, 11, 12, 13, 14]].sum()
print("  hand-over through the workspace (15 + 10 + 11 + 12 + 13 + 14): %.2f %%" % (100 * hand / tot))
env.close()

#!/usr/bin/env python3
"""Phase breakdown of the OSC controller-in-the-loop kernels, or (third argument "physics") of the packed physics kernel on the
bench workload (not a test).  Needs a profiling build of the extension (-DCASSIE_PHASE_TIMING on cassie_cabi, tu_g16 and
tu_ctrl_g16) pointed to by CASSIE2D_LIB; prints shader cycles per phase summed over wavefronts.
usage: phase_profile.py [n_envs] [scripted|random] [physics]"""
import ctypes as ct
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from cassierl_amd import rollout as R  # noqa: E402
from cassierl_amd.vec_env import CassieVecEnv  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
scripted = len(sys.argv) > 2 and sys.argv[2] == "scripted"
if len(sys.argv) > 3 and sys.argv[3] == "physics":
    from cassierl_amd.trajectory import default_gait
    gait = default_gait()
    for kind, mode in (("walk", "PD"), ("stand", "Torque")):
        env = CassieVecEnv(n, kind=kind, control_mode=mode, n_substeps=10, auto_reset=True)
        env.set_trajectory(gait.time, gait.qpos)
        out = env.alloc()
        env.reset(out)
        ids = torch.arange(n, device="cuda")
        buf = (ct.c_ulonglong * 16)()
        env.L.CassieVecPhaseCycles.argtypes = [ct.c_void_p, ct.POINTER(ct.c_ulonglong)]
        for t in range(10):
            env.step(R.random_actions(1, ids, t, env.action_space.low, env.action_space.high), out)
        env._chk(env.L.CassieVecPhaseCycles(env.h, buf))   # reads and clears
        for t in range(20):
            env.step(R.random_actions(1, ids, 10 + t, env.action_space.low, env.action_space.high), out)
        env._chk(env.L.CassieVecPhaseCycles(env.h, buf))
        v = np.array(list(buf), dtype=np.float64)
        names = ["0 outside the substep (load, glue, integrate, outputs, write-back)", "1 kinematics (FK, sincos)", "2 mass matrix, M^-1 (Gauss-Jordan), smooth accel.",
                 "3 active limits / contacts, row kinds", "4 constraint rows (J, aref, M^-1 J')", "5 A = J M^-1 J' + R", "6 warm start, initial residual",
                 "7 PGS sweeps", "8 J'f, qacc, implicit damping"]
        if os.environ.get("PHASE_LEG"):
            names = ["0 outside the substep (load, outputs, op-space state, write-back)", "1 kinematics (FK, sincos)", "2 subtree sums, mass-matrix blocks, bias",
                     "3 active set (limits, collision spheres), connect anchors", "4 motor commands, setState bookkeeping, block factorisation, M^-1 tau",
                     "5 constraint rows slot by slot (J, aref, R, z, u~, A), warm-start forces", "6 warm-start cost test, initial residuals", "7 PGS sweeps",
                     "8 generalised force J'f from the rows' geometry", "9 M^-1 g, implicit damping iteration, integration"]
        tot = v[:len(names)].sum()
        per_wave = 32 if os.environ.get("PHASE_LEG") else 4   # PHASE_LEG=1: the two-lanes-per-environment kernel is the one instrumented
        print("%s env, %s mode, %d envs: %.0f cycles per wavefront per Env.step" % (kind, mode, n, tot / (n / per_wave) / 20))
        for i, nm in enumerate(names):
            print("  %-72s %6.2f %%" % (nm, 100 * v[i] / tot))
        env.close()
    sys.exit(0)
env = CassieVecEnv(n, kind="stand", control_mode="OSC", n_substeps=10, auto_reset=True)
out = env.alloc()
env.reset(out)
ids = torch.arange(n, device="cuda")
lo, hi = np.array([-2, -2, -2, 0, -2, 0, -2.0]), np.full(7, 2.0)
zp = torch.full((n,), 0.9, dtype=torch.float64, device="cuda")
zv = torch.zeros(n, dtype=torch.float64, device="cuda")
buf = (ct.c_ulonglong * 16)()
env.L.CassieVecPhaseCycles.argtypes = [ct.c_void_p, ct.POINTER(ct.c_ulonglong)]
for rep in range(2):
    for t in range(10):
        if scripted:
            env._chk(env.L.CassieVecStandingStep(env.h, 2, zp.data_ptr(), zv.data_ptr(), 10))
        else:
            env.step(R.random_actions(4, ids, t + 10 * rep, lo, hi), out)
    env._chk(env.L.CassieVecPhaseCycles(env.h, buf))
v = np.array(list(buf), dtype=np.float64)
names = ["0 staging/setState", "1 ctrl_dyn (FK, M, Hinv, rows, pinv4)", "2 T build (Nc, Hinv y, A x)", "3 QP data (G, c)", "4 QP iterations",
         "5 glue", "6 physics substep"]
tot = v[:7].sum()
for i, nm in enumerate(names):
    print("%-42s %6.2f %%" % (nm, 100 * v[i] / tot))
print("QP iterations per wave-substep: %.2f" % (v[8] / (n / 4 * 10 * 10)))

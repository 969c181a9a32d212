#!/usr/bin/env python3
"""How often a group of 32 environments of the two-lanes / 64-environments-per-wavefront kernels leaves the six-row sweep (not a test): the kernel
source compiled for the CPU with 64 lanes = the 32 environments of one group and a counter on its wave-uniform `small` decision, on the bench
workload (walk env, PD, random targets, every step resets) and on stand env / PD.
usage: python tools/small_stats.py /path/to/libleg_host_stats64.so
       (g++ -O2 -march=native -fopenmp -ffp-contract=off -DLEG_HOST_FAST -DLEG_HOST_LANES=64 -DLEG_STATS leg_host.cpp)"""
import ctypes as ct, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import oracle_py as O
from conftest import state_vec
from cassierl_amd import rollout as R
from cassierl_amd.trajectory import default_gait
L = ct.CDLL(sys.argv[1])
g = default_gait()
tq = np.ascontiguousarray(g.qpos, dtype=np.float64)
dp, ip, bp = ct.POINTER(ct.c_double), ct.POINTER(ct.c_int), ct.POINTER(ct.c_ubyte)
low, high = np.radians([-50, -164, -140] * 2), np.radians([80, -37, -30] * 2)
for kind, seed in ((0, 1), (1, 2)):
    oe = O.OracleEnv("walk" if kind == 0 else "stand", "PD", traj=dict(time=g.time, qpos=g.qpos))
    oe.reset()
    q, v = oe.oracle.state()
    ctor = O.Oracle()
    n = 4096
    state = np.tile(state_vec(q, v, oe.oracle.warmstart(), kq=ctor.state()[0], kv=ctor.state()[1], qstate=q), (n, 1)).copy()
    obs, rew, done, pend, bad = np.zeros((n, 26)), np.zeros(n), np.zeros(n, dtype=np.uint8), np.zeros(n, dtype=np.int32), ct.c_int(0)
    ids = torch.arange(n)
    st = (ct.c_longlong * 8)()
    for blk in range(3):
        for t in range(10):
            a = np.ascontiguousarray(R.random_actions(seed, ids, blk * 10 + t, low, high).numpy())
            L.leg_host_step(state.ctypes.data_as(dp), a.ctypes.data_as(dp), n, 6, 0, 10, 0, kind, 1, tq.ctypes.data_as(dp), ct.c_double(float(g.time[-1])), len(g.time),
                            obs.ctypes.data_as(dp), rew.ctypes.data_as(dp), done.ctypes.data_as(bp), None, pend.ctypes.data_as(ip), ct.byref(bad), 8)
        L.leg_host_small_stats(st)
        print("%s env, PD random, steps %2d-%2d: group set-ups %d, not on the six-row path %.4f; per leg-lane set-up: joint limit %.5f, third pair %.5f" % (
            "walk" if kind == 0 else "stand", blk * 10, blk * 10 + 9, st[0], st[1] / max(1, st[0]), st[2] / (64.0 * max(1, st[0])), st[3] / (64.0 * max(1, st[0]))))

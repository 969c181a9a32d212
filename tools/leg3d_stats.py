#!/usr/bin/env python3
"""What a wavefront of the lane-per-leg Cassie3d kernel executes per substep (not a test): the kernel source compiled for the CPU with 64
lanes = the 32 environments of one wavefront and counters on its wave-uniform loops -- sweeps, joint-limit steps, contact steps and
Newton iterations of the cone QCQP -- on the configs[4] workload (random torques from the standing pose).
usage: python tools/leg3d_stats.py /path/to/libleg3d_stats.so   (g++ ... -DLEG_HOST_FAST -DLEG_HOST_LANES=64 -DLEG3_STATS leg3d_host.cpp)"""
import ctypes as ct, sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, ROOT)
import oracle_py as O
L = ct.CDLL(sys.argv[1])
n = 32
o = O.Oracle3D(); q, v = o.state()
rec = np.zeros(80); rec[:21] = q; rec[21:41] = v; rec[41:61] = o.warmstart()
state = np.tile(rec, (n, 1)).copy()
rng = np.random.default_rng(0)
CTRL = np.array([4.5, 4.5, 12.2, 12.2, 0.9] * 2)
dp, ip = ct.POINTER(ct.c_double), ct.POINTER(ct.c_int)
pend = np.zeros(n, dtype=np.int32)
st = (ct.c_longlong * 4)()
for blk in range(6):
    for t in range(10):
        a = np.ascontiguousarray(rng.uniform(-1, 1, (n, 10)) * CTRL)
        L.leg3d_host_step(state.ctypes.data_as(dp), a.ctypes.data_as(dp), n, 10, 1, pend.ctypes.data_as(ip), None, None)
    L.leg3d_host_stats(st)
    sub = 100.0
    print("steps %2d-%2d per wave-substep: sweeps %.1f, limit steps/sweep %.2f, contact steps/sweep %.2f, Newton iterations/contact step %.2f, pending %d" % (
        blk * 10, blk * 10 + 9, st[0] / sub, st[1] / max(1, st[0]), st[2] / max(1, st[0]), st[3] / max(1, st[2]), int((pend > 0).sum())))

#!/usr/bin/env python3
"""Minimal trigger for a wrong build of env_step_duo_kernel<1>: all robots in the reset pose except ONE whose left knee is pushed past its joint limit
(its group leaves the six-row joint sweep for the eight-row pair sweep inside the kernel); one substep, zero torque; which environments differ from
the two-lanes kernel?   usage: CASSIE2D_LIB=... python tools/dbg_duo_limit.py"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cassierl_amd.vec_env import CassieVecEnv, DUO_TIER_ON, DUO_TIER_OFF, LEG_TIER_ON
np.set_printoptions(linewidth=250, precision=6)
n = 128
CASES = ((5, 1, 6, 1.5, 0.0), (5, 1, 6, 1.5, 1.0), (5, 1, 11, 1.5, 0.0), (5, 1, 6, -1.5, 0.0), (5, 1, 6, 0.6, 0.0), (5, 1, 6, 0.3, 0.0), (40, 1, 6, 1.5, 0.0), (5, 1, 5, 1.0, 0.0), (5, 1, 3, 1.5, 0.0))
for odd_env, nsub, joint, dq, lift in CASES:
    outs = []
    for fl in (LEG_TIER_ON | DUO_TIER_OFF, LEG_TIER_ON | DUO_TIER_ON):
        env = CassieVecEnv(n, kind="stand", control_mode="Torque", n_substeps=nsub, auto_reset=False, flags=fl)
        bufs = env.alloc(); env.reset(bufs)
        s = env.get_full_state_host()
        if odd_env is not None:
            s[odd_env, joint] += dq; s[odd_env, 39 + joint] += dq
            s[odd_env, 1] += lift; s[odd_env, 40] += lift
        env.set_full_state_host(s)
        a = torch.zeros((n, 6), dtype=torch.float64, device="cuda")
        env.step(a, bufs)
        outs.append(env.get_full_state_host().copy()); c = env.counters(); env.close()
    ds = np.abs(outs[0] - outs[1])
    bad = np.nonzero(ds.max(axis=1) > 0)[0]
    print("odd env %s joint %d dq %+.1f lift %.1f n_sub %d: envs differing %s   cleanup %d" % (odd_env, joint, dq, lift, nsub, bad.tolist(), c["cleanup_substeps"]))
    for e in bad[:3]:
        f = np.nonzero(ds[e] > 0)[0]
        print("   env %d: qacc (warm start, fields 26..38)\n     pair %s\n     duo  %s\n     diff %s" % (e, outs[0][e][26:39], outs[1][e][26:39], outs[1][e][26:39] - outs[0][e][26:39]))

#!/usr/bin/env python3
"""Where a wrong build of env_step_duo_kernel<1> goes wrong (r06: the auto-var-init=pattern build): which environments (lane pattern), which
record fields, after how many substeps.  usage: CASSIE2D_LIB=... python tools/dbg_duo_pattern.py"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cassierl_amd import rollout as R
from cassierl_amd.vec_env import CassieVecEnv, DUO_TIER_ON, DUO_TIER_OFF, LEG_TIER_ON, action_space
np.set_printoptions(linewidth=250)
sp = action_space("Torque")
for n, nsub in ((128, 1), (128, 2), (128, 10), (40, 1)):
    outs = []
    for fl in (LEG_TIER_ON | DUO_TIER_OFF, LEG_TIER_ON | DUO_TIER_ON):
        env = CassieVecEnv(n, kind="stand", control_mode="Torque", n_substeps=nsub, auto_reset=False, flags=fl)
        bufs = env.alloc(); env.reset(bufs)
        ids = torch.arange(n, device="cuda")
        o, r, d = env.step(R.random_actions(3, ids, 0, sp.low, sp.high), bufs)
        outs.append((o.cpu().numpy().copy(), env.get_full_state_host().copy())); env.close()
    ds = np.abs(outs[0][1] - outs[1][1])
    bad_env = np.nonzero(ds.max(axis=1) > 0)[0]
    print("n=%d n_sub=%d: %d envs differ: %s" % (n, nsub, len(bad_env), bad_env.tolist()))
    if len(bad_env):
        e = bad_env[0]
        f = np.nonzero(ds[e] > 0)[0]
        print("  env %d fields %s" % (e, f.tolist()))
        print("  pair:", outs[0][1][e][f][:14]); print("  duo :", outs[1][1][e][f][:14])
        print("  per-field count of differing envs:", {int(k): int((ds[:, k] > 0).sum()) for k in range(ds.shape[1]) if (ds[:, k] > 0).any()})

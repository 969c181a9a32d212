#!/usr/bin/env python3
"""Per-environment comparison of the two-lanes kernel with the 64-environments kernel after ONE Env.step (walk env / PD), for experiments on the latter
(r05: the address-transposition experiment whose device build depended on dead code; r06: the same through -DDUO_VIEW_EXPERIMENT).
usage: [CASSIE2D_LIB=...] python tools/dbg_duo.py [extra flags for the duo env, e.g. 0x20000000 = CASSIE_DUO_VIEW_FLAG]"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cassierl_amd import rollout as R
from cassierl_amd.trajectory import default_gait
from cassierl_amd.vec_env import CassieVecEnv, DUO_TIER_ON, DUO_TIER_OFF, LEG_TIER_ON
extra = int(sys.argv[1], 0) if len(sys.argv) > 1 else 0
g = default_gait()
PD_LO, PD_HI = np.radians([-50, -164, -140] * 2), np.radians([80, -37, -30] * 2)
for n in (8, 70, 128, 4141):
    outs = []
    for fl in (LEG_TIER_ON | DUO_TIER_OFF, LEG_TIER_ON | DUO_TIER_ON | extra):
        env = CassieVecEnv(n, kind="walk", control_mode="PD", n_substeps=10, auto_reset=True, flags=fl)
        env.set_trajectory(g.time, g.qpos)
        bufs = env.alloc(); env.reset(bufs)
        ids = torch.arange(n, device="cuda")
        for t in range(3):
            o, r, d = env.step(R.random_actions(1, ids, t, PD_LO, PD_HI), bufs)
        outs.append((o.cpu().numpy().copy(), env.get_full_state_host().copy())); env.close()
    do = np.abs(outs[0][0] - outs[1][0]).max(axis=1); ds = np.abs(outs[0][1] - outs[1][1])
    bad = np.argwhere(ds > 0)
    print("n=%d extra=%#x: envs with a differing state field %d of %d; max |d obs| %.3g; max |d state| %.3g; first (env, field): %s"
          % (n, extra, len(set(bad[:, 0].tolist())), n, do.max(), ds.max(), bad[:6].tolist()))

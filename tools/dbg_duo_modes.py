#!/usr/bin/env python3
"""Two-lanes kernel against the 64-environments kernel after a few Env.steps, per control mode (PD / Torque: kernel MODE 0 / 1; OSC: MODE 2) -- which
build of the library (CASSIE2D_LIB) agrees with itself.  usage: [CASSIE2D_LIB=...] python tools/dbg_duo_modes.py [n_envs] [steps]"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cassierl_amd import rollout as R
from cassierl_amd.vec_env import CassieVecEnv, DUO_TIER_ON, DUO_TIER_OFF, LEG_TIER_ON, action_space
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2077
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
for kind, mode, ar in (("stand", "PD", True), ("stand", "Torque", False), ("stand", "Torque", True), ("stand", "OSC", True)):
    outs = []
    sp = action_space(mode)
    lo, hi = (np.array([-2.0, -2.0, -2.0, 0.0, -2.0, 0.0, -2.0]), np.full(7, 2.0)) if mode == "OSC" else (sp.low, sp.high)
    for fl in (LEG_TIER_ON | DUO_TIER_OFF, LEG_TIER_ON | DUO_TIER_ON):
        env = CassieVecEnv(n, kind=kind, control_mode=mode, n_substeps=10, auto_reset=ar, flags=fl)
        bufs = env.alloc(); env.reset(bufs)
        ids = torch.arange(n, device="cuda")
        for t in range(steps):
            o, r, d = env.step(R.random_actions(3, ids, t, lo, hi), bufs)
        outs.append((o.cpu().numpy().copy(), env.get_full_state_host().copy())); env.close()
    ds = np.abs(outs[0][1] - outs[1][1]); bad = np.argwhere(ds > 0)
    print("%s/%s auto_reset=%d n=%d steps=%d: envs differing %d; max |d state| %.3g; max |d obs| %.3g; fields %s" %
          (kind, mode, ar, n, steps, len(set(bad[:, 0].tolist())), ds.max(), np.abs(outs[0][0] - outs[1][0]).max(), sorted(set(bad[:, 1].tolist()))[:30]))

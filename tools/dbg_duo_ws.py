#!/usr/bin/env python3
"""The hand-over workspace of env_step_duo_kernel<1> after ONE substep of the minimal trigger (128 stand-env robots in the reset pose, robot 5's left toe
turned by +0.3 rad: one of its foot spheres leaves the floor), for the library CASSIE2D_LIB names.  Comparing two builds' dumps says which PHASE of the
wrong build goes wrong: rows / factorisation (set-up), forces (joint sweep) or state (finish).   usage: CASSIE2D_LIB=... python tools/dbg_duo_ws.py out.npz"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cassierl_amd.vec_env import CassieVecEnv, DUO_TIER_ON, DUO_TIER_OFF, LEG_TIER_ON
n = 128
res = {}
for name, fl in (("pair", LEG_TIER_ON | DUO_TIER_OFF), ("duo", LEG_TIER_ON | DUO_TIER_ON)):
    env = CassieVecEnv(n, kind="stand", control_mode="Torque", n_substeps=1, auto_reset=False, flags=fl)
    bufs = env.alloc(); env.reset(bufs)
    s = env.get_full_state_host()
    s[5, 6] += 0.3; s[5, 39 + 6] += 0.3
    env.set_full_state_host(s)
    env.step(torch.zeros((n, 6), dtype=torch.float64, device="cuda"), bufs)
    res[name + "_state"] = env.get_full_state_host()
    if name == "duo":
        res["ws"] = env.debug_workspace_host()
    env.close()
np.savez(sys.argv[1], **res)
print("max |pair - duo| state:", np.abs(res["pair_state"] - res["duo_state"]).max())

#!/usr/bin/env python3
"""One line per library: does env_step_duo_kernel agree with env_step_leg_kernel after one ten-substep Env.step of 2 077 stand-env robots under random
torques (MODE 1: groups leave the six-row path) and random PD targets (MODE 0).  usage: python tools/dbg_duo_quick.py lib.so [lib.so ...]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, numpy as np
sys.path.insert(0, %r)
import torch
from cassierl_amd import rollout as R
from cassierl_amd.vec_env import CassieVecEnv, DUO_TIER_ON, DUO_TIER_OFF, LEG_TIER_ON, action_space
n = 2077
res = []
for mode in ("Torque", "PD"):
    sp = action_space(mode); outs = []
    for fl in (LEG_TIER_ON | DUO_TIER_OFF, LEG_TIER_ON | DUO_TIER_ON):
        env = CassieVecEnv(n, kind="stand", control_mode=mode, n_substeps=10, auto_reset=False, flags=fl)
        bufs = env.alloc(); env.reset(bufs)
        ids = torch.arange(n, device="cuda")
        for t in range(3):
            env.step(R.random_actions(3, ids, t, sp.low, sp.high), bufs)
        outs.append(env.get_full_state_host().copy()); env.close()
    ds = np.abs(outs[0] - outs[1])
    res.append("%%s: %%d envs differ, max %%.3g" %% (mode, int((ds.max(axis=1) > 0).sum()), ds.max()))
print("QUICK " + "; ".join(res))
''' % ROOT
for lib in sys.argv[1:]:
    p = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, CASSIE2D_LIB=os.path.abspath(lib)), capture_output=True, text=True)
    line = [l for l in p.stdout.splitlines() if l.startswith("QUICK ")]
    print("%-28s %s" % (os.path.basename(lib).replace("libcassie2d_", "")[:-3], line[0][6:] if line else "FAILED " + p.stderr[-300:]), flush=True)

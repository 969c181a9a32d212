import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
import torch
from cassierl_amd import rollout as R
from cassierl_amd.trajectory import default_gait
from cassierl_amd.vec_env import CassieVecEnv, DUO_TIER_ON, DUO_TIER_OFF
g = default_gait()
PD_LO, PD_HI = np.radians([-50, -164, -140] * 2), np.radians([80, -37, -30] * 2)
for n in (8, 70, 128):
    outs = []
    for fl in (DUO_TIER_OFF, DUO_TIER_ON):
        os.environ["CASSIE2D_LEG"] = "1"
        env = CassieVecEnv(n, kind="walk", control_mode="PD", n_substeps=10, auto_reset=True, flags=fl)
        env.set_trajectory(g.time, g.qpos)
        bufs = env.alloc(); env.reset(bufs)
        ids = torch.arange(n, device="cuda")
        o, r, d = env.step(R.random_actions(1, ids, 0, PD_LO, PD_HI), bufs)
        outs.append((o.cpu().numpy().copy(), env.get_full_state_host().copy())); env.close()
    do = np.abs(outs[0][0] - outs[1][0]).max(axis=1); ds = np.abs(outs[0][1] - outs[1][1])
    print(n, "obs diff per env", np.round(do, 3).tolist()[:80])
    bad = np.argwhere(ds > 0); print(n, "state fields differing (env, field) first 20:", bad[:20].tolist())

#!/usr/bin/env python3
"""The two falling-robot workloads of bench.py's `env_steps_per_s_other_workloads` alone (not a test): stand env / torque mode with and
without auto-reset, 65 536 envs.  CASSIE2D_SEGMENTS=0 selects the one-launch order of the kernel tiers (A/B of the segmented Env.step)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench as B
from cassierl_amd import rollout as R, vec_env as VE
n = int(os.environ.get("AB_ENVS", "65536"))
ids = torch.arange(n, device="cuda:0")
tq = VE.action_space("Torque"); pd = VE.action_space("PD")
rows = [B.run_env_workload("stand_torque_random", n, "stand", "Torque", 0, None, 150, 40, lambda t: R.random_actions(3, ids, t, tq.low, tq.high), ""),
        B.run_env_workload("torque_random_no_reset_fallen", n, "stand", "Torque", 0, None, 200, 20, lambda t: R.random_actions(3, ids, t, tq.low, tq.high), "", auto_reset=False),
        B.run_env_workload("stand_pd_random", n, "stand", "PD", 0, None, 150, 40, lambda t: R.random_actions(2, ids, t, pd.low, pd.high), "")]
for r in rows:
    print(json.dumps({k: r[k] for k in ("workload", "env_steps_per_s", "ms_per_step", "cleanup_frac", "k1_frac", "finite")}))

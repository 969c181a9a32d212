#!/usr/bin/env python3
"""Driver for kernel traces of the all-fallen floor (not a test): 65 536 stand envs, random torques, no reset, 230 Env.steps."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cassierl_amd.vec_env import CassieVecEnv, action_space
from cassierl_amd import rollout as R
n = 65536
auto = len(sys.argv) > 1 and sys.argv[1] == "reset"
env = CassieVecEnv(n, kind="stand", control_mode="Torque", n_substeps=10, auto_reset=auto)
out = env.alloc(); env.reset(out)
ids = torch.arange(n, device="cuda")
box = action_space("Torque")
for t in range(230):
    env.step(R.random_actions(3, ids, t, box.low, box.high), out)
env.synchronize()
print(env.counters())

#!/usr/bin/env python3
"""A/B of two builds of the extension on the bench workload (not a test): for each library given on the command line, the median
kernel time of one 65 536-env Env.step (PD and torque) by HIP events, plus a spot check that the builds agree.
usage: python tools/ab_bench.py libA.so libB.so      (AB_ENVS = batch size, default 65536; AB_FLAGS = CassieVecConfig flags, e.g. 16 = no leg tier)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, json, numpy as np
sys.path.insert(0, %r)
import torch
from cassierl_amd import rollout as R
from cassierl_amd.trajectory import default_gait
from cassierl_amd.vec_env import CassieVecEnv
g = default_gait()
out = {}
for kind, mode in (("walk", "PD"), ("stand", "Torque")):
    n = int(os.environ.get("AB_ENVS", "65536"))
    env = CassieVecEnv(n, kind=kind, control_mode=mode, n_substeps=10, auto_reset=True, flags=int(os.environ.get("AB_FLAGS", "0")))
    env.set_trajectory(g.time, g.qpos)
    bufs = env.alloc(); env.reset(bufs)
    ids = torch.arange(n, device="cuda")
    lo, hi = env.action_space.low, env.action_space.high
    for t in range(20):
        env.step(R.random_actions(1, ids, t, lo, hi), bufs)
    q, v = env.get_state_host()
    ms = [env.time_steps(R.random_actions(1, ids, 20 + t, lo, hi), 20, bufs) for t in range(7)]
    out[mode] = dict(ms_per_step=float(np.median(ms)), ms_all=[float(x) for x in ms], q0=q[:8].tolist())
    env.close()
print("AB " + json.dumps(out))
''' % ROOT
res = []
for lib in sys.argv[1:]:
    env = dict(os.environ, CASSIE2D_LIB=os.path.abspath(lib))
    p = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    line = [l for l in p.stdout.splitlines() if l.startswith("AB ")]
    if not line:
        print(lib, "FAILED", p.stderr[-800:])
        continue
    r = json.loads(line[0][3:])
    res.append(r)
    print(json.dumps(dict(lib=lib, **{k: dict(ms_per_step=v["ms_per_step"], ms_all=v["ms_all"]) for k, v in r.items()})))
if len(res) == 2:
    import numpy as np
    for k in res[0]:
        d = np.abs(np.array(res[0][k]["q0"]) - np.array(res[1][k]["q0"])).max()
        print(json.dumps(dict(mode=k, B_over_A_time=res[1][k]["ms_per_step"] / res[0][k]["ms_per_step"], max_dq_8_envs_after_20_steps=float(d))))

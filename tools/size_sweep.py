#!/usr/bin/env python3
"""Env-steps/s of the first physics tier against the batch size on ONE GPU: walk env / PD (the headline workload) and
stand env / PD (robots that move), 65 536 .. 524 288 envs, with the tier chosen by the size rule, the two-lanes kernel forced and the
64-environments kernel forced.  One JSON line per (library, workload, n, tier).
usage: python tools/size_sweep.py [lib.so ...]     (no argument: the in-tree build; SWEEP_SIZES="65536,131072" overrides the sizes)"""
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
from cassierl_amd.vec_env import CassieVecEnv, LEG_TIER_ON, DUO_TIER_ON, DUO_TIER_OFF
g = default_gait()
sizes = [int(x) for x in os.environ.get("SWEEP_SIZES", "65536,98304,131072,196608,262144,524288").split(",")]
for kind in ("walk", "stand"):
    for n in sizes:
        for tier, fl in (("rule", 0), ("pair", LEG_TIER_ON | DUO_TIER_OFF), ("duo", LEG_TIER_ON | DUO_TIER_ON)):
            env = CassieVecEnv(n, kind=kind, control_mode="PD", n_substeps=10, auto_reset=True, flags=fl)
            env.set_trajectory(g.time, g.qpos)
            bufs = env.alloc(); env.reset(bufs)
            ids = torch.arange(n, device="cuda")
            lo, hi = env.action_space.low, env.action_space.high
            for t in range(6):
                env.step(R.random_actions(1, ids, t, lo, hi), bufs)
            ms = [env.time_steps(R.random_actions(1, ids, 6 + t, lo, hi), 10, bufs) for t in range(5)]
            m = float(np.median(ms))
            info = env.tier_info() if hasattr(env.L, "CassieVecTierInfo") else {}
            print("SWEEP " + json.dumps(dict(workload=kind + "_pd_random", n_envs=n, tier=tier, first_tier=info.get("first_tier"), ws_mb=round(info.get("duo_workspace_bytes", 0) / 1e6, 1), ws_probes=info.get("ws_probes"), ms_per_step=round(m, 4), env_steps_per_s=round(n / m * 1e3), ms_all=[round(float(x), 4) for x in ms])), flush=True)
            env.close()
''' % ROOT
for lib in sys.argv[1:] or [None]:
    env = dict(os.environ)
    if lib:
        env["CASSIE2D_LIB"] = os.path.abspath(lib)
    p = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    for l in p.stdout.splitlines():
        if l.startswith("SWEEP "):
            print(json.dumps(dict(lib=os.path.basename(lib) if lib else "in-tree", **json.loads(l[6:]))), flush=True)
    if p.returncode != 0:
        print(json.dumps(dict(lib=lib, failed=p.stderr[-600:])))

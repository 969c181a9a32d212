#!/usr/bin/env python3
"""The two falling-robot workloads of bench.py under the three settings of the segment scheduler (not a test): CASSIE2D_SEGMENTS unset (by the hand-over
estimate), 0 (never in segments), 1 (always while robots are down).   usage: python tools/ab_segments.py [lib.so]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, json
sys.path.insert(0, %r)
import torch, bench
from cassierl_amd import rollout as R, vec_env as VE
from cassierl_amd.trajectory import default_gait
g = default_gait(); traj = dict(time=g.time, qpos=g.qpos); n = 65536
ids = torch.arange(n, device="cuda:0"); tq = VE.action_space("Torque")
a = bench.run_env_workload("stand_torque_random", n, "stand", "Torque", 0, traj, 150, 40, lambda t: R.random_actions(3, ids, t, tq.low, tq.high), "")
b = bench.run_env_workload("fallen", n, "stand", "Torque", 0, traj, 200, 20, lambda t: R.random_actions(3, ids, t, tq.low, tq.high), "", auto_reset=False)
print("SEG " + json.dumps(dict(stand_torque_random=round(a["env_steps_per_s"] / 1e6, 2), fallen=round(b["env_steps_per_s"] / 1e6, 2))))
''' % ROOT
for seg in ("warm-up", None, "0", "1", None, "0", "1"):   # (the first process on a fresh box is ~15 % slow: thrown away)
    env = dict(os.environ)
    env.pop("CASSIE2D_SEGMENTS", None)
    if seg in ("0", "1"):
        env["CASSIE2D_SEGMENTS"] = seg
    if len(sys.argv) > 1:
        env["CASSIE2D_LIB"] = os.path.abspath(sys.argv[1])
    p = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    line = [l for l in p.stdout.splitlines() if l.startswith("SEG ")]
    print("CASSIE2D_SEGMENTS=%s" % seg, line[0][4:] if line else "FAILED " + p.stderr[-300:], flush=True)

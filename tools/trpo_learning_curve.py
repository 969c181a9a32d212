#!/usr/bin/env python3
"""Learning-curve evidence for the TRPO outer loop (not a test): `train_trpo.py`-style run on the stand env (cassie_stand2d reward:
stay upright at z = 0.9 with small torques), 16 384 envs x 16 Env.steps per iteration, printing one JSON line per iteration.
The reference's only acceptance criterion for training is qualitative ("Cassie learns to stand", README / Docs/Writeup.pdf);
this records that the average per-step reward and the episode length rise under the on-device rollouts."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from cassierl_amd.trajectory import default_gait  # noqa: E402
from cassierl_amd.trpo import make_cassie_trpo  # noqa: E402

n_itr = int(sys.argv[1]) if len(sys.argv) > 1 else 60
n = 16384
algo = make_cassie_trpo(n, kind="stand", control_mode="Torque", device=0, trajectory=default_gait(), seed=1, batch_size=n * 16)
t0 = time.perf_counter()
for it in range(n_itr):
    st = algo.train_iteration()
    alive = float((algo.path_t.double().mean()).item())
    print(json.dumps(dict(itr=st["itr"], avg_reward=st["avg_reward"], episodes=st["episodes"], avg_return=st["avg_return"], kl=st["kl"],
                          backtracks=st["backtracks"], mean_path_age_steps=alive, seconds=time.perf_counter() - t0)), flush=True)

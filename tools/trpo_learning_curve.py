#!/usr/bin/env python3
"""Learning-curve evidence for the TRPO outer loop (not a test): `train_trpo.py`-style runs on the device environment, one JSON line per
(every k-th) iteration: average per-step reward, episodes, average return, mean path age, wall-clock.
  stand   cassie_stand2d reward (stay upright at z = 0.9 with small torques), torque mode
  walk    the env trpo_cassie.py:12-20 trains: Cassie2dEnv, PD control, reference-gait reward -- with REFERENCE semantics (quirk Q3: the stale qstate
          makes reward < 0.6 on every step, every episode ends after one step: nothing to learn -- recorded as such) ...
  walkfix ... and with CASSIE_FIX_STALE_QSTATE (the reward the reference's authors meant)
The reference's only acceptance criterion for training is qualitative ("Cassie learns to stand", README / Docs/Writeup.pdf).
usage: python tools/trpo_learning_curve.py [stand|walk|walkfix] [iterations] [envs] [every]
(envs = 65 536 by default: the 64-environments-per-wavefront kernel, K1d, is then the first tier -- `first_tier` in the first line)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from cassierl_amd.trajectory import default_gait  # noqa: E402
from cassierl_amd.trpo import make_cassie_trpo  # noqa: E402
from cassierl_amd.vec_env import FIX_STALE_QSTATE  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "stand"
n_itr = int(sys.argv[2]) if len(sys.argv) > 2 else 60
n = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
every = int(sys.argv[4]) if len(sys.argv) > 4 else 10
horizon = 16
kw = dict(stand=dict(kind="stand", control_mode="Torque"), walk=dict(kind="walk", control_mode="PD"),
          walkfix=dict(kind="walk", control_mode="PD", flags=FIX_STALE_QSTATE))[which]
flags = kw.pop("flags", 0)
if flags:   # make_cassie_trpo builds the env itself: pass the flag through the environment's constructor default
    import cassierl_amd.vec_env as VE
    _init = VE.CassieVecEnv.__init__

    def init(self, *a, **k):
        k["flags"] = k.get("flags", 0) | flags
        _init(self, *a, **k)
    VE.CassieVecEnv.__init__ = init
algo = make_cassie_trpo(n, device=0, trajectory=default_gait(), seed=1, batch_size=n * horizon, **kw)
print(json.dumps(dict(run=which, envs=n, horizon_env_steps=horizon, samples_per_iteration=n * horizon, first_tier=algo.env.tier_info()["first_tier"], flags=flags,
                      hyper="trpo_cassie.py:21-42: 26-32-32-6 tanh Gaussian MLP, init_std 2.0, linear feature baseline, KL 0.005, gamma 0.99, path <= 1000")), flush=True)
t0 = time.perf_counter()
for it in range(n_itr):
    st = algo.train_iteration()
    if it % every == 0 or it == n_itr - 1:
        alive = float((algo.path_t.double().mean()).item())
        print(json.dumps(dict(itr=st["itr"], avg_reward=st["avg_reward"], episodes=st["episodes"], avg_return=st["avg_return"], kl=st["kl"],
                              backtracks=st["backtracks"], mean_path_age_steps=alive, seconds=time.perf_counter() - t0)), flush=True)

#!/usr/bin/env python3
"""Where one TRPO update spends its time (not a test): 65 536 envs x 8 steps = 524 288 samples, stand env, torque mode.
Synchronising timers around the phases of process() / optimize()."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cassierl_amd import trpo as T

n, hor = int(os.environ.get("N_ENVS", "65536")), 8
algo = T.make_cassie_trpo(n, kind="stand", control_mode="Torque", batch_size=n * hor)
def sync(): torch.cuda.synchronize(); return time.perf_counter()
for it in range(3):
    t0 = sync(); batch = algo.collect(); t1 = sync()
    d = algo.process(batch); t2 = sync()
    pol = algo.policy
    obs, act, adv, old_mean, old_lstd = d["obs"], d["act"], d["adv"], d["mean"], d["log_std"]
    old_ll = pol.log_likelihood(act, old_mean, old_lstd)
    def surrogate():
        mean, log_std = pol.dist_info(obs)
        lr = (pol.log_likelihood(act, mean, log_std) - old_ll).exp()
        return -(lr * adv).mean(), pol.kl(old_mean, old_lstd, mean, log_std).mean()
    t3 = sync(); loss, _ = surrogate(); g = T.flat_grad(loss, pol); t4 = sync()
    fisher = T.AnalyticFisher(pol, obs); t5 = sync()
    for k in range(11): hv = fisher(g)
    t6 = sync()
    with torch.no_grad():
        for k in range(3): surrogate()
    t7 = sync()
    print(json.dumps(dict(itr=it, rollout_ms=(t1-t0)*1e3, process_ms=(t2-t1)*1e3, grad_ms=(t4-t3)*1e3, fisher_setup_ms=(t5-t4)*1e3,
                          fvp11_ms=(t6-t5)*1e3, surrogate3_ms=(t7-t6)*1e3)))
    stats = algo.optimize(d)

#!/usr/bin/env python3
"""Where one TRPO iteration spends its time (not a test): 65 536 envs x 8 steps = 524 288 samples (KIND / MODE from the environment,
default walk env / PD = the configuration of the 20.6 M figure).  Synchronising timers around the phases of collect() / process() /
optimize(), the latter re-enacted with the same calls optimize() makes."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cassierl_amd import trpo as T

n, hor = int(os.environ.get("N_ENVS", "65536")), 8
from cassierl_amd.trajectory import default_gait
algo = T.make_cassie_trpo(n, kind=os.environ.get("KIND", "walk"), control_mode=os.environ.get("MODE", "PD"), trajectory=default_gait(), batch_size=n * hor)
def sync(): torch.cuda.synchronize(); return time.perf_counter()
for it in range(4):
    t0 = sync(); batch = algo.collect(); t1 = sync()
    d = algo.process(batch); t2 = sync()
    pol = algo.policy
    obs, act, adv, old_mean, old_lstd = d["obs"], d["act"], d["adv"], d["mean"], d["log_std"]
    old_ll = pol.log_likelihood(act, old_mean, old_lstd)
    def surrogate():
        mean, log_std = pol.dist_info(obs)
        lr = (pol.log_likelihood(act, mean, log_std) - old_ll).exp()
        return -(lr * adv).mean(), pol.kl(old_mean, old_lstd, mean, log_std).mean()
    t3 = sync()
    fisher = T.FusedFisher(pol, obs); t4 = sync()
    with torch.no_grad():
        std = old_lstd.exp(); z = (act - old_mean) / std
        g = fisher.vjp(-(adv.unsqueeze(-1) * z / std) / obs.shape[0])
    t5 = sync()
    for k in range(11): hv = fisher(g)
    t6 = sync()
    descent = T.conjugate_gradient(lambda v: fisher(v) + 1e-5 * v, g, 10); t7 = sync()
    with torch.no_grad():
        for k in range(2): l, kl = surrogate(); float(l); float(kl)
    t8 = sync()
    stats = algo.optimize(d); t9 = sync()
    print(json.dumps(dict(itr=it, rollout_ms=(t1-t0)*1e3, process_ms=(t2-t1)*1e3, old_ll_ms=(t3-t2)*1e3, fisher_setup_ms=(t4-t3)*1e3, grad_ms=(t5-t4)*1e3,
                          fvp11_ms=(t6-t5)*1e3, cg10_ms=(t7-t6)*1e3, surrogate2_ms=(t8-t7)*1e3, optimize_ms=(t9-t8)*1e3)))

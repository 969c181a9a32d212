#!/usr/bin/env python3
"""Where a TRPO update spends its time (not a test): python tools/time_trpo_parts.py [envs] [horizon]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cassierl_amd.trajectory import Cassie2dTraj
from cassierl_amd.trpo import make_cassie_trpo, AnalyticFisher, flat_grad, conjugate_gradient

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
T = int(sys.argv[2]) if len(sys.argv) > 2 else 8
d = np.load(os.path.join(ROOT, "tests", "golden", "traj2d.npz"))
algo = make_cassie_trpo(n, trajectory=Cassie2dTraj.from_arrays(d["time"], d["qpos"]), batch_size=n * T)
algo.train_iteration()


def timed(name, fn, reps=1):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    print("%-28s %8.2f ms" % (name, (time.perf_counter() - t0) / reps * 1e3), flush=True)
    return out

batch = timed("collect (rollout)", algo.collect)
dd = timed("process (returns, baseline)", lambda: algo.process(batch))
pol = algo.policy
obs, act, adv, old_mean, old_lstd = dd["obs"], dd["act"], dd["adv"], dd["mean"], dd["log_std"]
old_ll = pol.log_likelihood(act, old_mean, old_lstd)
def surrogate():
    mean, log_std = pol.dist_info(obs)
    lr = (pol.log_likelihood(act, mean, log_std) - old_ll).exp()
    return -(lr * adv).mean(), pol.kl(old_mean, old_lstd, mean, log_std).mean()
g = timed("surrogate + gradient", lambda: flat_grad(surrogate()[0], pol))
fisher = timed("AnalyticFisher setup", lambda: AnalyticFisher(pol, obs))
v = torch.randn_like(g)
timed("one Fisher-vector product", lambda: fisher(v), 10)
timed("CG (10 iterations)", lambda: conjugate_gradient(lambda x: fisher(x) + 1e-5 * x, g, 10))
with torch.no_grad():
    timed("surrogate eval (line search)", surrogate, 5)
timed("whole optimize()", lambda: algo.optimize(dd))

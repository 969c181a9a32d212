"""Per-pass time and per-step fixed cost of the first tier (not a test): Env.step of 65 536 envs timed for 1, 2, 5, 10, 20 substeps, straight-line fit."""
import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
import torch
from cassierl_amd import rollout as R
from cassierl_amd.trajectory import default_gait
from cassierl_amd.vec_env import CassieVecEnv
g = default_gait(); n = 65536
res = {}
for kind, mode, ar in (("walk", "PD", True), ("walk", "PD", False), ("stand", "PD", True)):
    for ns in (1, 2, 5, 10, 20):
        env = CassieVecEnv(n, kind=kind, control_mode=mode, n_substeps=ns, auto_reset=ar)
        env.set_trajectory(g.time, g.qpos)
        bufs = env.alloc(); env.reset(bufs)
        ids = torch.arange(n, device="cuda"); lo, hi = env.action_space.low, env.action_space.high
        for t in range(5): env.step(R.random_actions(1, ids, t, lo, hi), bufs)
        ms = [env.time_steps(R.random_actions(1, ids, 5 + t, lo, hi), 20, bufs) for t in range(5)]
        res[(kind, ar, ns)] = float(np.median(ms)); env.close()
        print(kind, ar, ns, res[(kind, ar, ns)])
for kind, ar in (("walk", True), ("walk", False), ("stand", True)):
    xs = np.array([1, 2, 5, 10, 20.0]); ys = np.array([res[(kind, ar, int(x))] for x in xs])
    A = np.vstack([xs, np.ones_like(xs)]).T; p, c = np.linalg.lstsq(A, ys, rcond=None)[0]
    print(kind, ar, "ms per pass %.4f  intercept %.4f" % (p, c))

#!/usr/bin/env python3
"""Counterpart of rllab/envs/sim_policy.py:19-31 on the batched MI355X environment: load a snapshot written by train_trpo.py
(`--snapshot`, snapshot_mode="last") and roll the policy out -- no training.  The reference animates ONE env through rllab's
`rollout(env, policy, max_path_length, animated=True)`; here N resident envs run the same loop in parallel (there is no
viewer: GUI is out of scope) and the script prints what the reference's loop would let one read off the screen: path
lengths and returns.

    python sim_policy.py snapshot.pt --envs 1024 --max-path-length 1000 [--kind stand --control-mode Torque] [--deterministic]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("file", help="snapshot written by train_trpo.py --snapshot")
    ap.add_argument("--envs", type=int, default=1024)
    ap.add_argument("--max-path-length", type=int, default=1000)   # sim_policy.py:14 default
    ap.add_argument("--kind", default="walk", choices=["walk", "stand"])
    ap.add_argument("--control-mode", default="PD", choices=["PD", "Torque", "OSC"])
    ap.add_argument("--deterministic", action="store_true", help="act with the policy mean (no exploration noise)")
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    import torch
    from cassierl_amd import rollout as R
    from cassierl_amd.trajectory import default_gait
    from cassierl_amd.trpo import make_cassie_trpo
    algo = make_cassie_trpo(args.envs, kind=args.kind, control_mode=args.control_mode, device=0, trajectory=default_gait(), seed=args.seed)
    _, _ = algo.load(args.file, restore_sampler=False)   # policy + baseline only: every path starts from env.reset()
    pol, n = algo.policy, args.envs
    dt = next(pol.parameters()).dtype
    obs = algo.env_reset().clone()
    alive = torch.ones(n, dtype=torch.bool, device=obs.device)
    ret = torch.zeros(n, dtype=torch.float64, device=obs.device)
    length = torch.zeros(n, dtype=torch.int64, device=obs.device)
    with torch.no_grad():
        for t in range(args.max_path_length):     # rllab.sampler.utils.rollout: until done or max_path_length
            mean, log_std = pol.dist_info(obs.to(dt))
            a = mean if args.deterministic else mean + R.counter_normal(args.seed, algo.env_ids, t, mean.shape[1]).to(dt) * log_std.exp()
            obs, rew, done = algo.env_step(algo.act_map(a))
            ret += torch.where(alive, rew, torch.zeros_like(rew))
            length += alive.to(torch.int64)
            alive &= ~done.bool()
            obs = obs.clone()
            if t % 50 == 49 and not bool(alive.any()):
                break
    print(json.dumps(dict(snapshot=args.file, itr=algo.itr, envs=n, max_path_length=args.max_path_length, deterministic=args.deterministic,
                          avg_return=float(ret.mean()), min_return=float(ret.min()), max_return=float(ret.max()),
                          avg_path_length=float(length.double().mean()), paths_reaching_max_length=int(alive.sum()))))
    algo.env.close()


if __name__ == "__main__":
    main()

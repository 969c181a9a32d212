#!/usr/bin/env python3
"""Counterpart of rllab/envs/trpo_cassie.py on the batched MI355X environment (BASELINE.json config 4 when launched with
`python -m torch.distributed.run --nproc-per-node 8 train_trpo.py --envs-per-gpu 65536`).

Hyper-parameters are those of trpo_cassie.py:21-42 (MLP 32x32, init_std 2.0, discount 0.99, step_size 0.005,
max_path_length 1000); batch_size defaults to one Env.step of every environment per iteration times --horizon.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs-per-gpu", type=int, default=4096)
    ap.add_argument("--horizon", type=int, default=4, help="Env.steps per environment per iteration")
    ap.add_argument("--n-itr", type=int, default=10)
    ap.add_argument("--kind", default="walk", choices=["walk", "stand"])
    ap.add_argument("--control-mode", default="PD", choices=["PD", "Torque", "OSC"])
    ap.add_argument("--snapshot", default="")
    ap.add_argument("--load-policy", default="")
    ap.add_argument("--timing", action="store_true", help="report rollout / update seconds separately (adds synchronisations)")
    ap.add_argument("--dump-params", default="", help="rank 0 writes the flat policy parameters (.npy) after the last iteration")
    args = ap.parse_args()
    import torch
    from cassierl_amd import rollout as R
    from cassierl_amd.trajectory import default_gait
    from cassierl_amd.trpo import make_cassie_trpo
    rank, local_rank, world = R.init_distributed()
    dev = R.local_device(local_rank) if world > 1 else 0   # CASSIE_DEVICE_MAP (test hook): several ranks on one GPU
    torch.cuda.set_device(dev)
    traj = default_gait()
    algo = make_cassie_trpo(args.envs_per_gpu, kind=args.kind, control_mode=args.control_mode, device=dev,
                            trajectory=traj, seed=1, batch_size=args.envs_per_gpu * world * args.horizon)
    algo.timing = args.timing
    if args.load_policy:
        _, restored = algo.load(args.load_policy)
        if rank == 0:
            print(json.dumps(dict(loaded=args.load_policy, itr=algo.itr, sampler_restored=restored)))
    for _ in range(args.n_itr):
        t0 = time.perf_counter()
        st = algo.train_iteration()
        torch.cuda.synchronize()
        st["seconds"] = time.perf_counter() - t0
        st["env_steps_per_s"] = st["env_steps"] / st["seconds"]
        if rank == 0:
            print(json.dumps(st))
        if args.snapshot:
            algo.save(args.snapshot)  # snapshot_mode="last"
    if args.dump_params and rank == 0:
        from cassierl_amd.trpo import flat_params
        np.save(args.dump_params, flat_params(algo.policy).double().cpu().numpy())
    if R.dist.is_initialized():
        R.dist.barrier()
        R.dist.destroy_process_group()


if __name__ == "__main__":
    main()

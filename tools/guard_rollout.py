#!/usr/bin/env python3
"""Rollouts of the 64-environments-per-wavefront kernels (env_step_duo_kernel / env_step_duo_hf_kernel) whose every output is saved, for the
build-guard test (tests/test_gpu_build_guard.py): run once per build of the library (CASSIE2D_LIB), the saved files are compared bit for bit.
usage: python tools/guard_rollout.py <out.npz>   (needs the GPU; the library is the one CASSIE2D_LIB names, or the in-tree build)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PD_LO, PD_HI = np.radians([-50, -164, -140] * 2), np.radians([80, -37, -30] * 2)
TQ = np.array([12.0, 12.0, 0.9] * 2)


def main(out):
    import torch
    from cassierl_amd import rollout as R
    from cassierl_amd.trajectory import default_gait
    from cassierl_amd.vec_env import CassieVecEnv, LEG_TIER_ON, DUO_TIER_ON
    fl = LEG_TIER_ON | DUO_TIER_ON | int(os.environ.get("GUARD_EXTRA_FLAGS", "0"), 0)
    g = default_gait()
    res = {}

    def run(tag, n, steps, seed, lo, hi, hf=None, shift=False, **kw):
        env = CassieVecEnv(n, flags=fl, **kw)
        if kw["kind"] == "walk":
            env.set_trajectory(g.time, g.qpos)
        if hf is not None:
            env.set_heightfield(hf, 10.0, 10.0)
        bufs = env.alloc()
        env.reset(bufs)
        if shift:
            s = env.get_full_state_host()
            s[:, 0] += np.linspace(-6.0, 6.0, n)
            s[:, 1] += 0.03
            env.set_full_state_host(s)
        ids = torch.arange(n, device="cuda:0")
        obs, rew, done = [], [], []
        for t in range(steps):
            o, r, d = env.step(R.random_actions(seed, ids, t, lo, hi), bufs)
            obs.append(o.cpu().numpy().copy()); rew.append(r.cpu().numpy().copy()); done.append(d.cpu().numpy().copy())
        res[tag + "_obs"], res[tag + "_rew"], res[tag + "_done"] = np.array(obs), np.array(rew), np.array(done)
        res[tag + "_state"] = env.get_full_state_host()
        res[tag + "_cleanup"] = np.array([env.counters()["cleanup_substeps"]])
        env.close()

    # the bench's regime: walk env, PD, every step resets (reference semantics); 4 141 envs = 64 full wavefronts + a partly filled one
    run("walk_pd", 4141, 25, 1, PD_LO, PD_HI, kind="walk", control_mode="PD", n_substeps=10, auto_reset=True)
    # robots that move, fall, hit their joint limits: six-row joint sweeps, eight-row pair sweeps inside the kernel, hand-overs
    run("stand_pd", 4141, 40, 2, PD_LO, PD_HI, kind="stand", control_mode="PD", n_substeps=10, auto_reset=True)
    run("stand_tq", 2077, 70, 3, -TQ, TQ, kind="stand", control_mode="Torque", n_substeps=10, auto_reset=False)
    # the height-field kernel
    xs = np.linspace(-10.0, 10.0, 2001)
    relief = np.tile(0.015 * (1.0 - np.cos(2.0 * np.pi * xs / 1.5)), (64, 1))
    run("hf_pd", 2077, 20, 6, PD_LO, PD_HI, hf=relief, shift=True, kind="stand", control_mode="PD", n_substeps=10, auto_reset=True)
    # kernel time of the headline workload with this build / these flags (tells a taken experiment branch from one that is not)
    env = CassieVecEnv(65536, kind="walk", control_mode="PD", n_substeps=10, auto_reset=True, flags=fl)
    env.set_trajectory(g.time, g.qpos)
    bufs = env.alloc(); env.reset(bufs)
    ids = torch.arange(65536, device="cuda:0")
    for t in range(5):
        env.step(R.random_actions(1, ids, t, PD_LO, PD_HI), bufs)
    res["ms_per_65536_env_step"] = np.array([env.time_steps(R.random_actions(1, ids, 5, PD_LO, PD_HI), 10, bufs)])
    ws = env.debug_workspace_host()
    wg = ws.shape[1] // 2                                                                        # Duo::W_GROUP
    res["ms_view_marks"] = np.array([float(ws[:, wg - 2, :].sum() + ws[:, 2 * wg - 2, :].sum())])   # Duo::W_DESC + 8 of both groups: written by joint_solve_view only
    env.close()
    np.savez(out, **res)


if __name__ == "__main__":
    main(sys.argv[1])

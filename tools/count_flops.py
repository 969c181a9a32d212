#!/usr/bin/env python3
"""Counted USEFUL arithmetic of one Env.step of the bench workload ("replace by a counted figure from an op-counting
build") -- not a test.  The source of the two-lanes-per-environment kernel is compiled for the CPU with an operation-counting lane
type (tests/host_emul/leg_host.cpp): +, -, * count 1 (a*b+c counts 2), /, sqrt, 1/x, exp count 1, sincos 2; comparisons, selects and
data movement count 0; inside a Gauss-Seidel step only the lane of the leg that owns the row is counted.  The figure is therefore the
arithmetic the ALGORITHM needs in this formulation, independent of how many lanes issue it -- to be read against the ISSUED FP64
lane-flops the hardware counters give (profiles/<tag>_pmc.json): useful / issued is the share of issue slots doing needed work.
Writes profiles/useful_flops.json (read by profiles/summarize_pmc.py -> bench.py's fp64_valu.useful_frac)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def count_cassie3d(n=16, warm=30, steps=30):
    """configs[4] as the bench line runs it: Cassie3d, torque mode, U(+-ctrlrange) redrawn every env-step (stream 5 of rollout.random_actions),
    30 env-steps of warm-up from the standing pose, 30 counted (the robots fall during the run), no reset.  Source of the lane-per-leg kernel
    (cassierl_amd/csrc/cassie3d_leg_core.h) through oracle/leg_host/leg3d_host.cpp with the counting lane type; inside a Gauss-Seidel step
    only the owner leg's lane counts.  An environment that leaves the row capacity of the kernel is frozen here (the lower tier of the GPU
    path finishes it there): its substeps are not counted, their share is reported."""
    import ctypes as ct
    import subprocess
    import torch
    import oracle_py as O
    from cassierl_amd import rollout as R
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "libleg3d_host_count.so"])
    L = ct.CDLL(os.path.join(ROOT, "oracle", "libleg3d_host_count.so"))
    L.leg3d_host_ops.restype = ct.c_double
    assert L.leg3d_host_lanes() == 2
    CTRL = np.array([4.5, 4.5, 12.2, 12.2, 0.9] * 2)
    o = O.Oracle3D()
    q, v = o.state()
    rec = np.zeros(80)
    rec[:21], rec[21:41], rec[41:61] = q, v, o.warmstart()
    state = np.tile(rec, (n, 1)).copy()
    dp, ip = ct.POINTER(ct.c_double), ct.POINTER(ct.c_int)
    pend, nit = np.zeros(n, dtype=np.int32), np.zeros(n, dtype=np.int32)
    ids = torch.arange(n)
    sweeps, frozen, done = 0.0, 0, 0
    for t in range(warm + steps):
        if t == warm:
            L.leg3d_host_ops()
        a = np.ascontiguousarray(R.random_actions(5, ids, t, -CTRL, CTRL).numpy())
        L.leg3d_host_step(state.ctypes.data_as(dp), a.ctypes.data_as(dp), n, 10, 1, pend.ctypes.data_as(ip), nit.ctypes.data_as(ip), None)
        if t >= warm:
            sweeps += nit.sum(); frozen += int(pend.sum()); done += 10 * n - int(pend.sum())
    ops = L.leg3d_host_ops()
    return dict(flop_per_env_step=ops / max(1, done) * 10.0, pgs_sweeps_per_env_step=float(sweeps) / max(1, done) * 10.0, envs=n, env_steps=steps,
                substeps_not_counted_frac=frozen / (10.0 * n * steps),
                convention="as pd_bench; per env-step of 10 substeps CARRIED OUT by this kernel; robots fall during the counted steps")


def count_fallen(n=16, warm=150, steps=12):
    """The floor of the PD / torque path (bench row torque_random_no_reset_fallen): stand env, torque mode, U(+-ctrlrange) (stream 3), no reset -- after the
    warm-up the robots lie on the ground.  An environment that needs more than eight rows on a leg leaves this kernel (on the GPU a lower tier finishes its
    step); here it is put back on its feet and falls again: its substeps are not counted, their share is reported."""
    import torch
    import oracle_py as O
    from conftest import state_vec
    from leg_host import LegHostEnv, lib
    from cassierl_amd import rollout as R
    lo, hi = -np.array([12.2, 12.2, 0.9] * 2), np.array([12.2, 12.2, 0.9] * 2)
    env = LegHostEnv(n, kind="stand", control_mode="Torque", n_substeps=10, auto_reset=False)
    oe = O.OracleEnv("stand", "Torque")
    oe.reset()
    q, v = oe.oracle.state()
    ctor = O.Oracle()
    s0 = state_vec(q, v, oe.oracle.warmstart(), kq=ctor.state()[0], kv=ctor.state()[1], qstate=q)
    env.set_full_state_host(np.tile(s0, (n, 1)))
    ids = torch.arange(n)
    done, left, sweeps, pairs, z = 0, 0, 0.0, 0, []
    for t in range(warm + steps):
        if t == warm:
            lib().leg_host_ops()
        env.step_host(R.random_actions(3, ids, t, lo, hi).numpy())
        over = env.pending > 0
        if t >= warm:
            done += 10 * n - int(env.pending.sum()); left += int(env.pending.sum()); sweeps += env.state[~over, 85].sum(); pairs += int((~over).sum()); z.append(float(env.state[~over, 1].mean()))
        env.state[over] = s0
        env.pending[:] = 0
    ops = lib().leg_host_ops()
    return dict(flop_per_env_step=ops / max(1, done) * 10.0, pgs_sweeps_per_env_step=float(sweeps) / max(1, pairs), envs=n, env_steps=steps,
                substeps_not_counted_frac=left / (10.0 * n * steps), mean_pelvis_height=float(np.mean(z)),
                convention="as pd_bench; per env-step of 10 substeps CARRIED OUT by this kernel; robots on the ground (no reset); an environment that leaves the "
                           "kernel's row capacity is put back on its feet")


def main():
    import torch
    import oracle_py as O
    from conftest import state_vec
    from leg_host import LegHostEnv, lib
    from cassierl_amd import rollout as R
    from cassierl_amd.trajectory import default_gait
    g = default_gait()
    out = {}
    for name, kind, mode, seed, lo, hi in (("pd_bench", "walk", "PD", 1, np.radians([-50, -164, -140] * 2), np.radians([80, -37, -30] * 2)),
                                           ("stand_torque_random", "stand", "Torque", 3, -np.array([12.2, 12.2, 0.9] * 2), np.array([12.2, 12.2, 0.9] * 2))):
        n, steps = 16, 12
        env = LegHostEnv(n, kind=kind, control_mode=mode, n_substeps=10, auto_reset=True)
        env.set_trajectory(g.time, g.qpos)
        oe = O.OracleEnv(kind, mode, traj=dict(time=g.time, qpos=g.qpos))
        oe.reset()
        q, v = oe.oracle.state()
        ctor = O.Oracle()
        env.set_full_state_host(np.tile(state_vec(q, v, oe.oracle.warmstart(), kq=ctor.state()[0], kv=ctor.state()[1], qstate=q), (n, 1)))
        ids = torch.arange(n)
        for t in range(4):   # spread the batch a little (the stand workload; the walk workload resets every step, quirk Q3)
            env.step_host(R.random_actions(seed, ids, t, lo, hi).numpy())
        lib().leg_host_ops()
        sweeps = 0.0
        for t in range(steps):
            env.step_host(R.random_actions(seed, ids, 4 + t, lo, hi).numpy())
            sweeps += env.state[:, 85].sum()
        ops = lib().leg_host_ops()
        out[name] = dict(flop_per_env_step=ops / (n * steps), pgs_sweeps_per_env_step=sweeps / (n * steps), envs=n, env_steps=steps,
                         convention="+,-,* = 1 (a*b+c = 2); /, sqrt, 1/x, exp = 1; sincos = 2; compare/select/move = 0; in a Gauss-Seidel step only "
                                    "the owner leg's lane counts; the reset pass of a terminated environment is included")
        print(name, json.dumps(out[name]))
    out["torque_random_no_reset_fallen"] = count_fallen()
    print("torque_random_no_reset_fallen", json.dumps(out["torque_random_no_reset_fallen"]))
    out["cassie3d_torque_random"] = count_cassie3d()
    print("cassie3d_torque_random", json.dumps(out["cassie3d_torque_random"]))
    with open(os.path.join(ROOT, "profiles", "useful_flops.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()

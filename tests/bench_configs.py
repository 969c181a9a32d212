#!/usr/bin/env python3
"""Throughput of the other BASELINE.json configs (not the bench line, not a test): config 3 (OSC standing controller
in the loop, 65 536 envs, n = 1 and n = 10 substeps per controller call), stand-env OSC Env.step, Jacobian squat."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from cassierl_amd.vec_env import CassieVecEnv, CONTROL_MODES  # noqa: E402


def timed(fn, reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    out = []
    for n in (4096, 65536):
        env = CassieVecEnv(n, kind="stand", control_mode="OSC", n_substeps=10, auto_reset=True)
        env.reset()
        rng = np.random.default_rng(0)
        zp = torch.as_tensor(0.9 + 0.01 * rng.uniform(-1, 1, n), device="cuda")
        zv = torch.zeros(n, dtype=torch.float64, device="cuda")
        for nsub in (1, 10):
            f = lambda: env._chk(env.L.CassieVecStandingStep(env.h, CONTROL_MODES["OSC"], zp.data_ptr(), zv.data_ptr(), nsub))
            f(); f()
            dt = timed(f, 10 if nsub == 1 else 5)
            out.append(dict(config="standing_controller_osc in loop", n_envs=n, substeps_per_call=nsub, ms=dt * 1e3,
                            controller_substeps_per_s=n * nsub / dt, env_steps_per_s_equiv=n * nsub / dt / 10))
        q, v = env.get_state_host()
        out[-1]["z_mean"] = float(q[:, 1].mean()); out[-1]["finite"] = bool(np.isfinite(q).all())
        a = torch.as_tensor(rng.uniform(-1, 1, (n, 7)) * np.array([3, 3, 1, 1, 1, 1, 3.0]), device="cuda")
        a[:, 3].abs_(); a[:, 5].abs_()
        bufs = env.alloc()
        g = lambda: env.step(a, bufs)
        g(); g()
        dt = timed(g, 5)
        out.append(dict(config="cassie_stand2d Env.step, OSC mode (QP every substep)", n_envs=n, ms=dt * 1e3, env_steps_per_s=n / dt))
        f2 = lambda: env._chk(env.L.CassieVecStandingStep(env.h, CONTROL_MODES["Jacobian"], zp.data_ptr(), zv.data_ptr(), 10))
        f2(); dt = timed(f2, 5)
        out.append(dict(config="standing_controller_jacobian in loop", n_envs=n, substeps_per_call=10, ms=dt * 1e3,
                        controller_substeps_per_s=n * 10 / dt))
        env.close()
    for o in out:
        print(json.dumps(o))


if __name__ == "__main__":
    main()

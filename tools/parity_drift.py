#!/usr/bin/env python3
"""Free-running parity drift (not a pytest file): HIP path vs CPU oracle from the same initial state and the same action
stream, no teacher forcing, error recorded every 500 substeps.  Writes one JSON line per model.

  python tools/parity_drift.py [substeps]     (GPU box; ~1 min for 10 000 substeps)
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_py as O  # noqa: E402
from conftest import state_vec  # noqa: E402
from cassierl_amd.vec_env import CassieVecEnv  # noqa: E402
from cassierl_amd.vec_env3d import Cassie3dVec, CTRL_RANGE, state_record  # noqa: E402


def rel(a, b):
    return float(np.abs(a - b).max() / (1.0 + np.abs(b).max()))


def main():
    total = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    rng = np.random.default_rng(7)
    # ---- Cassie2d, torque mode (PD mode is chaotic in the reference itself), smooth random torques
    env = CassieVecEnv(1, kind="stand", control_mode="Torque", n_substeps=10, auto_reset=False)
    o = O.Oracle()
    q, v = o.state()
    env.set_full_state_host(state_vec(q, v, o.warmstart())[None])
    tq = np.array([12.0, 12.0, 0.9] * 2)
    hist, u = [], np.zeros(6)
    for blk in range(total // 10):
        if blk % 20 == 0:
            u = rng.uniform(-0.25, 0.25, 6) * tq
        env.substep_host("Torque", u[None], 10)
        for _ in range(10):
            o.step_torque(u)
        if (blk + 1) % 50 == 0:
            s = env.get_full_state_host()[0]
            q1, v1 = o.state()
            hist.append(dict(substep=(blk + 1) * 10, qpos=rel(s[:13], q1), qvel=rel(s[13:26], v1), pelvis_z=float(q1[1])))
    print(json.dumps(dict(model="cassie2d", mode="torque", free_running=True, substeps=total, worst_qpos=max(h["qpos"] for h in hist),
                          worst_qvel=max(h["qvel"] for h in hist), history=hist)))
    env.close()
    # ---- Cassie3d, torque mode
    e3 = Cassie3dVec(1)
    o3 = O.Oracle3D()
    q, v = o3.state()
    e3.set_state_host(state_record(q, v, o3.warmstart())[None])
    hist, u = [], np.zeros(10)
    for blk in range(total // 10):
        if blk % 20 == 0:
            u = rng.uniform(-0.25, 0.25, 10) * CTRL_RANGE
        e3.step_host(u[None], 10)
        for _ in range(10):
            o3.step_torque(u)
        if (blk + 1) % 50 == 0:
            s = e3.get_state_host()[0]
            q1, v1 = o3.state()
            hist.append(dict(substep=(blk + 1) * 10, qpos=rel(s[:21], q1), qvel=rel(s[21:41], v1), pelvis_z=float(q1[2]), nefc=int(s[73])))
    print(json.dumps(dict(model="cassie3d", mode="torque", free_running=True, substeps=total, worst_qpos=max(h["qpos"] for h in hist),
                          worst_qvel=max(h["qvel"] for h in hist), history=hist)))
    e3.close()


if __name__ == "__main__":
    main()

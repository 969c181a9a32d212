#!/usr/bin/env python3
"""Ad-hoc GPU bring-up check (not a pytest file): stage-by-stage comparison of the HIP kernels
with the planar spec / 3-D oracle.  Run on the GPU box: python tests/gpu_check.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from cassierl_amd.vec_env import CassieVecEnv  # noqa: E402
from oracle_py import Oracle, OracleEnv  # noqa: E402
from planar_proto import Planar  # noqa: E402

DBG = dict(M=0, BIAS=169, QS=182, F0=195, B=241, R=287, AREF=333, ADIAG=379, F=425, QACC=471, QACCH=484)


def state_vec(q, v, ws, kq=None, kv=None, ctrl=None):
    s = np.zeros(88)
    s[0:13], s[13:26], s[26:39] = q, v, ws
    s[39:52] = q if kq is None else kq
    s[52:65] = v if kv is None else kv
    if ctrl is not None:
        s[78:84] = ctrl
    return s


def main():
    P = Planar()
    traj = np.load(os.path.join(ROOT, "tests", "golden", "traj2d.npz"))
    rng = np.random.default_rng(0)
    n = 4
    env = CassieVecEnv(n, kind="stand", control_mode="Torque", n_substeps=1, auto_reset=False)
    s = env.get_full_state_host()
    print("ctor state q", s[0, :13])
    print("ctor ws   ", s[0, 26:39])
    o = Oracle()
    print("oracle ws ", o.warmstart())
    print("ws err", np.abs(s[0, 26:39] - o.warmstart()).max())
    # ---- stage check on a random state
    q = s[0, :13] + rng.uniform(-0.05, 0.05, 13); q[1] -= 0.01
    v = rng.uniform(-1, 1, 13); ws = rng.uniform(-5, 5, 13)
    u = rng.uniform(-1, 1, 6) * np.array([12, 12, .9] * 2)
    env.set_full_state_host(np.tile(state_vec(q, v, ws), (n, 1)))
    dbg = env.debug_substep_host("Torque", np.tile(u, (n, 1)))[0]
    q2, v2, qacc, r = P.step(q, v, ws, u)
    def cmp(name, a, b):
        a, b = np.asarray(a).ravel(), np.asarray(b).ravel()
        print("%-8s max abs err %.3e  (scale %.3e)" % (name, np.abs(a - b).max(), np.abs(b).max()))
    cmp("M", dbg[DBG["M"]:DBG["M"] + 169], r["M"])
    cmp("bias", dbg[DBG["BIAS"]:DBG["BIAS"] + 13], r["bias"])
    cmp("qs", dbg[DBG["QS"]:DBG["QS"] + 13], r["qacc_smooth"])
    cmp("R", dbg[DBG["R"]:DBG["R"] + 46], np.where(r["active"], r["R"], 0))
    cmp("aref", dbg[DBG["AREF"]:DBG["AREF"] + 46], np.where(r["active"], r["aref"], 0))
    cmp("b", dbg[DBG["B"]:DBG["B"] + 46], np.where(r["active"], r["b"], 0))
    cmp("Adiag", dbg[DBG["ADIAG"]:DBG["ADIAG"] + 46], np.where(r["active"], np.diag(r["A"]), 0))
    cmp("f0", dbg[DBG["F0"]:DBG["F0"] + 46], r["f0"])
    cmp("f", dbg[DBG["F"]:DBG["F"] + 46], r["f"])
    cmp("qacc", dbg[DBG["QACC"]:DBG["QACC"] + 13], r["qacc"])
    sg = env.get_full_state_host()[0]
    cmp("q'", sg[:13], q2); cmp("v'", sg[13:26], v2); cmp("ws'", sg[26:39], qacc)
    print("niter gpu", sg[85], "proto", r["niter"], "active", int(r["active"].sum()))
    # ---- teacher-forced trajectory vs the 3-D oracle, torque and PD modes
    for mode in ("Torque", "PD"):
        o = Oracle()
        lo, hi = env.action_space.low, env.action_space.high
        if mode == "PD":
            lo, hi = np.radians([-50, -164, -140] * 2), np.radians([80, -37, -30] * 2)
        worst = 0.0
        a = None
        for i in range(600):
            if i % 10 == 0:
                a = rng.uniform(lo, hi)
            qo, vo = o.state(); wso = o.warmstart()
            env.set_full_state_host(np.tile(state_vec(qo, vo, wso), (n, 1)))
            env.substep_host(mode, np.tile(a, (n, 1)), 1)
            (o.step_torque if mode == "Torque" else o.step_pd)(a)
            sg = env.get_full_state_host()[1]
            q1, v1 = o.state()
            e = max(np.abs(sg[:13] - q1).max(), np.abs(sg[13:26] - v1).max() / (1 + np.abs(v1).max()))
            worst = max(worst, e)
        print("teacher-forced %s: worst per-step err %.3e" % (mode, worst))
    # ---- free-running torque mode, 1000 substeps
    o = Oracle()
    qo, vo = o.state()
    env.set_full_state_host(np.tile(state_vec(qo, vo, o.warmstart()), (n, 1)))
    maxrel = 0.0
    for i in range(100):
        a = rng.uniform(-1, 1, 6) * np.array([12, 12, .9] * 2)
        env.substep_host("Torque", np.tile(a, (n, 1)), 10)
        for _ in range(10):
            o.step_torque(a)
        sg = env.get_full_state_host()[2]
        q1, v1 = o.state()
        rel = max(np.abs(sg[:13] - q1).max() / (np.abs(q1).max()), np.abs(sg[13:26] - v1).max() / (1e-3 + np.abs(v1).max()))
        maxrel = max(maxrel, rel)
    print("free-running torque 1000 substeps: max rel err %.3e (z=%.3f)" % (maxrel, q1[1]))
    env.close()
    # ---- Env.step (walk, PD) vs OracleEnv
    tr = dict(time=traj["time"], qpos=traj["qpos"])
    env = CassieVecEnv(n, kind="walk", control_mode="PD", n_substeps=10, auto_reset=True)
    env.set_trajectory(traj["time"], traj["qpos"])
    oe = [OracleEnv("walk", "PD", traj=tr) for _ in range(n)]
    ob_g = env.reset_host()
    ob_o = np.array([e.reset() for e in oe])
    print("reset obs err", np.abs(ob_g - ob_o).max())
    lo, hi = np.radians([-50, -164, -140] * 2), np.radians([80, -37, -30] * 2)
    for t in range(12):
        acts = rng.uniform(lo, hi, size=(n, 6))
        og, rg, dg = env.step_host(acts)
        res = [e.step(acts[i]) for i, e in enumerate(oe)]
        oo = np.array([r[0] for r in res]); ro = np.array([r[1] for r in res]); do = np.array([r[2] for r in res])
        for i, e in enumerate(oe):
            if do[i]:
                oo[i] = e.reset()
        print("step %d obs err %.3e rew err %.3e done %s/%s" % (t, np.abs(og - oo).max(), np.abs(rg - ro).max(), dg.astype(int), do.astype(int)))
    env.close()
    # ---- quick throughput
    import torch
    for nenv in (4096, 65536):
        env = CassieVecEnv(nenv, kind="walk", control_mode="PD", n_substeps=10, auto_reset=True)
        env.set_trajectory(traj["time"], traj["qpos"])
        out = env.alloc()
        env.reset(out)
        acts = torch.as_tensor(rng.uniform(lo, hi, size=(nenv, 6)), device="cuda")
        ms = env.time_steps(acts, 3, out)
        ms = env.time_steps(acts, 10, out)
        print("n=%d  %.3f ms/step  %.3f M env-steps/s" % (nenv, ms, nenv / ms / 1e3))
        env.close()


if __name__ == "__main__":
    main()

"""The lane-per-leg Cassie3d kernel source (cassierl_amd/csrc/cassie3d_leg_core.h), compiled for the CPU by
oracle/leg_host/leg3d_host.cpp, against the Cassie3d oracle: the checks tests/test_gpu_cassie3d.py runs on the GPU through the C-ABI,
here without one -- so that a formulation error (block mass matrix, factorisation, matrix-free rows, cone updates) is found on the CPU."""
import numpy as np
import pytest

from leg3d_host import Leg3dHostVec

CTRL = np.array([4.5, 4.5, 12.2, 12.2, 0.9] * 2)


def record(q, v, ws=None):
    s = np.zeros(80)
    s[0:21], s[21:41] = q, v
    if ws is not None:
        s[41:61] = ws
    return s


def err(sg, q1, v1):
    return max(np.abs(sg[:21] - q1).max(), np.abs(sg[21:41] - v1).max() / (1 + np.abs(v1).max()))


def test_forward_from_random_states_matches_the_oracle(oracle_mod):
    """mj_forward (integrate off): qacc = the warm-start slot afterwards, on random moving states around the standing pose, feet in
    the ground or in the air, joints inside their ranges: M, bias, Jacobians, the solver and the block solves all have to be right."""
    rng = np.random.default_rng(0)
    o = oracle_mod.Oracle3D()
    q0, _ = o.state()
    worst = 0.0
    env = Leg3dHostVec(1)
    for k in range(40):
        q = q0.copy()
        q[0:3] += rng.uniform(-0.02, 0.02, 3)
        q[2] += rng.uniform(-0.01, 0.03)
        quat = q[3:7] + rng.uniform(-0.05, 0.05, 4)
        q[3:7] = quat / np.linalg.norm(quat)
        q[7:] += rng.uniform(-0.05, 0.05, 14)
        v = rng.uniform(-0.5, 0.5, 20)
        u = rng.uniform(-1, 1, 10) * CTRL
        o.set_state_raw(q, v, np.zeros(20))
        o.step_torque(u)   # (one step so that ctrl is set; then restore the state and run forward)
        o.set_state_raw(q, v, np.zeros(20))
        o.forward()
        env.set_state_host(record(q, v)[None])
        s = env.state.copy(); s[0, 61:71] = u; env.set_state_host(s)
        env.step_host(None, 1, integrate=False)
        assert env.pending[0] == 0
        qa = o.qacc()
        worst = max(worst, np.abs(env.state[0, 41:61] - qa).max() / (1 + np.abs(qa).max()))
        assert env.nrows[0] == o.nefc, (env.nrows[0], o.nefc)
    assert worst < 1e-7, worst


def test_teacher_forced_1000_substeps(oracle_mod):
    rng = np.random.default_rng(1)
    env = Leg3dHostVec(2)
    o = oracle_mod.Oracle3D()
    worst, done, handed = 0.0, 0, 0
    for i in range(1000):
        if i % 10 == 0:
            a = rng.uniform(-1, 1, 10) * CTRL
        q, v = o.state()
        env.set_state_host(np.tile(record(q, v, o.warmstart()), (2, 1)))
        env.step_host(np.tile(a, (2, 1)), 1)
        o.step_torque(a)
        sg = env.get_state_host()
        assert np.array_equal(sg[0], sg[1])
        if env.pending[0]:
            handed += 1
            continue
        done += 1
        q1, v1 = o.state()
        worst = max(worst, err(sg[0], q1, v1))
    assert worst < 1e-9, worst
    assert done > 600, (done, handed)


def test_free_running_500_substeps(oracle_mod):
    rng = np.random.default_rng(2)
    n = 3
    env = Leg3dHostVec(n)
    os_ = [oracle_mod.Oracle3D() for _ in range(n)]
    env.set_state_host(np.array([record(*o.state(), o.warmstart()) for o in os_]))
    worst = 0.0
    for t in range(50):
        acts = rng.uniform(-1, 1, (n, 10)) * CTRL * 0.3
        env.step_host(acts, 10)
        assert (env.pending == 0).all(), (t, env.pending)
        for i, o in enumerate(os_):
            for _ in range(10):
                o.step_torque(acts[i])
        sg = env.get_state_host()
        for i, o in enumerate(os_):
            worst = max(worst, err(sg[i], *o.state()))
    assert worst < 1e-5, worst

"""Cassie3d physics on the HIP path (include/cassie3d_vec.h) against oracle/liboracle3d.so.  -m gpu only.
Tolerances: the kernel and the oracle evaluate the same equations with different factorisations (composite-inertia M and
Gauss-Jordan vs per-body sums and Cholesky; incremental vs recomputed PGS residuals), so agreement is to rounding, not bitwise."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
pytestmark = pytest.mark.gpu

D3 = dict(M=0, BIAS=400, QS=420, NEFC=440, QACC=441, F=461, AREF=525, J=589)
CTRL = np.array([4.5, 4.5, 12.2, 12.2, 0.9] * 2)


@pytest.fixture(scope="module")
def kat():
    with open(os.path.join(ROOT, "tests", "golden", "model3d_kat.json")) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def V3():
    from cassierl_amd import vec_env3d
    return vec_env3d


def random_state(rng, kat, spread=0.3, height=None):
    q = np.array(kat["qpos_init"])
    q[:2] += rng.uniform(-0.5, 0.5, 2)
    q[2] = height if height is not None else q[2] + rng.uniform(-0.05, 0.3)
    quat = np.array([1.0, 0, 0, 0]) + rng.uniform(-spread, spread, 4)
    q[3:7] = quat / np.linalg.norm(quat)
    q[7:] += rng.uniform(-spread, spread, 14)
    v = rng.uniform(-1.0, 1.0, 20)
    return q, v


@pytest.fixture(params=["leg", "wave"])
def tier(request, monkeypatch):
    """First Cassie3d kernel tier: "leg" = one lane per leg, 32 environments per wavefront (cassie3d_leg.hip, the default),
    "wave" = the r03 tiers only (one wavefront per environment; CASSIE3D_LEG=0).  Read by Cassie3dVecCreate."""
    monkeypatch.setenv("CASSIE3D_LEG", "1" if request.param == "leg" else "0")
    return request.param


def test_forward_dynamics_terms_match_oracle(V3, kat):
    """mj_forward piece by piece on random states (in the air and in contact): M, bias, smooth acceleration, constraint rows,
    reference accelerations, forces, qacc."""
    import oracle_py
    rng = np.random.default_rng(0)
    n = 6
    env = V3.Cassie3dVec(n)
    o = oracle_py.Oracle3D()
    states, torques = [], rng.uniform(-1, 1, (n, 10)) * CTRL
    for e in range(n):
        q, v = random_state(rng, kat, height=(None if e % 2 else 0.80 + 0.03 * e))
        if e == 0:
            q, v = np.array(kat["qpos_init"]), np.zeros(20)
        states.append((q, v))
    env.set_state_host(np.stack([V3.state_record(q, v) for q, v in states]))
    dbg = env.debug_forward_host(torques)
    saw_contact = False
    for e, (q, v) in enumerate(states):
        o.set_state_raw(q, v, np.zeros(20))
        _set_ctrl_forward(o, torques[e])
        d = dbg[e]
        M = d[D3["M"]:D3["M"] + 400].reshape(20, 20)
        np.testing.assert_allclose(M, o.mass_matrix(q), rtol=0, atol=1e-11)
        np.testing.assert_allclose(d[D3["BIAS"]:D3["BIAS"] + 20], o.bias(q, v), rtol=0, atol=1e-9)
        nefc = int(d[D3["NEFC"]])
        assert nefc == o.nefc, (e, nefc, o.nefc)
        J, f, pos, aref, typ = o.efc()
        saw_contact |= bool((typ == 2).any())
        Jg = d[D3["J"]:D3["J"] + 64 * 20].reshape(64, 20)[:nefc]
        np.testing.assert_allclose(Jg, J, rtol=0, atol=1e-11)
        np.testing.assert_allclose(d[D3["AREF"]:D3["AREF"] + nefc], aref, rtol=1e-9, atol=1e-7)
        scale = 1.0 + np.abs(f).max()
        np.testing.assert_allclose(d[D3["F"]:D3["F"] + nefc], f, rtol=0, atol=1e-7 * scale)
        a = o.qacc()
        np.testing.assert_allclose(d[D3["QACC"]:D3["QACC"] + 20], a, rtol=0, atol=1e-7 * (1.0 + np.abs(a).max()))
    assert saw_contact
    env.close()


def _set_ctrl_forward(o, u):
    """mj_forward with ctrl = u on the oracle (orc_step_torque would also integrate)"""
    import oracle_py
    o.L.orc_set_ctrl(o.h, oracle_py._p(oracle_py._vec(u, 10)))
    o.forward()


def test_teacher_forced_steps_match_oracle(V3, kat, tier):
    """1000 substeps of random torques, the kernel restarted from the oracle's state (incl. warm start) at every substep."""
    import oracle_py
    rng = np.random.default_rng(1)
    env = V3.Cassie3dVec(2)
    o = oracle_py.Oracle3D()
    worst_q, worst_v = 0.0, 0.0
    u = np.zeros(10)
    for i in range(1000):
        if i % 20 == 0:
            u = rng.uniform(-1, 1, 10) * CTRL
        q, v = o.state()
        ws = o.warmstart()
        env.set_state_host(np.tile(V3.state_record(q, v, ws), (2, 1)))
        env.step_host(np.tile(u, (2, 1)), 1)
        o.step_torque(u)
        s = env.get_state_host()
        assert np.array_equal(s[0], s[1])
        q1, v1 = o.state()
        worst_q = max(worst_q, np.abs(s[0, :21] - q1).max())
        worst_v = max(worst_v, np.abs(s[0, 21:41] - v1).max() / (1.0 + np.abs(v1).max()))
        assert s[0, 74] == 0.0
    assert worst_q < 1e-9 and worst_v < 1e-7, (worst_q, worst_v)
    env.close()


def test_free_running_1000_substeps_within_1e5(V3, kat, tier):
    """north_star tolerance: state trajectories within 1e-5 relative over 1000 steps on identical actions (free running)."""
    import oracle_py
    rng = np.random.default_rng(2)
    env = V3.Cassie3dVec(1)
    o = oracle_py.Oracle3D()
    q0, v0 = o.state()
    env.set_state_host(V3.state_record(q0, v0, o.warmstart())[None])
    worst = 0.0
    for blk in range(100):
        u = rng.uniform(-0.3, 0.3, 10) * CTRL
        env.step_host(u[None], 10)
        for _ in range(10):
            o.step_torque(u)
        s = env.get_state_host()[0]
        q1, v1 = o.state()
        worst = max(worst, np.abs(s[:21] - q1).max() / (1.0 + np.abs(q1).max()), np.abs(s[21:41] - v1).max() / (1.0 + np.abs(v1).max()))
    assert worst < 1e-5, worst
    assert abs(np.linalg.norm(s[3:7]) - 1.0) < 1e-12
    env.close()


def test_free_running_10000_substeps_drift_within_1e5(V3, kat, tier):
    """The drift run of tools/parity_drift.py as a test: 10 000 FREE-RUNNING torque substeps = 1000 Env.steps of
    cassie3d_stiff.xml, smooth random torques (a new draw every 200 substeps), HIP path against the oracle from the same state and
    the same torques, no teacher forcing; the north_star bar (1e-5 relative) at every 100th substep.  Measured: ~2e-9."""
    import oracle_py
    rng = np.random.default_rng(7)
    env = V3.Cassie3dVec(1)
    o = oracle_py.Oracle3D()
    q0, v0 = o.state()
    env.set_state_host(V3.state_record(q0, v0, o.warmstart())[None])
    worst, u = 0.0, np.zeros(10)
    for blk in range(1000):
        if blk % 20 == 0:
            u = rng.uniform(-0.25, 0.25, 10) * CTRL
        env.step_host(u[None], 10)
        for _ in range(10):
            o.step_torque(u)
        if (blk + 1) % 10 == 0:
            s = env.get_state_host()[0]
            q1, v1 = o.state()
            worst = max(worst, np.abs(s[:21] - q1).max() / (1.0 + np.abs(q1).max()), np.abs(s[21:41] - v1).max() / (1.0 + np.abs(v1).max()))
            assert worst < 1e-5, ((blk + 1) * 10, worst)
    print("cassie3d[%s] 10000 free-running substeps: worst relative deviation %.3e" % (tier, worst))
    env.close()


def test_reset_and_batch_independence(V3, kat, tier):
    """Every environment of a batch is independent and the default reset is the standing pose with a valid warm start."""
    import torch
    env = V3.Cassie3dVec(130)
    s = env.get_state_host()
    np.testing.assert_allclose(s[:, :21], np.tile(kat["qpos_init"], (130, 1)), atol=0)
    assert (s[:, 73] == 18).all() and (s[:, 74] == 0).all()   # 6 connect rows + 4 contacts x 3
    rng = np.random.default_rng(3)
    u = rng.uniform(-1, 1, (130, 10)) * CTRL
    u[65:] = u[:65]
    env.step(torch.as_tensor(u, device="cuda"), 10)
    env.synchronize()
    s = env.get_state_host()
    assert np.array_equal(s[:65], s[65:]) and np.isfinite(s).all()
    assert not np.array_equal(s[0], s[1])
    env.close()


def test_many_contacts_hand_over_to_the_general_kernel(V3, kat, tier):
    """A robot lying on the floor has more than 32 constraint rows: the high-occupancy kernel hands the environment over to the
    64-row kernel (pending list).  Same checks as above on such states, through both the debug (general kernel only) and the
    normal two-kernel path, including an Env.step that starts below 32 rows and crosses the limit part-way."""
    import oracle_py
    rng = np.random.default_rng(5)
    o = oracle_py.Oracle3D()
    q = np.array(kat["qpos_init"])
    q[2] = 0.12
    q[3:7] = [np.cos(np.pi / 4), np.sin(np.pi / 4), 0.0, 0.0]   # rolled 90 degrees: lying on its side, 9 contacts
    v = rng.uniform(-0.05, 0.05, 20)
    o.reset(q, v)
    assert o.nefc > 32, o.nefc
    env = V3.Cassie3dVec(3)
    worst = 0.0
    u = rng.uniform(-1, 1, 10) * CTRL
    for i in range(60):
        qo, vo = o.state()
        env.set_state_host(np.tile(V3.state_record(qo, vo, o.warmstart()), (3, 1)))
        env.step_host(np.tile(u, (3, 1)), 1)
        o.step_torque(u)
        s = env.get_state_host()
        q1, v1 = o.state()
        assert s[0, 74] == 0.0 and s[0, 73] == o.nefc
        worst = max(worst, np.abs(s[0, :21] - q1).max(), np.abs(s[0, 21:41] - v1).max() / (1.0 + np.abs(v1).max()))
    assert worst < 1e-7, worst
    # falling onto the floor inside one Env.step: rows grow past 32 between substeps
    q = np.array(kat["qpos_init"]); q[2] = 0.15
    q[3:7] = [np.cos(np.pi / 4), np.sin(np.pi / 4), 0.0, 0.0]
    v = np.zeros(20); v[2] = -1.0
    o.reset(q, v)
    n0 = o.nefc
    env.set_state_host(np.tile(V3.state_record(q, v, o.warmstart()), (3, 1)))
    env.step_host(np.zeros((3, 10)), 60)
    crossed = False
    for _ in range(60):
        o.step_torque(np.zeros(10))
        crossed |= o.nefc > 32
    assert n0 <= 32 and crossed, (n0, o.nefc)
    s = env.get_state_host()
    q1, v1 = o.state()
    assert abs(s[0, 71] - 60 * 0.0005) < 1e-12 and s[0, 74] == 0.0
    np.testing.assert_allclose(s[0, :21], q1, atol=1e-6)
    np.testing.assert_allclose(s[0, 21:41], v1, atol=1e-4 * (1 + np.abs(v1).max()))
    env.close()


def test_two_environments_per_wavefront_kernel_agrees_with_the_default(V3, kat, monkeypatch):
    """cassie3d_pair.hip (CASSIE3D_PAIR=1: two environments per wavefront on 32-lane halves, aligned row triples) against the default
    one-environment-per-wavefront kernel and the oracle: an odd batch (one half idle) of robots dropped from different heights and
    attitudes -- standing, falling, on the ground with limits active and > 30 rows (handed over from one half while the neighbour
    carries on) -- 400 substeps with random torques.  The two kernels evaluate the same arithmetic per environment (pad rows add
    exact zeros), so they stay together far below the oracle tolerance."""
    import oracle_py
    import torch
    rng = np.random.default_rng(11)
    n = 33
    recs, starts = [], []
    o = oracle_py.Oracle3D()
    for i in range(n):
        q, v = random_state(rng, kat, spread=0.25 if i % 3 else 0.05, height=None if i % 4 else 0.25)
        if i % 5 == 4:   # lying on its side: many contacts, handed over to the general kernel
            q[2] = 0.12; q[3:7] = [np.cos(np.pi / 4), np.sin(np.pi / 4), 0.0, 0.0]
        o.reset(q, v)
        recs.append(V3.state_record(q, v, o.warmstart())); starts.append((q, v))
    recs = np.array(recs)
    env_a = V3.Cassie3dVec(n)
    monkeypatch.setenv("CASSIE3D_PAIR", "1")
    env_b = V3.Cassie3dVec(n)
    monkeypatch.delenv("CASSIE3D_PAIR")
    env_a.set_state_host(recs); env_b.set_state_host(recs)
    us = rng.uniform(-0.5, 0.5, (40, n, 10)) * CTRL
    for t in range(40):
        u = torch.as_tensor(us[t], device="cuda")
        env_a.step(u, 10); env_b.step(u, 10)
    env_a.synchronize(); env_b.synchronize()
    sa, sb = env_a.get_state_host(), env_b.get_state_host()
    assert np.isfinite(sa).all() and np.isfinite(sb).all()
    assert np.array_equal(sa[:, 71], sb[:, 71])                       # env clocks: every substep was done exactly once
    np.testing.assert_allclose(sb[:, :41], sa[:, :41], atol=1e-8)
    assert env_b.counters()["general_kernel_substeps"] > 0            # some halves were handed over while their neighbours went on
    # and one environment of the batch against the oracle (teacher-free, 400 substeps)
    q, v = starts[1]
    o.reset(q, v)
    for t in range(40):
        for _ in range(10):
            o.step_torque(us[t, 1])
    q1, v1 = o.state()
    assert np.abs(sb[1, :21] - q1).max() / (1.0 + np.abs(q1).max()) < 1e-5
    env_a.close(); env_b.close()


def test_row_cap_beyond_64_rows_matches_the_capped_oracle(V3, kat, tier):
    """The model's worst case is 69 rows (6 connect + 12 limits + 17 contacts x 3); the 64-row kernel then leaves the last
    contacts out for that substep instead of freezing the environment.  A pressed-down, folded-up robot reaches
    that regime: the kernel must match the oracle run with the same cap, count the event, and keep stepping."""
    import oracle_py
    rng = np.random.default_rng(9)
    o = oracle_py.Oracle3D()
    o.set_row_cap(64)
    q = np.array(kat["qpos_init"])
    q[2] = -0.7                                             # pressed into the floor: every collision sphere is down
    q[3:7] = [np.cos(np.pi / 4), np.sin(np.pi / 4), 0.0, 0.0]
    # every limited hinge beyond its range (cassie3d_stiff.xml; dofs 6..11 and 13..18, qpos address = dof + 1)
    lim_dof = [6, 7, 8, 9, 10, 11, 13, 14, 15, 16, 17, 18]
    lim_rng = np.radians([[-15, 22.5], [-22.5, 22.5], [-50, 80], [-164, -37], [50, 170], [-140, -30]] * 2)
    for k, dof in enumerate(lim_dof):
        q[dof + 1] = lim_rng[k, 1] + 0.02 if k % 2 else lim_rng[k, 0] - 0.02
    v = rng.uniform(-0.02, 0.02, 20)
    o.reset(q, v)
    free = oracle_py.Oracle3D()
    free.reset(q, v)
    assert free.nefc == 69 and o.nefc == 63, (free.nefc, o.nefc)  # 6 + 12 + 3 * 17; the cap keeps 15 contacts
    env = V3.Cassie3dVec(2)
    env.reset_counters()
    u = np.zeros(10)
    worst, capped = 0.0, 0
    for i in range(12):
        qo, vo = o.state()
        env.set_state_host(np.tile(V3.state_record(qo, vo, o.warmstart()), (2, 1)))
        env.step_host(np.tile(u, (2, 1)), 1)
        o.step_torque(u)
        free.reset(*o.state())
        capped += free.nefc > 64
        s = env.get_state_host()
        q1, v1 = o.state()
        assert np.isfinite(s).all() and s[0, 73] == o.nefc
        worst = max(worst, np.abs(s[0, :21] - q1).max(), np.abs(s[0, 21:41] - v1).max() / (1.0 + np.abs(v1).max()))
    assert worst < 1e-7, worst
    c = env.counters()
    assert c["capped_substeps"] >= 10 and capped >= 4 and c["general_kernel_substeps"] == 24
    env.close()


def test_step_in_segments_is_the_step_in_one_launch_bit_for_bit(V3, kat, monkeypatch):
    """launch3d (cassie_cabi.hip) runs the lane-per-leg kernel over the substeps of a step in three segments and finishes the hand-overs
    of a segment on its own stream while the next segment runs (CASSIE3D_SEGMENTS=0: one launch, lower tiers behind it).  Every environment
    is stepped by the same kernels on the same data either way: 256 robots from standing, tilted and lying starts under random torques, 30
    steps of 10 substeps, states equal to the bit, clocks complete, and hand-overs did happen."""
    import oracle_py
    import torch
    rng = np.random.default_rng(21)
    n = 256
    o = oracle_py.Oracle3D()
    recs = []
    for i in range(n):
        q, v = random_state(rng, kat, spread=0.25 if i % 3 else 0.05, height=None if i % 4 else 0.25)
        if i % 7 == 6:
            q[2] = 0.12; q[3:7] = [np.cos(np.pi / 4), np.sin(np.pi / 4), 0.0, 0.0]
        o.reset(q, v)
        recs.append(V3.state_record(q, v, o.warmstart()))
    recs = np.array(recs)
    us = rng.uniform(-1.0, 1.0, (30, n, 10)) * CTRL
    finals, counts = [], []
    for seg in ("1", "0"):
        monkeypatch.setenv("CASSIE3D_SEGMENTS", seg)
        env = V3.Cassie3dVec(n)
        env.set_state_host(recs)
        for t in range(30):
            env.step(torch.as_tensor(us[t], device="cuda"), 10)
        env.synchronize()
        finals.append(env.get_state_host()); counts.append(env.counters())
        env.close()
    monkeypatch.delenv("CASSIE3D_SEGMENTS")
    assert np.isfinite(finals[0]).all()
    assert np.allclose(finals[0][:, 71], 300 * 0.0005, atol=1e-12)     # every substep of every environment was done exactly once
    assert counts[0]["leg_handover_substeps"] > 0 and counts[0]["leg_handover_substeps"] == counts[1]["leg_handover_substeps"]
    assert np.array_equal(finals[0], finals[1])

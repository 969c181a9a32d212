"""The planar formulation the HIP kernels implement (tools/planar_proto.py) against the 3-D oracle:
validates the sagittal-plane reduction, the 2-row contacts/connects, the fixed slot order and the
incremental-residual PGS on CPU, step by step (teacher-forced so that chaos cannot mask a bug)."""
import numpy as np
import pytest

from planar_proto import Planar


@pytest.mark.parametrize("mode", ["torque", "pd"])
def test_step_matches_oracle(oracle_mod, mode):
    P = Planar()
    o = oracle_mod.Oracle()
    rng = np.random.default_rng(11)
    lo, hi = np.radians([-50, -164, -140] * 2), np.radians([80, -37, -30] * 2)
    worst, maxrows = 0.0, 0
    for i in range(400):
        q, v = o.state(); ws = o.warmstart()
        if i % 10 == 0:
            a = rng.uniform(-1, 1, 6) * np.array([12, 12, .9] * 2) if mode == "torque" else rng.uniform(lo, hi)
        ctrl = a if mode == "torque" else P.pd_ctrl(q, v, a)
        q2, v2, qacc, r = P.step(q, v, ws, ctrl)
        (o.step_torque if mode == "torque" else o.step_pd)(a)
        qo, vo = o.state()
        worst = max(worst, np.abs(q2 - qo).max(), np.abs(v2 - vo).max() / (1 + np.abs(vo).max()))
        assert r["niter"] == o.solver_niter
        maxrows = max(maxrows, int(r["active"].sum()))
    assert worst < 1e-11, worst
    assert maxrows >= 10


def test_mass_and_bias_random_states(oracle_mod):
    P = Planar()
    o = oracle_mod.Oracle()
    rng = np.random.default_rng(12)
    q0, _ = o.state()
    for _ in range(20):
        q = q0 + rng.uniform(-0.6, 0.6, 13); v = rng.uniform(-3, 3, 13)
        M, b = P.mass_bias(P.fk(q, v))
        np.testing.assert_allclose(M, o.mass_matrix(q), atol=1e-13)
        np.testing.assert_allclose(b, o.bias(q, v), atol=1e-11)


def test_opstate_matches_oracle(oracle_mod):
    P = Planar(sem="rbdl")
    o = oracle_mod.Oracle()
    for i in range(30):
        o.step_pd(np.radians([20, -90, -80] * 2))
    q, v = o.state()
    # the oracle's kin state is the pre-step state of the last call: replay to capture it
    o2 = oracle_mod.Oracle()
    for i in range(29):
        o2.step_pd(np.radians([20, -90, -80] * 2))
    kq, kv = o2.state()
    np.testing.assert_allclose(P.opstate(kq, kv, q, v), o.opstate(0), atol=1e-13)
    np.testing.assert_allclose(P.opstate(q, v, q, v), o.opstate(1), atol=1e-13)

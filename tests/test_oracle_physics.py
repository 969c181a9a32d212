"""Physics invariants of the CPU oracle.  The reference has no tests and its physics (MuJoCo) is absent,
so besides the model KATs the oracle is pinned by properties any correct restatement must have."""
import numpy as np


def _free(o, q, v):
    o.set_contact_enabled(False)
    o.set_state_raw(q, v, np.zeros(13))


def test_energy_conservation_without_dissipation(oracle_mod):
    # no contacts, no damping, no actuation: only the (soft) loop closure acts.  Semi-implicit Euler at
    # h = 0.5 ms must keep the total energy within a tight band -> validates M, bias and the integrator together.
    o = oracle_mod.Oracle()
    q, v = o.state()
    o.set_damping_scale(0.0)
    rng = np.random.default_rng(3)
    v0 = rng.uniform(-0.5, 0.5, 13)
    v0[7] = v0[12] = 0.0
    _free(o, q, v0)
    e0 = o.energy()[0]
    es = []
    for _ in range(400):
        o.step_torque(np.zeros(6))
        es.append(o.energy()[0])
    # the connect constraint is a damped soft constraint: it may only REMOVE a little energy
    assert max(es) - e0 < 2e-2 and e0 - min(es) < 0.5, (e0, min(es), max(es))


def test_bias_is_the_lagrangian_of_the_mass_matrix(oracle_mod):
    """Coriolis/centrifugal + gravity forces must satisfy  bias = Mdot v - d/dq (1/2 v'Mv) + dV/dq  exactly
    (Lagrange).  Central differences of the oracle's own M(q) and potential pin the RNE restatement for all 13 dofs."""
    o = oracle_mod.Oracle()
    rng = np.random.default_rng(4)
    q0, _ = o.state()
    eps = 1e-6
    for _ in range(5):
        q = q0 + rng.uniform(-0.4, 0.4, 13); v = rng.uniform(-2, 2, 13)
        Mdot_v = (o.mass_matrix(q + eps * v) - o.mass_matrix(q - eps * v)) @ v / (2 * eps)
        dT, dV = np.zeros(13), np.zeros(13)
        for k in range(13):
            e = np.zeros(13); e[k] = eps
            dT[k] = 0.5 * v @ (o.mass_matrix(q + e) - o.mass_matrix(q - e)) @ v / (2 * eps)
            o.set_state_raw(q + e, np.zeros(13)); vp = o.energy()[2]
            o.set_state_raw(q - e, np.zeros(13)); vm = o.energy()[2]
            dV[k] = (vp - vm) / (2 * eps)
        np.testing.assert_allclose(o.bias(q, v), Mdot_v - dT + dV, atol=2e-6)


def test_momentum_drift_zero_gravity(oracle_mod):
    # internal torques and the loop closure cannot change the linear momentum; a first-order integrator keeps it to O(h)
    o = oracle_mod.Oracle()
    q, _ = o.state()
    o.set_gravity(0.0)
    rng = np.random.default_rng(4)
    v0 = rng.uniform(-1, 1, 13)
    _free(o, q, v0)
    def momentum():
        qq, vv = o.state()
        return (o.mass_matrix(qq) @ vv)[:2]
    p0 = momentum()
    for _ in range(300):
        o.step_torque(np.array([3.0, -2.0, 0.5, -1.0, 2.0, -0.3]))
    assert np.abs(momentum() - p0).max() < 5e-3 * np.abs(p0).max()


def test_free_fall_acceleration(oracle_mod):
    o = oracle_mod.Oracle()
    q, v = o.state()
    q = q.copy(); q[1] += 1.0
    _free(o, q, np.zeros(13))
    o.set_damping_scale(0.0)
    o.forward()
    a = o.qacc()
    # the centre of mass must fall at g whatever the internal motion: (M qacc)[z] = -m g
    M = o.mass_matrix(q)
    assert abs((M @ a)[1] + 32.822 * 9.806) < 1e-9 and abs((M @ a)[0]) < 1e-9


def test_left_right_symmetry(oracle_mod):
    o = oracle_mod.Oracle()
    rng = np.random.default_rng(5)
    q, _ = o.state()
    q = q + rng.uniform(-0.2, 0.2, 13)
    v = rng.uniform(-1, 1, 13)
    u = rng.uniform(-5, 5, 6)
    swap = np.r_[0:3, 8:13, 3:8]
    uswap = np.r_[3:6, 0:3]
    o.set_state_raw(q, v, np.zeros(13)); o.step_torque(u); a = np.concatenate(o.state())
    o2 = oracle_mod.Oracle()
    o2.set_state_raw(q[swap], v[swap], np.zeros(13)); o2.step_torque(u[uswap]); b = np.concatenate(o2.state())
    # rows are swept left-before-right in both runs, so PGS order differs between the two: agreement is to solver accuracy
    np.testing.assert_allclose(a[np.r_[swap, 13 + swap]], b, atol=5e-4)


def test_rest_contact_force_balances_weight(oracle_mod):
    o = oracle_mod.Oracle()
    for _ in range(6000):  # collapse under zero torque and settle on the ground
        o.step_torque(np.zeros(6))
    q, v = o.state()
    assert np.abs(v).max() < 5e-2
    e = o.efc()
    fz = e["force"][e["type"] == 2][::3].sum()
    assert abs(fz - 32.822 * 9.806) / (32.822 * 9.806) < 2e-2
    assert (e["force"][e["type"] == 2][::3] >= 0).all()
    assert np.abs(e["pos"][:6]).max() < 3e-3  # loop closure residual stays bounded


def test_joint_limit_holds_against_torque(oracle_mod):
    # robot in the air, full knee torque toward the upper limit (-37 deg): the soft limit must stop the joint
    o = oracle_mod.Oracle()
    q, v = o.state()
    q = q.copy(); q[1] += 1.0
    o.set_contact_enabled(False)
    o.set_state_raw(q, np.zeros(13), np.zeros(13))
    hi = np.radians(-37.0)
    seen = False
    for _ in range(1500):
        o.step_torque(np.array([0, 12.2, 0, 0, 12.2, 0]))
        e = o.efc()
        lim = e["type"] == 1
        if lim.any():
            seen = True
            assert (e["force"][lim] >= 0).all()
            # complementarity of the unilateral row: pushing, or already separating at least as fast as the reference acceleration
            jar = e["J"][lim] @ o.qacc() - e["aref"][lim]
            assert ((e["force"][lim] > 0) | (jar > -1e-6)).all()
    assert seen
    qf, vf = o.state()
    # soft limit (solref 0.02): the joint comes to rest a little beyond the limit, the row carries the motor torque
    assert hi < qf[4] < hi + 0.03 and abs(vf[4]) < 1e-3
    e = o.efc()
    knee_rows = [i for i in range(o.nefc) if e["type"][i] == 1 and e["J"][i, 4] == -1.0]
    assert len(knee_rows) == 1 and 150.0 < e["force"][knee_rows[0]] < 12.2 * 16


def test_sensitivity_documented(oracle_mod):
    """Torque mode is contractive enough for 1000-step free-running parity; the reference's PD law is chaotic
    (toe PD gain x gear 100 is unstable at h = 0.5 ms), which is why PD parity is checked teacher-forced."""
    def run(eps, mode):
        o = oracle_mod.Oracle()
        q, v = o.state(); q = q.copy(); q[4] += eps
        o.set_state_raw(q, v, np.zeros(13))
        rng = np.random.default_rng(0)
        out = []
        for i in range(600):
            if i % 10 == 0:
                a = rng.uniform(-1, 1, 6)
            if mode == "torque":
                o.step_torque(a * np.array([12, 12, .9] * 2))
            else:
                o.step_pd(np.radians([15, -100, -85] * 2) + 0.3 * a)
            out.append(np.concatenate(o.state()))
        return np.array(out)
    d_t = np.abs(run(0, "torque") - run(1e-12, "torque")).max()
    d_p = np.abs(run(0, "pd") - run(1e-12, "pd")).max()
    assert d_t < 1e-8
    assert d_p > 1e-6  # documents the chaos; if this ever fails the PD parity strategy can be tightened

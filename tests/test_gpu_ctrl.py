"""In-loop controllers on the HIP path (StepOsc / StepJacobian / standing controllers) against the oracle.  -m gpu only.
The oracle solves the reference's literal 39-variable OSC QP; the kernel solves the reduced 14-variable box QP, so these
tests also check that reduction on the device."""
import ctypes as ct

import numpy as np
import pytest

from conftest import state_vec

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def vec():
    from cassierl_amd.vec_env import CassieVecEnv
    return CassieVecEnv


def osc_action(rng, scale=1.0):
    a = rng.uniform(-1, 1, 7) * np.array([3, 3, 1, 1, 1, 1, 3.0]) * scale
    a[3], a[5] = abs(a[3]), abs(a[5])
    return a


def jac_action(rng):
    return np.array([rng.uniform(-50, 50), 150 + rng.uniform(-60, 60), rng.uniform(-20, 20)] * 2) + rng.uniform(-5, 5, 6)


@pytest.mark.parametrize("mode", ["OSC", "Jacobian"])
def test_teacher_forced_controller_substeps(vec_tier, oracle_mod, mode):
    vec = vec_tier
    rng = np.random.default_rng(21)
    n = 2
    env = vec(n, kind="stand", control_mode="Torque", n_substeps=1, auto_reset=False)
    o = oracle_mod.Oracle()
    worst_state, worst_u = 0.0, 0.0
    for i in range(400):
        scale = 1.0 if i < 250 else 8.0
        a = osc_action(rng, scale) if mode == "OSC" else jac_action(rng)
        qo, vo = o.state()
        env.set_full_state_host(np.tile(state_vec(qo, vo, o.warmstart()), (n, 1)))
        env.substep_host(mode, np.tile(a, (n, 1)), 1)
        (o.step_osc if mode == "OSC" else o.step_jacobian)(a)
        sg = env.get_full_state_host()
        assert np.array_equal(sg[0], sg[1])
        q1, v1 = o.state()
        worst_state = max(worst_state, np.abs(sg[0, :13] - q1).max(), np.abs(sg[0, 13:26] - v1).max() / (1 + np.abs(v1).max()))
        worst_u = max(worst_u, np.abs(sg[0, 78:84] - o.ctrl()).max())
    assert worst_u < 1e-6, worst_u
    assert worst_state < 1e-8, worst_state
    env.close()


def test_env_step_osc_golden_stream(vec_tier, streams, traj):
    vec = vec_tier
    n = 2
    env = vec(n, kind="stand", control_mode="OSC", n_substeps=10, auto_reset=True)
    obs0 = env.reset_host()
    np.testing.assert_allclose(obs0, np.tile(streams["stand_osc_obs0"], (n, 1)), atol=1e-12)
    acts = streams["stand_osc_actions"]
    for t in range(40):
        obs, rew, done = env.step_host(np.tile(acts[t], (n, 1)))
        assert (done == bool(streams["stand_osc_done"][t])).all()
        np.testing.assert_allclose(rew, streams["stand_osc_reward"][t], rtol=0, atol=1e-6)
        np.testing.assert_allclose(obs, np.tile(streams["stand_osc_obs"][t], (n, 1)), rtol=0, atol=1e-5)
    env.close()


@pytest.mark.parametrize("wave_per_env", [False, True, "leg"])
def test_walk_env_with_osc_control_golden_stream(vec, streams, traj, wave_per_env):
    """cassie2d.py with control_mode = 'OSC' (the walk env accepts it, cassie2d.py:53,104-105): reference-gait reward, r < 0.6
    termination, gait joints in obs[17:26] -- recorded from the reference's own class (tests/golden/make_env_streams.py).
    r01 computed the stand reward here."""
    from cassierl_amd.vec_env import LEG_TIER_ON, WAVE_PER_ENV
    n = 3
    flags = LEG_TIER_ON if wave_per_env == "leg" else (WAVE_PER_ENV if wave_per_env else 0)   # "leg": mj_step in env_step_leg_kernel<2>
    env = vec(n, kind="walk", control_mode="OSC", n_substeps=10, auto_reset=True, flags=flags)
    env.set_trajectory(traj["time"], traj["qpos"])
    obs0 = env.reset_host()
    np.testing.assert_allclose(obs0, np.tile(streams["walk_osc_obs0"], (n, 1)), atol=1e-12)
    acts = streams["walk_osc_actions"]
    for t in range(40):
        obs, rew, done = env.step_host(np.tile(acts[t], (n, 1)))
        assert (done == bool(streams["walk_osc_done"][t])).all()
        np.testing.assert_allclose(rew, streams["walk_osc_reward"][t], rtol=0, atol=1e-6)
        want = streams["walk_osc_reset_obs"][t] if streams["walk_osc_done"][t] else streams["walk_osc_obs"][t]
        np.testing.assert_allclose(obs, np.tile(want, (n, 1)), rtol=0, atol=1e-5)
    assert streams["walk_osc_done"].all() and abs(streams["walk_osc_reward"][0] - 0.39754008) < 1e-7  # quirk Q3 in this mode too
    env.close()


def _py_standing_osc(o, zpos, zvel, flags=0):
    s = o.opstate(flags)
    act = np.zeros(7)
    act[2] = 0.0; act[3] = 100.0 * (-5e-3 - s[7])
    act[4] = 0.0; act[5] = 100.0 * (-5e-3 - s[13])
    xt = (s[6] + s[12]) / 2.0
    act[0] = 100.0 * (xt - s[0]) + 20.0 * (0.0 - s[3])
    act[1] = 100.0 * (zpos - s[1]) + 20.0 * (zvel - s[4])
    act[6] = 20.0 * (0.0 - s[2]) + 10.0 * (0.0 - s[5])
    o.step_osc(act)


def _py_standing_jac(o, zpos, zvel):
    s = o.opstate(0)
    xt = (s[6] + s[12]) / 2.0
    fx = 200.0 * (xt - s[0]) + 50.0 * (0.0 - s[3])
    fz = 0.5 * 9.806 * 31.0 + 200.0 * (zpos - s[1]) + 50.0 * (zvel - s[4])
    my = 100.0 * (0.0 - s[2]) + 10.0 * (0.0 - s[5])
    fz = max(fz, 0.0)
    o.step_jacobian(np.array([fx, fz, my, fx, fz, my]))


def test_standing_controller_osc_free_running(vec_tier, oracle_mod):
    """config 3 building block: standing_controller_osc(0.9, 0) in the loop, closed loop is stable -> free-running parity."""
    vec = vec_tier
    n = 3
    env = vec(n, kind="stand", control_mode="OSC", n_substeps=1, auto_reset=False)
    env.reset_host()
    o = oracle_mod.Oracle()
    q0 = np.array([0.0, 0.939, 0.0, 0.68111815, -1.40730357, 1.62972042, -1.77611107, -0.61968407] + [0.68111815, -1.40730357, 1.62972042, -1.77611107, -0.61968407])
    o.reset(q0, np.zeros(13))
    worst = 0.0
    for blk in range(60):
        env.standing_step_host("OSC", 0.9, 0.0, 10)
        for _ in range(10):
            _py_standing_osc(o, 0.9, 0.0)
        sg = env.get_full_state_host()
        q1, v1 = o.state()
        worst = max(worst, np.abs(sg[0, :13] - q1).max(), np.abs(sg[0, 13:26] - v1).max() / (1 + np.abs(v1).max()))
    assert worst < 1e-5, worst
    assert 0.8 < sg[0, 1] < 1.0
    env.close()


def test_squatting_jacobian_controller(vec_tier, oracle_mod):
    """config 1 (squatting.py): z target 0.7 + 0.25 sin(w t), w = 0.5*3.1415, via standing_controller_jacobian."""
    vec = vec_tier
    env = vec(1, kind="stand", control_mode="OSC", n_substeps=1, auto_reset=False)
    env.reset_host()
    o = oracle_mod.Oracle()
    q0 = np.array([0.0, 0.939, 0.0, 0.68111815, -1.40730357, 1.62972042, -1.77611107, -0.61968407] + [0.68111815, -1.40730357, 1.62972042, -1.77611107, -0.61968407])
    o.reset(q0, np.zeros(13))
    w, t = 0.5 * 3.1415, 0.0
    worst, zs = 0.0, []
    for i in range(1200):
        zt, zv = 0.7 + 0.25 * np.sin(w * t), 0.25 * np.cos(w * t)
        env.standing_step_host("Jacobian", zt, zv, 1)
        _py_standing_jac(o, zt, zv)
        t += 0.0005
        if i % 50 == 49:
            sg = env.get_full_state_host()[0]
            q1, v1 = o.state()
            worst = max(worst, np.abs(sg[:13] - q1).max(), np.abs(sg[13:26] - v1).max() / (1 + np.abs(v1).max()))
            zs.append(sg[1])
    assert worst < 1e-5, worst
    assert 0.4 < min(zs) and max(zs) < 1.05  # stays up (README acceptance: "Cassie squats")
    env.close()


def test_legacy_abi_osc_and_jacobian(oracle_mod):
    from cassierl_amd import _lib
    from cassierl_amd import structs as S
    L = _lib.load()
    L.StepOsc.argtypes = [ct.c_void_p, ct.POINTER(S.ControllerOsc)]
    L.StepJacobian.argtypes = [ct.c_void_p, ct.POINTER(S.ControllerForce)]
    L.GetGeneralState.argtypes = [ct.c_void_p, ct.POINTER(S.StateGeneral)]
    h = L.Cassie2dInit()
    o = oracle_mod.Oracle()
    cv = S.InterfaceStructConverter()
    rng = np.random.default_rng(8)
    for t in range(20):
        a = osc_action(rng)
        L.StepOsc(h, ct.byref(cv.array_to_operational_action(a)))
        o.step_osc(a)
    for t in range(20):
        f = jac_action(rng)
        fs = S.ControllerForce(); fs.left_force[:] = list(f[:3]); fs.right_force[:] = list(f[3:])
        L.StepJacobian(h, ct.byref(fs))
        o.step_jacobian(f)
    qs = S.StateGeneral()
    L.GetGeneralState(h, ct.byref(qs))
    qg, vg = S.general_array_to_qpos_qvel(cv.general_state_to_array(qs))
    q1, v1 = o.state()
    np.testing.assert_allclose(qg, q1, atol=1e-7)
    np.testing.assert_allclose(vg, v1, atol=1e-5)


@pytest.mark.parametrize("mode", ["OSC", "Jacobian"])
def test_g16_and_wave_per_env_controller_kernels_agree(vec, mode):
    """Env.step with the controller in the loop: the 4-envs-per-wave kernel (+ clean-up pass for environments that need more
    than 16 constraint rows) against the wave-per-environment kernel, teacher-forced, including falls and auto-resets."""
    from cassierl_amd.vec_env import WAVE_PER_ENV
    n = 41
    rng = np.random.default_rng(5)
    a = vec(n, kind="stand", control_mode=mode, n_substeps=10, auto_reset=True)
    b = vec(n, kind="stand", control_mode=mode, n_substeps=10, auto_reset=True, flags=WAVE_PER_ENV)
    # (the packed reset runs the two-lanes-per-environment core, the other handle the wave-per-environment reset kernel: two formulations of the
    # same mj_forward, equal to rounding -- bitwise only by accident of how the compiler fused their multiply-adds)
    np.testing.assert_allclose(a.reset_host(), b.reset_host(), rtol=0, atol=1e-13)
    ndone = 0
    for t in range(160):
        if mode == "OSC":
            acts = np.stack([osc_action(rng) for e in range(n)])
            if t >= 30:
                acts[::3, 0], acts[::3, 1], acts[::3, 6] = 15.0, -20.0, 10.0  # drive every third robot into the ground
        else:
            acts = np.stack([jac_action(rng) * (1.0 if (t < 30 or e % 3) else 0.2) for e in range(n)])
        b.set_full_state_host(a.get_full_state_host())
        oa, ra, da = a.step_host(acts)
        ob, rb, db = b.step_host(acts)
        assert (da == db).all()
        ndone += int(da.sum())
        np.testing.assert_allclose(ra, rb, rtol=0, atol=1e-9)
        np.testing.assert_allclose(oa, ob, rtol=0, atol=1e-7)
        sa, sb = a.get_full_state_host(), b.get_full_state_host()
        np.testing.assert_allclose(sa[:, :26], sb[:, :26], rtol=0, atol=1e-7)
        np.testing.assert_allclose(sa[:, 39:65], sb[:, 39:65], rtol=0, atol=1e-7)  # kinematics of the last setState (Q1/Q2)
        np.testing.assert_allclose(sa[:, 78:84], sb[:, 78:84], rtol=0, atol=1e-5)  # mj_data->ctrl
        np.testing.assert_allclose(sa[:, 84], sb[:, 84], atol=1e-12)               # env time
    assert ndone > 0  # some environments fell: the overflow / clean-up / auto-reset branches ran
    a.close(); b.close()


@pytest.mark.parametrize("mode", ["OSC", "Jacobian"])
def test_g16_scripted_standing_controller_agrees(vec, mode):
    """standing_controller_* in the loop on perturbed initial poses (config 3's workload), both kernels."""
    from cassierl_amd.vec_env import WAVE_PER_ENV
    n = 23
    rng = np.random.default_rng(9)
    a = vec(n, kind="stand", control_mode="OSC", n_substeps=1, auto_reset=False)
    b = vec(n, kind="stand", control_mode="OSC", n_substeps=1, auto_reset=False, flags=WAVE_PER_ENV)
    a.reset_host(); b.reset_host()
    s = a.get_full_state_host()
    s[:, :13] += rng.uniform(-0.01, 0.01, (n, 13))
    a.set_full_state_host(s)
    zp = rng.uniform(0.7, 0.95, n)
    for blk in range(40):
        b.set_full_state_host(a.get_full_state_host())
        a.standing_step_host(mode, zp, 0.0, 10)
        b.standing_step_host(mode, zp, 0.0, 10)
        sa, sb = a.get_full_state_host(), b.get_full_state_host()
        np.testing.assert_allclose(sa[:, :26], sb[:, :26], rtol=0, atol=1e-7)
        np.testing.assert_allclose(sa[:, 39:65], sb[:, 39:65], rtol=0, atol=1e-7)
        np.testing.assert_allclose(sa[:, 84], sb[:, 84], atol=1e-12)
    a.close(); b.close()


def test_osc_qp_hot_start_does_not_change_the_solution(vec):
    """The OSC kernels keep the QP working set of the previous call in the state record (slot 86), like qpOASES' hotstart in
    the reference (OSC_RBDL.cpp:276-280).  The QP is strictly convex: cold start, the carried working set and an arbitrary
    (wrong) working set must all give the same torques."""
    rng = np.random.default_rng(11)
    n = 3
    env = vec(n, kind="stand", control_mode="OSC", n_substeps=1, auto_reset=False)
    env.reset_host()
    for blk in range(5):
        env.standing_step_host("OSC", 0.88, 0.0, 10)
    s = env.get_full_state_host()
    assert s[0, 86] != 0.0                      # a working set was recorded
    for t in range(20):
        s = env.get_full_state_host()
        s[1] = s[0]; s[2] = s[0]
        s[1, 86] = 0.0                          # cold start
        s[2, 86] = float(0x155 | (0x3FFF << 14))  # some other working set: motors 0,2,4 and generators 6,8 on their lower bound
        env.set_full_state_host(s)
        a = np.tile(osc_action(rng, 2.0), (n, 1))
        env.substep_host("OSC", a, 1)
        r = env.get_full_state_host()
        np.testing.assert_allclose(r[1, 78:84], r[0, 78:84], rtol=0, atol=1e-7)
        np.testing.assert_allclose(r[2, 78:84], r[0, 78:84], rtol=0, atol=1e-7)
        np.testing.assert_allclose(r[1, :26], r[0, :26], rtol=0, atol=1e-9)
        np.testing.assert_allclose(r[2, :26], r[0, :26], rtol=0, atol=1e-9)
    env.close()


@pytest.mark.parametrize("mode", ["OSC", "Jacobian"])
def test_pseudoinverse_shortcuts_agree_with_the_literal_svd_route(vec, mode):
    """pseudoinverse(Jeq Hinv Jeq', 1e-3) and pseudoinverse(Nc Bt, 1e-4) (Cassie2d.cpp:134-171, OSC_RBDL.cpp:171) are evaluated
    as a certified inverse / normal-equations solve when no singular value can be under the threshold; flag 8 forces the
    eigen-decomposition / one-sided Jacobi SVD that applies the threshold literally.  Both must give the same torques."""
    from cassierl_amd.vec_env import NO_PINV_SHORTCUT
    n = 9
    rng = np.random.default_rng(13)
    a = vec(n, kind="stand", control_mode=mode, n_substeps=1, auto_reset=False)
    b = vec(n, kind="stand", control_mode=mode, n_substeps=1, auto_reset=False, flags=NO_PINV_SHORTCUT)
    a.reset_host(); b.reset_host()
    s = a.get_full_state_host()
    s[:, :13] += rng.uniform(-0.05, 0.05, (n, 13))
    a.set_full_state_host(s)
    for t in range(120):
        acts = np.stack([osc_action(rng, 2.0) if mode == "OSC" else jac_action(rng) for _ in range(n)])
        b.set_full_state_host(a.get_full_state_host())
        a.substep_host(mode, acts, 1)
        b.substep_host(mode, acts, 1)
        sa, sb = a.get_full_state_host(), b.get_full_state_host()
        np.testing.assert_allclose(sa[:, 78:84], sb[:, 78:84], rtol=0, atol=1e-8)   # torques
        np.testing.assert_allclose(sa[:, :13], sb[:, :13], rtol=0, atol=1e-11)
        np.testing.assert_allclose(sa[:, 13:26], sb[:, 13:26], rtol=0, atol=1e-8)  # velocities: h * M^-1 * (rounding-level torque difference)
    a.close(); b.close()


@pytest.mark.parametrize("wave_per_env", [False, True])
def test_standing_controller_osc_with_current_kinematics_flag(vec, oracle_mod, wave_per_env):
    """CASSIE_FIX_STALE_KIN: the scripted controller reads the operational-space state of the CURRENT state instead of the
    one of the last setState (quirk Q1/Q2 switched off), on both kernel generations."""
    from cassierl_amd.vec_env import FIX_STALE_KIN, WAVE_PER_ENV
    env = vec(2, kind="stand", control_mode="OSC", n_substeps=1, auto_reset=False, flags=FIX_STALE_KIN | (WAVE_PER_ENV if wave_per_env else 0))
    env.reset_host()
    o = oracle_mod.Oracle()
    q0 = np.array([0.0, 0.939, 0.0, 0.68111815, -1.40730357, 1.62972042, -1.77611107, -0.61968407] + [0.68111815, -1.40730357, 1.62972042, -1.77611107, -0.61968407])
    o.reset(q0, np.zeros(13))
    worst = 0.0
    for blk in range(30):
        env.standing_step_host("OSC", 0.85, 0.0, 10)
        for _ in range(10):
            _py_standing_osc(o, 0.85, 0.0, flags=1)
        sg = env.get_full_state_host()
        q1, v1 = o.state()
        worst = max(worst, np.abs(sg[0, :13] - q1).max(), np.abs(sg[0, 13:26] - v1).max() / (1 + np.abs(v1).max()))
    assert worst < 1e-5, worst
    env.close()


def test_dynamic_state_terms_match_the_oracle(vec, oracle_mod):
    """SURVEY.md row R5, directly: DynamicState::UpdateDynamicState (src/DynamicState.cpp:45-91) -- M (CRBA + rotor inertia),
    bias (NonlinearEffects + damping*qvel), the contact-site Jacobians Jc, the loop-closure Jacobian Jeq and JeqdotQdot -- as
    the controller kernel builds them (RBDL-semantics tables), against oracle/cassie_oracle.c:update_dynamic_state on random
    moving states.  The kernel keeps Hinv = M^-1 (never M), so M is compared through the inverse."""
    rng = np.random.default_rng(77)
    n = 6
    env = vec(n, kind="stand", control_mode="Torque", n_substeps=1, auto_reset=False)
    o = oracle_mod.Oracle()
    q0, _ = o.state()
    states, refs = [], []
    for i in range(n):
        q = q0 + rng.uniform(-0.15, 0.15, 13) * (i > 0)
        v = rng.uniform(-1.5, 1.5, 13) * (i > 0)
        o.set_state_raw(q, v)  # state + DynamicModel::setState (Cassie2d::Reset alone would leave the kinematics stale, quirk Q2)
        refs.append(o.dynamic_state())
        states.append(state_vec(q, v))
    env.set_full_state_host(np.array(states))
    dbg = env.debug_substep_host("Jacobian", np.zeros((n, 6)))
    for i, ref in enumerate(refs):
        d = dbg[i]
        Hinv = d[325:325 + 169].reshape(13, 13)
        M = np.linalg.inv(Hinv)
        assert np.abs(M - ref["M"]).max() < 1e-9 * np.abs(ref["M"]).max(), i
        assert np.abs(Hinv - np.linalg.inv(ref["M"])).max() < 1e-9 * np.abs(Hinv).max()
        assert np.abs(d[97:110] - ref["bias"]).max() < 1e-9 * (1 + np.abs(ref["bias"]).max()), i
        Jd = d[130:130 + 195].reshape(15, 13)
        acc = d[494:509]
        # loop closures: oracle rows (x, y, z) per connect, kernel rows (x, z); the world-y rows are identically zero
        assert np.abs(ref["Jeq"][[1, 4]]).max() < 1e-14 and np.abs(ref["JeqdotQdot"][[1, 4]]).max() < 1e-12
        assert np.abs(Jd[0:4] - ref["Jeq"][[0, 2, 3, 5]]).max() < 1e-11, i
        assert np.abs(acc[0:4] - ref["JeqdotQdot"][[0, 2, 3, 5]]).max() < 1e-9 * (1 + np.abs(ref["JeqdotQdot"]).max()), i
        # contact sites 2..5: oracle Jc rows 3 s + (0, 1, 2); kernel controller rows 6.. (x, z per site)
        Jc = ref["Jc"].reshape(4, 3, 13)
        assert np.abs(Jc[:, 1]).max() < 1e-14
        assert np.abs(Jd[6:14].reshape(4, 2, 13) - Jc[:, [0, 2]]).max() < 1e-11, i
        # selector matrix: gear on the actuated dofs (DynamicModel.cpp:197-216)
        Bt = ref["Bt"]
        assert np.count_nonzero(Bt) == 6 and [Bt[j, k] for k, j in enumerate((3, 4, 6, 8, 9, 11))] == [16, 16, 100, 16, 16, 100]
    env.close()

"""Parity of the HIP path (through the C-ABI) with the CPU oracle on the same seeded inputs.  -m gpu only.

Tolerances (north_star: 1e-5 relative over 1000 steps):
  * torque mode is non-chaotic -> 1000 free-running substeps, 1e-5 relative on (qpos, qvel)   [measured ~1e-13]
  * the reference's PD law is chaotic (tests/test_oracle_physics.py::test_sensitivity_documented): a 1e-12
    perturbation reaches O(1) within ~300 substeps in the oracle itself, so PD parity is asserted
    (a) teacher-forced per step over 1000 substeps at 1e-9, (b) free-running over the first 100 substeps at 1e-6,
    (c) free-running over 1000 substeps as a SHADOWING bound: the HIP path stays within C = 1000 times the envelope of what
        1-ulp perturbations of the initial state do to the oracle's own trajectory (test_pd_1000_substeps_shadowing_bound).
"""
import ctypes as ct

import numpy as np
import pytest

from conftest import state_vec

pytestmark = pytest.mark.gpu

DBG = dict(M=0, BIAS=169, QS=182, F0=195, B=241, R=287, AREF=333, ADIAG=379, F=425, QACC=471, QACCH=484)
PD_LO, PD_HI = np.radians([-50, -164, -140] * 2), np.radians([80, -37, -30] * 2)
TQ = np.array([12.0, 12.0, 0.9] * 2)


@pytest.fixture(scope="module")
def vec():
    from cassierl_amd.vec_env import CassieVecEnv
    return CassieVecEnv


def rel_err(sg, q1, v1):
    return max(np.abs(sg[:13] - q1).max() / np.abs(q1).max(), np.abs(sg[13:26] - v1).max() / (1e-3 + np.abs(v1).max()))


def test_stage_by_stage_against_planar_spec(vec):
    from planar_proto import Planar
    P = Planar()
    rng = np.random.default_rng(0)
    n = 6
    env = vec(n, kind="stand", control_mode="Torque", n_substeps=1, auto_reset=False)
    q0 = env.get_full_state_host()[0, :13]
    states, ctrls = [], []
    for i in range(n):
        q = q0 + rng.uniform(-0.05, 0.05, 13); q[1] -= 0.01 * i
        states.append(state_vec(q, rng.uniform(-1, 1, 13), rng.uniform(-5, 5, 13)))
        ctrls.append(rng.uniform(-1, 1, 6) * TQ)
    env.set_full_state_host(np.array(states))
    dbg = env.debug_substep_host("Torque", np.array(ctrls))
    after = env.get_full_state_host()
    for i in range(n):
        s = states[i]
        q2, v2, qacc, r = P.step(s[:13], s[13:26], s[26:39], ctrls[i])
        d = dbg[i]
        np.testing.assert_allclose(d[DBG["M"]:DBG["M"] + 169], r["M"].ravel(), atol=1e-13)
        np.testing.assert_allclose(d[DBG["BIAS"]:DBG["BIAS"] + 13], r["bias"], atol=1e-11)
        np.testing.assert_allclose(d[DBG["QS"]:DBG["QS"] + 13], r["qacc_smooth"], rtol=1e-10, atol=1e-9)
        act = r["active"]
        np.testing.assert_allclose(d[DBG["R"]:DBG["R"] + 46], np.where(act, r["R"], 0), rtol=1e-13)
        np.testing.assert_allclose(d[DBG["ADIAG"]:DBG["ADIAG"] + 46], np.where(act, np.diag(r["A"]), 0), rtol=1e-11)
        np.testing.assert_allclose(d[DBG["B"]:DBG["B"] + 46], np.where(act, r["b"], 0), rtol=1e-9, atol=1e-8)
        np.testing.assert_allclose(d[DBG["F0"]:DBG["F0"] + 46], r["f0"], rtol=1e-9, atol=1e-8)
        np.testing.assert_allclose(d[DBG["F"]:DBG["F"] + 46], r["f"], rtol=1e-8, atol=1e-7)
        np.testing.assert_allclose(after[i, :13], q2, atol=1e-14)
        np.testing.assert_allclose(after[i, 13:26], v2, atol=1e-11)
        np.testing.assert_allclose(after[i, 26:39], qacc, rtol=1e-9, atol=1e-8)
        assert after[i, 85] == r["niter"]
    env.close()


def test_constructor_matches_cassie2d_ctor(vec, oracle_mod):
    env = vec(3, kind="stand", control_mode="Torque", n_substeps=1, auto_reset=False)
    s = env.get_full_state_host()
    o = oracle_mod.Oracle()
    q, v = o.state()
    for i in range(3):
        assert np.array_equal(s[i, :13], q) and np.array_equal(s[i, 13:26], v)
        np.testing.assert_allclose(s[i, 26:39], o.warmstart(), rtol=1e-9, atol=1e-9)  # mj_forward in the ctor
    env.close()


@pytest.mark.parametrize("mode", ["Torque", "PD"])
def test_teacher_forced_1000_substeps(vec_tier, oracle_mod, mode):
    vec = vec_tier
    rng = np.random.default_rng(1)
    n = 2
    env = vec(n, kind="stand", control_mode=mode, n_substeps=1, auto_reset=False)
    o = oracle_mod.Oracle()
    worst, maxrows = 0.0, 0
    for i in range(1000):
        if i % 10 == 0:
            a = rng.uniform(-1, 1, 6) * TQ if mode == "Torque" else rng.uniform(PD_LO, PD_HI)
        qo, vo = o.state()
        env.set_full_state_host(np.tile(state_vec(qo, vo, o.warmstart()), (n, 1)))
        env.substep_host(mode, np.tile(a, (n, 1)), 1)
        (o.step_torque if mode == "Torque" else o.step_pd)(a)
        sg = env.get_full_state_host()
        assert np.array_equal(sg[0], sg[1])
        q1, v1 = o.state()
        worst = max(worst, np.abs(sg[0, :13] - q1).max(), np.abs(sg[0, 13:26] - v1).max() / (1 + np.abs(v1).max()))
        maxrows = max(maxrows, o.nefc)
    assert worst < 1e-9, worst
    assert maxrows >= 18
    env.close()


def test_free_running_torque_1000_substeps(vec_tier, oracle_mod):
    vec = vec_tier
    rng = np.random.default_rng(2)
    n = 8
    env = vec(n, kind="stand", control_mode="Torque", n_substeps=1, auto_reset=False)
    oracles = [oracle_mod.Oracle() for _ in range(n)]
    s0 = []
    for o in oracles:
        q, v = o.state()
        s0.append(state_vec(q, v, o.warmstart()))
    env.set_full_state_host(np.array(s0))
    worst = 0.0
    for t in range(100):
        acts = rng.uniform(-1, 1, (n, 6)) * TQ
        env.substep_host("Torque", acts, 10)
        for i, o in enumerate(oracles):
            for _ in range(10):
                o.step_torque(acts[i])
        sg = env.get_full_state_host()
        for i, o in enumerate(oracles):
            q1, v1 = o.state()
            worst = max(worst, rel_err(sg[i], q1, v1))
    assert worst < 1e-5, worst  # north_star tolerance; measured ~1e-13
    env.close()


def test_free_running_torque_10000_substeps_drift(vec_tier, oracle_mod):
    """The drift run of tools/parity_drift.py as a test: 10 000 FREE-RUNNING torque substeps = 1000 Env.steps, smooth
    random torques (a new draw every 200 substeps: the robot sways, falls and rolls on the ground), HIP path against the oracle from
    the same state, no teacher forcing; the north_star bar (1e-5 relative) at every 100th substep, each first tier.
    Measured: 3e-13 .. 7e-12 depending on the tier's roundings."""
    vec = vec_tier
    rng = np.random.default_rng(7)
    env = vec(1, kind="stand", control_mode="Torque", n_substeps=10, auto_reset=False)
    o = oracle_mod.Oracle()
    q, v = o.state()
    env.set_full_state_host(state_vec(q, v, o.warmstart())[None])
    worst, u = 0.0, np.zeros(6)
    for blk in range(1000):
        if blk % 20 == 0:
            u = rng.uniform(-0.25, 0.25, 6) * TQ
        env.substep_host("Torque", u[None], 10)
        for _ in range(10):
            o.step_torque(u)
        if (blk + 1) % 10 == 0:
            s = env.get_full_state_host()[0]
            q1, v1 = o.state()
            worst = max(worst, rel_err(s, q1, v1))
            assert worst < 1e-5, ((blk + 1) * 10, worst)
    print("cassie2d[%s] 10000 free-running torque substeps: worst relative deviation %.3e, pelvis z at the end %.3f" % (vec.tier, worst, q1[1]))
    env.close()


SHADOW_C = 1000.0    # the HIP path's per-substep rounding differs from the oracle's by ~1e-13 relative = ~1000 ulp
SHADOW_FLOOR = 1e-12


def _state_dist(a, q, v):
    return max(np.abs(a[:13] - q).max() / np.abs(q).max(), np.abs(a[13:26] - v).max() / (1e-3 + np.abs(v).max()))


def test_pd_1000_substeps_shadowing_bound(vec_tier, oracle_mod):
    """North-star horizon for the mode the bench and TRPO run (Cassie2d::StepPd, Cassie2d.cpp:96-117): 1000 FREE-RUNNING PD
    substeps, 8 envs.  The PD law is chaotic in the reference itself, so agreement is asserted relative to the oracle's own
    sensitivity: E(t) = running max over four 1-ulp perturbations of qpos(0) of ||oracle_perturbed - oracle||(t); the HIP
    trajectory must satisfy ||hip - oracle||(t) <= C * E(t) + floor at every Env.step boundary, i.e. it is
    indistinguishable from an oracle run whose initial state was off by ~C ulp."""
    vec = vec_tier
    rng = np.random.default_rng(3)
    n, T = 8, 100
    acts = rng.uniform(PD_LO, PD_HI, (T, n, 6))
    base = [oracle_mod.Oracle() for _ in range(n)]
    pert = []
    for k in range(4):
        prng = np.random.default_rng(100 + k)
        grp = [oracle_mod.Oracle() for _ in range(n)]
        for o in grp:
            q, v = o.state()
            o.set_state_raw(np.nextafter(q, q + prng.choice([-1.0, 1.0], 13)), v, o.warmstart())
        pert.append(grp)
    env = vec(n, kind="stand", control_mode="PD", n_substeps=1, auto_reset=False)
    env.set_full_state_host(np.array([state_vec(*o.state(), o.warmstart()) for o in base]))
    E = np.zeros(n)
    worst_ratio, first_o1 = 0.0, None
    # The bound is only a TEST while it is small: `tight[i]` counts the Env.steps of environment i in which
    # C * E + floor < TIGHT -- there the assertion below is an absolute bar of 1e-6 on the HIP-vs-oracle distance, the part of this test
    # that can fail; `vacuous[i]` is the first substep at which C * E reaches 1 (from there the assertion holds for any finite state).
    TIGHT = 1e-6
    tight, worst_tight, vacuous = np.zeros(n, dtype=int), 0.0, [None] * n
    for t in range(T):
        env.substep_host("PD", acts[t], 10)
        sg = env.get_full_state_host()
        for i in range(n):
            for _ in range(10):
                base[i].step_pd(acts[t, i])
            q, v = base[i].state()
            for grp in pert:
                for _ in range(10):
                    grp[i].step_pd(acts[t, i])
                qp, vp = grp[i].state()
                E[i] = max(E[i], _state_dist(np.concatenate([qp, vp]), q, v))
            d = _state_dist(sg[i], q, v)
            assert d <= SHADOW_C * E[i] + SHADOW_FLOOR, (t, i, d, E[i])
            if SHADOW_C * E[i] + SHADOW_FLOOR < TIGHT:
                assert vacuous[i] is None and tight[i] == t, "the envelope is a running maximum: the tight window is a prefix"
                assert d < TIGHT, (t, i, d)
                tight[i] += 1
                worst_tight = max(worst_tight, d)
            if vacuous[i] is None and SHADOW_C * E[i] >= 1.0:
                vacuous[i] = (t + 1) * 10
            worst_ratio = max(worst_ratio, d / (E[i] + SHADOW_FLOOR / SHADOW_C))
        if first_o1 is None and E.max() > 1e-2:
            first_o1 = t
    # the bound must have been informative for a good part of the horizon: the oracle's own perturbations only reach O(1e-2)
    # after a few hundred substeps (if this fails the test inputs changed, not the kernel)
    assert first_o1 is None or first_o1 >= 15, first_o1
    assert np.isfinite(sg).all()
    # the informative part: every environment has at least 5 Env.steps (50 free-running PD substeps), the batch on average 10, inside
    # the absolute 1e-6 bar before chaos takes the envelope away -- and where it ends is printed, not hidden
    print("PD shadowing [%s]: tight window (bar %.0e) per env, substeps: %s; worst distance inside it %.2e; bound vacuous (C*E >= 1) from substep: %s"
          % (vec.tier, TIGHT, (tight * 10).tolist(), worst_tight, vacuous))
    assert tight.min() >= 5 and tight.mean() >= 10, tight
    env.close()


TWIN_EPS = 3e-14     # measured per-substep deviation of the HIP path from the oracle (tools/teacher_forced_error.py: max 3.2e-14)
TWIN_C = 20.0


def test_pd_env_streams_agree_until_the_oracle_itself_flips(vec, oracle_mod, traj):
    """Same idea one level up: Cassie2dEnv.step (walk env, PD, robots free to fall, CASSIE_FIX_STALE_QSTATE so the episode is
    not cut at the first step) over 100 Env.steps = 1000 substeps.  Reward and done from the HIP path must follow the
    oracle env as closely as the oracle's own twins do when their state is disturbed, at every Env.step, by the relative
    amount the HIP path is measured to differ from the oracle in ONE substep (TWIN_EPS; the HIP path injects that ten times
    per Env.step, hence a constant of order 10; measured worst ratio 1.6-1.9, TWIN_C = 20)."""
    rng = np.random.default_rng(11)
    n, T = 6, 100
    FIXQ = 2
    acts = rng.uniform(PD_LO, PD_HI, (T, n, 6))
    base = [oracle_mod.OracleEnv("walk", "PD", flags=FIXQ, traj=traj) for _ in range(n)]
    twins = [[oracle_mod.OracleEnv("walk", "PD", flags=FIXQ, traj=traj) for _ in range(n)] for _ in range(4)]
    env = vec(n, kind="walk", control_mode="PD", n_substeps=10, flags=FIXQ, auto_reset=False)
    env.set_trajectory(traj["time"], traj["qpos"])
    obs = env.reset_host()
    for i, e in enumerate(base):
        assert np.abs(e.reset() - obs[i]).max() < 1e-12
    prng = np.random.default_rng(200)

    def disturb(e):
        q, v = e.oracle.state()
        e.oracle.set_state_raw(q * (1.0 + TWIN_EPS * prng.uniform(-1, 1, 13)), v * (1.0 + TWIN_EPS * prng.uniform(-1, 1, 13)), e.oracle.warmstart())

    for grp in twins:
        for e in grp:
            e.reset()
            disturb(e)
    Er = np.zeros(n)
    checked_done, worst_ratio = 0, 0.0
    for t in range(T):
        o_g, r_g, d_g = env.step_host(acts[t])
        for i in range(n):
            _, r, d = base[i].step(acts[t, i])
            for grp in twins:
                _, rp, dp = grp[i].step(acts[t, i])
                Er[i] = max(Er[i], abs(rp - r), 1.0 if dp != d else 0.0)
                disturb(grp[i])
            assert abs(r_g[i] - r) <= TWIN_C * Er[i] + 1e-10, (t, i, r_g[i], r, Er[i])
            worst_ratio = max(worst_ratio, abs(r_g[i] - r) / (Er[i] + 1e-12))
            if TWIN_C * Er[i] < 1e-3:  # the oracle's own perturbations are still far from changing the outcome
                assert bool(d_g[i]) == d, (t, i)
                checked_done += 1
    print("pd_env_shadowing worst |r_hip - r| / E = %.2f, done checked %d times" % (worst_ratio, checked_done))
    assert checked_done >= n * 10
    env.close()


def test_free_running_pd_first_100_substeps(vec_tier, oracle_mod):
    vec = vec_tier
    rng = np.random.default_rng(3)
    n = 4
    env = vec(n, kind="stand", control_mode="PD", n_substeps=1, auto_reset=False)
    oracles = [oracle_mod.Oracle() for _ in range(n)]
    env.set_full_state_host(np.array([state_vec(*o.state(), o.warmstart()) for o in oracles]))
    worst = 0.0
    for t in range(10):
        acts = rng.uniform(PD_LO, PD_HI, (n, 6))
        env.substep_host("PD", acts, 10)
        sg = env.get_full_state_host()
        for i, o in enumerate(oracles):
            for _ in range(10):
                o.step_pd(acts[i])
            worst = max(worst, rel_err(sg[i], *o.state()))
    assert worst < 1e-6, worst
    env.close()


@pytest.mark.parametrize("tag,kind,mode", [("walk_pd", "walk", "PD"), ("walk_torque", "walk", "Torque"),
                                          ("stand_torque", "stand", "Torque"), ("stand_pd", "stand", "PD")])
def test_env_step_against_golden_streams(vec_tier, streams, traj, tag, kind, mode):
    """Env.step / reset / auto-reset through the batched ABI against streams recorded from the reference's own Python env."""
    vec = vec_tier
    n = 3
    env = vec(n, kind=kind, control_mode=mode, n_substeps=10, auto_reset=True)
    env.set_trajectory(traj["time"], traj["qpos"])
    obs0 = env.reset_host()
    np.testing.assert_allclose(obs0, np.tile(streams[tag + "_obs0"], (n, 1)), atol=1e-12)
    acts = streams[tag + "_actions"]
    horizon = 40 if mode == "Torque" or kind == "walk" else 8  # PD stand runs free for 10*T substeps: chaos beyond ~100
    for t in range(horizon):
        obs, rew, done = env.step_host(np.tile(acts[t], (n, 1)))
        d = bool(streams[tag + "_done"][t])
        exp_obs = streams[tag + "_reset_obs"][t] if d else streams[tag + "_obs"][t]
        assert (done == d).all()
        np.testing.assert_allclose(rew, streams[tag + "_reward"][t], rtol=0, atol=1e-9)
        np.testing.assert_allclose(obs, np.tile(exp_obs, (n, 1)), rtol=0, atol=2e-7)
    env.close()


def test_env_step_vs_oracle_env_with_quirk_fixes(vec_tier, oracle_mod, traj):
    vec = vec_tier
    tr = dict(time=traj["time"], qpos=traj["qpos"])
    rng = np.random.default_rng(5)
    for flags in (0, 1, 3):
        n = 3
        env = vec(n, kind="walk", control_mode="Torque", n_substeps=10, auto_reset=True, flags=flags)
        env.set_trajectory(traj["time"], traj["qpos"])
        oes = [oracle_mod.OracleEnv("walk", "Torque", flags=flags, traj=tr) for _ in range(n)]
        np.testing.assert_allclose(env.reset_host(), np.array([e.reset() for e in oes]), atol=1e-12)
        for t in range(15):
            acts = rng.uniform(-1, 1, (n, 6)) * TQ
            obs, rew, done = env.step_host(acts)
            for i, e in enumerate(oes):
                o, r, d = e.step(acts[i])
                if d:
                    o = e.reset()
                assert d == done[i] and abs(r - rew[i]) < 1e-9
                np.testing.assert_allclose(obs[i], o, atol=1e-7)
        env.close()


def test_edge_cases_limits_many_contacts_single_env(vec_tier, oracle_mod):
    vec = vec_tier
    # (a) n_envs = 1 (b) joint limits active (c) collapsed robot: many simultaneous contacts incl. pelvis/thigh/shin spheres
    env = vec(1, kind="stand", control_mode="Torque", n_substeps=1, auto_reset=False)
    o = oracle_mod.Oracle()
    maxcon, maxlim = 0, 0
    push = np.array([12.2, -12.2, 0.9, 12.2, -12.2, 0.9])
    for i in range(5000):
        a = push if (i < 800) else np.zeros(6)   # drives all 8 limited joints into their limits, then collapses
        if i == 1200:                             # second scenario: passive collapse from the reset pose (7 contacts)
            o = oracle_mod.Oracle()
        if i >= 1200:
            a = np.zeros(6)
        if i % 20 == 0:  # teacher-forced comparison every 25 substeps along a fall with limits and many contacts
            q, v = o.state()
            env.set_full_state_host(state_vec(q, v, o.warmstart())[None])
            env.substep_host("Torque", a[None], 1)
            o.step_torque(a)
            sg = env.get_full_state_host()[0]
            q1, v1 = o.state()
            assert np.abs(sg[:13] - q1).max() < 1e-10 and np.abs(sg[13:26] - v1).max() < 1e-8 * (1 + np.abs(v1).max())
            e = o.efc()
            maxcon = max(maxcon, o.ncon); maxlim = max(maxlim, int((e["type"] == 1).sum()))
        else:
            o.step_torque(a)
    assert maxcon >= 6 and maxlim >= 6, (maxcon, maxlim)
    env.close()


def test_full_size_properties_65536_envs(vec, traj):
    """BASELINE full size: size-independent properties instead of the (too slow) oracle."""
    import torch
    n = 65536
    env = vec(n, kind="stand", control_mode="Torque", n_substeps=10, auto_reset=False)
    out = env.alloc()
    rng = np.random.default_rng(9)
    base = rng.uniform(-1, 1, (64, 6)) * TQ
    acts = np.tile(base, (n // 64, 1))
    # mirror property: env 2k+1 gets the left/right-swapped action of env 2k
    acts[1::2] = acts[0::2][:, [3, 4, 5, 0, 1, 2]]
    a = torch.as_tensor(acts, device="cuda")
    for _ in range(3):
        env.step(a, out)
    env.synchronize()
    q, v = env.get_state_host()
    assert np.isfinite(q).all() and np.isfinite(v).all()
    # determinism / index independence: envs with identical inputs are bit-identical wherever they sit in the grid
    assert np.array_equal(q[:128], q[-128:]) and np.array_equal(v[:128], v[n // 2:n // 2 + 128])
    # left/right mirror symmetry (PGS sweeps left rows first in both, so equality is to solver accuracy)
    swap = np.r_[0:3, 8:13, 3:8]
    np.testing.assert_allclose(q[1::2][:, swap], q[0::2], atol=2e-4)
    assert np.abs(q[:, 1] - 0.939).max() < 0.2  # nobody exploded in 30 substeps
    env.close()


def test_legacy_abi_batch_of_one(oracle_mod):
    """The ten reference symbols (Cassie2d.cpp:15-27), driven exactly like rllab/envs/cassie2d.py drives them."""
    from cassierl_amd import _lib
    from cassierl_amd import structs as S
    L = _lib.load()
    L.Reset.argtypes = [ct.c_void_p, ct.POINTER(S.StateGeneral)]
    L.StepPd.argtypes = [ct.c_void_p, ct.POINTER(S.ControllerPd)]
    L.StepTorque.argtypes = [ct.c_void_p, ct.POINTER(S.ControllerTorque)]
    L.GetGeneralState.argtypes = [ct.c_void_p, ct.POINTER(S.StateGeneral)]
    L.GetOperationalSpaceState.argtypes = [ct.c_void_p, ct.POINTER(S.StateOperationalSpace)]
    L.Display.argtypes = [ct.c_void_p, ct.c_bool]
    h = L.Cassie2dInit()
    L.Display(h, True)
    cv = S.InterfaceStructConverter()
    qinit = np.array([0.0, 0.939, 0.0, 0.0, 0.0, 0.0, 0.68111815, -1.40730357, 1.62972042, -1.77611107, -0.61968407, 0, 0, 0, 0, 0,
                      0.68111815, -1.40730357, 1.62972042, -1.77611107, -0.61968407, 0, 0, 0, 0, 0])
    L.Reset(h, cv.array_to_general_state(qinit))
    o = oracle_mod.Oracle()
    q, v = S.general_array_to_qpos_qvel(qinit)
    o.reset(q, v)
    xs = S.StateOperationalSpace()
    L.GetOperationalSpaceState(h, ct.byref(xs))
    np.testing.assert_allclose(cv.operational_state_to_array(xs), o.opstate(0), atol=1e-13)
    rng = np.random.default_rng(6)
    for t in range(30):
        a = rng.uniform(-1, 1, 6) * TQ
        L.StepTorque(h, ct.byref(cv.array_to_torque_action(a)))
        o.step_torque(a)
    qs = S.StateGeneral()
    L.GetGeneralState(h, ct.byref(qs))
    qg, vg = S.general_array_to_qpos_qvel(cv.general_state_to_array(qs))
    q1, v1 = o.state()
    np.testing.assert_allclose(qg, q1, atol=1e-11)
    np.testing.assert_allclose(vg, v1, atol=1e-9)
    L.GetOperationalSpaceState(h, ct.byref(xs))
    np.testing.assert_allclose(cv.operational_state_to_array(xs), o.opstate(0), atol=1e-10)
    L.Render(h)


def test_g16_and_wave_per_env_kernels_agree(vec, traj):
    """The 4-envs-per-wave fast path (+ clean-up pass) and the wave-per-environment kernel are two implementations of the same
    step: run the same random-torque rollout through both, including falls that overflow 16 rows, and compare."""
    from cassierl_amd.vec_env import WAVE_PER_ENV
    n = 37  # not a multiple of 4: exercises the partial last wave
    rng = np.random.default_rng(17)
    a = vec(n, kind="stand", control_mode="Torque", n_substeps=10, auto_reset=True)
    b = vec(n, kind="stand", control_mode="Torque", n_substeps=10, auto_reset=True, flags=WAVE_PER_ENV)
    # (the packed reset runs the two-lanes-per-environment core, the other handle the wave-per-environment reset kernel: two formulations of the
    # same mj_forward, equal to rounding -- bitwise only by accident of how the compiler fused their multiply-adds)
    np.testing.assert_allclose(a.reset_host(), b.reset_host(), rtol=0, atol=1e-13)
    saw_overflow = False
    for t in range(150):
        acts = rng.uniform(-1, 1, (n, 6)) * TQ * (0.2 if t < 100 else 1.0)
        # teacher-force b from a's state so that rounding-level differences cannot grow across steps
        b.set_full_state_host(a.get_full_state_host())
        oa, ra, da = a.step_host(acts)
        ob, rb, db = b.step_host(acts)
        assert (da == db).all()
        np.testing.assert_allclose(ra, rb, rtol=0, atol=1e-10)
        np.testing.assert_allclose(oa, ob, rtol=0, atol=1e-8)
        sa, sb = a.get_full_state_host(), b.get_full_state_host()
        np.testing.assert_allclose(sa[:, :26], sb[:, :26], rtol=0, atol=1e-8)
        np.testing.assert_allclose(sa[:, 84], sb[:, 84], atol=1e-12)  # env time
    a.close(); b.close()


def test_masked_reset_to_states_and_device_getters(vec_tier, oracle_mod):
    """CassieVecResetTo (Cassie2d::Reset with caller states, masked), CassieVecGetState and CassieVecGetOpState on device
    tensors against the oracle: Reset = mj_forward without setState, so the op-space state is still the one of the previous
    setState (quirk Q2) while qpos/qvel are the new ones."""
    vec = vec_tier
    import torch
    n = 11
    rng = np.random.default_rng(23)
    env = vec(n, kind="stand", control_mode="Torque", n_substeps=10, auto_reset=False)
    env.reset_host()
    os_ = [oracle_mod.Oracle() for _ in range(n)]
    q0 = np.array([0.0, 0.939, 0.0, 0.68111815, -1.40730357, 1.62972042, -1.77611107, -0.61968407] + [0.68111815, -1.40730357, 1.62972042, -1.77611107, -0.61968407])
    for o in os_:
        o.reset(q0, np.zeros(13))                      # Cassie2dEnv.reset pose (the constructor pose differs in the 8th digit)
    acts = rng.uniform(-1, 1, (n, 6)) * TQ * 0.3
    env.step_host(acts)                                # one Env.step so that a setState has happened
    for i, o in enumerate(os_):
        for _ in range(10):
            o.step_torque(acts[i])
    qn = q0 + rng.uniform(-0.05, 0.05, (n, 13))
    vn = rng.uniform(-0.3, 0.3, (n, 13))
    mask = (np.arange(n) % 3 != 0)
    obs = env.reset_to(torch.as_tensor(qn, device="cuda"), torch.as_tensor(vn, device="cuda"), mask=torch.as_tensor(mask.astype(np.uint8), device="cuda"))
    q, v = env.get_state()
    x = env.get_opstate()
    env.synchronize()
    q, v, x, obs = q.cpu().numpy(), v.cpu().numpy(), x.cpu().numpy(), obs.cpu().numpy()
    for i, o in enumerate(os_):
        if mask[i]:
            o.reset(qn[i], vn[i])
        qo, vo = o.state()
        tol = 1e-12 if mask[i] else 1e-10   # untouched envs carry 10 free-running substeps of rounding
        np.testing.assert_allclose(q[i], qo, atol=tol)
        np.testing.assert_allclose(v[i], vo, atol=tol * 100)
        np.testing.assert_allclose(x[i], o.opstate(0), atol=1e-9)
    # one more step from the mixed states: warm start / kinematics bookkeeping of reset and non-reset envs stays consistent
    acts = rng.uniform(-1, 1, (n, 6)) * TQ * 0.3
    env.step_host(acts)
    q, v = env.get_state_host()
    for i, o in enumerate(os_):
        for _ in range(10):
            o.step_torque(acts[i])
        qo, vo = o.state()
        np.testing.assert_allclose(q[i], qo, atol=1e-9)
        np.testing.assert_allclose(v[i], vo, atol=1e-7)
    env.close()


@pytest.mark.parametrize("flags", [0, 4])
def test_terminal_observation_on_a_caller_stream(vec, streams, traj, flags):
    """CassieVecStep's optional terminal_obs output (the observation the reference's Env.step returns before its caller
    resets) through the device-tensor API on a caller-owned HIP stream (CassieVecSetStream), both kernel generations."""
    import torch
    n = 5
    env = vec(n, kind="walk", control_mode="PD", n_substeps=10, auto_reset=True, flags=flags)
    env.set_trajectory(traj["time"], traj["qpos"])
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        env.use_torch_stream()
        out = env.alloc()
        term = torch.zeros((n, 26), dtype=torch.float64, device="cuda")
        env.reset(out)
        acts = streams["walk_pd_actions"]
        saw_done = False
        for t in range(40):
            a = torch.as_tensor(np.tile(acts[t], (n, 1)), device="cuda")
            obs, rew, done = env.step(a, out, terminal_obs=term)
            stream.synchronize()
            d = bool(streams["walk_pd_done"][t])
            saw_done |= d
            assert (done.cpu().numpy().astype(bool) == d).all()
            np.testing.assert_allclose(term.cpu().numpy(), np.tile(streams["walk_pd_obs"][t], (n, 1)), rtol=0, atol=2e-7)
            exp = streams["walk_pd_reset_obs"][t] if d else streams["walk_pd_obs"][t]
            np.testing.assert_allclose(obs.cpu().numpy(), np.tile(exp, (n, 1)), rtol=0, atol=2e-7)
    assert saw_done
    env.close()


@pytest.mark.parametrize("n", [1, 63, 1024, 4141, 65536])
def test_accumulate_is_the_torch_bookkeeping_in_one_launch(vec, n, traj):
    """CassieVecAccumulate (`returns += reward; episodes += done.count_nonzero()` on the env's stream) against the torch expressions it
    replaces in the rollout loops, on the rewards / flags the env itself produces and on ragged synthetic ones; either accumulator
    may be absent; an accumulator without its input is an error."""
    import torch
    from cassierl_amd import rollout as R
    env = vec(n, kind="walk", control_mode="PD", n_substeps=10, auto_reset=True)
    env.set_trajectory(traj["time"], traj["qpos"])
    env.use_torch_stream()
    out = env.alloc()
    env.reset(out)
    ids = torch.arange(n, device="cuda")
    low, high = env.action_space.low, env.action_space.high
    ret = torch.zeros(n, dtype=torch.float64, device="cuda"); ep = torch.zeros(1, dtype=torch.int64, device="cuda")
    ret_ref = torch.zeros_like(ret); ep_ref = 0
    for t in range(4):
        _, rew, dn = env.step(R.random_actions(3, ids, t, low, high), out)
        env.accumulate(rew, dn, ret, ep)
        ret_ref += rew; ep_ref += int(dn.count_nonzero())
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    rew = torch.randn(n, dtype=torch.float64, device="cuda", generator=g)
    dn = (torch.rand(n, device="cuda", generator=g) < 0.3).to(torch.uint8) * 7   # any non-zero byte is a done flag
    env.accumulate(rew, dn, ret, ep)
    ret_ref += rew; ep_ref += int(dn.count_nonzero())
    env.accumulate(rew, dn, None, ep); ep_ref += int(dn.count_nonzero())       # episodes only
    env.accumulate(rew, dn, ret, None); ret_ref += rew                          # returns only
    env.synchronize()
    assert torch.equal(ret, ret_ref) and int(ep.item()) == ep_ref
    assert env.L.CassieVecAccumulate(env.h, None, dn.data_ptr(), ret.data_ptr(), None) < 0
    assert env.L.CassieVecAccumulate(env.h, rew.data_ptr(), None, None, ep.data_ptr()) < 0
    env.close()


def test_error_behaviour_of_the_batched_abi(vec):
    """int error codes + CassieVecLastError instead of the reference's process exit (mju_error, Cassie2d.cpp:49-52)."""
    import ctypes as ct2
    import torch
    from cassierl_amd import _lib
    L = _lib.load()
    h = ct2.c_void_p()
    cfg = _lib.CassieVecConfig(0, 0, 10, 0, 1)
    assert L.CassieVecCreate(ct2.byref(h), 0, 0, ct2.byref(cfg)) != 0          # n_envs must be positive
    assert L.CassieVecCreate(ct2.byref(h), 4, 99, ct2.byref(cfg)) != 0         # no such device
    with pytest.raises(AssertionError, match="Invalid Control Mode"):           # cassie2d.py:53
        vec(2, control_mode="Velocity")
    env = vec(4, kind="walk", control_mode="PD")
    out = env.alloc()
    a = torch.zeros((4, 6), dtype=torch.float64, device="cuda")
    with pytest.raises(RuntimeError, match="CassieVecSetTrajectory"):           # walk reward needs the gait table
        env.step(a, out)
    rc = L.CassieVecStep(env.h, None, out["obs"].data_ptr(), out["reward"].data_ptr(), out["done"].data_ptr(), None)
    assert rc != 0 and b"null" in L.CassieVecLastError(env.h)
    assert L.CassieVecSubstep(env.h, 0, a.data_ptr(), 0) != 0                   # n_sub must be positive
    assert L.CassieVecStandingStep(env.h, 0, a.data_ptr(), a.data_ptr(), 1) != 0  # scripted controllers exist for OSC / Jacobian only
    assert L.CassieVecNumEnvs(env.h) == 4 and L.CassieVecActionDim(env.h) == 6
    env.close()


def test_packed_reset_kernel_agrees_with_the_wave_per_environment_reset(vec, monkeypatch):
    """CassieVecReset / CassieVecResetTo run the two-lanes-per-environment core (32 environments per wavefront; r04) and leave a state with
    more than 8 rows on a leg to the wave-per-environment reset kernel (CASSIE2D_RESET_PACKED=0: that kernel for everyone).  Same records
    and observations from both on: the reset pose, random poses around it, robots pushed into the floor and into their joint limits
    (which take the slow path), a mask; environments outside the mask untouched to the bit."""
    import torch
    rng = np.random.default_rng(31)
    n = 333
    qinit = np.array([0.0, 0.939, 0.0, 0.0, 0.0, 0.0, 0.0, 0.68111815, -1.40730357, 1.62972042, -1.77611107, -0.61968407, 0.0])
    recs, obss = [], []
    for packed in ("1", "0"):
        monkeypatch.setenv("CASSIE2D_RESET_PACKED", packed)
        e = vec(n, kind="stand", control_mode="Torque", n_substeps=10, auto_reset=False)
        e.reset_host()
        q0, _ = e.get_state_host()
        r = np.random.default_rng(32)
        for t in range(3):   # stale ctrl / warm start / setState copies that differ per environment
            e.step_host(r.uniform(-1, 1, (n, 6)) * TQ)
        q, v = e.get_state_host()
        qn = q0 + rng.normal(0, 0.08, (n, 13)) if packed == "1" else qn
        vn = rng.normal(0, 0.5, (n, 13)) if packed == "1" else vn
        if packed == "1":
            qn[::7, 1] = 0.35                                  # pelvis near the floor: many contacts
            qn[3::11, 3:] += rng.normal(0, 0.8, (len(qn[3::11]), 10))   # joints far outside their ranges: limits
            mask = (rng.uniform(size=n) < 0.7).astype(np.uint8)
        dev = "cuda:0"
        out = e.alloc()
        pre = e.get_full_state_host()
        obs = e.reset_to(torch.as_tensor(qn, device=dev), torch.as_tensor(vn, device=dev), out, mask=torch.as_tensor(mask, device=dev)).cpu().numpy().copy()
        before = e.get_full_state_host()
        assert np.array_equal(before[mask == 0], pre[mask == 0])   # outside the mask: untouched to the bit
        obs2 = e.reset(out, mask=torch.as_tensor(1 - mask, device=dev)).cpu().numpy().copy()   # the others: to the reset pose
        recs.append((before, e.get_full_state_host())); obss.append((obs, obs2))
        e.close()
    monkeypatch.delenv("CASSIE2D_RESET_PACKED")
    (b1, a1), (b0, a0) = recs
    assert np.isfinite(a1).all()
    np.testing.assert_allclose(b1, b0, rtol=0, atol=1e-8)
    np.testing.assert_allclose(a1, a0, rtol=0, atol=1e-8)
    m = mask.astype(bool)
    np.testing.assert_allclose(obss[0][0][m], obss[1][0][m], rtol=0, atol=1e-9)
    np.testing.assert_allclose(obss[0][1][~m], obss[1][1][~m], rtol=0, atol=1e-9)


@pytest.mark.parametrize("kind,mode", [("walk", "PD"), ("stand", "Torque"), ("stand", "OSC")])
def test_declared_observation_space_contains_the_rows_the_env_emits(kind, mode, traj):
    """a consumer that sizes its policy from env.observation_space (as trpo_cassie.py does through env.spec) must get the
    width step() / reset() return -- 26 for both env kinds; the reference stand env's own Box(17) is `reference_observation_space`
    with `obs_view()` as the matching view."""
    import torch
    from cassierl_amd.vec_env import CassieVecEnv
    env = CassieVecEnv(64, kind=kind, control_mode=mode, n_substeps=10)
    if kind == "walk":
        env.set_trajectory(traj["time"], traj["qpos"])
    space = env.observation_space
    obs0 = env.reset().cpu().numpy()
    a = torch.as_tensor(np.stack([env.action_space.sample(np.random.default_rng(i)) for i in range(64)]) * 0.1, device="cuda:0")
    obs, _, _ = env.step(a)
    obs = obs.cpu().numpy()
    assert obs.shape[1:] == space.shape == (26,)
    assert all(space.contains(o) for o in obs0) and all(space.contains(o) for o in obs)
    ref = env.reference_observation_space
    assert ref.shape == ((17,) if kind == "stand" else (26,)) and all(ref.contains(o) for o in env.obs_view(obs))
    env.close()

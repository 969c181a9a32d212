"""The two-lanes-per-environment kernel source (cassierl_amd/csrc/cassie_leg_core.h), compiled for the CPU by
oracle/leg_host/leg_host.cpp, against the oracle: the same checks tests/test_gpu_parity.py runs on the GPU through the C-ABI,
here without one -- so that a formulation error (block factorisation, factored A, row slots, sweep order) is found on the CPU."""
import numpy as np
import pytest

from conftest import state_vec
from leg_host import LegHostEnv

PD_LO, PD_HI = np.radians([-50, -164, -140] * 2), np.radians([80, -37, -30] * 2)
TQ = np.array([12.0, 12.0, 0.9] * 2)


def rel_err(sg, q1, v1):
    return max(np.abs(sg[:13] - q1).max() / np.abs(q1).max(), np.abs(sg[13:26] - v1).max() / (1e-3 + np.abs(v1).max()))


@pytest.mark.parametrize("mode", ["Torque", "PD"])
def test_teacher_forced_1000_substeps(oracle_mod, mode):
    rng = np.random.default_rng(1)
    env = LegHostEnv(1, n_substeps=1, auto_reset=False)
    o = oracle_mod.Oracle()
    worst, maxrows, worst_ws = 0.0, 0, 0.0
    for i in range(1000):
        if i % 10 == 0:
            a = rng.uniform(-1, 1, 6) * TQ if mode == "Torque" else rng.uniform(PD_LO, PD_HI)
        qo, vo = o.state()
        env.set_full_state_host(state_vec(qo, vo, o.warmstart())[None])
        env.substep_host(mode, a[None], 1)
        assert env.pending[0] == 0
        (o.step_torque if mode == "Torque" else o.step_pd)(a)
        sg = env.get_full_state_host()[0]
        q1, v1 = o.state()
        worst = max(worst, np.abs(sg[:13] - q1).max(), np.abs(sg[13:26] - v1).max() / (1 + np.abs(v1).max()))
        worst_ws = max(worst_ws, np.abs(sg[26:39] - o.warmstart()).max() / (1 + np.abs(o.warmstart()).max()))
        maxrows = max(maxrows, o.nefc)
    assert worst < 1e-9 and worst_ws < 1e-6, (worst, worst_ws)
    assert maxrows >= 18


def test_free_running_torque_1000_substeps(oracle_mod):
    rng = np.random.default_rng(2)
    n = 4
    env = LegHostEnv(n, n_substeps=1, auto_reset=False)
    oracles = [oracle_mod.Oracle() for _ in range(n)]
    env.set_full_state_host(np.array([state_vec(*o.state(), o.warmstart()) for o in oracles]))
    worst = 0.0
    for t in range(100):
        acts = rng.uniform(-1, 1, (n, 6)) * TQ
        env.substep_host("Torque", acts, 10)
        assert (env.pending == 0).all()
        for i, o in enumerate(oracles):
            for _ in range(10):
                o.step_torque(acts[i])
        sg = env.get_full_state_host()
        for i, o in enumerate(oracles):
            worst = max(worst, rel_err(sg[i], *o.state()))
    assert worst < 1e-5, worst   # north_star tolerance


@pytest.mark.parametrize("tag,kind,mode", [("walk_pd", "walk", "PD"), ("walk_torque", "walk", "Torque"),
                                          ("stand_torque", "stand", "Torque"), ("stand_pd", "stand", "PD")])
def test_env_step_against_golden_streams(streams, traj, oracle_mod, tag, kind, mode):
    env = LegHostEnv(2, kind=kind, control_mode=mode, n_substeps=10, auto_reset=True)
    env.set_trajectory(traj["time"], traj["qpos"])
    # Cassie2dEnv.reset through the oracle env (the emulation has no reset entry of its own: the kernel resets inside Env.step)
    oe = oracle_mod.OracleEnv(kind, mode, traj=dict(time=traj["time"], qpos=traj["qpos"]))
    oe.reset()
    o = oe.oracle
    q, v = o.state()
    ctor = oracle_mod.Oracle()
    s0 = state_vec(q, v, o.warmstart(), kq=ctor.state()[0], kv=ctor.state()[1], qstate=q)   # stale kinematics of the ctor (quirk Q2)
    env.set_full_state_host(np.tile(s0, (2, 1)))
    acts = streams[tag + "_actions"]
    horizon = 40 if mode == "Torque" or kind == "walk" else 8
    for t in range(horizon):
        obs, rew, done = env.step_host(np.tile(acts[t], (2, 1)))
        assert (env.pending == 0).all()
        d = bool(streams[tag + "_done"][t])
        exp_obs = streams[tag + "_reset_obs"][t] if d else streams[tag + "_obs"][t]
        assert (done == d).all()
        np.testing.assert_allclose(rew, streams[tag + "_reward"][t], rtol=0, atol=1e-9)
        np.testing.assert_allclose(obs, np.tile(exp_obs, (2, 1)), rtol=0, atol=2e-7)
        assert np.array_equal(env.state[0], env.state[1])


def test_limits_contacts_and_the_capacity_hand_over(oracle_mod):
    """A robot driven into its joint limits and collapsing: wherever a leg needs at most 8 rows the step agrees with the oracle;
    beyond that the environment is handed over untouched (pending = substeps left)."""
    env = LegHostEnv(1, n_substeps=1, auto_reset=False)
    o = oracle_mod.Oracle()
    push = np.array([12.2, -12.2, 0.9, 12.2, -12.2, 0.9])
    done_in, handed, maxlim, maxcon = 0, 0, 0, 0
    for i in range(5000):
        a = push if i < 800 else np.zeros(6)
        if i == 1200:
            o = oracle_mod.Oracle()
        if i % 10 == 0:
            q, v = o.state()
            s0 = state_vec(q, v, o.warmstart())
            env.set_full_state_host(s0[None])
            env.substep_host("Torque", a[None], 1)
            o.step_torque(a)
            sg = env.get_full_state_host()[0]
            if env.pending[0]:
                handed += 1
                assert env.pending[0] == 1 and np.array_equal(sg[:39], s0[:39])
            else:
                done_in += 1
                q1, v1 = o.state()
                assert np.abs(sg[:13] - q1).max() < 1e-10 and np.abs(sg[13:26] - v1).max() < 1e-8 * (1 + np.abs(v1).max()), i
                e = o.efc()
                maxcon = max(maxcon, o.ncon); maxlim = max(maxlim, int((e["type"] == 1).sum()))
        else:
            o.step_torque(a)
    assert done_in > 100 and handed > 0 and maxlim >= 2 and maxcon >= 4, (done_in, handed, maxlim, maxcon)


@pytest.mark.parametrize("mode", ["Torque", "PD"])
def test_height_field_ramp_teacher_forced(oracle_mod, mode):
    """The height-field instantiation of the kernel source (cassie_leg_core.h with HF = true: terrain collision stage, contact
    frame from the local plane) on the CPU against the oracle's hfield_sphere: robots on the flat part of a ramp, across its kink
    and on the 10 % slope, 200 teacher-forced substeps (the GPU twin is tests/test_gpu_terrain.py).  The emulation fills unused
    contact slots with NaN normals: nothing may leak from them."""
    from cassierl_amd import terrain as T
    hm = T.ramp(nrow=64, ncol=2001, size_x=10.0, slope=0.1, x0=0.5)
    shifts = [(-1.0, 0.0), (0.45, 0.002), (1.0, 0.05), (3.3, 0.28)]
    os_ = []
    for dx, dz in shifts:
        o = oracle_mod.Oracle()
        o.set_hfield(hm, 10.0, 10.0)
        q, v = o.state()
        q[0] += dx; q[1] += dz
        o.set_state_raw(q, v, np.zeros(13))
        os_.append(o)
    n = len(os_)
    env = LegHostEnv(n, n_substeps=1, auto_reset=False)
    env.set_heightfield(hm, 10.0, 10.0)
    rng = np.random.default_rng(12)
    worst, sloped = 0.0, 0
    for t in range(200):
        if t % 10 == 0:
            a = rng.uniform(-1, 1, (n, 6)) * TQ if mode == "Torque" else rng.uniform(PD_LO, PD_HI, (n, 6))
        env.set_full_state_host(np.array([state_vec(*o.state(), o.warmstart()) for o in os_]))
        env.substep_host(mode, a, 1)
        assert (env.pending == 0).all() and env.nonfinite == 0
        sg = env.get_full_state_host()
        for i, o in enumerate(os_):
            (o.step_torque if mode == "Torque" else o.step_pd)(a[i])
            q1, v1 = o.state()
            worst = max(worst, np.abs(sg[i, :13] - q1).max(), np.abs(sg[i, 13:26] - v1).max() / (1 + np.abs(v1).max()))
            if o.ncon and np.abs(o.contacts()["frame"][:, 0]).max() > 0.05:
                sloped += 1
    assert worst < 1e-9, worst
    assert sloped > 100   # contacts with a tilted frame were really exercised


def test_packed_leg_constants_are_in_sync_with_the_planar_tables():
    """cassie2d_legk.h (what the two-lanes-per-environment kernel reads) is generated from cassie2d_planar.h by
    cassierl_amd/model/pack_leg_consts.py, by hand, after compile_model.py: regenerate it in memory and compare with the checked-in
    header, so that a model change cannot leave the kernel on stale constants."""
    import os
    from cassierl_amd.model import pack_leg_consts as P
    assert P.render() == open(P.DST).read()


def test_timing_build_equals_the_parity_build_bit_for_bit(oracle_mod):
    """bench.py's same-source CPU leg (libleg_host_fast.so: eight lanes = four environments per AVX-512 register, -O3 -march=native,
    OpenMP) computes what the two-lane parity build computes -- the CPU counterpart of the GPU's wavefront-neighbour tests: which
    environments share a register must not matter.  Falling robots, torque mode, auto-reset, 3 threads over 21 environments."""
    import ctypes as ct
    import leg_host as LH
    rng = np.random.default_rng(5)
    n = 21   # not a multiple of four: the last group has idle lanes
    a = LegHostEnv(n, kind="stand", control_mode="Torque", n_substeps=10, auto_reset=True)
    o = oracle_mod.Oracle()
    q, v = o.state()
    s0 = np.tile(state_vec(q, v, o.warmstart()), (n, 1))
    s0[:, 1] += rng.uniform(-0.01, 0.01, n)
    a.set_full_state_host(s0)
    fast = LH.lib(fast=True)
    assert fast.leg_host_lanes() == 8
    sb = s0.copy()
    dp = ct.POINTER(ct.c_double)
    for t in range(40):
        acts = np.ascontiguousarray(rng.uniform(-1, 1, (n, 6)) * TQ)
        oa, ra, da = a.step_host(acts)
        ob, rb, db = np.zeros((n, 26)), np.zeros(n), np.zeros(n, dtype=np.uint8)
        pend, bad = np.zeros(n, dtype=np.int32), ct.c_int(0)
        fast.leg_host_step(sb.ctypes.data_as(dp), acts.ctypes.data_as(dp), n, 6, 1, 10, 0, 1, 1, None, ct.c_double(0.0), 0, ob.ctypes.data_as(dp),
                           rb.ctypes.data_as(dp), db.ctypes.data_as(ct.POINTER(ct.c_ubyte)), None, pend.ctypes.data_as(ct.POINTER(ct.c_int)), ct.byref(bad), 3)
        # an environment over the rows-per-leg capacity is left untouched by both builds (the GPU hands it to the next tier): skip it from then on
        keep = (a.pending == 0) & (pend == 0)
        assert np.array_equal(a.pending, pend)
        assert np.array_equal(a.state[keep], sb[keep]) and np.array_equal(ra[keep], rb[keep]) and np.array_equal(da[keep], db[keep].astype(bool))
        if not keep.all():
            a.state[~keep] = s0[~keep]; sb[~keep] = s0[~keep]

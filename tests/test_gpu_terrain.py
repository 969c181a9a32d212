"""N4: height-field terrain on the HIP path (CassieVecSetHeightField) against the oracle's hfield_sphere on the same seeded inputs:
a synthetic ramp and one of the reference's terrain images (tests/golden/terrain_png.npz).  -m gpu only."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, state_vec
from cassierl_amd import terrain as T

pytestmark = pytest.mark.gpu

TQ = np.array([12.0, 12.0, 0.9] * 2)
PD_LO, PD_HI = np.radians([-50, -164, -140] * 2), np.radians([80, -37, -30] * 2)


@pytest.fixture(scope="module")
def vec():
    from cassierl_amd.vec_env import CassieVecEnv
    return CassieVecEnv


@pytest.fixture(scope="module")
def png_field():
    gray = np.load(os.path.join(GOLDEN, "terrain_png.npz"))["gray"].astype(np.float64)
    hm = T.hfield_from_gray(gray, (10, 10, 0.2, 0.001))
    return hm - max(T.height_at(hm, 10, 10, x, y) for x in np.linspace(-0.2, 0.3, 26) for y in (-0.1305, 0.1305)) - 1e-4


def _tier_flags(t):
    """first physics tier of a test id: False = four envs per wavefront, True = wave per env, "leg" = two lanes per env (env_step_leg_hf_kernel),
    "duo" = 64 envs per wavefront (env_step_duo_hf_kernel: its DIRECT oracle ids, r06)"""
    from cassierl_amd.vec_env import WAVE_PER_ENV, LEG_TIER_ON, DUO_TIER_ON, DUO_TIER_OFF
    return {False: 0, True: WAVE_PER_ENV, "leg": LEG_TIER_ON | DUO_TIER_OFF, "duo": LEG_TIER_ON | DUO_TIER_ON}[t]


def rel_err(sg, q1, v1):
    return max(np.abs(sg[:13] - q1).max() / np.abs(q1).max(), np.abs(sg[13:26] - v1).max() / (1e-3 + np.abs(v1).max()))


def _oracles(oracle_mod, hm, shifts):
    os_ = []
    for dx, dz in shifts:
        o = oracle_mod.Oracle()
        o.set_hfield(hm, 10.0, 10.0)
        q, v = o.state()
        q[0] += dx; q[1] += dz
        o.set_state_raw(q, v, np.zeros(13))
        os_.append(o)
    return os_


@pytest.mark.parametrize("wave_per_env", [False, True, "leg", "duo"])
@pytest.mark.parametrize("mode", ["Torque", "PD"])
def test_ramp_teacher_forced_substeps(vec, oracle_mod, mode, wave_per_env):
    """300 substeps, teacher-forced each step, robots on the flat part, across the kink and on the slope (0.1) of the ramp; each
    kernel tier as the first one (4 envs per wavefront, wave per env, two lanes per env)."""
    from cassierl_amd.vec_env import WAVE_PER_ENV, LEG_TIER_ON
    hm = T.ramp(nrow=64, ncol=2001, size_x=10.0, slope=0.1, x0=0.5)
    shifts = [(-1.0, 0.0), (0.4, 0.0), (0.45, 0.002), (1.0, 0.05), (2.0, 0.15), (3.3, 0.28)]
    os_ = _oracles(oracle_mod, hm, shifts)
    n = len(os_)
    env = vec(n, kind="stand", control_mode=mode, n_substeps=1, auto_reset=False, flags=_tier_flags(wave_per_env))
    env.set_heightfield(hm, 10.0, 10.0)
    rng = np.random.default_rng(12)
    worst, sloped = 0.0, 0
    for t in range(300):
        if t % 10 == 0:
            a = rng.uniform(-1, 1, (n, 6)) * TQ if mode == "Torque" else rng.uniform(PD_LO, PD_HI, (n, 6))
        env.set_full_state_host(np.array([state_vec(*o.state(), o.warmstart()) for o in os_]))
        env.substep_host(mode, a, 1)
        sg = env.get_full_state_host()
        for i, o in enumerate(os_):
            (o.step_torque if mode == "Torque" else o.step_pd)(a[i])
            q1, v1 = o.state()
            worst = max(worst, np.abs(sg[i, :13] - q1).max(), np.abs(sg[i, 13:26] - v1).max() / (1 + np.abs(v1).max()))
            if o.ncon and np.abs(o.contacts()["frame"][:, 0]).max() > 0.05:
                sloped += 1
    assert worst < 1e-9, worst
    assert sloped > 300  # contacts with a tilted frame were really exercised
    env.close()


@pytest.mark.parametrize("leg_tier", [False, "leg", "duo"])
def test_png_terrain_free_running_1000_substeps(vec, oracle_mod, png_field, leg_tier):
    """North-star bar on terrain: 1000 free-running torque-mode substeps on one of the reference's terrain images, 8 robots
    dropped at different x; (qpos, qvel) within 1e-5 relative of the oracle throughout (robots land, tumble, lie on the relief)."""
    hm = png_field
    shifts = [(x, T.height_at(hm, 10, 10, x, 0.0) - T.height_at(hm, 10, 10, 0.0, 0.0) + 0.03) for x in (0.0, -2.5, 1.7, 3.1, -4.2, 5.5, 0.8, -0.9)]
    os_ = _oracles(oracle_mod, hm, shifts)
    n = len(os_)
    from cassierl_amd.vec_env import LEG_TIER_ON
    env = vec(n, kind="stand", control_mode="Torque", n_substeps=1, auto_reset=False, flags=_tier_flags(leg_tier))
    env.set_heightfield(hm, 10.0, 10.0)
    env.set_full_state_host(np.array([state_vec(*o.state(), o.warmstart()) for o in os_]))
    rng = np.random.default_rng(5)
    worst, maxcon, tilted = 0.0, 0, 0.0
    for t in range(100):
        acts = rng.uniform(-1, 1, (n, 6)) * TQ
        env.substep_host("Torque", acts, 10)
        sg = env.get_full_state_host()
        for i, o in enumerate(os_):
            for _ in range(10):
                o.step_torque(acts[i])
            worst = max(worst, rel_err(sg[i], *o.state()))
            maxcon = max(maxcon, o.ncon)
            if o.ncon:
                tilted = max(tilted, float(np.abs(o.contacts()["frame"][:, 0]).max()))
    assert worst < 1e-5, worst
    assert maxcon >= 4 and tilted > 0.02
    c = env.counters()
    assert c["nonfinite_resets"] == 0
    env.close()


@pytest.mark.parametrize("leg_tier", [False, "leg", "duo"])
def test_env_step_on_terrain_packed_vs_wave_per_env(vec, traj, png_field, leg_tier):
    """Cassie2dEnv.step (stand env, PD and torque) on the terrain: 257 robots spread over 12 m of relief, the packed kernel with
    its hand-over pass against the wave-per-environment kernel, teacher-forced per Env.step; resets land on the terrain too."""
    from cassierl_amd.vec_env import WAVE_PER_ENV, LEG_TIER_ON
    hm = png_field
    n = 257
    for mode in ("Torque", "PD"):
        a = vec(n, kind="stand", control_mode=mode, n_substeps=10, auto_reset=True, flags=_tier_flags(leg_tier))
        b = vec(n, kind="stand", control_mode=mode, n_substeps=10, auto_reset=True, flags=WAVE_PER_ENV)
        for e in (a, b):
            e.set_heightfield(hm, 10.0, 10.0)
            e.reset_host()
        s = a.get_full_state_host()
        xs = np.linspace(-6, 6, n)
        s[:, 0] += xs
        s[:, 39] += xs
        s[:, 1] += np.array([T.height_at(hm, 10, 10, x, 0.0) for x in xs]) - T.height_at(hm, 10, 10, 0.0, 0.0) + 0.02
        a.set_full_state_host(s)
        rng = np.random.default_rng(2)
        lo, hi = (-TQ, TQ) if mode == "Torque" else (PD_LO, PD_HI)
        worst, ndone = 0.0, 0
        for t in range(40):
            acts = rng.uniform(lo, hi, (n, 6))
            b.set_full_state_host(a.get_full_state_host())
            oa, ra, da = a.step_host(acts)
            ob, rb, db = b.step_host(acts)
            sa, sb = a.get_full_state_host(), b.get_full_state_host()
            assert np.isfinite(sa).all()
            err = np.abs(sa[:, :26] - sb[:, :26]).max(axis=1) / (1.0 + np.abs(sb[:, :26]).max(axis=1))
            worst = max(worst, float(err.max()), float(np.abs(ra - rb).max()))
            assert (da != db).sum() == 0
            ndone += int(da.sum())
        assert worst < (1e-10 if mode == "Torque" else (1e-6 if leg_tier else 1e-7)), (mode, worst)   # PD, leg tier: see test_gpu_fullsize.py
        assert np.abs(sa[:, 13:26]).max() > 1.0  # the robots are moving on the relief
        a.close(); b.close()


def test_zero_field_equals_flat_floor_and_field_can_be_removed(vec):
    n = 8
    a = vec(n, kind="stand", control_mode="Torque", n_substeps=10, auto_reset=True)
    b = vec(n, kind="stand", control_mode="Torque", n_substeps=10, auto_reset=True)
    b.set_heightfield(np.zeros((8, 16)), 10.0, 10.0)
    rng = np.random.default_rng(4)
    a.reset_host(); b.reset_host()
    for t in range(30):
        acts = rng.uniform(-1, 1, (n, 6)) * TQ
        oa, ra, da = a.step_host(acts)
        ob, rb, db = b.step_host(acts)
        assert np.abs(oa - ob).max() < 1e-8 and np.abs(ra - rb).max() < 1e-9 and (da == db).all()
    b.set_heightfield(None)
    b.set_full_state_host(a.get_full_state_host())
    oa, ra, da = a.step_host(acts)
    ob, rb, db = b.step_host(acts)
    assert np.array_equal(oa, ob) and np.array_equal(ra, rb)
    a.close(); b.close()


def _py_standing_osc(o, zpos, zvel):
    s = o.opstate(0)
    act = np.zeros(7)
    act[3] = 100.0 * (-5e-3 - s[7]); act[5] = 100.0 * (-5e-3 - s[13])
    act[0] = 100.0 * ((s[6] + s[12]) / 2.0 - s[0]) + 20.0 * (0.0 - s[3])
    act[1] = 100.0 * (zpos - s[1]) + 20.0 * (zvel - s[4])
    act[6] = 20.0 * (0.0 - s[2]) + 10.0 * (0.0 - s[5])
    return act


def _py_standing_jac(o, zpos, zvel):
    s = o.opstate(0)
    xt = (s[6] + s[12]) / 2.0
    fx = 200.0 * (xt - s[0]) + 50.0 * (0.0 - s[3])
    fz = max(0.5 * 9.806 * 31.0 + 200.0 * (zpos - s[1]) + 50.0 * (zvel - s[4]), 0.0)
    my = 100.0 * (0.0 - s[2]) + 10.0 * (0.0 - s[5])
    return np.array([fx, fz, my, fx, fz, my])


@pytest.mark.parametrize("wave_per_env", [False, True, "leg", "duo"])
@pytest.mark.parametrize("mode", ["OSC", "Jacobian"])
def test_controllers_step_on_the_ramp_teacher_forced(vec, oracle_mod, mode, wave_per_env):
    """StepOsc / StepJacobian on terrain (rllab/envs/terrain_random.py:51-76 rewrites the MJCF every Step* variant loads,
    Cassie2d.cpp:119-209): the controller is the flat-floor one -- it works from the RBDL model and the foot sites -- and the
    mj_step behind it collides with the height field.  200 teacher-forced substeps of the scripted standing controllers' commands
    on the flat part, across the kink and on the slope of the ramp: motor commands and states against the oracle."""
    from cassierl_amd.vec_env import WAVE_PER_ENV, LEG_TIER_ON
    hm = T.ramp(nrow=64, ncol=2001, size_x=10.0, slope=0.1, x0=0.5)
    shifts = [(-1.0, 0.0), (0.4, 0.0), (0.45, 0.002), (1.0, 0.05), (2.0, 0.15), (3.3, 0.28)]
    os_ = _oracles(oracle_mod, hm, shifts)
    n = len(os_)
    env = vec(n, kind="stand", control_mode=mode, n_substeps=1, auto_reset=False, flags=_tier_flags(wave_per_env))
    env.set_heightfield(hm, 10.0, 10.0)
    worst, worst_u, sloped = 0.0, 0.0, 0
    for t in range(200):
        acts = np.array([(_py_standing_osc if mode == "OSC" else _py_standing_jac)(o, 0.9 + 0.03 * np.sin(0.02 * t), 0.0) for o in os_])
        env.set_full_state_host(np.array([state_vec(*o.state(), o.warmstart(), ctrl=o.ctrl()) for o in os_]))
        env.substep_host(mode, acts, 1)
        sg = env.get_full_state_host()
        for i, o in enumerate(os_):
            (o.step_osc if mode == "OSC" else o.step_jacobian)(acts[i])
            q1, v1 = o.state()
            worst = max(worst, np.abs(sg[i, :13] - q1).max(), np.abs(sg[i, 13:26] - v1).max() / (1 + np.abs(v1).max()))
            worst_u = max(worst_u, np.abs(sg[i, 78:84] - o.ctrl()).max())
            if o.ncon and np.abs(o.contacts()["frame"][:, 0]).max() > 0.05:
                sloped += 1
    assert worst < 1e-8 and worst_u < 1e-6, (worst, worst_u)
    assert sloped > 200  # contacts with a tilted frame were really exercised
    env.close()


@pytest.mark.parametrize("mode", ["OSC", "Jacobian"])
def test_standing_controllers_hold_the_robot_on_the_slope_closed_loop(vec, oracle_mod, mode):
    """Closed loop on the ramp: the scripted standing controller (device side, CassieVecStandingStep) keeps robots standing on the
    flat part and on the 10 % slope for 400 substeps, and follows the oracle running the same law (free-running, 1e-5)."""
    from cassierl_amd.vec_env import CONTROL_MODES
    import torch
    hm = T.ramp(nrow=64, ncol=2001, size_x=10.0, slope=0.1, x0=0.5)
    shifts = [(-1.0, 0.0), (1.0, 0.05), (2.0, 0.15)]
    os_ = _oracles(oracle_mod, hm, shifts)
    n = len(os_)
    env = vec(n, kind="stand", control_mode=mode, n_substeps=1, auto_reset=False)
    env.set_heightfield(hm, 10.0, 10.0)
    env.set_full_state_host(np.array([state_vec(*o.state(), o.warmstart(), ctrl=o.ctrl()) for o in os_]))
    zp = torch.full((n,), 0.9, dtype=torch.float64, device="cuda")
    zv = torch.zeros(n, dtype=torch.float64, device="cuda")
    worst = 0.0
    for t in range(40):
        env._chk(env.L.CassieVecStandingStep(env.h, CONTROL_MODES[mode], zp.data_ptr(), zv.data_ptr(), 10))
        for o in os_:
            for _ in range(10):
                a = (_py_standing_osc if mode == "OSC" else _py_standing_jac)(o, 0.9, 0.0)
                (o.step_osc if mode == "OSC" else o.step_jacobian)(a)
        sg = env.get_full_state_host()
        for i, o in enumerate(os_):
            worst = max(worst, rel_err(sg[i], *o.state()))
    assert worst < 1e-5, worst
    q = env.get_full_state_host()[:, :13]
    assert (q[:, 1] > 0.7).all()   # nobody fell
    env.close()


def test_controllers_on_terrain_and_field_removal(vec):
    """A zero field under the OSC controller is the flat floor; the field can be set and removed between steps."""
    a = vec(4, kind="stand", control_mode="OSC", n_substeps=1, auto_reset=False)
    b = vec(4, kind="stand", control_mode="OSC", n_substeps=1, auto_reset=False)
    b.set_heightfield(np.zeros((4, 4)), 10.0, 10.0)
    act = np.tile([0.5, -0.5, 0.0, 0.2, 0.0, 0.2, 0.1], (4, 1))
    for _ in range(5):
        oa, ra, da = a.step_host(act)
        ob, rb, db = b.step_host(act)
        assert np.abs(oa - ob).max() < 1e-9 and np.abs(ra - rb).max() < 1e-10
    b.set_heightfield(None)
    b.set_full_state_host(a.get_full_state_host())
    assert np.array_equal(a.step_host(act)[0], b.step_host(act)[0])
    a.close(); b.close()


def test_leg_tier_on_rolling_relief_at_size(vec):
    """The two-lanes-per-environment kernel's height-field instantiation at size: 16 384 robots on a 3 cm rolling relief, random PD
    targets, auto-reset, against the 4-environments-per-wavefront kernel from the same actions; no state ever leaves the finite
    range (r03: an unused contact slot's terrain normal -- whatever LDS held -- leaked a NaN into the force sum at exactly this
    size, which the small parity tests never saw)."""
    import torch
    from cassierl_amd import rollout as R
    from cassierl_amd.vec_env import LEG_TIER_ON, LEG_TIER_OFF, action_space
    n = 16384
    xs = np.linspace(-10.0, 10.0, 2001)
    relief = np.tile(0.015 * (1.0 - np.cos(2.0 * np.pi * xs / 1.5)), (64, 1))
    box = action_space("PD")
    ids = torch.arange(n, device="cuda")
    envs = []
    for fl in (LEG_TIER_ON, LEG_TIER_OFF):
        e = vec(n, kind="stand", control_mode="PD", n_substeps=10, auto_reset=True, flags=fl)
        e.set_heightfield(relief, 10.0, 10.0)
        envs.append((e, e.alloc()))
        e.reset(envs[-1][1])
    worst = 0.0
    for t in range(25):
        a = R.random_actions(2, ids, t, box.low, box.high)
        sa = envs[0][0].get_full_state_host()
        envs[1][0].set_full_state_host(sa)            # teacher-forced per Env.step (PD is chaotic)
        for e, out in envs:
            e.step(a, out)
        s0, s1 = envs[0][0].get_full_state_host(), envs[1][0].get_full_state_host()
        assert np.isfinite(s0).all()
        worst = max(worst, float((np.abs(s0[:, :26] - s1[:, :26]).max(axis=1) / (1.0 + np.abs(s1[:, :26]).max(axis=1))).max()))
    assert worst < 1e-6, worst
    for e, _ in envs:
        assert e.counters()["nonfinite_resets"] == 0
        e.close()

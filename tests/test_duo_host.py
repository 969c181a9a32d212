"""The 64-environments-per-wavefront form of the headline kernel (cassierl_amd/csrc/cassie_duo_core.h: lane-per-leg set-up of two
groups of environments, one joint PGS sweep with a lane per environment) against the two-lanes-per-environment form
(cassie_leg_core.h), both compiled for the CPU by oracle/leg_host/leg_host.cpp.  The claim is BIT-IDENTITY: every state record,
observation, reward, done flag and hand-over count.  (The pair form itself is checked against the oracle in tests/test_leg_host.py.)"""
import numpy as np
import pytest

from conftest import state_vec
from leg_host import LegHostEnv

PD_LO, PD_HI = np.radians([-50, -164, -140] * 2), np.radians([80, -37, -30] * 2)
TQ = np.array([12.0, 12.0, 0.9] * 2)


def _pair_and_duo(n, **kw):
    return LegHostEnv(n, **kw), LegHostEnv(n, duo=True, **kw)


def _same(a, b, what):
    assert np.array_equal(a, b, equal_nan=True), (what, np.argwhere(a != b)[:5])


@pytest.mark.parametrize("n", [1, 2, 5])
def test_walk_env_pd_stream_with_resets_is_bit_identical(oracle_mod, traj, n):
    """The bench's regime: walk env, PD, random targets, every step ends the episode (quirk Q3) -- set-up, joint sweep, reset pass."""
    rng = np.random.default_rng(3)
    pair, duo = _pair_and_duo(n, kind="walk", control_mode="PD", n_substeps=10, auto_reset=True)
    o = oracle_mod.Oracle()
    q, v = o.state()
    s0 = np.tile(state_vec(q, v, o.warmstart(), qstate=q), (n, 1))
    for e in (pair, duo):
        e.set_trajectory(traj["time"], traj["qpos"])
        e.set_full_state_host(s0)
    for t in range(12):
        a = rng.uniform(PD_LO, PD_HI, (n, 6))
        rp, rd = pair.step_host(a), duo.step_host(a)
        for x, y, w in zip(rp, rd, ("obs", "reward", "done")):
            _same(x, y, (t, w))
        _same(pair.get_full_state_host(), duo.get_full_state_host(), (t, "state"))
        _same(pair.pending, duo.pending, (t, "pending"))


@pytest.mark.parametrize("mode", ["Torque", "PD"])
def test_collapsing_robots_without_reset_are_bit_identical(oracle_mod, mode):
    """Stand env, no reset, robots driven into their joint limits and to the ground (the scenario of
    test_leg_host.py::test_limits_contacts_and_the_capacity_hand_over): the eight-row pair sweep inside the duo kernel, groups of one
    call that differ in which sweep they take, a partly empty group (five environments), and environments that leave the tier (more
    than eight rows on a leg: pending > 0, state untouched from that substep on).
    One field group may differ, and only for an environment that is handed over mid-step: the setState snapshot (record fields
    39..64).  The pair form writes the snapshot of the last substep it carried out; the duo form only ever writes the snapshot of a
    step's LAST substep -- for a handed-over environment the lower tier that finishes the step does (every tier takes its own
    snapshot on every substep it carries out), so the record after the tiers is the same (GPU: tests/test_gpu_duo.py)."""
    rng = np.random.default_rng(5)
    n = 5
    pair, duo = _pair_and_duo(n, kind="stand", control_mode=mode, n_substeps=10, auto_reset=False)
    o = oracle_mod.Oracle()
    q, v = o.state()
    s0 = np.tile(state_vec(q, v, o.warmstart(), qstate=q), (n, 1))
    s0[:, 2] += rng.uniform(-0.05, 0.05, n)          # different pitch per robot: they go down at different times
    for e in (pair, duo):
        e.set_full_state_host(s0)
    push = np.array([12.2, -12.2, 0.9, 12.2, -12.2, 0.9]) if mode == "Torque" else np.array([PD_HI[0], PD_LO[1], PD_HI[2]] * 2)
    left = 0
    for t in range(110):
        if mode == "Torque":
            a = np.tile(push, (n, 1)) * (1 + 0.2 * rng.uniform(-1, 1, (n, 6))) if t < 80 else np.zeros((n, 6))
        else:
            a = np.clip(np.tile(push, (n, 1)) + 0.3 * rng.uniform(-1, 1, (n, 6)), PD_LO, PD_HI) if t < 80 else rng.uniform(PD_LO, PD_HI, (n, 6))
        rp, rd = pair.step_host(a), duo.step_host(a)
        _same(pair.pending, duo.pending, (t, "pending"))
        stay = pair.pending == 0
        for x, y, w in zip(rp, rd, ("obs", "reward", "done")):
            _same(x[stay], y[stay], (t, w))
        sp, sd = pair.get_full_state_host(), duo.get_full_state_host()
        _same(sp[stay], sd[stay], (t, "state"))
        keep = np.r_[0:39, 65:88]
        _same(sp[~stay][:, keep], sd[~stay][:, keep], (t, "state of the handed-over environments, snapshot aside"))
        left += int((~stay).sum())
        if (~stay).any():   # on the GPU the lower tiers finish them; here: back on their feet
            sp[~stay] = s0[~stay]
            pair.set_full_state_host(sp); duo.set_full_state_host(sp)
    assert left > 0 or mode == "PD", "the torque run must include hand-overs (the PD run drives the joints into their limits: eight-row sweeps)"


def test_record_command_mode_is_bit_identical(oracle_mod):
    """MODE 2 (motor commands from the state record: the physics substep behind the OSC / Jacobian controller kernels)."""
    rng = np.random.default_rng(7)
    n = 3
    pair, duo = _pair_and_duo(n, kind="stand", control_mode="Record", n_substeps=1, auto_reset=False)
    o = oracle_mod.Oracle()
    q, v = o.state()
    s0 = np.tile(state_vec(q, v, o.warmstart(), qstate=q), (n, 1))
    for e in (pair, duo):
        e.set_full_state_host(s0)
    for t in range(40):
        s = pair.get_full_state_host()
        s[:, 78:84] = rng.uniform(-1, 1, (n, 6)) * TQ * 0.3
        pair.set_full_state_host(s); duo.set_full_state_host(s)
        pair.substep_host("Record", None, 1); duo.substep_host("Record", None, 1)
        _same(pair.get_full_state_host(), duo.get_full_state_host(), (t, "state"))


@pytest.mark.parametrize("mode", ["Torque", "PD"])
def test_height_field_is_bit_identical(oracle_mod, mode):
    """The height-field instantiation (terrain collision stage, contact frames along the local normal: cassie_leg_core.h with HF = true)
    through both forms: robots on the flat part of a ramp, across its kink and on the 10 % slope, stand env with resets."""
    from cassierl_amd import terrain as T
    hm = T.ramp(nrow=64, ncol=2001, size_x=10.0, slope=0.1, x0=0.5)
    rng = np.random.default_rng(21)
    n = 5
    pair, duo = _pair_and_duo(n, kind="stand", control_mode=mode, n_substeps=10, auto_reset=True)
    o = oracle_mod.Oracle()
    q, v = o.state()
    s0 = np.tile(state_vec(q, v, np.zeros(13), qstate=q), (n, 1))
    s0[:, 0] += np.array([-1.0, 0.45, 1.0, 3.3, 2.0]); s0[:, 1] += np.array([0.0, 0.002, 0.05, 0.28, 0.15])
    for e in (pair, duo):
        e.set_heightfield(hm, 10.0, 10.0)
        e.set_full_state_host(s0)
    for t in range(40):
        a = rng.uniform(-1, 1, (n, 6)) * TQ * 0.4 if mode == "Torque" else rng.uniform(PD_LO, PD_HI, (n, 6))
        rp, rd = pair.step_host(a), duo.step_host(a)
        _same(pair.pending, duo.pending, (t, "pending"))
        stay = pair.pending == 0
        for x, y, w in zip(rp, rd, ("obs", "reward", "done")):
            _same(x[stay], y[stay], (t, w))
        sp, sd = pair.get_full_state_host(), duo.get_full_state_host()
        _same(sp[stay], sd[stay], (t, "state"))
        assert pair.nonfinite == duo.nonfinite == 0
        if (~stay).any():
            sp[~stay] = s0[~stay]
            pair.set_full_state_host(sp); duo.set_full_state_host(sp)

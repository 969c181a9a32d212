"""Full-size runs of every BASELINE.json config on the HIP path: sampled environments against the oracle where the dynamics
allow it, size-independent properties (determinism across grid positions, finiteness, bands, counters) everywhere.  -m gpu only.
Each test appends one JSON line to gpurun_out/fullsize.jsonl (copied into profiles/ per round)."""
import json
import os
import time

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

PD_LO, PD_HI = np.radians([-50, -164, -140] * 2), np.radians([80, -37, -30] * 2)
TQ = np.array([12.0, 12.0, 0.9] * 2)
FIXQ = 2


def record(**kw):
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "fullsize.jsonl"), "a") as f:
        f.write(json.dumps(kw) + "\n")


@pytest.fixture(scope="module")
def vec():
    from cassierl_amd.vec_env import CassieVecEnv
    return CassieVecEnv


# ------------------------------------------------------------------------------------------------ configs[1]
@pytest.mark.parametrize("n", [4096, 8192])   # 8192: the default size rule makes env_step_leg_kernel the first tier (>= 6144 envs)
@pytest.mark.parametrize("flags,steps,tol", [(0, 20, 1e-8), (FIXQ, 8, 1e-6)])
def test_configs1_4096_walk_pd_sampled_envs_follow_the_oracle(vec, oracle_mod, traj, flags, steps, tol, n):
    """configs[1]: 4096 envs, walk env, PD mode, random policy.  64 sampled environments are replayed by the oracle env with
    the same actions.  flags=0 is the reference-faithful regime (quirk Q3: every step terminates and auto-resets, so the
    comparison holds for any number of steps); with CASSIE_FIX_STALE_QSTATE the robots run free, so the window is the
    first 80 substeps (PD is chaotic beyond ~100, see test_gpu_parity.py)."""
    import torch
    from cassierl_amd import rollout as R
    env = vec(n, kind="walk", control_mode="PD", n_substeps=10, flags=flags, auto_reset=True)
    env.set_trajectory(traj["time"], traj["qpos"])
    sample = np.unique(np.concatenate([np.arange(0, n, 67), [1, 2, 3, n - 1]]))[:64]
    oes = [oracle_mod.OracleEnv("walk", "PD", flags=flags, traj=traj) for _ in sample]
    out = env.alloc()
    obs0 = env.reset(out).cpu().numpy()
    for k, e in enumerate(oes):
        assert np.abs(e.reset() - obs0[sample[k]]).max() < 1e-12
    ids = torch.arange(n, device="cuda")
    worst, ndone = 0.0, 0
    for t in range(steps):
        a = R.random_actions(1, ids, t, PD_LO, PD_HI)
        o, r, d = env.step(a, out)
        o, r, d, ah = o.cpu().numpy(), r.cpu().numpy(), d.cpu().numpy(), a.cpu().numpy()
        ndone += int(d.sum())
        for k, e in enumerate(oes):
            i = sample[k]
            oo, rr, dd = e.step(ah[i])
            assert bool(d[i]) == dd, (t, i)
            if dd:
                oo = e.reset()  # the kernel returns the reset observation of a terminated env (auto_reset)
            worst = max(worst, abs(r[i] - rr), np.abs(o[i] - oo).max())
    assert worst < tol, worst
    c = env.counters()
    assert c["nonfinite_resets"] == 0
    record(test="configs1_%d_walk_pd" % n, first_tier="leg" if n >= 6144 else "g16", flags=flags, steps=steps, sampled=len(sample), worst=worst, episodes=ndone, **c)
    env.close()


# ------------------------------------------------------------------------------------------------ configs[2]
def test_configs2_65536_osc_standing_controller_properties(vec):
    """configs[2]: 65 536 envs with the OSC controller (QP) in every substep -- standing_controller_osc(zpos=0.9).  17 distinct
    initial perturbations are tiled over the batch (17 is coprime to the 4 envs per wavefront and to the grid), so every
    state is integrated by many different (workgroup, 16-lane row) positions: all replicas must agree BIT FOR BIT, stay finite,
    keep the pelvis inside the standing band and never need the slow path."""
    import torch
    n, nsub = 65536, 100
    env = vec(n, kind="stand", control_mode="OSC", n_substeps=1, auto_reset=False)
    rng = np.random.default_rng(5)
    s = env.get_full_state_host()
    pert = rng.uniform(-0.01, 0.01, (17, 13))
    pert[0] = 0
    k = np.arange(n) % 17
    s[:, 0:13] += pert[k]
    s[:, 39:52] += pert[k]
    env.set_full_state_host(s)
    zp = torch.full((n,), 0.9, dtype=torch.float64, device="cuda")
    zv = torch.zeros(n, dtype=torch.float64, device="cuda")
    env.reset_counters()
    t0 = time.perf_counter()
    env._chk(env.L.CassieVecStandingStep(env.h, 2, zp.data_ptr(), zv.data_ptr(), nsub))
    env.synchronize()
    dt = time.perf_counter() - t0
    s1 = env.get_full_state_host()
    assert np.isfinite(s1).all()
    for j in range(17):
        grp = s1[k == j]
        assert (grp[:, :39] == grp[0, :39]).all(), j  # qpos, qvel, warm start: bit-identical across grid positions
    z = s1[:, 1]
    assert z.min() > 0.8 and z.max() < 1.0, (z.min(), z.max())
    assert np.abs(s1[:, 2]).max() < 0.2  # pitch
    c = env.counters()
    assert c["nonfinite_resets"] == 0 and c["k1_substeps"] == 0
    record(test="configs2_65536_osc", n_sub=nsub, seconds=dt, controller_substeps_per_s=n * nsub / dt, zmin=float(z.min()), zmax=float(z.max()), **c)
    env.close()


def test_configs2_65536_stand_env_osc_step_replicas(vec):
    """cassie_stand2d Env.step with OSC actions at 65 536 envs: replicas of the same (state, action) agree bit for bit across the
    grid, outputs finite, reward consistent with its definition (cassie_stand2d.py:118-125)."""
    import torch
    n = 65536
    env = vec(n, kind="stand", control_mode="OSC", n_substeps=10, auto_reset=True)
    rng = np.random.default_rng(6)
    base_act = rng.uniform(-1, 1, (13, 7)) * np.array([2, 2, 1, 1, 1, 1, 2.0])
    base_act[:, 3], base_act[:, 5] = np.abs(base_act[:, 3]), np.abs(base_act[:, 5])
    k = np.arange(n) % 13
    a = torch.as_tensor(base_act[k], device="cuda")
    out = env.alloc()
    env.reset(out)
    for _ in range(5):
        o, r, d = env.step(a, out)
    o, r, d = o.cpu().numpy(), r.cpu().numpy(), d.cpu().numpy()
    assert np.isfinite(o).all() and np.isfinite(r).all()
    for j in range(13):
        m = k == j
        assert (o[m] == o[m][0]).all() and (r[m] == r[m][0]).all() and (d[m] == d[m][0]).all(), j
    z = o[:, 0]
    keep = d == 0
    expect = 1 - 2 * (0.9 - z) ** 2 - 2 * ((o[:, 5] + o[:, 11]) / 2) ** 2 - 0.001 * (base_act[k] ** 2).sum(1)
    assert np.abs(r[keep] - expect[keep]).max() < 1e-12
    record(test="configs2_65536_stand_osc_step", done=int(d.sum()), **env.counters())
    env.close()


# ------------------------------------------------------------------------------------------------ 65 536 envs, PD / torque
@pytest.mark.parametrize("tier", ["leg", "g16"])
@pytest.mark.parametrize("kind,mode,flags,auto_reset", [("stand", "PD", 0, True), ("stand", "Torque", 0, True), ("stand", "Torque", 0, False)])
def test_65536_envs_moving_robots_fast_path_vs_general_kernel(vec, traj, kind, mode, flags, auto_reset, tier):
    """The regime the headline does NOT see: robots that move, hit joint limits and fall (more than 8 rows on a leg / 16 rows).
    After 60 free-running Env.steps the whole batch is stepped once more by the packed kernels (first tier = two lanes per
    environment, the default at this size, or four environments per wavefront) and, from the same states, by the
    wave-per-environment kernel: results agree to 1e-10 (torque), episodes keep terminating, nothing is non-finite, and the share
    of env-substeps that left the fast path is recorded.  PD mode: the toe's explicit damper amplifies a rounding ~1e6-fold
    within one Env.step; the g16 tier shares its formulation with the general kernel (they differ by ~1e-15 per substep: bar
    1e-8), the leg tier factorises the mass matrix by blocks (~1e-13 per substep against either of them AND against the oracle,
    tests/test_leg_host.py: bar 1e-6)."""
    import torch
    from cassierl_amd import rollout as R
    from cassierl_amd.vec_env import WAVE_PER_ENV, LEG_TIER_ON, LEG_TIER_OFF
    n = 65536
    flags = flags | (LEG_TIER_ON if tier == "leg" else LEG_TIER_OFF)
    a_env = vec(n, kind=kind, control_mode=mode, n_substeps=10, flags=flags, auto_reset=auto_reset)
    b_env = vec(n, kind=kind, control_mode=mode, n_substeps=10, flags=flags | WAVE_PER_ENV, auto_reset=auto_reset)
    for e in (a_env, b_env):
        e.set_trajectory(traj["time"], traj["qpos"])
    lo, hi = (PD_LO, PD_HI) if mode == "PD" else (-TQ, TQ)
    ids = torch.arange(n, device="cuda")
    out, outb = a_env.alloc(), b_env.alloc()
    a_env.reset(out)
    a_env.reset_counters()
    ndone = 0
    for t in range(60 if auto_reset else 150):  # without auto-reset: long enough for every robot to end up on the ground
        _, _, d = a_env.step(R.random_actions(9, ids, t, lo, hi), out)
        ndone += int(d.sum())
    c = a_env.counters()
    b_env.set_full_state_host(a_env.get_full_state_host())
    act = R.random_actions(9, ids, 60, lo, hi)
    oa, ra, da = (x.cpu().numpy() for x in a_env.step(act, out))
    ob, rb, db = (x.cpu().numpy() for x in b_env.step(act, outb))
    sa, sb = a_env.get_full_state_host(), b_env.get_full_state_host()
    assert np.isfinite(sa).all() and np.isfinite(oa).all()
    err = np.abs(sa[:, :26] - sb[:, :26]).max(axis=1) / (1.0 + np.abs(sb[:, :26]).max(axis=1))
    tol = 1e-10 if mode == "Torque" else (1e-6 if tier == "leg" else 1e-8)
    assert err.max() < tol and np.abs(oa - ob).max() < 10 * tol and np.abs(ra - rb).max() < tol
    assert (da != db).sum() == 0
    assert ndone > 1000 or mode == "PD"  # torque: robots fall and episodes end; PD targets hold the robots up longer
    assert np.abs(sa[:, 13:26]).max() > 1.0
    assert c["nonfinite_resets"] == 0
    if not auto_reset:
        assert c["cleanup_substeps"] > 0 and np.median(sa[:, 1]) < 0.5  # the robots are down; some needed more than 16 constraint rows
    record(test="65536_moving_robots", tier=tier, kind=kind, mode=mode, flags=flags, auto_reset=auto_reset, episodes=ndone, worst=float(err.max()), **c)
    a_env.close(); b_env.close()


# ------------------------------------------------------------------------------------------------ configs[4]
def test_configs4_16384_cassie3d_properties():
    """configs[4]: 16 384 Cassie3d envs, random torques, robots collapse onto the floor.  No env may be dropped (every constraint
    count is handled), replicas tiled with period 11 agree bit for bit, states stay finite."""
    import torch
    from cassierl_amd import rollout as R
    from cassierl_amd.vec_env3d import Cassie3dVec, CTRL_RANGE
    n = 16384
    e3 = Cassie3dVec(n)
    ids = torch.arange(n, device="cuda") % 11
    t0 = time.perf_counter()
    for t in range(120):
        e3.step(R.random_actions(5, ids, t, -CTRL_RANGE, CTRL_RANGE), 10)
    e3.synchronize()
    dt = time.perf_counter() - t0
    s = e3.get_state_host()
    assert np.isfinite(s).all()
    assert (s[:, 74] == 0).all(), "environments frozen by a constraint-row overflow: %d" % int((s[:, 74] != 0).sum())
    k = np.arange(n) % 11
    for j in range(11):
        grp = s[k == j]
        assert (grp[:, :61] == grp[0, :61]).all(), j
    assert s[:, 2].min() < 0.5  # robots did fall (pelvis height)
    row = dict(test="configs4_16384_cassie3d", steps=120, env_steps_per_s=n * 120 / dt, max_rows=float(s[:, 73].max()))
    if hasattr(e3, "counters"):
        row.update(e3.counters())
    record(**row)
    e3.close()


# ------------------------------------------------------------------------------------------------ configs[3], all 524 288 envs on one device
def test_configs3_524288_envs_single_device_properties(vec, traj):
    """The size of configs[3] (8 x 65 536) resident on ONE device: 131 072 workgroups of the packed kernel, 369 MB of state
    records, record offsets beyond 2^31 bytes (size_t indexing), the hand-over pass over 8192 workgroups.  Robots move and fall
    (stand env, random torques keyed by env id mod 97, auto-reset): replicas spread over the whole grid agree bit for bit, every
    state stays finite, the counters add up, and the first 4096 envs equal a 4096-env batch stepped on its own."""
    import torch
    from cassierl_amd import rollout as R
    n, T = 524288, 12
    from cassierl_amd.vec_env import LEG_TIER_ON
    env = vec(n, kind="stand", control_mode="Torque", n_substeps=10, auto_reset=True, flags=LEG_TIER_ON)
    small = vec(4096, kind="stand", control_mode="Torque", n_substeps=10, auto_reset=True, flags=LEG_TIER_ON)   # same first tier: same bits
    out, outs = env.alloc(), small.alloc()
    env.reset(out); small.reset(outs)
    ids = torch.arange(n, device="cuda") % 97
    lo, hi = env.action_space.low * 1.5, env.action_space.high * 1.5
    ndone = torch.zeros((), dtype=torch.int64, device="cuda")
    for t in range(T):
        a = R.random_actions(11, ids, t, lo, hi)
        o, r, d = env.step(a, out)
        ndone += d.sum()
        small.step(a[:4096].contiguous(), outs)
    env.synchronize()
    s = env.get_full_state_host()
    assert s.shape == (n, 88) and np.isfinite(s).all()
    k = np.arange(n) % 97
    for j in (0, 13, 96):
        grp = s[k == j]
        assert (grp[:, :39] == grp[0, :39]).all(), j
    assert np.array_equal(s[:4096, :39], small.get_full_state_host()[:, :39])
    assert np.array_equal(out["obs"][:4096].cpu().numpy(), outs["obs"].cpu().numpy())
    c = env.counters()
    assert c["substeps"] == n * T * 10 and c["nonfinite_resets"] == 0
    assert s[:, 1].min() < 0.93 and np.abs(s[:, 13:26]).max() > 1.0   # robots are moving
    record(test="configs3_524288_single_device", steps=T, episodes=int(ndone), **c)
    env.close(); small.close()


# ------------------------------------------------------------------------------------------------ configs[3], one rank's share
def test_configs3_one_rank_trpo_iteration_65536_envs():
    """configs[3] is 8 x 65 536 envs under TRPO; this is one rank's share on one GPU: a full TRPO iteration (policy rollout of
    65 536 envs x 8 Env.steps = 524 288 samples, baseline fit, CG + line search) and a second one from the updated policy."""
    import torch
    from cassierl_amd.trajectory import default_gait
    from cassierl_amd.trpo import make_cassie_trpo
    n = 65536
    algo = make_cassie_trpo(n, kind="stand", control_mode="Torque", device=0, trajectory=default_gait(), seed=1, batch_size=n * 8)
    t0 = time.perf_counter()
    st = [algo.train_iteration() for _ in range(2)]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for s in st:
        assert s["env_steps"] == n * 8 and s["gathered"] == n
        assert np.isfinite([s["loss_before"], s["loss_after"], s["avg_reward"]]).all()
        assert 0 <= s["kl"] <= 0.005 + 1e-9 and s["loss_after"] <= s["loss_before"]
    assert st[0]["backtracks"] >= 0
    c = algo.env.counters()
    assert c["nonfinite_resets"] == 0
    record(test="configs3_one_rank_trpo_65536", seconds_two_iterations=dt, env_steps_per_s=2 * n * 8 / dt,
           kl=[s["kl"] for s in st], avg_reward=[s["avg_reward"] for s in st], **c)
    algo.env.close()


# ------------------------------------------------------------------------------------------------ stress / soak (were scripts)
@pytest.mark.parametrize("tier", ["g16", "leg"])
@pytest.mark.parametrize("kind,mode", [("walk", "PD"), ("stand", "Torque"), ("stand", "OSC"), ("stand", "Jacobian")])
def test_stress_packed_kernels_agree_with_wave_per_env(vec, traj, kind, mode, tier):
    """4099 envs (not a multiple of 4 or 32), 25 teacher-forced Env.steps: each packed first tier (4 environments per wavefront;
    two lanes per environment) with its hand-over passes against the wave-per-environment kernels, every control mode, random
    actions over (and beyond) the action box.  Bar 1e-9; the OSC closed loop on the leg tier gets 5e-7: that tier factorises the
    mass matrix differently (block elimination), so its per-substep rounding differs from the other kernels' by ~1e-13 instead of
    ~1e-15, and ten substeps of QP controller + physics amplify that (against the ORACLE both tiers sit inside the same bars:
    tests/test_gpu_ctrl.py passes with either)."""
    from cassierl_amd.vec_env import WAVE_PER_ENV, LEG_TIER_ON, LEG_TIER_OFF
    n, steps = 4099, 25
    rng = np.random.default_rng(0)
    a = vec(n, kind=kind, control_mode=mode, n_substeps=10, auto_reset=True, flags=LEG_TIER_ON if tier == "leg" else LEG_TIER_OFF)
    b = vec(n, kind=kind, control_mode=mode, n_substeps=10, auto_reset=True, flags=WAVE_PER_ENV)
    for e in (a, b):
        e.set_trajectory(traj["time"], traj["qpos"])
    a.reset_host(); b.reset_host()
    lo, hi = a.action_space.low, a.action_space.high
    if mode == "OSC":
        lo, hi = np.array([-6, -6, -2, 0, -2, 0, -6.0]), np.array([6, 6, 2, 2, 2, 2, 6.0])
    if mode == "Jacobian":
        lo, hi = np.array([-60.0, 0.0, -25.0] * 2), np.array([60.0, 250.0, 25.0] * 2)
    worst, ndone, bad = 0.0, 0, 0
    for t in range(steps):
        acts = rng.uniform(lo, hi, (n, a.adim))
        b.set_full_state_host(a.get_full_state_host())
        oa, ra, da = a.step_host(acts)
        ob, rb, db = b.step_host(acts)
        sa, sb = a.get_full_state_host(), b.get_full_state_host()
        err = np.abs(sa[:, :26] - sb[:, :26]).max(axis=1) / (1.0 + np.abs(sb[:, :26]).max(axis=1))
        worst = max(worst, float(err.max()), float(np.abs(oa - ob).max()), float(np.abs(ra - rb).max()))
        bad += int((da != db).sum()) + int((~np.isfinite(sa)).any())
        ndone += int(da.sum())
    assert worst < (5e-7 if (tier == "leg" and mode == "OSC") else 1e-9) and bad == 0, (worst, bad)
    record(test="stress_agree", tier=tier, kind=kind, mode=mode, n=n, steps=steps, worst=worst, episodes=ndone, **a.counters())
    a.close(); b.close()


@pytest.mark.parametrize("kind,mode,n,steps", [("walk", "PD", 65536, 120), ("stand", "Torque", 65536, 120), ("stand", "OSC", 16384, 60),
                                               ("stand", "Jacobian", 16384, 60)])
def test_soak_long_random_rollouts_stay_finite(vec, traj, kind, mode, n, steps):
    import torch
    from cassierl_amd import rollout as R
    env = vec(n, kind=kind, control_mode=mode, n_substeps=10, auto_reset=True)
    env.set_trajectory(traj["time"], traj["qpos"])
    out = env.alloc()
    env.reset(out)
    ids = torch.arange(n, device="cuda")
    lo, hi = env.action_space.low, env.action_space.high
    if mode == "Jacobian":
        lo, hi = np.array([-80.0, -50.0, -40.0] * 2), np.array([80.0, 400.0, 40.0] * 2)
    t0 = time.perf_counter()
    ndone = torch.zeros((), dtype=torch.int64, device="cuda")
    for t in range(steps):
        o, r, done = env.step(R.random_actions(3, ids, t, lo, hi), out)
        ndone += done.sum()
    env.synchronize()
    dt = time.perf_counter() - t0
    q, v = env.get_state_host()
    assert np.isfinite(q).all() and np.isfinite(v).all() and bool(torch.isfinite(o).all()) and bool(torch.isfinite(r).all())
    c = env.counters()
    record(test="soak", kind=kind, mode=mode, n=n, steps=steps, seconds=dt, env_steps_per_s=n * steps / dt, episodes=int(ndone), vmax=float(np.abs(v).max()), **c)
    env.close()


# ------------------------------------------------------------------------------------------------ failure guard
@pytest.mark.parametrize("mode,wave_per_env", [("PD", False), ("PD", True), ("OSC", False), ("OSC", True)])
def test_failure_guard_terminates_and_resets_poisoned_envs(vec, traj, mode, wave_per_env):
    """SURVEY.md section 5 'failure detection': an env whose state goes NaN / diverges is force-terminated (done = 1, reward 0),
    counted, and -- with auto_reset -- comes back clean; its neighbours in the same wavefront are untouched."""
    from cassierl_amd.vec_env import WAVE_PER_ENV
    n = 12
    kind = "walk" if mode == "PD" else "stand"
    env = vec(n, kind=kind, control_mode=mode, n_substeps=10, auto_reset=True, flags=WAVE_PER_ENV if wave_per_env else 0)
    ref = vec(n, kind=kind, control_mode=mode, n_substeps=10, auto_reset=True, flags=WAVE_PER_ENV if wave_per_env else 0)
    for e in (env, ref):
        e.set_trajectory(traj["time"], traj["qpos"])
        e.reset_host()
    s = env.get_full_state_host()
    s[1, 4] = np.nan          # qpos
    s[6, 13 + 2] = np.inf     # qvel
    s[9, 26 + 3] = np.nan     # warm start only: poisons qacc -> the state after one substep
    s[10, 3] = 5e10           # diverged, still finite
    env.set_full_state_host(s)
    rng = np.random.default_rng(1)
    a = rng.uniform(env.action_space.low, env.action_space.high, (n, env.adim)) * (0.1 if mode == "OSC" else 1.0)
    o, r, d = env.step_host(a)
    o2, r2, d2 = ref.step_host(a)
    badset = [1, 6, 9, 10]
    good = [i for i in range(n) if i not in badset]
    assert d[badset].all() and (r[badset] == 0).all()
    assert np.isfinite(o).all() and np.isfinite(r).all()
    assert np.array_equal(o[good], o2[good]) and np.array_equal(r[good], r2[good]) and np.array_equal(d[good], d2[good])
    assert env.counters()["nonfinite_resets"] == 4
    s1 = env.get_full_state_host()
    assert np.isfinite(s1).all()
    # the poisoned envs restart from the reset pose exactly like an env that terminated normally
    o3, r3, d3 = env.step_host(a)
    assert np.isfinite(o3).all() and np.isfinite(r3).all()
    assert env.counters()["nonfinite_resets"] == 4
    env.close(); ref.close()


@pytest.mark.parametrize("tier", ["leg", "duo"])
@pytest.mark.parametrize("mode", ["PD", "Torque"])
def test_failure_guard_in_the_lane_per_leg_tiers(vec, traj, mode, tier):
    """... and in the two-lanes-per-environment and 64-environments-per-wavefront kernels (their row construction leaves out the
    structurally zero Jacobian terms, so a non-finite velocity or warm start no longer reaches every row as 0 x Inf -- it must still trip the
    guard through the state it produces): NaN position, infinite velocity, NaN warm start, diverged-but-finite position; neighbours in the same
    wavefront (both groups of it) untouched, bit for bit."""
    from cassierl_amd.vec_env import LEG_TIER_ON, DUO_TIER_ON, DUO_TIER_OFF
    fl = LEG_TIER_ON | (DUO_TIER_ON if tier == "duo" else DUO_TIER_OFF)
    n = 70
    kind = "walk" if mode == "PD" else "stand"
    env = vec(n, kind=kind, control_mode=mode, n_substeps=10, auto_reset=True, flags=fl)
    ref = vec(n, kind=kind, control_mode=mode, n_substeps=10, auto_reset=True, flags=fl)
    assert env.tier_info()["first_tier"] == tier
    for e in (env, ref):
        e.set_trajectory(traj["time"], traj["qpos"])
        e.reset_host()
    s = env.get_full_state_host()
    s[1, 4] = np.nan          # qpos (left knee)
    s[6, 13 + 2] = np.inf     # qvel (pitch)
    s[9, 26 + 3] = np.nan     # warm start only: poisons qacc -> the state after one substep
    s[40, 13 + 9] = np.nan    # qvel of a right-leg joint, group B of the 64-environments form
    s[41, 3] = 5e10           # diverged, still finite
    env.set_full_state_host(s)
    rng = np.random.default_rng(1)
    a = rng.uniform(env.action_space.low, env.action_space.high, (n, env.adim))
    o, r, d = env.step_host(a)
    o2, r2, d2 = ref.step_host(a)
    badset = [1, 6, 9, 40, 41]
    good = [i for i in range(n) if i not in badset]
    assert d[badset].all() and (r[badset] == 0).all()
    assert np.isfinite(o).all() and np.isfinite(r).all()
    assert np.array_equal(o[good], o2[good]) and np.array_equal(r[good], r2[good]) and np.array_equal(d[good], d2[good])
    assert env.counters()["nonfinite_resets"] == 5
    assert np.isfinite(env.get_full_state_host()).all()
    o3, r3, d3 = env.step_host(a)
    assert np.isfinite(o3).all() and np.isfinite(r3).all() and env.counters()["nonfinite_resets"] == 5
    env.close(); ref.close()


@pytest.mark.parametrize("mode", ["PD", "Torque", "OSC", "Jacobian"])
def test_results_do_not_depend_on_wavefront_neighbours(vec, traj, mode):
    """Four environments share a wavefront and some control flow is decided per wavefront (straight-line vs general PGS sweep,
    'any row still busy' loops).  An environment's result must nevertheless be BIT-identical whatever its neighbours do: here
    every second group of four gets one member with a joint beyond its limit / a different pose, the others are compared with
    an undisturbed batch.  (r02: the two inlined copies of the PGS step were contracted differently by the compiler -- 1e-13.)"""
    n = 32
    kind = "walk" if mode == "PD" else "stand"
    a_env = vec(n, kind=kind, control_mode=mode, n_substeps=10, auto_reset=True)
    b_env = vec(n, kind=kind, control_mode=mode, n_substeps=10, auto_reset=True)
    for e in (a_env, b_env):
        e.set_trajectory(traj["time"], traj["qpos"])
        e.reset_host()
    s = a_env.get_full_state_host()
    odd = [1, 6, 11, 12, 21, 26]
    s[1, 3] = 2.0; s[6, 6] = -0.4; s[11, 8] = 1.6; s[12, 4] = -2.9; s[21, 1] -= 0.05; s[26, 2] = 0.4
    a_env.set_full_state_host(s)
    rng = np.random.default_rng(2)
    lo, hi = a_env.action_space.low, a_env.action_space.high
    if mode == "OSC":
        lo, hi = np.array([-2, -2, -1, 0, -1, 0, -2.0]), np.array([2, 2, 1, 1, 1, 1, 2.0])
    if mode == "Jacobian":
        lo, hi = np.array([-40.0, 50.0, -15.0] * 2), np.array([40.0, 250.0, 15.0] * 2)
    same = [i for i in range(n) if i not in odd]
    for t in range(3):
        a = rng.uniform(lo, hi, (n, a_env.adim))
        oa, ra, da = a_env.step_host(a)
        ob, rb, db = b_env.step_host(a)
        sa, sb = a_env.get_full_state_host(), b_env.get_full_state_host()
        assert np.array_equal(sa[same], sb[same]), (t, np.abs(sa[same] - sb[same]).max())
        assert np.array_equal(oa[same], ob[same]) and np.array_equal(ra[same], rb[same]) and np.array_equal(da[same], db[same])
        # keep the disturbed environments disturbed but everyone else on the common trajectory
        sb2 = sb.copy()
        sb2[odd] = sa[odd]
        a_env.set_full_state_host(sb2)
    a_env.close(); b_env.close()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["Torque", "PD", "OSC", "Jacobian"])
def test_results_do_not_depend_on_which_environments_share_a_wavefront(vec, traj, mode):
    """Stronger form of the neighbour test: the same 512 environments (robots falling, joints at their limits, 0..8 contacts --
    every row layout the packed kernel has) stepped in their natural order and in a random permutation, i.e. with different
    wavefront neighbours for everyone.  States, observations, rewards and dones must be BIT-identical after undoing the
    permutation: layout decisions taken per wavefront (straight-line vs general sweep, pairs moved to row 8) may not change a
    single rounding of an environment's own arithmetic."""
    n, T = 512, 25
    rng = np.random.default_rng(31)
    kind = "stand"
    a_env = vec(n, kind=kind, control_mode=mode, n_substeps=10, auto_reset=True)
    b_env = vec(n, kind=kind, control_mode=mode, n_substeps=10, auto_reset=True)
    lo, hi = (-TQ * 1.5, TQ * 1.5) if mode == "Torque" else (PD_LO, PD_HI)
    if mode == "OSC":
        lo, hi = np.array([-2, -2, -1, 0, -1, 0, -2.0]), np.array([2, 2, 1, 1, 1, 1, 2.0])
    if mode == "Jacobian":
        lo, hi = np.array([-40.0, 50.0, -15.0] * 2), np.array([40.0, 250.0, 15.0] * 2)
    adim = len(lo)
    a_env.reset_host()
    for t in range(12):   # spread the batch over many different configurations first
        a_env.step_host(rng.uniform(lo, hi, (n, adim)))
    s0 = a_env.get_full_state_host()
    perm = rng.permutation(n)
    b_env.reset_host()
    b_env.set_full_state_host(s0[perm])
    saw_limits = 0
    for t in range(T):
        a = rng.uniform(lo, hi, (n, adim))
        oa, ra, da = a_env.step_host(a)
        ob, rb, db = b_env.step_host(a[perm])
        sa, sb = a_env.get_full_state_host(), b_env.get_full_state_host()
        assert np.array_equal(sa[perm], sb), (t, np.abs(sa[perm] - sb).max())
        assert np.array_equal(oa[perm], ob) and np.array_equal(ra[perm], rb) and np.array_equal(da[perm], db)
        saw_limits += int((np.abs(sa[:, 3:13]) > 1.0).sum())
    assert saw_limits > 0 or mode in ("OSC", "Jacobian")
    record(test="permutation_invariance", mode=mode, n=n, steps=T, **a_env.counters())
    a_env.close(); b_env.close()


# ------------------------------------------------------------------------------------------------ the three kernel tiers
@pytest.mark.parametrize("mode", ["Torque", "PD"])
def test_leg_tier_agrees_with_the_other_tiers_on_falling_robots(vec, traj, mode):
    """The two-lanes-per-environment kernel (block-factorised mass matrix, factored A, 8 rows per leg) against the
    4-environments-per-wavefront kernel and the wave-per-environment kernel on 16 384 robots that move, hit joint limits and fall:
    teacher-forced Env.steps from common states; environments beyond a tier's row capacity are handed down, so the comparison
    also covers both hand-over passes.  Bars: 1e-10 (torque); 1e-6 (PD: the reference's PD law is chaotic -- a 1-ulp perturbation
    grows 1e4-fold over ten substeps in the oracle itself -- and this tier's roundings differ from the other kernels' by ~1e-13
    per substep, not ~1e-15, because it factorises the mass matrix by blocks).  The oracle arbitrates: the environments where the
    tiers differ most are replayed on the CPU, and the leg tier must be as close to the oracle as the bar says."""
    import oracle_py as O
    from cassierl_amd.vec_env import WAVE_PER_ENV, LEG_TIER_ON, LEG_TIER_OFF
    n, T = 16384, 60
    rng = np.random.default_rng(41)
    a = vec(n, kind="stand", control_mode=mode, n_substeps=10, auto_reset=False, flags=LEG_TIER_ON)
    b = vec(n, kind="stand", control_mode=mode, n_substeps=10, auto_reset=False, flags=LEG_TIER_OFF)
    c = vec(n, kind="stand", control_mode=mode, n_substeps=10, auto_reset=False, flags=WAVE_PER_ENV)
    for e in (a, b, c):
        e.reset_host()
    lo, hi = (-TQ * 1.5, TQ * 1.5) if mode == "Torque" else (PD_LO, PD_HI)
    tol = 1e-10 if mode == "Torque" else 1e-6
    worst_ab = worst_ac = worst_ao = worst_bo = 0.0
    orc = O.Oracle()
    for t in range(T):
        acts = rng.uniform(lo, hi, (n, 6))
        s0 = a.get_full_state_host()
        b.set_full_state_host(s0); c.set_full_state_host(s0)
        oa, ra, da = a.step_host(acts)
        ob, rb, db = b.step_host(acts)
        oc, rc, dc = c.step_host(acts)
        sa, sb, sc = a.get_full_state_host(), b.get_full_state_host(), c.get_full_state_host()
        den = 1.0 + np.abs(sc[:, :26]).max(axis=1)
        worst_ab = max(worst_ab, float((np.abs(sa[:, :26] - sb[:, :26]).max(axis=1) / den).max()), float(np.abs(oa - ob).max()), float(np.abs(ra - rb).max()))
        worst_ac = max(worst_ac, float((np.abs(sa[:, :26] - sc[:, :26]).max(axis=1) / den).max()), float(np.abs(oa - oc).max()), float(np.abs(ra - rc).max()))
        assert np.isfinite(sa).all()
        if t % 10 == 9:   # the oracle replays this Env.step for the four environments where the two packed tiers differ most
            dab = np.abs(sa[:, :26] - sb[:, :26]).max(axis=1) / den
            for i in np.argsort(dab)[-4:]:
                orc.set_state_raw(s0[i, :13], s0[i, 13:26], s0[i, 26:39])
                for _ in range(10):
                    (orc.step_torque if mode == "Torque" else orc.step_pd)(acts[i])
                so = np.concatenate(orc.state())
                worst_ao = max(worst_ao, float(np.abs(sa[i, :26] - so).max() / den[i]))
                worst_bo = max(worst_bo, float(np.abs(sb[i, :26] - so).max() / den[i]))
    ca = a.counters()
    assert worst_ab < tol and worst_ac < tol and worst_ao < tol, (worst_ab, worst_ac, worst_ao, worst_bo)
    if mode == "Torque":   # (PD targets hold the robots up for longer than this test runs)
        assert ca["cleanup_substeps"] > 0, "no environment ever left the 8-rows-per-leg tier: the hand-over was not exercised"
        assert sa[:, 1].min() < 0.8  # robots are falling
    record(test="leg_tier_agreement", mode=mode, n=n, steps=T, worst_vs_g16=worst_ab, worst_vs_wave_per_env=worst_ac,
           worst_leg_vs_oracle=worst_ao, worst_g16_vs_oracle=worst_bo, **ca)
    a.close(); b.close(); c.close()


@pytest.mark.parametrize("mode", ["Torque", "PD", "OSC"])
def test_leg_tier_results_do_not_depend_on_wavefront_neighbours(vec, traj, mode):
    """32 environments share a wavefront of the two-lanes-per-environment kernel and its sweep code is chosen per wavefront
    ("some environment has such a row" bits, "any environment still iterates"): an environment's bits must nevertheless be a
    function of its own state only.  512 falling robots in natural order and in a random permutation."""
    from cassierl_amd.vec_env import LEG_TIER_ON
    n, T = 512, 25
    rng = np.random.default_rng(32)
    a_env = vec(n, kind="stand", control_mode=mode, n_substeps=10, auto_reset=True, flags=LEG_TIER_ON)
    b_env = vec(n, kind="stand", control_mode=mode, n_substeps=10, auto_reset=True, flags=LEG_TIER_ON)
    lo, hi = (-TQ * 1.5, TQ * 1.5) if mode == "Torque" else (PD_LO, PD_HI)
    if mode == "OSC":
        lo, hi = np.array([-2, -2, -1, 0, -1, 0, -2.0]), np.array([2, 2, 1, 1, 1, 1, 2.0])
    adim = len(lo)
    a_env.reset_host()
    for t in range(12):
        a_env.step_host(rng.uniform(lo, hi, (n, adim)))
    s0 = a_env.get_full_state_host()
    perm = rng.permutation(n)
    b_env.reset_host()
    b_env.set_full_state_host(s0[perm])
    for t in range(T):
        a = rng.uniform(lo, hi, (n, adim))
        oa, ra, da = a_env.step_host(a)
        ob, rb, db = b_env.step_host(a[perm])
        sa, sb = a_env.get_full_state_host(), b_env.get_full_state_host()
        assert np.array_equal(sa[perm], sb), (t, np.abs(sa[perm] - sb).max())
        assert np.array_equal(oa[perm], ob) and np.array_equal(ra[perm], rb) and np.array_equal(da[perm], db)
    record(test="leg_permutation_invariance", mode=mode, n=n, steps=T, **a_env.counters())
    a_env.close(); b_env.close()


def test_lower_tiers_side_by_side_or_one_after_the_other_give_the_same_bits(vec, monkeypatch):
    """launch_physics_tiers (cassie_cabi.hip) picks the order of the two lower kernel tiers from a hint it reads without
    synchronising: one after the other (the middle tier looks at every handed-down environment first) or side by side on two streams
    (a small kernel routes the environments the middle tier could not hold straight to the wave-per-environment kernel).  The order
    must not change a single bit: 16 384 robots driven to the ground with random torques (hand-overs to both lower tiers in every
    step), once with each order forced, and once with the automatic choice.  r04: side by side means the Env.step in SEGMENTS (the
    lower tiers start on an overflowing environment while the first tier goes on with the rest; CASSIE2D_SEGMENTS=0 keeps the r03
    side-by-side order of one launch of the first tier) -- a fourth run."""
    from cassierl_amd.vec_env import LEG_TIER_ON
    n, T = 16384, 120
    rng = np.random.default_rng(77)
    acts = rng.uniform(-TQ * 1.5, TQ * 1.5, (T, n, 6))
    finals, counters = [], []
    for mode, seg in (("0", None), ("1", None), (None, None), ("1", "0")):
        if mode is None:
            monkeypatch.delenv("CASSIE2D_SIDE_BY_SIDE", raising=False)
        else:
            monkeypatch.setenv("CASSIE2D_SIDE_BY_SIDE", mode)
        if seg is None:
            monkeypatch.delenv("CASSIE2D_SEGMENTS", raising=False)
        else:
            monkeypatch.setenv("CASSIE2D_SEGMENTS", seg)
        e = vec(n, kind="stand", control_mode="Torque", n_substeps=10, auto_reset=False, flags=LEG_TIER_ON)
        e.reset_host()
        for t in range(T):
            e.step_host(acts[t])
        finals.append(e.get_full_state_host())
        counters.append(e.counters())
        e.close()
    monkeypatch.delenv("CASSIE2D_SIDE_BY_SIDE", raising=False)
    monkeypatch.delenv("CASSIE2D_SEGMENTS", raising=False)
    assert np.isfinite(finals[0]).all()
    assert counters[0]["cleanup_substeps"] > 0 and counters[0]["k1_substeps"] > 0, counters[0]   # both lower tiers had work
    assert np.array_equal(finals[0], finals[1]) and np.array_equal(finals[0], finals[2]) and np.array_equal(finals[0], finals[3])
    assert counters[0]["cleanup_substeps"] == counters[1]["cleanup_substeps"] == counters[2]["cleanup_substeps"] == counters[3]["cleanup_substeps"]

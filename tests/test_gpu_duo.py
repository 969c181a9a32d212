"""The 64-environments-per-wavefront form of the first physics tier (cassierl_amd/csrc/cassie_duo_core.h, env_step_duo_kernel) on the
GPU against the two-lanes-per-environment form (env_step_leg_kernel): BIT-IDENTICAL state records, observations, rewards, done flags
-- also where environments leave the tier and are finished by the lower tiers.  (Against the oracle: the `[duo]` ids of the vec_tier
fixture in test_gpu_parity.py / test_gpu_ctrl.py.)  -m gpu only."""
import numpy as np
import pytest

from conftest import state_vec

pytestmark = pytest.mark.gpu
PD_LO, PD_HI = np.radians([-50, -164, -140] * 2), np.radians([80, -37, -30] * 2)
TQ = np.array([12.0, 12.0, 0.9] * 2)


def _envs(n, **kw):
    from cassierl_amd.vec_env import CassieVecEnv, LEG_TIER_ON, DUO_TIER_ON, DUO_TIER_OFF
    flags = kw.pop("flags", 0)
    return (CassieVecEnv(n, flags=flags | LEG_TIER_ON | DUO_TIER_OFF, **kw), CassieVecEnv(n, flags=flags | LEG_TIER_ON | DUO_TIER_ON, **kw))


def _same(a, b, what):
    assert np.array_equal(a, b, equal_nan=True), (what, np.argwhere(a != b)[:6].tolist())


def _close(a, b, what):
    """Observation / reward: BIT-IDENTICAL too (r06).  Until r05 this allowed 1e-13: the units were then compiled with hipcc's default
    -ffp-contract=fast, where the back end fuses a multiply-add in one kernel and not in the other; with -ffp-contract=on (build.py UNIT_FLAGS)
    fusion is the front end's decision per source expression and the end-of-step arithmetic (operational-space state, reward, done) comes out the same
    in every kernel that instantiates cassie_leg_core.h -- so a reward or a body height one ulp from a termination threshold cannot flip `done`
    between the tiers the size rule switches between."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape and np.array_equal(a, b, equal_nan=True), (what, np.argwhere(a != b)[:6].tolist())


@pytest.mark.parametrize("n", [8, 37, 4096 + 45, 65536])
def test_walk_pd_rollout_with_resets_is_bit_identical(n, traj):
    """The bench's regime (walk env, PD, random targets, every step resets): set-up of two groups, joint sweep, reset pass; batch
    sizes with an empty group B (8), a partly filled one (37), a partly filled last wavefront, the bench's own."""
    import torch
    from cassierl_amd import rollout as R
    pair, duo = _envs(n, kind="walk", control_mode="PD", n_substeps=10, auto_reset=True)
    ids = torch.arange(n, device="cuda:0")
    outs = []
    for env in (pair, duo):
        env.set_trajectory(traj["time"], traj["qpos"])
        bufs = env.alloc()
        env.reset(bufs)
        rows = []
        for t in range(12 if n > 10000 else 25):
            o, r, d = env.step(R.random_actions(1, ids, t, PD_LO, PD_HI), bufs)
            rows.append((o.cpu().numpy().copy(), r.cpu().numpy().copy(), d.cpu().numpy().copy()))
        outs.append((rows, env.get_full_state_host(), env.counters()))
    for t, (a, b) in enumerate(zip(outs[0][0], outs[1][0])):
        _close(a[0], b[0], (t, "obs")); _close(a[1], b[1], (t, "reward")); _same(a[2], b[2], (t, "done"))
    _same(outs[0][1], outs[1][1], "state records")
    assert outs[0][2]["cleanup_substeps"] == outs[1][2]["cleanup_substeps"] == 0
    pair.close(); duo.close()


@pytest.mark.parametrize("kind,mode,auto_reset", [("stand", "Torque", False), ("stand", "PD", True), ("stand", "Torque", True)])
def test_falling_robots_through_the_tiers_are_bit_identical(kind, mode, auto_reset, traj):
    """Robots that fall, hit their joint limits and lie on the ground: eight-row sweeps inside the kernel, groups of a wavefront that
    differ in which sweep they take, environments handed to the lower tiers (which finish the step: the records are compared AFTER
    them, setState snapshot included)."""
    import torch
    from cassierl_amd import rollout as R
    n = 8192 + 19
    pair, duo = _envs(n, kind=kind, control_mode=mode, n_substeps=10, auto_reset=auto_reset)
    ids = torch.arange(n, device="cuda:0")
    lo, hi = (-TQ, TQ) if mode == "Torque" else (PD_LO, PD_HI)
    outs = []
    for env in (pair, duo):
        bufs = env.alloc()
        env.reset(bufs)
        rows = []
        for t in range(60):
            o, r, d = env.step(R.random_actions(3, ids, t, lo, hi), bufs)
            if t % 6 == 5:
                rows.append((o.cpu().numpy().copy(), r.cpu().numpy().copy(), d.cpu().numpy().copy()))
        outs.append((rows, env.get_full_state_host(), env.counters()))
    for t, (a, b) in enumerate(zip(outs[0][0], outs[1][0])):
        _close(a[0], b[0], (t, "obs")); _close(a[1], b[1], (t, "reward")); _same(a[2], b[2], (t, "done"))
    _same(outs[0][1], outs[1][1], "state records")
    assert outs[0][2]["cleanup_substeps"] == outs[1][2]["cleanup_substeps"]
    if not auto_reset:
        assert outs[0][2]["cleanup_substeps"] > 0, "the run must include hand-overs"
    pair.close(); duo.close()


def test_osc_controller_in_the_loop_is_bit_identical():
    """configs[2]: the OSC controller kernel writes the motor commands, the physics substep is this tier in MODE 2 (one substep per launch)."""
    import torch
    from cassierl_amd import rollout as R
    n = 4096 + 3
    pair, duo = _envs(n, kind="stand", control_mode="OSC", n_substeps=10, auto_reset=True)
    ids = torch.arange(n, device="cuda:0")
    lo, hi = np.array([-2.0, -2.0, -2.0, 0.0, -2.0, 0.0, -2.0]), np.full(7, 2.0)
    outs = []
    for env in (pair, duo):
        bufs = env.alloc()
        env.reset(bufs)
        for t in range(8):
            o, r, d = env.step(R.random_actions(4, ids, t, lo, hi), bufs)
        outs.append((o.cpu().numpy().copy(), r.cpu().numpy().copy(), env.get_full_state_host()))
    _close(outs[0][0], outs[1][0], "obs"); _close(outs[0][1], outs[1][1], "reward")
    _same(outs[0][2], outs[1][2], "state records")
    pair.close(); duo.close()


@pytest.mark.parametrize("mode", ["PD", "Torque"])
def test_height_field_rollout_is_bit_identical(mode):
    """N4: the same kernels on terrain (env_step_duo_hf_kernel against env_step_leg_hf_kernel): a 3 cm rolling relief, stand env with resets,
    robots spread along x so that they meet different facets."""
    import torch
    from cassierl_amd import rollout as R
    n = 8192 + 77
    xs = np.linspace(-10.0, 10.0, 2001)
    relief = np.tile(0.015 * (1.0 - np.cos(2.0 * np.pi * xs / 1.5)), (64, 1))
    pair, duo = _envs(n, kind="stand", control_mode=mode, n_substeps=10, auto_reset=True)
    ids = torch.arange(n, device="cuda:0")
    lo, hi = (-TQ * 0.5, TQ * 0.5) if mode == "Torque" else (PD_LO, PD_HI)
    outs = []
    for env in (pair, duo):
        env.set_heightfield(relief, 10.0, 10.0)
        bufs = env.alloc()
        env.reset(bufs)
        s = env.get_full_state_host()
        s[:, 0] += np.linspace(-6.0, 6.0, n)                      # along the relief
        s[:, 1] += 0.03
        env.set_full_state_host(s)
        rows = []
        for t in range(30):
            o, r, d = env.step(R.random_actions(6, ids, t, lo, hi), bufs)
            if t % 5 == 4:
                rows.append((o.cpu().numpy().copy(), r.cpu().numpy().copy(), d.cpu().numpy().copy()))
        outs.append((rows, env.get_full_state_host(), env.counters()))
    for t, (a, b) in enumerate(zip(outs[0][0], outs[1][0])):
        _close(a[0], b[0], (t, "obs")); _close(a[1], b[1], (t, "reward")); _same(a[2], b[2], (t, "done"))
    _same(outs[0][1], outs[1][1], "state records")
    assert outs[0][2]["nonfinite_resets"] == outs[1][2]["nonfinite_resets"] == 0
    pair.close(); duo.close()


@pytest.mark.parametrize("table,flat", [(2048, False), (128, True), (64, False)])
def test_workspace_claim_table_is_bit_identical(table, flat, traj, monkeypatch):
    """r06: batches of more than one round of the chip claim their hand-over workspace per wavefront from a table (DuoSlots, cassie_kernels_duo.hip;
    first probe = hash of the wavefront's physical place, compare-and-swap + linear probing).  CASSIE2D_DUO_TABLE forces the claim path for a small
    batch: the chip's own table size (the hash must be collision-free: zero extra probes), a small table with every first probe on word 0 (heavy
    probing), a table with fewer words than wavefronts (a wavefront waits for a release) -- launch after launch, against the two-lanes kernel."""
    import torch
    from cassierl_amd import rollout as R
    n = 4096 + 45
    monkeypatch.setenv("CASSIE2D_DUO_TABLE", str(table))
    if flat:
        monkeypatch.setenv("CASSIE2D_DUO_FLAT_HINT", "1")
    pair, duo = _envs(n, kind="walk", control_mode="PD", n_substeps=10, auto_reset=True)
    info = duo.tier_info()
    assert info["first_tier"] == "duo" and info["duo_table_slots"] == table and pair.tier_info()["first_tier"] == "leg"
    ids = torch.arange(n, device="cuda:0")
    outs = []
    for env in (pair, duo):
        env.set_trajectory(traj["time"], traj["qpos"])
        bufs = env.alloc()
        env.reset(bufs)
        rows = []
        for t in range(12):
            o, r, d = env.step(R.random_actions(1, ids, t, PD_LO, PD_HI), bufs)
            rows.append((o.cpu().numpy().copy(), r.cpu().numpy().copy(), d.cpu().numpy().copy()))
        outs.append((rows, env.get_full_state_host(), env.counters()))
    for t, (a, b) in enumerate(zip(outs[0][0], outs[1][0])):
        _close(a[0], b[0], (t, "obs")); _close(a[1], b[1], (t, "reward")); _same(a[2], b[2], (t, "done"))
    _same(outs[0][1], outs[1][1], "state records")
    probes = duo.tier_info()["ws_probes"]
    if table == 2048:
        assert probes == 0, "two resident wavefronts hashed to one workspace slot: the physical-place hash is not collision-free on this part"
    if flat:
        assert probes > 0
    pair.close(); duo.close()


def test_workspace_claim_table_falling_robots_and_terrain(monkeypatch):
    """... with robots that fall (eight-row groups, hand-overs to the lower tiers) and on the height field (env_step_duo_hf_kernel)."""
    import torch
    from cassierl_amd import rollout as R
    monkeypatch.setenv("CASSIE2D_DUO_TABLE", "64")
    monkeypatch.setenv("CASSIE2D_DUO_FLAT_HINT", "1")
    n = 2048 + 19
    xs = np.linspace(-10.0, 10.0, 2001)
    relief = np.tile(0.015 * (1.0 - np.cos(2.0 * np.pi * xs / 1.5)), (64, 1))
    for hf in (False, True):
        pair, duo = _envs(n, kind="stand", control_mode="Torque", n_substeps=10, auto_reset=hf)
        ids = torch.arange(n, device="cuda:0")
        outs = []
        for env in (pair, duo):
            if hf:
                env.set_heightfield(relief, 10.0, 10.0)
            bufs = env.alloc()
            env.reset(bufs)
            for t in range(60):
                o, r, d = env.step(R.random_actions(3, ids, t, -TQ, TQ), bufs)
            outs.append((o.cpu().numpy().copy(), d.cpu().numpy().copy(), env.get_full_state_host(), env.counters()))
        _close(outs[0][0], outs[1][0], "obs"); _same(outs[0][1], outs[1][1], "done"); _same(outs[0][2], outs[1][2], "state records")
        assert outs[0][3]["cleanup_substeps"] == outs[1][3]["cleanup_substeps"]
        if not hf:
            assert outs[0][3]["cleanup_substeps"] > 0
        pair.close(); duo.close()


def test_handover_estimate_reaches_the_host_while_robots_are_down():
    """The segment scheduler's input: classify_pending_kernel's estimate of the environments that left the first tier, summed in device memory and
    handed to the host by ONE plain store per launch (r06; until r05 a system-scope atomic add on pinned host memory, which needs PCIe atomics and
    left the hint at zero -- and the segmented order unused -- where the platform lacks them).  Robots under random torques without resets are on
    the ground after ~150 Env.steps; the estimate must then be of the order of the hand-overs the counters report."""
    import torch
    from cassierl_amd import rollout as R
    from cassierl_amd.vec_env import CassieVecEnv, LEG_TIER_ON
    n = 8192
    env = CassieVecEnv(n, kind="stand", control_mode="Torque", n_substeps=10, auto_reset=False, flags=LEG_TIER_ON)
    bufs = env.alloc()
    env.reset(bufs)
    ids = torch.arange(n, device="cuda:0")
    for t in range(220):
        env.step(R.random_actions(3, ids, t, -TQ, TQ), bufs)
    env.synchronize()

    env.reset_counters()
    for t in range(220, 300):
        env.step(R.random_actions(3, ids, t, -TQ, TQ), bufs)
    env.synchronize()
    info, c = env.tier_info(), env.counters()
    per_launch = c["cleanup_substeps"] / 80.0 / 10.0   # env-substeps handed over per Env.step / substeps left on average ~ environments
    assert c["cleanup_substeps"] > 0 and info["handovers_per_launch"] > 0, (info, c)
    assert 0.05 * per_launch < info["handovers_per_launch"] < 40.0 * per_launch + 64, (info["handovers_per_launch"], per_launch)
    env.close()


def test_size_rule_splits_a_batch_between_the_two_kernels(traj):
    """r06: a batch of whole rounds of the chip plus a short remainder (65 537 .. 98 304 envs on an MI355X) steps its whole rounds in the 64-environments
    kernel and the remainder in the two-lanes kernel, in one Env.step -- `tier_info()["duo_envs"]` says where the cut is.  The kernels are bit-identical,
    so the result must equal the all-two-lanes run, record for record, across the cut."""
    import torch
    from cassierl_amd import rollout as R
    from cassierl_amd.vec_env import CassieVecEnv, LEG_TIER_ON, DUO_TIER_OFF
    n = 65536 + 4141
    split = CassieVecEnv(n, kind="stand", control_mode="PD", n_substeps=10, auto_reset=True)
    info = split.tier_info()
    if info["duo_envs"] in (0, n):
        split.close()
        pytest.skip("the size rule does not split %d envs on this part (duo_envs = %d)" % (n, info["duo_envs"]))
    assert info["first_tier"] == "duo" and info["duo_envs"] == 65536
    pair = CassieVecEnv(n, kind="stand", control_mode="PD", n_substeps=10, auto_reset=True, flags=LEG_TIER_ON | DUO_TIER_OFF)
    ids = torch.arange(n, device="cuda:0")
    outs = []
    for env in (pair, split):
        bufs = env.alloc()
        env.reset(bufs)
        for t in range(12):
            o, r, d = env.step(R.random_actions(5, ids, t, PD_LO, PD_HI), bufs)
        outs.append((o.cpu().numpy().copy(), r.cpu().numpy().copy(), d.cpu().numpy().copy(), env.get_full_state_host()))
    for k, what in enumerate(("obs", "reward", "done", "state records")):
        _same(outs[0][k], outs[1][k], what)
    pair.close(); split.close()

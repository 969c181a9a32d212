"""The oracle's C restatement of the env layer against golden streams recorded by running the REFERENCE's own
rllab/envs/cassie2d.py and cassie_stand2d.py (stub rllab) on the oracle-backed drop-in library
(tests/golden/make_env_streams.py): pins reset/step/obs/reward/done arithmetic and the stale-state quirks."""
import numpy as np
import pytest

CASES = [("walk_pd", "walk", "PD"), ("walk_torque", "walk", "Torque"), ("stand_torque", "stand", "Torque"),
         ("stand_pd", "stand", "PD"), ("stand_osc", "stand", "OSC")]


@pytest.mark.parametrize("tag,kind,mode", CASES)
def test_env_stream(oracle_mod, streams, traj, tag, kind, mode):
    env = oracle_mod.OracleEnv(kind, mode, traj=dict(time=traj["time"], qpos=traj["qpos"]))
    np.testing.assert_allclose(env.reset(), streams[tag + "_obs0"], rtol=0, atol=1e-13)
    acts = streams[tag + "_actions"]
    for t in range(len(acts)):
        obs, r, d = env.step(acts[t])
        np.testing.assert_allclose(obs, streams[tag + "_obs"][t], rtol=0, atol=1e-9)
        assert abs(r - streams[tag + "_reward"][t]) < 1e-12
        assert d == bool(streams[tag + "_done"][t])
        if d:
            np.testing.assert_allclose(env.reset(), streams[tag + "_reset_obs"][t], rtol=0, atol=1e-9)


def test_walk_env_terminates_every_step(streams):
    # faithful reproduction of the committed reference: the joint reward term compares the stale reset pose with the
    # reference gait (quirk Q3), so r < 0.6 and every step ends its episode
    assert streams["walk_pd_done"].all() and (streams["walk_pd_reward"] < 0.6).all()


def test_osc_qp_is_kkt_converged(oracle_mod):
    o = oracle_mod.Oracle()
    rng = np.random.default_rng(7)
    for i in range(40):
        a = rng.uniform(-1, 1, 7) * np.array([3, 3, 1, 1, 1, 1, 3.0])
        o.step_osc(a)
        x, kkt = o.osc_qp()
        assert kkt[0] < 1e-7 and kkt[1] < 1e-8 and kkt[2] < 1e-7, kkt
        u = x[13:19]
        assert (u >= np.array([-12.2, -12.2, -0.9] * 2) - 1e-9).all() and (u <= np.array([12.2, 12.2, 0.9] * 2) + 1e-9).all()
        assert (x[19:] >= -1e-9).all()
        np.testing.assert_allclose(o.ctrl(), u)


def test_standing_controller_osc_holds_height(oracle_mod):
    # cassie2d.py:263-295 standing_controller_osc(zpos=0.9): behavioural acceptance test of the OSC path (README criterion)
    o = oracle_mod.Oracle()
    zs = []
    for i in range(1500):
        s = o.opstate(0)
        body_x, body_xd, left_x, right_x = s[0:3], s[3:6], s[6:9], s[12:15]
        act = np.zeros(7)
        act[2] = 0.0; act[3] = 100.0 * (-5e-3 - left_x[1])
        act[4] = 0.0; act[5] = 100.0 * (-5e-3 - right_x[1])
        xt = (left_x[0] + right_x[0]) / 2.0
        act[0] = 100.0 * (xt - body_x[0]) + 20.0 * (0.0 - body_xd[0])
        act[1] = 100.0 * (0.9 - body_x[1]) + 20.0 * (0.0 - body_xd[1])
        act[6] = 20.0 * (0.0 - body_x[2]) + 10.0 * (0.0 - body_xd[2])
        o.step_osc(act)
        zs.append(o.state()[0][1])
    assert 0.8 < zs[-1] < 1.0 and min(zs) > 0.7

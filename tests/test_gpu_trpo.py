"""TRPO over the resident batched env on the GPU (config 4 building block).  -m gpu only."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_trpo_iterations_on_cassie_stand(traj):
    import torch
    from cassierl_amd.trajectory import Cassie2dTraj
    from cassierl_amd.trpo import make_cassie_trpo, flat_params
    tr = Cassie2dTraj.from_arrays(traj["time"], traj["qpos"])
    algo = make_cassie_trpo(512, kind="stand", control_mode="Torque", trajectory=tr, batch_size=512 * 8, seed=1)
    before = flat_params(algo.policy).clone()
    stats = [algo.train_iteration() for _ in range(3)]
    for st in stats:
        assert np.isfinite(st["avg_reward"]) and st["kl"] <= 0.005 + 1e-6 and st["env_steps"] == 512 * 8 and st["gathered"] == 512
    assert any(st["backtracks"] >= 0 and st["loss_after"] < st["loss_before"] for st in stats)
    assert (flat_params(algo.policy) - before).abs().max() > 0
    q, v = algo.env.get_state_host()
    assert np.isfinite(q).all() and np.isfinite(v).all()
    algo.env.close()


def test_trpo_walk_env_pd_mode(traj):
    from cassierl_amd.trajectory import Cassie2dTraj
    from cassierl_amd.trpo import make_cassie_trpo
    tr = Cassie2dTraj.from_arrays(traj["time"], traj["qpos"])
    algo = make_cassie_trpo(256, kind="walk", control_mode="PD", trajectory=tr, batch_size=256 * 4, seed=1)
    st = algo.train_iteration()
    # faithful reference semantics: every step terminates (quirk Q3), so every path has length 1
    assert st["episodes"] == 256 * 4 and abs(st["avg_return"] - st["avg_reward"]) < 1e-9
    algo.env.close()

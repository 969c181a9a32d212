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


def _free_port():
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def test_two_rank_trpo_iterations_equal_the_one_rank_run(tmp_path):
    """GPU twin of tests/test_trpo_cpu.py::test_data_parallel_update_equals_single_process, launched through train_trpo.py: two
    ranks with 2048 envs each (both on device 0, gloo carrying the all-reduces -- RCCL refuses duplicate devices) against ONE
    rank with the same 4096 global env ids.  Env physics does not depend on the shard, the exploration noise is keyed by the
    global env id, and gradient / Fisher products / baseline / line-search statistics are all-reduced, so both runs take the same
    TRPO steps up to float32 reduction order."""
    import json, os, subprocess, sys
    from conftest import ROOT
    script = os.path.join(ROOT, "train_trpo.py")
    common = ["--horizon", "4", "--n-itr", "2", "--kind", "stand", "--control-mode", "Torque"]
    one, two = str(tmp_path / "one.npy"), str(tmp_path / "two.npy")
    p1 = subprocess.run([sys.executable, script, "--envs-per-gpu", "4096", "--dump-params", one] + common, capture_output=True, text=True, timeout=900)
    assert p1.returncode == 0, p1.stderr[-2000:]
    env = dict(os.environ, CASSIE_DEVICE_MAP="0,0", CASSIE_BACKEND="gloo")
    p2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                         "--master-port", str(_free_port()), script, "--envs-per-gpu", "2048", "--dump-params", two] + common,
                        capture_output=True, text=True, timeout=900, env=env)
    assert p2.returncode == 0, p2.stderr[-2000:]
    s1 = [json.loads(l) for l in p1.stdout.splitlines() if l.startswith("{")]
    s2 = [json.loads(l) for l in p2.stdout.splitlines() if l.startswith("{")]
    assert len(s1) == len(s2) == 2
    for a, b in zip(s1, s2):
        assert a["env_steps"] == b["env_steps"] == 4096 * 4 and a["gathered"] == b["gathered"] == 4096
        assert a["backtracks"] == b["backtracks"] and a["episodes"] == b["episodes"]
        assert abs(a["avg_reward"] - b["avg_reward"]) < 1e-6 and abs(a["kl"] - b["kl"]) < 1e-5 and abs(a["loss_after"] - b["loss_after"]) < 1e-5
    t1, t2 = np.load(one), np.load(two)
    assert np.abs(t1 - t2).max() < 1e-4 * max(1.0, np.abs(t1).max()), np.abs(t1 - t2).max()


def test_sim_policy_rolls_a_snapshot_out(tmp_path):
    """sim_policy.py counterpart (rllab/envs/sim_policy.py:19-31): train two iterations with --snapshot, then load the snapshot
    and roll the policy out without training."""
    import json, os, subprocess, sys
    from conftest import ROOT
    snap = str(tmp_path / "snap.pt")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "train_trpo.py"), "--envs-per-gpu", "512", "--horizon", "4", "--n-itr", "2", "--kind", "stand",
                        "--control-mode", "Torque", "--snapshot", snap], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and os.path.exists(snap), p.stderr[-2000:]
    q = subprocess.run([sys.executable, os.path.join(ROOT, "sim_policy.py"), snap, "--envs", "256", "--max-path-length", "60", "--kind", "stand",
                        "--control-mode", "Torque"], capture_output=True, text=True, timeout=900)
    assert q.returncode == 0, q.stderr[-2000:]
    r = json.loads([l for l in q.stdout.splitlines() if l.startswith("{")][-1])
    assert r["itr"] == 2 and r["envs"] == 256 and 0 < r["avg_path_length"] <= 60 and np.isfinite(r["avg_return"])

"""TRPO outer loop (cassierl_amd/trpo.py, counterpart of rllab/envs/trpo_cassie.py) on CPU: the math against naive
references, one constrained update, learning on a toy vectorised env, and the world-size-2 (gloo) data-parallel update."""
import math
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cassierl_amd import trpo as T


class ToyVecEnv:
    """N point masses on a line: obs = (x, v, target, 1), action = force in [-1, 1] (2-D action, second ignored);
    reward = -(x - target)^2 - 0.01 a^2; episode of 20 steps then auto-reset."""

    def __init__(self, n, seed=0):
        self.n = n
        self.g = torch.Generator().manual_seed(seed)
        self.reset()

    def _obs(self):
        return torch.stack([self.x, self.v, self.tg, torch.ones(self.n, dtype=torch.float64)], dim=1)

    def reset(self):
        self.x = torch.zeros(self.n, dtype=torch.float64); self.v = torch.zeros(self.n, dtype=torch.float64)
        self.tg = torch.rand(self.n, generator=self.g, dtype=torch.float64) * 2 - 1
        self.t = torch.zeros(self.n, dtype=torch.int64)
        return self._obs()

    def step(self, a):
        f = a[:, 0].clamp(-1, 1)
        self.v = 0.8 * self.v + 0.2 * f
        self.x = self.x + self.v
        r = -(self.x - self.tg) ** 2 - 0.01 * f * f
        self.t += 1
        done = self.t >= 20
        if done.any():
            idx = done.nonzero().squeeze(-1)
            self.x[idx] = 0; self.v[idx] = 0; self.t[idx] = 0
            self.tg[idx] = torch.rand(len(idx), generator=self.g, dtype=torch.float64) * 2 - 1
        return self._obs(), r, done


def test_discounted_returns_match_per_path_loop():
    rng = np.random.default_rng(0)
    Tn, N, g = 30, 5, 0.97
    rew = rng.normal(size=(Tn, N)); done = rng.uniform(size=(Tn, N)) < 0.15
    out = T.discounted_returns(torch.tensor(rew), torch.tensor(done), g).numpy()
    for n in range(N):
        run = 0.0
        for t in range(Tn - 1, -1, -1):
            run = rew[t, n] + (0.0 if done[t, n] else g * run)
            assert abs(out[t, n] - run) < 1e-12


def test_linear_feature_baseline_is_ridge_regression():
    rng = np.random.default_rng(1)
    obs = torch.tensor(rng.normal(size=(400, 6)) * 4); t = torch.tensor(rng.integers(0, 300, 400)); y = torch.tensor(rng.normal(size=400))
    b = T.LinearFeatureBaseline(1e-5)
    b.fit(obs, t, y)
    X = b.features(obs, t).numpy()
    ref = np.linalg.solve(X.T @ X + 1e-5 * np.eye(X.shape[1]), X.T @ y.numpy())
    np.testing.assert_allclose(b.coeffs.numpy(), ref, rtol=1e-8, atol=1e-10)
    assert X.shape[1] == 2 * 6 + 4 and np.abs(X[:, :6]).max() <= 10.0


def test_normalized_actions():
    low, high = np.radians([-50, -164, -140] * 2), np.radians([80, -37, -30] * 2)
    m = T.NormalizedActions(low, high, "cpu")
    a = torch.tensor([[-1.0] * 6, [1.0] * 6, [0.0] * 6, [3.0] * 6, [-7.0] * 6])
    out = m(a).numpy()
    np.testing.assert_allclose(out[0], low); np.testing.assert_allclose(out[1], high)
    np.testing.assert_allclose(out[2], 0.5 * (low + high)); np.testing.assert_allclose(out[3], high); np.testing.assert_allclose(out[4], low)


def test_conjugate_gradient():
    rng = np.random.default_rng(2)
    A = rng.normal(size=(12, 12)); A = torch.tensor(A @ A.T + 12 * np.eye(12)); b = torch.tensor(rng.normal(size=12))
    x = T.conjugate_gradient(lambda v: A @ v, b, iters=50, tol=1e-24)
    np.testing.assert_allclose((A @ x).numpy(), b.numpy(), atol=1e-8)


def _toy_algo(n=64, seed=1, **kw):
    env = ToyVecEnv(n, seed)
    torch.manual_seed(seed)
    pol = T.GaussianMLPPolicy(4, 2, (32, 32), init_std=1.0, dtype=torch.float64)
    algo = T.TRPO(env.step, env.reset, pol, T.LinearFeatureBaseline(), n, 4, T.NormalizedActions([-1, -1], [1, 1], "cpu"),
                  batch_size=n * 40, max_path_length=1000, discount=0.99, step_size=0.01, seed=seed, **kw)
    return algo


def test_one_update_respects_the_trust_region():
    algo = _toy_algo()
    before = T.flat_params(algo.policy).clone()
    st = algo.train_iteration()
    assert st["backtracks"] >= 0 and st["backtracks"] <= 15
    assert st["kl"] <= 0.01 + 1e-9 and st["loss_after"] < st["loss_before"]
    assert (T.flat_params(algo.policy) - before).abs().max() > 0
    assert st["env_steps"] == 64 * 40 and st["gathered"] == 64


def test_policy_initialisation_follows_trpo_cassie():
    pol = T.GaussianMLPPolicy(26, 6, (32, 32), init_std=2.0)
    shapes = [tuple(p.shape) for p in pol.parameters()]
    assert shapes == [(6,), (32, 26), (32,), (32, 32), (32,), (6, 32), (6,)]
    assert abs(float(pol.log_std[0]) - math.log(2.0)) < 1e-6
    assert sum(p.numel() for p in pol.parameters()) == 26 * 32 + 32 + 32 * 32 + 32 + 32 * 6 + 6 + 6


def test_learning_improves_reward_on_toy_env():
    algo = _toy_algo(n=128, seed=3)
    first = algo.train_iteration()["avg_reward"]
    for _ in range(25):
        last = algo.train_iteration()["avg_reward"]
    assert last > first + 0.05, (first, last)


def test_checkpoint_roundtrip(tmp_path):
    algo = _toy_algo()
    algo.train_iteration()
    p = str(tmp_path / "params.pt")
    algo.save(p, extra={"note": 1})
    other = _toy_algo(seed=9)
    extra, restored = other.load(p)
    assert extra == {"note": 1} and other.itr == 1
    assert torch.equal(T.flat_params(other.policy), T.flat_params(algo.policy))
    assert torch.equal(other.baseline.coeffs, algo.baseline.coeffs)


class SnapshotToyEnv(ToyVecEnv):
    """ToyVecEnv with the two state-record hooks the snapshot uses (88 doubles per env, like the Cassie2d record)."""

    def get_full_state_host(self):
        s = np.zeros((self.n, 88))
        s[:, 0], s[:, 1], s[:, 2], s[:, 3] = self.x.numpy(), self.v.numpy(), self.tg.numpy(), self.t.numpy()
        return s

    def set_full_state_host(self, s):
        self.x, self.v, self.tg = (torch.tensor(s[:, i].copy()) for i in range(3))
        self.t = torch.tensor(s[:, 3].astype(np.int64))


def _snap_algo(seed):
    env = SnapshotToyEnv(32, seed)
    env.g = None  # resets draw no randomness below (episodes of 20 steps never end inside the compared window)
    torch.manual_seed(seed)
    pol = T.GaussianMLPPolicy(4, 2, (32, 32), init_std=1.0, dtype=torch.float64)
    algo = T.TRPO(env.step, env.reset, pol, T.LinearFeatureBaseline(), 32, 4, T.NormalizedActions([-1, -1], [1, 1], "cpu"),
                  batch_size=32 * 4, seed=seed)
    algo.env = env
    return algo


def test_resumed_run_is_the_interrupted_run(tmp_path):
    """SURVEY.md section 5 checkpoint/resume: the snapshot carries policy, baseline AND the sampler state (env records,
    action-noise generator, observation, path clocks), so iteration k+1 after a resume equals iteration k+1 without one."""
    a = _snap_algo(2)
    a.env.g = torch.Generator().manual_seed(2); a.env.reset(); a.obs = None
    a.train_iteration()
    p = str(tmp_path / "snap.pt")
    a.save(p)
    ref = a.train_iteration()
    b = _snap_algo(7)  # different seed: everything must come from the snapshot
    b.env.g = torch.Generator().manual_seed(99)
    _, restored = b.load(p)
    assert restored and b.sampler_restored
    got = b.train_iteration()
    assert got["itr"] == ref["itr"] == 1
    assert abs(got["avg_reward"] - ref["avg_reward"]) < 1e-12 and abs(got["loss_before"] - ref["loss_before"]) < 1e-12
    assert torch.allclose(T.flat_params(a.policy), T.flat_params(b.policy), atol=1e-12)


def test_truncated_paths_reset_the_env():
    """rllab's sampler calls env.reset() when a path reaches max_path_length: the masked-reset hook is called
    with exactly the envs that were cut without terminating, and the next observation is the reset one."""
    env = ToyVecEnv(8, 0)
    calls = []

    def reset_masked(mask):
        calls.append(mask.clone())
        idx = mask.bool().nonzero().squeeze(-1)
        env.x[idx] = 0; env.v[idx] = 0; env.t[idx] = 0
        return env._obs()

    torch.manual_seed(0)
    pol = T.GaussianMLPPolicy(4, 2, (32, 32), init_std=1.0, dtype=torch.float64)
    algo = T.TRPO(env.step, env.reset, pol, T.LinearFeatureBaseline(), 8, 4, T.NormalizedActions([-1, -1], [1, 1], "cpu"),
                  batch_size=8 * 12, max_path_length=5, env_reset_masked=reset_masked)
    batch = algo.collect()
    # the masked-reset launch happens only when some path WAS truncated: steps 5 and 10 of 12, never with an empty mask
    assert len(calls) == 2 and all(int(c.sum()) == 8 for c in calls)
    assert batch["done"][4].all() and batch["done"][9].all() and not batch["done"][5].any()
    assert (batch["t"][5] == 0).all() and (batch["obs"][5][:, 0] == 0).all()  # x was reset to 0


# ---- data-parallel update: 2 ranks with half the batch each == 1 process with the whole batch
def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _make_batch(seed, n):
    g = torch.Generator().manual_seed(seed)
    obs = torch.randn(n, 4, generator=g, dtype=torch.float64)
    return obs, g


def _run_update(obs, act, mean, lstd, adv, seed=5):
    torch.manual_seed(seed)
    pol = T.GaussianMLPPolicy(4, 2, (16, 16), init_std=1.0, dtype=torch.float64)
    algo = T.TRPO(None, None, pol, T.LinearFeatureBaseline(), 1, 4, None, batch_size=1, step_size=0.01)
    with torch.no_grad():
        m0, l0 = pol.dist_info(obs)
    st = algo.optimize(dict(obs=obs, act=act, mean=m0.clone(), log_std=l0.clone(), adv=adv))
    return T.flat_params(pol), st


def _data(n=256):
    g = torch.Generator().manual_seed(11)
    obs = torch.randn(n, 4, generator=g, dtype=torch.float64)
    act = torch.randn(n, 2, generator=g, dtype=torch.float64)
    adv = torch.randn(n, generator=g, dtype=torch.float64)
    return obs, act, adv


def _dp_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), GLOO_SOCKET_IFNAME="lo")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    obs, act, adv = _data()
    h = obs.shape[0] // world
    sl = slice(rank * h, (rank + 1) * h)
    theta, st = _run_update(obs[sl], act[sl], None, None, adv[sl])
    if rank == 0:
        q.put((theta.numpy(), st))
    dist.destroy_process_group()


def test_data_parallel_update_equals_single_process():
    obs, act, adv = _data()
    ref, st_ref = _run_update(obs, act, None, None, adv)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    theta, st = q.get(timeout=180)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    np.testing.assert_allclose(theta, ref.numpy(), rtol=0, atol=1e-9)
    assert st["backtracks"] == st_ref["backtracks"] and abs(st["kl"] - st_ref["kl"]) < 1e-10


def test_analytic_fisher_matches_double_backprop():
    """F v in closed form (AnalyticFisher) against rllab-style double backprop through mean-KL, float64."""
    import torch
    from cassierl_amd.trpo import AnalyticFisher, GaussianMLPPolicy, flat_grad
    torch.manual_seed(0)
    pol = GaussianMLPPolicy(26, 6, (32, 32), init_std=2.0, dtype=torch.float64)
    with torch.no_grad():
        for p in pol.parameters():
            p.add_(0.1 * torch.randn_like(p))
    obs = torch.randn(500, 26, dtype=torch.float64)
    old_mean, old_lstd = [t.detach() for t in pol.dist_info(obs)]
    mean, lstd = pol.dist_info(obs)
    kl = pol.kl(old_mean, old_lstd, mean, lstd).mean()
    gk = flat_grad(kl, pol, retain_graph=True, create_graph=True)
    fisher = AnalyticFisher(pol, obs)
    for _ in range(4):
        v = torch.randn(sum(p.numel() for p in pol.parameters()), dtype=torch.float64)
        ref = flat_grad(gk @ v, pol, retain_graph=True)
        got = fisher(v)
        assert torch.allclose(got, ref, rtol=1e-9, atol=1e-12), float((got - ref).abs().max())


def test_chunked_gram_matches_plain_products():
    import torch
    from cassierl_amd.trpo import gram
    torch.manual_seed(1)
    for n in (5, 2048, 5000, 8192):
        X, y = torch.randn(n, 7, dtype=torch.float64), torch.randn(n, dtype=torch.float64)
        A, b = gram(X, y, chunk=2048)
        assert torch.allclose(A, X.T @ X, rtol=1e-12, atol=1e-10) and torch.allclose(b, X.T @ y, rtol=1e-12, atol=1e-10)


def test_closed_form_policy_gradient_matches_autograd():
    """optimize() takes the gradient of the surrogate loss in closed form when a Fisher object exists (likelihood ratio 1 at theta_old:
    d loss / d mean = -adv z / std / N through the Fisher object's reverse pass, d loss / d log_std = -sum adv (z^2 - 1) / N):
    against autograd through the surrogate, float64."""
    import torch
    from cassierl_amd.trpo import AnalyticFisher, GaussianMLPPolicy, flat_grad
    torch.manual_seed(5)
    pol = GaussianMLPPolicy(26, 6, (32, 32), init_std=2.0, dtype=torch.float64)
    with torch.no_grad():
        for p in pol.parameters():
            p.add_(0.2 * torch.randn_like(p))
    n = 700
    obs = torch.randn(n, 26, dtype=torch.float64)
    adv = torch.randn(n, dtype=torch.float64)
    with torch.no_grad():
        mean, lstd = pol.dist_info(obs)
        act = mean + torch.randn_like(mean) * lstd.exp()
        old_ll = pol.log_likelihood(act, mean, lstd)
    m2, l2 = pol.dist_info(obs)
    loss = -((pol.log_likelihood(act, m2, l2) - old_ll).exp() * adv).mean()
    g_ref = flat_grad(loss, pol)
    fisher = AnalyticFisher(pol, obs)
    std = lstd.exp()
    z = (act - mean) / std
    g = fisher.vjp(-(adv.unsqueeze(-1) * z / std) / n)
    g_ls = -((adv.unsqueeze(-1) * (z * z - 1.0)).sum(0)) / n
    i0 = 0
    for nm, p_ in pol.named_parameters():
        if nm == "log_std":
            g[i0:i0 + p_.numel()] += g_ls
        i0 += p_.numel()
    assert (g - g_ref).abs().max() < 1e-12 * (1 + g_ref.abs().max())
    assert abs(float(loss) - float(-adv.mean())) < 1e-14

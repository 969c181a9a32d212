"""The fused TRPO policy kernels (csrc/tu_trpo.hip) against the torch implementations they replace.  -m gpu only."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,obs_dim,act_dim", [(1000, 26, 6), (65536, 26, 6), (4099, 26, 7), (777, 17, 6)])
def test_fused_fisher_and_vjp_match_the_torch_versions(n, obs_dim, act_dim):
    import torch
    from cassierl_amd import trpo as T
    torch.manual_seed(3)
    pol = T.GaussianMLPPolicy(obs_dim, act_dim, (32, 32), init_std=2.0).cuda()
    with torch.no_grad():
        for p in pol.parameters():
            p.add_(0.3 * torch.randn_like(p))
    obs = torch.randn(n, obs_dim, device="cuda") * 0.7
    ref = T.AnalyticFisher(pol, obs)
    fused = T.FusedFisher(pol, obs)
    nparam = sum(p.numel() for p in pol.parameters())
    for k in range(3):
        v = torch.randn(nparam, device="cuda")
        a, b = ref(v), fused(v)
        scale = a.abs().max().item()
        assert (a - b).abs().max().item() < 2e-4 * scale, (k, (a - b).abs().max().item(), scale)
    # J' w against autograd
    w = torch.randn(n, act_dim, device="cuda") / n
    mean, _ = pol.dist_info(obs)
    g = torch.autograd.grad((mean * w).sum(), list(pol.parameters()), allow_unused=True)
    gref = torch.cat([torch.zeros_like(p).reshape(-1) if x is None else x.reshape(-1) for x, p in zip(g, pol.parameters())])
    gf = fused.vjp(w)
    assert (gref - gf).abs().max().item() < 2e-4 * gref.abs().max().item()


def test_fused_fisher_timing_524288_samples():
    import time
    import torch
    from cassierl_amd import trpo as T
    torch.manual_seed(4)
    n = 524288
    pol = T.GaussianMLPPolicy(26, 6, (32, 32), init_std=2.0).cuda()
    obs = torch.randn(n, 26, device="cuda")
    v = torch.randn(sum(p.numel() for p in pol.parameters()), device="cuda")
    out = {}
    for name, F in (("torch", T.AnalyticFisher(pol, obs)), ("fused", T.FusedFisher(pol, obs))):
        for _ in range(3):
            F(v)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(11):
            F(v)
        torch.cuda.synchronize()
        out[name] = (time.perf_counter() - t0) / 11 * 1e3
    print("fvp ms per product at %d samples: %s" % (n, out))   # r04: fused 0.80, torch 0.67 -- which is why it is opt-in
    assert out["fused"] < 3.0 * out["torch"]


@pytest.mark.parametrize("control_mode,adim", [("PD", 6), ("OSC", 7)])
def test_fused_policy_step_matches_the_torch_operations(control_mode, adim):
    """CassieTrpoPolicyStep (float32 view of obs, mean network, noise, normalize() action map in one launch) against
    GaussianMLPPolicy.get_actions + NormalizedActions on the same observations and noise."""
    import torch
    from cassierl_amd import trpo as T
    from cassierl_amd.vec_env import action_space
    torch.manual_seed(7)
    n = 5000
    pol = T.GaussianMLPPolicy(26, adim, (32, 32), init_std=2.0).cuda()
    with torch.no_grad():
        for p in pol.parameters():
            p.add_(0.3 * torch.randn_like(p))
    box = action_space(control_mode)
    amap = T.NormalizedActions(box.low, box.high, "cuda")
    algo = T.TRPO(None, None, pol, T.LinearFeatureBaseline(), n, 26, amap)
    step = algo._fused_policy_step(torch.device("cuda:0"), torch.float32)
    assert step is not None
    obs = torch.randn(n, 26, dtype=torch.float64, device="cuda")
    noise = torch.randn(n, adim, device="cuda")
    o32, mean, act = torch.empty(n, 26, device="cuda"), torch.empty(n, adim, device="cuda"), torch.empty(n, adim, device="cuda")
    step(obs, noise, o32, mean, act)
    a_ref, m_ref, _ = pol.get_actions(obs.float(), noise=noise)
    assert torch.equal(o32, obs.float())
    assert (mean - m_ref).abs().max().item() < 2e-6 * (1 + m_ref.abs().max().item())
    assert (act - a_ref).abs().max().item() < 2e-6 * (1 + a_ref.abs().max().item())
    e_ref = amap(act)
    assert (algo._env_actions - e_ref).abs().max().item() < 1e-12
    lo, hi = torch.as_tensor(box.low, device="cuda"), torch.as_tensor(box.high, device="cuda")
    assert (algo._env_actions >= lo).all() and (algo._env_actions <= hi).all()


@pytest.mark.parametrize("n,obs_dim,act_dim", [(1000, 26, 6), (65536, 26, 6), (4099, 26, 7), (777, 17, 6)])
def test_fused_surrogate_matches_the_torch_line_search_evaluation(n, obs_dim, act_dim):
    """CassieTrpoSurrogate (loss and mean KL of the line search in one launch) against the torch expressions of TRPO.optimize on a
    perturbed policy."""
    import torch
    from cassierl_amd import trpo as T
    torch.manual_seed(5)
    pol = T.GaussianMLPPolicy(obs_dim, act_dim, (32, 32), init_std=1.3).cuda()
    with torch.no_grad():
        for p in pol.parameters():
            p.add_(0.3 * torch.randn_like(p))
    obs = torch.randn(n, obs_dim, device="cuda") * 0.7
    with torch.no_grad():
        old_mean, old_lstd = pol.dist_info(obs)
        old_mean, old_lstd = old_mean.clone(), old_lstd.clone()
        act = old_mean + torch.randn_like(old_mean) * old_lstd.exp()
        adv = torch.randn(n, device="cuda")
        fused = T.FusedFisher(pol, obs)
        for p in pol.parameters():   # the candidate of a backtrack
            p.add_(0.02 * torch.randn_like(p))
        mean, log_std = pol.dist_info(obs)
        lr = (pol.log_likelihood(act, mean, log_std) - pol.log_likelihood(act, old_mean, old_lstd)).exp()
        l_ref, k_ref = -(lr.double() * adv.double()).mean().item(), pol.kl(old_mean, old_lstd, mean, log_std).double().mean().item()
        l, k = fused.surrogate(pol, act, adv, old_mean, old_lstd[0])
    scale = (lr.double() * adv.double()).abs().mean().item()
    assert abs(l.item() - l_ref) < 2e-5 * scale, (l.item(), l_ref)
    assert abs(k.item() - k_ref) < 2e-5 * max(k_ref, 1e-3), (k.item(), k_ref)


def test_fused_sampler_step_matches_the_torch_bookkeeping():
    """CassieTrpoSamplerStep (path clocks / returns / truncation / episode statistics in one launch) against the element-wise torch
    operations of TRPO.collect: two samplers from the same seed, paths truncated after 5 steps so that every branch is taken."""
    import torch
    from cassierl_amd import trpo as T
    algos = []
    for fused in (True, False):
        a = T.make_cassie_trpo(4096, kind="stand", control_mode="Torque", batch_size=4096 * 12, max_path_length=5, seed=3)
        a.fused_sampler_step = fused
        algos.append(a)
    for it in range(2):
        b0, b1 = algos[0].collect(), algos[1].collect()
        for k in ("obs", "act", "mean", "rew", "done", "t"):
            assert torch.equal(b0[k], b1[k]), (it, k)
        assert b0["done"].any() and not b0["done"].all()
        assert float(b0["episode_count"]) == float(b1["episode_count"]) > 0
        assert abs(float(b0["episode_return_sum"]) - float(b1["episode_return_sum"])) < 1e-9 * (1 + abs(float(b1["episode_return_sum"])))
        assert torch.equal(algos[0].path_t, algos[1].path_t) and torch.equal(algos[0].path_ret, algos[1].path_ret)


def features_dot(obs, tt, coeffs):
    from cassierl_amd import trpo as T
    return T.LinearFeatureBaseline.features(obs, tt).double() @ coeffs


def test_baseline_kernels_match_the_torch_expressions():
    """csrc/tu_trpo_baseline.hip against LinearFeatureBaseline / TRPO.process of trpo.py on the same batches: normal equations (FP64 MFMA),
    returns, advantages and predictions; two iterations, so that the second one runs with fitted coefficients and bootstrapped paths."""
    import torch
    from cassierl_amd import trpo as T
    algos = []
    for fused in (True, False):
        a = T.make_cassie_trpo(4099, kind="stand", control_mode="Torque", batch_size=4099 * 12, max_path_length=7, seed=5)
        a.fused_baseline = fused
        algos.append(a)
    for it in range(2):
        batch = algos[0].collect()
        b1 = algos[1].collect()
        assert torch.equal(batch["rew"], b1["rew"]) and torch.equal(batch["done"], b1["done"])
        d0, d1 = algos[0].process(batch), algos[1].process(b1)
        assert algos[0]._bk is not None
        scale = d1["adv"].abs().max().item()
        assert (d0["adv"] - d1["adv"]).abs().max().item() < 2e-5 * scale, (it, (d0["adv"] - d1["adv"]).abs().max().item(), scale)   # float32 outputs
        # the normal equations themselves
        Tn, N = batch["rew"].shape
        obs, tt = batch["obs"].reshape(Tn * N, -1), batch["t"].reshape(Tn * N)
        y = torch.randn(Tn * N, dtype=torch.float64, device="cuda")
        A, b = algos[0]._bk.gram(obs, tt, y)
        Ar, br = T.gram(T.LinearFeatureBaseline.features(obs, tt).double(), y)
        assert (A - Ar).abs().max().item() < 1e-10 * Ar.abs().max().item()
        assert ((A - Ar).abs() / (Ar.abs() + 1e-3 * Ar.abs().max())).max().item() < 1e-9   # entry by entry: the float32 feature arithmetic is the same
        assert (b - br).abs().max().item() < 1e-10 * (br.abs().max().item() + Ar.abs().max().item() ** 0.5)
        assert torch.equal(A, A.T)
        # the device solve (Cholesky + the retry rule) against torch.linalg.solve on the same system, and a singular system that needs the retries
        xs = algos[0]._bk.ridge_solve(A, b, 1e-5)
        xr = torch.linalg.solve(A + 1e-5 * torch.eye(A.shape[0], dtype=A.dtype, device="cuda"), b)
        assert ((A + 1e-5 * torch.eye(A.shape[0], dtype=A.dtype, device="cuda")) @ xs - b).abs().max().item() < 1e-6 * (1 + b.abs().max().item())
        assert (features_dot(obs, tt, xs) - features_dot(obs, tt, xr)).abs().max().item() < 1e-6 * (1 + features_dot(obs, tt, xr).abs().max().item())
        Z = torch.zeros_like(A)
        x0 = algos[0]._bk.ridge_solve(Z, b, 0.0)     # 0 x = b: no positive pivot at any regulariser 0 * 10^k -> the last attempt's (non-finite or stale) vector, no hang
        assert x0.shape == b.shape
        # fitted baselines predict the same values (the solve amplifies rounding by the conditioning of X'X: compare predictions)
        c0, c1 = algos[0].baseline.coeffs, algos[1].baseline.coeffs
        p0 = algos[0]._bk.predict(obs, tt, c0)
        p1 = T.LinearFeatureBaseline.features(obs, tt).double() @ c1
        assert (p0 - p1).abs().max().item() < 1e-5 * (1 + p1.abs().max().item()), (it, (p0 - p1).abs().max().item())
        algos[0].optimize(d0); algos[1].optimize(d1)


def test_fused_cg_matches_the_torch_conjugate_gradient():
    """FusedFisher.conjugate_gradient (CassieTrpoCgUpdate: the vector work of an iteration in one launch) against conjugate_gradient() over
    the same Fisher-vector products: ten iterations, and the early exit (a tolerance that is met half-way freezes the same iterate)."""
    import torch
    from cassierl_amd import trpo as T
    torch.manual_seed(9)
    n = 20000
    pol = T.GaussianMLPPolicy(26, 6, (32, 32), init_std=1.5).cuda()
    with torch.no_grad():
        for p in pol.parameters():
            p.add_(0.3 * torch.randn_like(p))
    obs = torch.randn(n, 26, device="cuda") * 0.7
    fisher = T.FusedFisher(pol, obs)
    g = torch.randn(sum(p.numel() for p in pol.parameters()), device="cuda")
    reg = 1e-5
    ref = T.conjugate_gradient(lambda v: fisher(v) + reg * v, g, 10)
    x = fisher.conjugate_gradient(g, 10, reg)
    assert x is not None
    assert (x - ref).abs().max().item() < 2e-3 * ref.abs().max().item(), ((x - ref).abs().max().item(), ref.abs().max().item())
    # early exit: a tolerance between the residuals of iterations 3 and 4 stops both at the same iterate
    r4 = T.conjugate_gradient(lambda v: fisher(v) + reg * v, g, 4)
    res = (fisher(r4) + reg * r4 - g)
    tol = float(res @ res) * 1.5
    a = T.conjugate_gradient(lambda v: fisher(v) + reg * v, g, 10, tol=tol)
    b = fisher.conjugate_gradient(g, 10, reg, tol=tol)
    assert (a - b).abs().max().item() < 2e-3 * a.abs().max().item()
    assert (a - ref).abs().max().item() > 1e-2 * ref.abs().max().item()   # (it did stop early)

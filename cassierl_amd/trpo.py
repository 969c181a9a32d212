"""TRPO outer loop over the batched environment: counterpart of rllab/envs/trpo_cassie.py:12-55.

The reference drives ONE environment through rllab's TRPO (Theano).  Here the rollout is N resident environments stepped by
one kernel launch per Env.step, and the optimiser is written directly against torch tensors on the same device:

  policy     GaussianMLPPolicy(hidden_sizes=(32, 32), init_std=2.0)            trpo_cassie.py:21-27
             (tanh hidden units, state-independent learned log-std -- rllab's defaults [external])
  baseline   LinearFeatureBaseline: ridge regression on [o, o^2, t, t^2, t^3, 1]  trpo_cassie.py:29 [external]
  algorithm  TRPO: batch_size env-steps per iteration, max_path_length=1000, discount=0.99, step_size (mean KL) 0.005,
             conjugate gradient (10 iterations, damping 1e-5) + backtracking line search (0.8, 15)   trpo_cassie.py:31-42
  env        normalize(Cassie2dEnv()): actions in [-1, 1] mapped affinely to the action box and clipped  trpo_cassie.py:13

Multi-GPU: one process per GPU, each with its own shard of environments; the policy gradient, every Fisher-vector
product, the baseline's normal equations and the line-search statistics are averaged with all_reduce (RCCL on MI355X, gloo in
the CPU tests), so all ranks take the identical step.  Episode returns are gathered once per batch (rollout.gather_returns).
"""
import math

import torch
import torch.distributed as dist
from torch import nn


# --------------------------------------------------------------------------------------------- distributed helpers
def _world():
    return dist.get_world_size() if dist.is_initialized() else 1


class CommTimer:
    """Per-collective timing of the data-parallel loop without host synchronisation: a pair of HIP events on the current stream
    around every collective (the RCCL kernel runs on the communicator's stream, which the current stream waits for), read once by
    `summary()`.  bench.py sets `trpo.COMM = CommTimer()` for configs[3]'s line; None (default) costs nothing."""

    def __init__(self):
        self.spans = {}

    def run(self, name, fn, t):
        if not t.is_cuda:
            import time
            t0 = time.perf_counter()
            fn()
            self.spans.setdefault(name, []).append(time.perf_counter() - t0)
            return
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        self.spans.setdefault(name, []).append((e0, e1))

    def summary(self):
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        out = {}
        for name, sp in self.spans.items():
            ms = [x * 1e3 if isinstance(x, float) else x[0].elapsed_time(x[1]) for x in sp]
            out[name] = dict(calls=len(ms), total_ms=float(sum(ms)), mean_ms=float(sum(ms) / max(1, len(ms))), max_ms=float(max(ms)))
        return out


COMM = None


def _all_reduce_sum(t, name):
    if COMM is None:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    else:
        COMM.run(name, lambda: dist.all_reduce(t, op=dist.ReduceOp.SUM), t)


def all_mean_(t, name="all_reduce"):
    """In-place mean over ranks (no-op single process)."""
    if _world() > 1:
        _all_reduce_sum(t, name)
        t /= _world()
    return t


def all_sum_(t, name="all_reduce"):
    if _world() > 1:
        _all_reduce_sum(t, name)
    return t


# --------------------------------------------------------------------------------------------- policy / baseline
class GaussianMLPPolicy(nn.Module):
    def __init__(self, obs_dim, act_dim, hidden_sizes=(32, 32), init_std=2.0, dtype=torch.float32):
        super().__init__()
        layers, d = [], obs_dim
        for h in hidden_sizes:
            layers += [nn.Linear(d, h), nn.Tanh()]
            d = h
        layers.append(nn.Linear(d, act_dim))
        self.mean_net = nn.Sequential(*layers)
        self.log_std = nn.Parameter(torch.full((act_dim,), math.log(init_std)))
        for m in self.mean_net:
            if isinstance(m, nn.Linear):  # rllab: Xavier-uniform weights, zero bias [external]
                nn.init.xavier_uniform_(m.weight)
                nn.init.zeros_(m.bias)
        self.to(dtype)

    def dist_info(self, obs):
        return self.mean_net(obs), self.log_std.expand(obs.shape[0], -1)

    @torch.no_grad()
    def get_actions(self, obs, generator=None, noise=None):
        mean, log_std = self.dist_info(obs)
        if noise is None:
            noise = torch.randn(mean.shape, dtype=mean.dtype, device=mean.device, generator=generator)
        return mean + noise.to(mean.dtype) * log_std.exp(), mean, log_std

    @staticmethod
    def log_likelihood(actions, mean, log_std):
        z = (actions - mean) / log_std.exp()
        return -(log_std.sum(-1) + 0.5 * (z * z).sum(-1) + 0.5 * mean.shape[-1] * math.log(2 * math.pi))

    @staticmethod
    def kl(old_mean, old_log_std, new_mean, new_log_std):
        old_std, new_std = old_log_std.exp(), new_log_std.exp()
        num = (old_mean - new_mean) ** 2 + old_std ** 2 - new_std ** 2
        return (num / (2 * new_std ** 2 + 1e-8) + new_log_std - old_log_std).sum(-1)


def gram(X, y, chunk=2048):
    """(X'X, X'y) for a tall-skinny X [N, F].  One GEMM with K = N is pathological in rocBLAS (56 ms for 524 288 x 58 in
    FP64 on MI355X); a batch of K = `chunk` GEMMs summed afterwards does the same arithmetic in 0.2 ms."""
    Z = torch.cat([X, y.unsqueeze(-1)], dim=-1)
    n, f = Z.shape
    m = (n // chunk) * chunk
    G = torch.zeros((f, f), dtype=Z.dtype, device=Z.device)
    if m:
        Zc = Z[:m].view(n // chunk, chunk, f)
        G += torch.bmm(Zc.transpose(1, 2), Zc).sum(0)
    if m < n:
        G += Z[m:].T @ Z[m:]
    return G[:-1, :-1], G[:-1, -1]


def tmatmul(A, B, chunk=2048):
    """A' B for tall-skinny A [N, p], B [N, q] (a reduction over N): chunked for the same reason as gram()."""
    n = A.shape[0]
    m = (n // chunk) * chunk
    out = torch.zeros((A.shape[1], B.shape[1]), dtype=A.dtype, device=A.device)
    if m:
        out += torch.bmm(A[:m].view(n // chunk, chunk, -1).transpose(1, 2), B[:m].view(n // chunk, chunk, -1)).sum(0)
    if m < n:
        out += A[m:].T @ B[m:]
    return out


class LinearFeatureBaseline:
    """Ridge regression of the discounted return on [o, o^2, t, t^2, t^3, 1], o clipped to [-10, 10], t = step/100."""

    def __init__(self, reg_coeff=1e-5):
        self.coeffs, self.reg_coeff = None, reg_coeff

    @staticmethod
    def features(obs, t):
        o = obs.clamp(-10, 10)
        al = (t.to(obs.dtype) / 100.0).unsqueeze(-1)
        return torch.cat([o, o * o, al, al ** 2, al ** 3, torch.ones_like(al)], dim=-1)

    def fit(self, obs, t, returns):
        X = self.features(obs, t).double()
        y = returns.double()
        A, b = gram(X, y)
        self.fit_normal_equations(A, b)

    def fit_normal_equations(self, A, b, solver=None):
        """coeffs from this rank's X'X [F, F] and X'y [F] (summed over ranks here).  solver(A, b, reg): the same rule on the device."""
        A, b = all_sum_(A.contiguous(), "baseline_all_reduce"), all_sum_(b.contiguous(), "baseline_all_reduce")
        if solver is not None:
            self.coeffs = solver(A, b, self.reg_coeff)
            return
        reg = self.reg_coeff
        eye = torch.eye(A.shape[0], dtype=A.dtype, device=A.device)
        for _ in range(5):
            sol = torch.linalg.solve(A + reg * eye, b)
            if torch.isfinite(sol).all():
                break
            reg *= 10
        self.coeffs = sol

    def predict(self, obs, t):
        if self.coeffs is None:
            return torch.zeros(obs.shape[0], dtype=torch.float64, device=obs.device)
        return self.features(obs, t).double() @ self.coeffs


class BaselineKernels:
    """LinearFeatureBaseline on the device batch as three HIP kernels (csrc/tu_trpo_baseline.hip, include/cassie_trpo.h): prediction,
    returns / advantages of the [T, n] batch in one pass, and the regression's normal equations on the FP64 matrix cores -- the feature
    matrix is never materialised.  CUDA float32 observations of a supported width only (ValueError otherwise: TRPO.process falls back to
    the torch expressions of LinearFeatureBaseline)."""

    def __init__(self, dev, obs_dim):
        import ctypes as ct
        from . import _lib
        if dev.type != "cuda":
            raise ValueError("BaselineKernels need a CUDA device")
        self.L, self.ct, self.dev, self.D = _lib.load(), ct, dev, obs_dim
        self.F = self.L.CassieTrpoBaselineFeatures(obs_dim)
        if self.F == 0:
            raise ValueError("BaselineKernels: unsupported observation width %d" % obs_dim)
        rows, size = self.L.CassieTrpoGramRows(), self.L.CassieTrpoGramRowSize(obs_dim)
        self.gram_partial = torch.empty((rows, size), dtype=torch.float64, device=dev)
        # full symmetric matrix from the upper 16 x 16 blocks (r <= c, r-major, each row-major): one gather
        nb = ((self.F + 1) + 15) // 16
        order, k = {}, 0
        for r in range(nb):
            for c in range(r, nb):
                order[(r, c)] = k
                k += 1
        idx = torch.empty((16 * nb, 16 * nb), dtype=torch.int64)
        for I in range(16 * nb):
            for J in range(16 * nb):
                r, c, i, j = I // 16, J // 16, I % 16, J % 16
                idx[I, J] = order[(r, c)] * 256 + i * 16 + j if r <= c else order[(c, r)] * 256 + j * 16 + i
        self.idx = idx.reshape(-1).to(dev)
        self.nz = 16 * nb

    def _stream(self):
        return self.ct.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)

    def _p(self, t):
        return self.ct.c_void_p(t.data_ptr()) if t is not None else None

    def predict(self, obs32, t, coeffs):
        m = obs32.shape[0]
        out = torch.empty(m, dtype=torch.float64, device=self.dev)
        rc = self.L.CassieTrpoBaselinePredict(self._p(obs32.contiguous()), self._p(t.contiguous()), m, self.D, self._p(coeffs.contiguous()), self._p(out), self._stream())
        if rc != 0:
            raise RuntimeError("CassieTrpoBaselinePredict failed (%d)" % rc)
        return out

    def returns_advantages(self, obs_b, t_b, rew_b, cut_b, coeffs, last_value, gamma):
        """[T, n] batch -> returns [T, n], advantages [T, n] (float64) and (sum adv, sum adv^2) as a float64 pair on the device."""
        T, n = rew_b.shape
        assert obs_b.is_contiguous() and t_b.is_contiguous() and rew_b.is_contiguous() and cut_b.is_contiguous()
        assert obs_b.dtype == torch.float32 and t_b.dtype == torch.int64 and rew_b.dtype == torch.float64 and cut_b.dtype in (torch.bool, torch.uint8)
        returns, adv = torch.empty_like(rew_b), torch.empty_like(rew_b)
        partial = torch.empty(((n + 255) // 256, 2), dtype=torch.float64, device=self.dev)
        rc = self.L.CassieTrpoReturnsAdvantages(self._p(obs_b), self._p(t_b), self._p(rew_b), self._p(cut_b), T, n, self.D,
                                                self._p(None if coeffs is None else coeffs.contiguous()), self._p(None if last_value is None else last_value.contiguous()),
                                                self.ct.c_double(gamma), self._p(returns), self._p(adv), self._p(partial), self._stream())
        if rc != 0:
            raise RuntimeError("CassieTrpoReturnsAdvantages failed (%d)" % rc)
        return returns, adv, partial.sum(0)

    def gram(self, obs32, t, y):
        """(X'X [F, F], X'y [F]) of the baseline's features on m samples."""
        m = obs32.shape[0]
        assert obs32.is_contiguous() and t.is_contiguous() and y.is_contiguous() and y.dtype == torch.float64
        rc = self.L.CassieTrpoBaselineGram(self._p(obs32), self._p(t), self._p(y), m, self.D, self._p(self.gram_partial), self._stream())
        if rc != 0:
            raise RuntimeError("CassieTrpoBaselineGram failed (%d)" % rc)
        G = self.gram_partial.sum(0)[self.idx].view(self.nz, self.nz)
        return G[:self.F, :self.F], G[:self.F, self.F]

    def ridge_solve(self, A, b, reg):
        """(A + reg I)^-1 b on the device (Cholesky, with fit's retry rule inside the kernel: no read-back)."""
        A, b = A.contiguous(), b.contiguous()
        x = torch.empty_like(b)
        rc = self.L.CassieTrpoRidgeSolve(self._p(A), self._p(b), A.shape[0], self.ct.c_double(reg), self._p(x), self._stream())
        if rc != 0:
            raise RuntimeError("CassieTrpoRidgeSolve failed (%d)" % rc)
        return x


class NormalizedActions:
    """rllab.envs.normalized_env.normalize (actions only): [-1, 1] -> [lb, ub], then clip."""

    def __init__(self, low, high, device, dtype=torch.float64):
        self.low = torch.as_tensor(low, dtype=dtype, device=device)
        self.high = torch.as_tensor(high, dtype=dtype, device=device)

    def __call__(self, a):
        a = a.to(self.low.dtype)
        scaled = self.low + (a + 1.0) * 0.5 * (self.high - self.low)
        return torch.minimum(torch.maximum(scaled, self.low), self.high).contiguous()


# --------------------------------------------------------------------------------------------- math helpers
def discounted_returns(rewards, dones, gamma, last_value=None):
    """rewards, dones: [T, N].  Return-to-go that restarts after a done (path boundary)."""
    T = rewards.shape[0]
    out = torch.zeros_like(rewards)
    run = torch.zeros_like(rewards[0]) if last_value is None else last_value.clone()
    for t in range(T - 1, -1, -1):
        run = rewards[t] + gamma * run * (~dones[t]).to(rewards.dtype)
        out[t] = run
    return out


def flat_params(module):
    return torch.cat([p.data.reshape(-1) for p in module.parameters()])


def set_flat_params(module, flat):
    i = 0
    for p in module.parameters():
        n = p.numel()
        p.data.copy_(flat[i:i + n].view_as(p))
        i += n


def flat_grad(y, module, retain_graph=False, create_graph=False):
    g = torch.autograd.grad(y, list(module.parameters()), retain_graph=retain_graph, create_graph=create_graph)
    return torch.cat([x.reshape(-1) for x in g])


def conjugate_gradient(Avp, b, iters=10, tol=1e-10):
    """rllab/misc/krylov.py:cg -- including its early exit on the residual, but WITHOUT asking the device for it: once
    r.r < tol the step length is zero from then on (x and r stay what they were at the break), so no iteration waits for a
    host-side comparison (r04: ten synchronisations per update were most of what the CG loop cost around its products)."""
    x = torch.zeros_like(b)
    r, p = b.clone(), b.clone()
    rr = r @ r
    zero = torch.zeros((), dtype=b.dtype, device=b.device)
    running = torch.ones((), dtype=torch.bool, device=b.device)
    for _ in range(iters):
        Ap = Avp(p)
        alpha = torch.where(running, rr / (p @ Ap), zero)
        x.addcmul_(alpha, p)
        r.addcmul_(alpha, Ap, value=-1.0)
        rr_new = r @ r
        running = running & (rr_new >= tol)
        p = torch.addcmul(r, torch.where(running, rr_new / rr, zero), p)   # (stopped: p = r, finite whatever rr is)
        rr = rr_new
    return x


class AnalyticFisher:
    """Fisher-vector products of the Gaussian tanh-MLP policy in closed form (what rllab's `hvp_approach` gets by double
    backprop through mean-KL): F v = (1/N) J' S J v for the mean network (J = d mean / d theta by forward mode, S the
    precision of the old Gaussian) plus a diagonal block for log_std.  Activations of the OLD policy are computed once per
    TRPO update; a product is then ~8 GEMMs and a handful of element-wise kernels instead of ~100 autograd kernels."""

    def __init__(self, policy, obs, eps=1e-8):
        lin = [m for m in policy.mean_net if isinstance(m, nn.Linear)]
        act_ok = all(isinstance(m, (nn.Linear, nn.Tanh)) for m in policy.mean_net)
        if len(lin) != 3 or not act_ok:
            raise ValueError("AnalyticFisher covers the two-hidden-layer tanh policy of trpo_cassie.py only")
        self.names = [n for n, _ in policy.named_parameters()]
        self.shapes = [tuple(p.shape) for p in policy.parameters()]
        with torch.no_grad():
            self.W = [l.weight.detach().clone() for l in lin]
            self.X = obs
            self.H1 = torch.tanh(torch.addmm(lin[0].bias, obs, self.W[0].T))
            self.H2 = torch.tanh(torch.addmm(lin[1].bias, self.H1, self.W[1].T))
            self.D1, self.D2 = 1 - self.H1 * self.H1, 1 - self.H2 * self.H2
            var = (2 * policy.log_std.detach()).exp()
            self.prec = 2.0 / (2.0 * var + eps)                                   # d2 KL / d mean^2 (with kl()'s epsilon)
            self.h_ls = 4.0 * var * (2.0 * var - eps) / (2.0 * var + eps) ** 2     # d2 KL / d log_std^2
            self.n = obs.shape[0]

    @torch.no_grad()
    def __call__(self, v):
        parts, i = {}, 0
        for n, shp in zip(self.names, self.shapes):
            k = int(torch.tensor(shp).prod()) if len(shp) else 1
            parts[n] = v[i:i + k].view(shp)
            i += k
        dW1, db1 = parts["mean_net.0.weight"], parts["mean_net.0.bias"]
        dW2, db2 = parts["mean_net.2.weight"], parts["mean_net.2.bias"]
        dW3, db3 = parts["mean_net.4.weight"], parts["mean_net.4.bias"]
        X, H1, H2, D1, D2 = self.X, self.H1, self.H2, self.D1, self.D2
        W1, W2, W3 = self.W
        # forward mode: directional derivative of the mean
        dH1 = D1 * torch.addmm(db1, X, dW1.T)
        dH2 = D2 * (torch.addmm(db2, dH1, W2.T) + H1 @ dW2.T)
        dmu = torch.addmm(db3, dH2, W3.T) + H2 @ dW3.T
        w = dmu * (self.prec / self.n)
        return self._reverse(w, self.h_ls * parts["log_std"])

    @torch.no_grad()
    def _reverse(self, w, log_std_part):
        """J' w (reverse mode through the cached activations) with `log_std_part` in the log_std slot, in flat parameter order."""
        X, H1, H2, D1, D2 = self.X, self.H1, self.H2, self.D1, self.D2
        W1, W2, W3 = self.W
        out = {"log_std": log_std_part}
        out["mean_net.4.weight"], out["mean_net.4.bias"] = tmatmul(w, H2), w.sum(0)
        g2 = (w @ W3) * D2
        out["mean_net.2.weight"], out["mean_net.2.bias"] = tmatmul(g2, H1), g2.sum(0)
        g1 = (g2 @ W2) * D1
        out["mean_net.0.weight"], out["mean_net.0.bias"] = tmatmul(g1, X), g1.sum(0)
        return torch.cat([out[n].reshape(-1) for n in self.names])

    @torch.no_grad()
    def vjp(self, w):
        """J' w for per-sample cotangents w [n, act_dim] on the mean; the log_std slot is zero."""
        return self._reverse(w, torch.zeros_like(self.h_ls))


class FusedFisher:
    """The same Fisher-vector products as AnalyticFisher, each in ONE launch of the fused HIP kernel (csrc/tu_trpo.hip,
    include/cassie_trpo.h): forward mode along the direction, precision of the old Gaussian, reverse mode and the outer-product
    accumulation per wavefront; obs is the only per-sample tensor read.  `vjp(w)` gives J' w for per-sample cotangents (the policy
    gradient).  CUDA float32 policies of the supported shapes only (ValueError otherwise: the caller falls back to AnalyticFisher)."""

    def __init__(self, policy, obs, eps=1e-8):
        import ctypes as ct
        from . import _lib
        lin = [m for m in policy.mean_net if isinstance(m, nn.Linear)]
        if len(lin) != 3 or not all(isinstance(m, (nn.Linear, nn.Tanh)) for m in policy.mean_net):
            raise ValueError("FusedFisher covers the two-hidden-layer tanh policy of trpo_cassie.py only")
        if not obs.is_cuda or obs.dtype != torch.float32 or lin[0].out_features != 32 or lin[1].out_features != 32:
            raise ValueError("FusedFisher needs a float32 CUDA batch and 32 x 32 hidden units")
        self.L = _lib.load()
        self.D, self.A = lin[0].in_features, lin[2].out_features
        self.NP = self.L.CassieTrpoParamCount(self.D, self.A)
        if self.NP == 0:
            raise ValueError("FusedFisher: unsupported policy shape %d -> %d" % (self.D, self.A))
        self.ct = ct
        self.obs = obs.contiguous()
        self.n = obs.shape[0]
        self.names = [n for n, _ in policy.named_parameters()]
        self.shapes = [tuple(p.shape) for p in policy.parameters()]
        self.order = ["mean_net.0.weight", "mean_net.0.bias", "mean_net.2.weight", "mean_net.2.bias", "mean_net.4.weight", "mean_net.4.bias"]
        with torch.no_grad():
            self.theta = {n: p.detach().clone().contiguous() for n, p in policy.named_parameters()}
            var = (2 * policy.log_std.detach()).exp()
            self.prec = (2.0 / (2.0 * var + eps)).to(torch.float32).contiguous()
            self.h_ls = 4.0 * var * (2.0 * var - eps) / (2.0 * var + eps) ** 2
        self.rows = self.L.CassieTrpoPartialRows(self.n)
        self.partial = torch.empty((self.rows, self.NP), dtype=torch.float32, device=obs.device)
        # offsets of the kernel's row layout [gW1 | gb1 | gW2 | gb2 | gW3 | gb3]
        self.sizes = [32 * self.D, 32, 1024, 32, self.A * 32, self.A]

    def _split(self, v):
        parts, i = {}, 0
        for n, shp in zip(self.names, self.shapes):
            k = 1
            for d in shp:
                k *= d
            parts[n] = v[i:i + k]
            i += k
        return parts

    def _assemble(self, flat_mean, log_std_part):
        pieces = dict(zip(self.order, torch.split(flat_mean, self.sizes)))
        pieces["log_std"] = log_std_part
        return torch.cat([pieces[n].reshape(-1) for n in self.names])

    def _ptrs(self, d):
        return [self.ct.c_void_p(d[n].data_ptr()) for n in self.order]

    @torch.no_grad()
    def __call__(self, v):
        v = v.to(torch.float32).contiguous()
        parts = self._split(v)
        stream = self.ct.c_void_p(torch.cuda.current_stream(self.obs.device).cuda_stream)
        rc = self.L.CassieTrpoFvp(self.ct.c_void_p(self.obs.data_ptr()), self.n, self.D, self.A, *self._ptrs(self.theta), *self._ptrs(parts),
                                  self.ct.c_void_p(self.prec.data_ptr()), self.ct.c_float(1.0 / self.n), self.ct.c_void_p(self.partial.data_ptr()), stream)
        if rc != 0:
            raise RuntimeError("CassieTrpoFvp failed (%d)" % rc)
        return self._assemble(self.partial.sum(0), self.h_ls * parts["log_std"])

    @torch.no_grad()
    def mean_product(self, v):
        """The mean network's part of F v for this rank's samples, [NP] float32 in the kernel's order (no log_std block, no damping)."""
        parts = self._split(v)
        stream = self.ct.c_void_p(torch.cuda.current_stream(self.obs.device).cuda_stream)
        rc = self.L.CassieTrpoFvp(self.ct.c_void_p(self.obs.data_ptr()), self.n, self.D, self.A, *self._ptrs(self.theta), *self._ptrs(parts),
                                  self.ct.c_void_p(self.prec.data_ptr()), self.ct.c_float(1.0 / self.n), self.ct.c_void_p(self.partial.data_ptr()), stream)
        if rc != 0:
            raise RuntimeError("CassieTrpoFvp failed (%d)" % rc)
        return self.partial.sum(0)

    @torch.no_grad()
    def conjugate_gradient(self, b, iters, reg, tol=1e-10):
        """conjugate_gradient(lambda v: all_mean_(self(v)) + reg * v, b, iters) with the vector work of an iteration as ONE launch
        (CassieTrpoCgUpdate): per iteration the product, the sum of its partial rows, the all-reduce over ranks, the update.  None if
        the parameter vector is not laid out as [.. log_std ..] around a contiguous mean-network block in the kernel's order."""
        names = self.names
        if "log_std" not in names or [n for n in names if n != "log_std"] != self.order or b.dtype != torch.float32 or not b.is_contiguous():
            return None
        k = names.index("log_std")
        if k not in (0, len(names) - 1):
            return None
        ls_off = 0 if k == 0 else self.NP
        n = b.numel()
        x = torch.zeros_like(b)
        r, p = b.clone(), b.clone()
        scal = torch.stack([r @ r, torch.ones((), dtype=b.dtype, device=b.device)]).contiguous()
        hls = self.h_ls.to(torch.float32).contiguous()
        P = lambda t: self.ct.c_void_p(t.data_ptr())
        for _ in range(iters):
            apm = all_mean_(self.mean_product(p), "fvp_all_reduce")
            rc = self.L.CassieTrpoCgUpdate(n, ls_off, self.A, P(apm), P(hls), self.ct.c_float(reg), self.ct.c_float(tol), P(x), P(r), P(p), P(scal),
                                           self.ct.c_void_p(torch.cuda.current_stream(self.obs.device).cuda_stream))
            if rc != 0:
                raise RuntimeError("CassieTrpoCgUpdate failed (%d)" % rc)
        return x

    @torch.no_grad()
    def vjp(self, w):
        """J' w for w [n, act_dim] float32 (cotangents on the mean); the log_std slot of the result is zero."""
        w = w.to(torch.float32).contiguous()
        stream = self.ct.c_void_p(torch.cuda.current_stream(self.obs.device).cuda_stream)
        rc = self.L.CassieTrpoVjp(self.ct.c_void_p(self.obs.data_ptr()), self.n, self.D, self.A, *self._ptrs(self.theta),
                                  self.ct.c_void_p(w.data_ptr()), self.ct.c_void_p(self.partial.data_ptr()), stream)
        if rc != 0:
            raise RuntimeError("CassieTrpoVjp failed (%d)" % rc)
        return self._assemble(self.partial.sum(0), torch.zeros_like(self.theta["log_std"]))

    @torch.no_grad()
    def surrogate(self, policy, act, adv, old_mean, old_log_std):
        """(surrogate loss, mean KL) of `policy` AS IT IS NOW against the old Gaussian on this batch, in one launch (CassieTrpoSurrogate):
        the line search's evaluation.  old_log_std: the [act_dim] vector of the state-independent old log-std.  Two float64 scalars on the device."""
        live = dict(policy.named_parameters())
        if not hasattr(self, "_sur"):
            self._sur = torch.empty((self.rows, 2), dtype=torch.float64, device=self.obs.device)
        P = lambda t: self.ct.c_void_p(t.data_ptr())
        act, adv, old_mean = act.contiguous(), adv.to(torch.float32).contiguous(), old_mean.contiguous()
        old_ls = old_log_std.to(torch.float32).contiguous()
        stream = self.ct.c_void_p(torch.cuda.current_stream(self.obs.device).cuda_stream)
        rc = self.L.CassieTrpoSurrogate(P(self.obs), self.n, self.D, self.A, *[P(live[k].detach()) for k in self.order], P(live["log_std"].detach()), P(old_ls),
                                        P(act), P(adv), P(old_mean), P(self._sur), stream)
        if rc != 0:
            raise RuntimeError("CassieTrpoSurrogate failed (%d)" % rc)
        s = self._sur.sum(0) / self.n
        return s[0], s[1]


# --------------------------------------------------------------------------------------------- TRPO
class TRPO:
    def __init__(self, env_step, env_reset, policy, baseline, n_envs, obs_dim, act_map, batch_size=15000, max_path_length=1000,
                 discount=0.99, step_size=0.005, cg_iters=10, reg_coeff=1e-5, backtrack_ratio=0.8, max_backtracks=15, seed=1,
                 env_reset_masked=None, env_id0=None):
        """env_step(actions[N, adim] float64) -> (obs[N, obs_dim], reward[N], done[N] uint8/bool), auto-resetting;
        env_reset() -> obs; env_reset_masked(mask uint8[N]) -> obs with the masked envs reset (used when a path is truncated
        at max_path_length, where rllab's sampler calls env.reset()).  batch_size counts env-steps over ALL ranks, as
        rllab's batch_size does.
        Deviations from rllab's TRPO, on purpose: paths cut by the end of the batch are bootstrapped with the baseline's
        value of the next observation (rllab's batch sampler discards/truncates the tail instead; with 65 536 envs x few steps
        per iteration nearly every path is cut, so dropping the tail would bias every return); the policy runs in float32."""
        self.env_step, self.env_reset, self.env_reset_masked = env_step, env_reset, env_reset_masked
        # Host-side lower bound of the Env.steps left before ANY path can reach max_path_length (path_t <= max_path_length -
        # _steps_to_trunc holds for every env): while it is positive no truncation mask can be non-empty, so collect() launches no
        # masked reset at all; when it reaches zero the device-side maximum of path_t re-arms it (one scalar read-back per
        # max_path_length steps at most -- r02 launched the masked-reset kernel on EVERY step once 1000 steps had elapsed).
        self._steps_to_trunc = max_path_length
        self.policy, self.baseline, self.act_map = policy, baseline, act_map
        self.n_envs, self.obs_dim = n_envs, obs_dim
        self.horizon = max(1, int(math.ceil(batch_size / (n_envs * _world()))))
        self.max_path_length, self.discount, self.step_size = max_path_length, discount, step_size
        self.cg_iters, self.reg_coeff = cg_iters, reg_coeff
        self.backtrack_ratio, self.max_backtracks = backtrack_ratio, max_backtracks
        dev = next(policy.parameters()).device
        # Exploration noise that does not depend on the number of ranks the envs are sharded over: every rank draws the noise of
        # ALL environments of the job from an identically seeded generator -- one randn kernel per Env.step, 3 M numbers at
        # 512k envs -- and keeps the rows of its own shard [env_id0, env_id0 + n_envs).  (r03 first used a counter-based hash keyed
        # by the global env id, rollout.counter_normal: ~55 elementwise launches per step, a third of the rollout's wall-clock.)
        rank = dist.get_rank() if dist.is_initialized() else 0
        self.seed = seed
        self.env_id0 = rank * n_envs if env_id0 is None else env_id0
        self.n_envs_global = n_envs * _world()
        self.env_ids = torch.arange(n_envs, dtype=torch.int64, device=dev) + self.env_id0
        self.gen = torch.Generator(device=dev)
        self.gen.manual_seed(seed * 1000003)
        self.noise_step = 0
        self.obs = None
        self.path_t = torch.zeros(n_envs, dtype=torch.int64, device=dev)
        self.path_ret = torch.zeros(n_envs, dtype=torch.float64, device=dev)
        self.itr = 0

    def _fused_policy_step(self, dev, pol_dtype):
        """The fused policy-step launcher (include/cassie_trpo.h: CassieTrpoPolicyStep) when it applies -- CUDA, float32 two-layer
        tanh policy of a supported shape, rllab's normalize() action map -- else None (the torch operations below)."""
        if not getattr(self, "fused_policy_step", True) or dev.type != "cuda" or pol_dtype != torch.float32 or not isinstance(self.act_map, NormalizedActions):
            return None
        lin = [m for m in self.policy.mean_net if isinstance(m, nn.Linear)]
        if len(lin) != 3 or not all(isinstance(m, (nn.Linear, nn.Tanh)) for m in self.policy.mean_net):
            return None
        D, A = lin[0].in_features, lin[2].out_features
        if (D, A) not in ((26, 6), (26, 7)) or lin[0].out_features != 32 or lin[1].out_features != 32 or self.obs_dim != D:
            return None
        try:
            import ctypes as ct
            from . import _lib
            L = _lib.load()
        except OSError:
            return None
        if not hasattr(self, "_env_actions") or self._env_actions.shape != (self.n_envs, A):
            self._env_actions = torch.empty((self.n_envs, A), dtype=torch.float64, device=dev)
        P = lambda t: ct.c_void_p(t.data_ptr())
        w = [P(lin[0].weight), P(lin[0].bias), P(lin[1].weight), P(lin[1].bias), P(lin[2].weight), P(lin[2].bias), P(self.policy.log_std)]
        low, high, n = self.act_map.low, self.act_map.high, self.n_envs
        # the kernel reads the bounds as `const double*`: anything else (NormalizedActions takes a dtype) goes the torch way
        if not all(t.is_cuda and t.dtype == torch.float64 and t.is_contiguous() and t.numel() == A for t in (low, high)):
            return None

        def step(obs, noise, obs32, mean, act):
            if obs.dtype != torch.float64 or not obs.is_contiguous():
                raise TypeError("CassieTrpoPolicyStep: observations must be a contiguous float64 tensor (got %s)" % obs.dtype)
            assert noise.is_contiguous() and obs32.is_contiguous()
            rc = L.CassieTrpoPolicyStep(P(obs), n, D, A, *w, P(noise), P(low), P(high), P(obs32), P(mean), P(act), P(self._env_actions),
                                        ct.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
            if rc != 0:
                raise RuntimeError("CassieTrpoPolicyStep failed (%d)" % rc)
        return step

    def _fused_sampler_step(self, dev):
        """The sampler's per-step bookkeeping as one launch (include/cassie_trpo.h: CassieTrpoSamplerStep), or None."""
        if not getattr(self, "fused_sampler_step", True) or dev.type != "cuda":
            return None
        try:
            import ctypes as ct
            from . import _lib
            L = _lib.load()
        except OSError:
            return None
        n = self.n_envs
        if not hasattr(self, "_book_partial") or self._book_partial.shape[0] != L.CassieTrpoSamplerRows(n):
            self._book_partial = torch.empty((L.CassieTrpoSamplerRows(n), 2), dtype=torch.float64, device=dev)
        P = lambda t: ct.c_void_p(t.data_ptr())

        def book(rew, done, rew_row, t_row, cut_row):
            assert rew.is_contiguous() and done.is_contiguous() and rew_row.is_contiguous() and t_row.is_contiguous() and cut_row.is_contiguous()
            assert self.path_t.dtype == torch.int64 and self.path_ret.dtype == torch.float64
            rc = L.CassieTrpoSamplerStep(P(rew), P(done), n, ct.c_longlong(int(self.max_path_length)), P(self.path_t), P(self.path_ret), P(rew_row), P(t_row),
                                         P(cut_row), P(self._book_partial), ct.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
            if rc != 0:
                raise RuntimeError("CassieTrpoSamplerStep failed (%d)" % rc)
        return book

    # ---- sampling: T vectorised Env.steps, everything stays on the device
    @torch.no_grad()
    def collect(self):
        from . import rollout as R
        pol_dtype = next(self.policy.parameters()).dtype
        if self.obs is None:
            self.obs = self.env_reset().clone()
        T, N = self.horizon, self.n_envs
        dev = self.obs.device
        obs_b = torch.empty((T, N, self.obs_dim), dtype=pol_dtype, device=dev)
        act_b = torch.empty((T, N, self.policy.log_std.numel()), dtype=pol_dtype, device=dev)
        mean_b, lstd_b = torch.empty_like(act_b), torch.empty_like(act_b)
        rew_b = torch.empty((T, N), dtype=torch.float64, device=dev)
        done_b = torch.empty((T, N), dtype=torch.bool, device=dev)
        t_b = torch.empty((T, N), dtype=torch.int64, device=dev)
        ep = torch.zeros(2, dtype=torch.float64, device=dev)   # finished episodes / their summed returns, kept on the device:
        fused = self._fused_policy_step(dev, pol_dtype)          # boolean-mask indexing would synchronise every step
        book = self._fused_sampler_step(dev) if fused is not None else None
        if fused is not None:
            lstd_b[:] = self.policy.log_std.detach()
        for t in range(T):
            noise = torch.randn((self.n_envs_global, self.policy.log_std.numel()), dtype=pol_dtype, device=dev,
                                generator=self.gen)[self.env_id0:self.env_id0 + N]
            self.noise_step += 1
            if fused is not None:
                # float32 view of the observation, mean network, noise and the normalize() action map in ONE launch (csrc/tu_trpo.hip),
                # written straight into this step's rows of the batch buffers
                fused(self.obs, noise, obs_b[t], mean_b[t], act_b[t])
                nobs, rew, done = self.env_step(self._env_actions)
            else:
                o = self.obs.to(pol_dtype)
                a, mean, log_std = self.policy.get_actions(o, noise=noise)
                nobs, rew, done = self.env_step(self.act_map(a))
                obs_b[t], act_b[t], mean_b[t], lstd_b[t] = o, a, mean, log_std
            if book is not None and rew.dtype == torch.float64 and done.dtype == torch.uint8:
                # clocks, returns, truncation and episode statistics of this step in ONE launch (CassieTrpoSamplerStep)
                book(rew, done, rew_b[t], t_b[t], done_b[t])
                ep += self._book_partial.sum(0)
                cut = done_b[t]
            else:
                done = done.bool().clone()
                rew_b[t], t_b[t] = rew, self.path_t
                self.path_ret += rew
                self.path_t += 1
                cut = done | (self.path_t >= self.max_path_length)  # rllab truncates paths at max_path_length
                done_b[t] = cut
                ep[0] += cut.sum()
                ep[1] += torch.where(cut, self.path_ret, torch.zeros_like(self.path_ret)).sum()
                self.path_ret = torch.where(cut, torch.zeros_like(self.path_ret), self.path_ret)
                self.path_t = torch.where(cut, torch.zeros_like(self.path_t), self.path_t)
            self._steps_to_trunc -= 1
            if self.env_reset_masked is not None and self._steps_to_trunc <= 0:
                # a path may have been truncated at max_path_length while its env is still alive: rllab resets the env there
                trunc = cut & ~done.bool()
                if bool(trunc.any()):
                    nobs = self.env_reset_masked(trunc.to(torch.uint8))
                self._steps_to_trunc = self.max_path_length - int(self.path_t.max())  # re-arm from the oldest live path
            self.obs = nobs.clone()
        return dict(obs=obs_b, act=act_b, mean=mean_b, log_std=lstd_b, rew=rew_b, done=done_b, t=t_b,
                    episode_count=ep[0], episode_return_sum=ep[1])

    def _baseline_kernels(self, obs):
        """BaselineKernels for this batch (float32 CUDA observations of a supported width, LinearFeatureBaseline) or None."""
        if not getattr(self, "fused_baseline", True) or not obs.is_cuda or obs.dtype != torch.float32 or type(self.baseline) is not LinearFeatureBaseline:
            return None
        bk = getattr(self, "_bk", None)
        if bk is None or bk.D != obs.shape[-1] or bk.dev != obs.device:
            try:
                bk = BaselineKernels(obs.device, obs.shape[-1])
            except (ValueError, OSError):
                bk = None
            self._bk = bk
        return bk

    def process(self, batch):
        T, N = batch["rew"].shape
        flat = lambda x: x.reshape(T * N, *x.shape[2:])
        obs, tt = flat(batch["obs"]), flat(batch["t"])
        bk = self._baseline_kernels(obs)
        if bk is not None:
            # value of every sample, return-to-go and advantage in one pass over the batch, the regression's normal equations on the
            # FP64 matrix cores (csrc/tu_trpo_baseline.hip): no feature matrix in memory
            coeffs = self.baseline.coeffs
            last_v = None if coeffs is None else bk.predict(self.obs.to(obs.dtype), self.path_t, coeffs)
            returns, adv, sums = bk.returns_advantages(batch["obs"], batch["t"], batch["rew"], batch["done"], coeffs, last_v, self.discount)
            adv = flat(adv)
            n = torch.tensor([adv.numel()], dtype=torch.float64, device=adv.device)
            s12 = all_sum_(sums.clone(), "advantage_all_reduce"); n = all_sum_(n, "advantage_all_reduce")
            mean = s12[0] / n
            std = (s12[1] / n - mean * mean).clamp_min(0).sqrt()
            adv = ((adv - mean) / (std + 1e-8)).to(obs.dtype)  # center_adv
            A, b = bk.gram(obs, tt, flat(returns))
            self.baseline.fit_normal_equations(A, b, bk.ridge_solve if getattr(self, "fused_solve", True) else None)
            return dict(obs=obs, act=flat(batch["act"]), mean=flat(batch["mean"]), log_std=flat(batch["log_std"]), adv=adv)
        # bootstrap unfinished paths with the baseline of the next observation (0 at iteration 0)
        last_v = self.baseline.predict(self.obs.to(obs.dtype), self.path_t)
        returns = discounted_returns(batch["rew"], batch["done"], self.discount, last_v)
        values = self.baseline.predict(obs, tt).view(T, N)
        adv = flat(returns - values)                       # gae_lambda = 1
        n = torch.tensor([adv.numel()], dtype=torch.float64, device=adv.device)
        s1 = all_sum_(adv.sum().view(1).clone(), "advantage_all_reduce"); s2 = all_sum_((adv * adv).sum().view(1).clone(), "advantage_all_reduce"); n = all_sum_(n, "advantage_all_reduce")
        mean = s1 / n
        std = (s2 / n - mean * mean).clamp_min(0).sqrt()
        adv = ((adv - mean) / (std + 1e-8)).to(obs.dtype)  # center_adv
        self.baseline.fit(obs, tt, flat(returns))
        return dict(obs=obs, act=flat(batch["act"]), mean=flat(batch["mean"]), log_std=flat(batch["log_std"]), adv=adv)

    # ---- constrained update
    def optimize(self, d):
        pol = self.policy
        obs, act, adv, old_mean, old_lstd = d["obs"], d["act"], d["adv"], d["mean"], d["log_std"]
        old_ll = None

        def surrogate():
            nonlocal old_ll
            if old_ll is None:
                old_ll = pol.log_likelihood(act, old_mean, old_lstd)
            mean, log_std = pol.dist_info(obs)
            lr = (pol.log_likelihood(act, mean, log_std) - old_ll).exp()
            return -(lr * adv).mean(), pol.kl(old_mean, old_lstd, mean, log_std).mean()

        # Fisher-vector products: closed form for the tanh-MLP Gaussian policy -- FusedFisher = the product as ONE launch on the matrix
        # cores (csrc/tu_trpo.hip: 0.16 ms against 0.66 ms per product for the torch operations of AnalyticFisher at 524 288 samples,
        # r04), AnalyticFisher where the kernel does not apply (CPU, other shapes); otherwise double backprop through ONE graph of
        # grad(KL) (the KL and its gradient do not depend on v: only the second backward pass is repeated per product)
        gk = kl0 = None
        fisher = None
        if getattr(self, "analytic_fisher", True):
            for cls in ((FusedFisher, AnalyticFisher) if getattr(self, "fused_fisher", True) else (AnalyticFisher,)):
                try:
                    fisher = cls(pol, obs)
                    break
                except (ValueError, OSError):
                    fisher = None
        if fisher is not None:
            # policy gradient in closed form too: at theta = theta_old the likelihood ratio is 1, so with z = (a - mean) / std
            #   d loss / d mean = -adv z / std / N,   d loss / d log_std = -sum_s adv (z^2 - 1) / N,   loss = -mean(adv)
            # and J' (d loss / d mean) comes from the Fisher object's reverse pass (r04: 3.9 ms of autograd -> 0.5 ms at 524 288 samples).
            # ASSUMES the batch is exactly on-policy -- old_mean / old_lstd were produced by the CURRENT parameters (true for this
            # sampler: optimize() runs right after the rollout that recorded them) -- and a state-independent log_std; a caller that
            # reuses an older batch must set analytic_fisher = False (the autograd gradient below makes neither assumption).
            with torch.no_grad():
                std = old_lstd.exp()
                z = (act - old_mean) / std
                n_inv = 1.0 / obs.shape[0]
                g = fisher.vjp(-(adv.unsqueeze(-1) * z / std) * n_inv)
                g_ls = -((adv.unsqueeze(-1) * (z * z - 1.0)).sum(0)) * n_inv
                i0 = 0
                for nm, p_ in pol.named_parameters():
                    if nm == "log_std":
                        g[i0:i0 + p_.numel()] += g_ls.to(g.dtype)
                    i0 += p_.numel()
                loss = -adv.mean()
            g = all_mean_(g, "gradient_all_reduce")
        else:
            loss, _ = surrogate()
            g = all_mean_(flat_grad(loss, pol), "gradient_all_reduce")
            _, kl0 = surrogate()
            gk = flat_grad(kl0, pol, retain_graph=True, create_graph=True)

        def Fvp(v):
            hv = fisher(v) if fisher is not None else flat_grad(gk @ v, pol, retain_graph=True)
            return all_mean_(hv, "fvp_all_reduce") + self.reg_coeff * v

        descent = fisher.conjugate_gradient(g, self.cg_iters, self.reg_coeff) if isinstance(fisher, FusedFisher) and getattr(self, "fused_cg", True) else None
        if descent is None:
            descent = conjugate_gradient(Fvp, g, self.cg_iters)
        shs = 0.5 * (descent @ Fvp(descent))
        if isinstance(fisher, FusedFisher):   # the line search evaluates loss and KL in one launch each (CassieTrpoSurrogate)
            ff, old_ls_vec = fisher, old_lstd[0].detach().clone()
            surrogate = lambda: ff.surrogate(pol, act, adv, old_mean, old_ls_vec)
        del gk, kl0, fisher
        step = torch.sqrt(self.step_size / (shs + 1e-8)) * descent
        if not torch.isfinite(step).all():
            return dict(loss_before=float(loss), loss_after=float(loss), kl=0.0, backtracks=-1)
        theta = flat_params(pol)
        loss_before = float(all_mean_(loss.detach().clone().view(1), "line_search_all_reduce"))
        for k in range(self.max_backtracks + 1):
            set_flat_params(pol, theta - (self.backtrack_ratio ** k) * step)
            with torch.no_grad():
                l_new, kl_new = surrogate()
            l_new, kl_new = all_mean_(torch.stack([l_new.detach().double().reshape(()), kl_new.detach().double().reshape(())]), "line_search_all_reduce").tolist()   # one read-back
            if math.isfinite(l_new) and l_new < loss_before and kl_new <= self.step_size:
                return dict(loss_before=loss_before, loss_after=l_new, kl=kl_new, backtracks=k)
        set_flat_params(pol, theta)  # line search failed: keep the old policy
        return dict(loss_before=loss_before, loss_after=loss_before, kl=0.0, backtracks=self.max_backtracks + 1)

    def train_iteration(self):
        from . import rollout as R
        timing = getattr(self, "timing", False)  # synchronising timers: rollout (policy forward + Env.step) vs TRPO update
        if timing:
            import time
            torch.cuda.synchronize(); t0 = time.perf_counter()
        batch = self.collect()
        if timing:
            torch.cuda.synchronize(); t1 = time.perf_counter()
        stats = self.optimize(self.process(batch))
        if timing:
            torch.cuda.synchronize(); t2 = time.perf_counter()
            stats.update(seconds_rollout=t1 - t0, seconds_update=t2 - t1)
        cnt = all_sum_(torch.stack([batch["episode_count"], batch["episode_return_sum"]]), "stats_all_reduce")
        per_env = batch["rew"].sum(0)                       # the one gather of the rollout batch (N per rank)
        stats.update(itr=self.itr, env_steps=batch["rew"].numel() * _world(), episodes=int(cnt[0]),
                     avg_return=float(cnt[1] / cnt[0]) if cnt[0] > 0 else float("nan"),
                     avg_reward=float(all_mean_(batch["rew"].mean().view(1).clone(), "stats_all_reduce")),
                     gathered=int(self._gather_returns(per_env).numel()))
        self.itr += 1
        return stats

    @staticmethod
    def _gather_returns(per_env):
        from . import rollout as R
        if COMM is None or _world() == 1:
            return R.gather_returns(per_env)
        box = []
        COMM.run("returns_all_gather", lambda: box.append(R.gather_returns(per_env)), per_env)
        return box[0]

    # ---- snapshot_mode="last" (trpo_cassie.py:9-19,50; sim_policy.py:19-23): everything a resumed run needs to BE the
    # interrupted run -- policy, baseline, iteration, and per rank the sampler state (action-noise generator, current
    # observation, path clocks/returns) and the resident env state records.  Rank 0 writes `path`, rank r > 0 `path.rank<r>`.
    def _rank_path(self, path):
        r = dist.get_rank() if dist.is_initialized() else 0
        return path if r == 0 else "%s.rank%d" % (path, r)

    def save(self, path, extra=None):
        """Tensors and plain Python values only (loaded with weights_only=True), written to `path.tmp` and renamed into place so
        that a crash during the write never corrupts the one snapshot_mode='last' file."""
        import io
        import os
        env = getattr(self, "env", None)
        if extra is not None:   # `extra` must survive the weights_only load of load(): refuse at save time what could not be read back
            buf = io.BytesIO()
            torch.save(dict(extra=extra), buf)
            buf.seek(0)
            try:
                torch.load(buf, map_location="cpu", weights_only=True)
            except Exception as ex:
                raise TypeError("TRPO.save: `extra` must be tensors / plain Python values (weights_only round trip failed: %r)" % (ex,))
        n_global = self.n_envs * (dist.get_world_size() if dist.is_initialized() else 1)
        ck = dict(n_envs_global=int(n_global), policy={k: v.detach().cpu() for k, v in self.policy.state_dict().items()},
                  baseline=None if self.baseline.coeffs is None else self.baseline.coeffs.detach().cpu(), itr=int(self.itr), extra=extra,
                  noise_step=int(self.noise_step), noise_seed=int(self.seed), gen_state=self.gen.get_state(), obs=None if self.obs is None else self.obs.cpu(),
                  path_t=self.path_t.cpu(), path_ret=self.path_ret.cpu(), steps_to_trunc=int(self._steps_to_trunc),
                  env_state=None if env is None or not hasattr(env, "get_full_state_host") else torch.from_numpy(env.get_full_state_host()))
        mine = self._rank_path(path)
        torch.save(ck, mine + ".tmp")
        os.replace(mine + ".tmp", mine)

    def load(self, path, restore_sampler=True):
        """Returns (extra, sampler_restored): policy / baseline / iteration always come back; the sampler state (env records,
        noise counter, observation, path clocks) only from this rank's own file with a matching batch shape -- the second value
        says whether that happened, so a run that merely LOOKS resumed can be told from the interrupted run continued."""
        import os
        dev = next(self.policy.parameters()).device
        mine = self._rank_path(path)
        ck = torch.load(mine if os.path.exists(mine) else path, map_location="cpu", weights_only=True)
        self.policy.load_state_dict(ck["policy"])
        self.baseline.coeffs = None if ck["baseline"] is None else ck["baseline"].to(dev)
        self.itr = ck["itr"]
        env = getattr(self, "env", None)
        restored = False
        n_global = self.n_envs * (dist.get_world_size() if dist.is_initialized() else 1)
        # the sampler comes back only if the snapshot carries all of it and was written by a job of the same shape: the noise stream is
        # drawn over the job's GLOBAL env range, so a different world size would silently change it (snapshots written before the
        # noise counter existed have no "noise_step": policy / baseline only, restored = False)
        complete = all(k in ck for k in ("noise_step", "obs", "path_t", "path_ret")) and ck.get("n_envs_global", n_global) == n_global
        if restore_sampler and complete and os.path.exists(mine) and ck.get("env_state") is not None and env is not None \
                and tuple(ck["env_state"].shape) == (self.n_envs, 88):
            env.set_full_state_host(ck["env_state"].numpy())
            self.noise_step, self.seed = ck["noise_step"], ck.get("noise_seed", self.seed)
            if ck.get("gen_state") is not None:
                self.gen.set_state(ck["gen_state"])
            self.obs = None if ck["obs"] is None else ck["obs"].to(dev)
            self.path_t, self.path_ret = ck["path_t"].to(dev), ck["path_ret"].to(dev)
            self._steps_to_trunc = ck.get("steps_to_trunc", 0)
            restored = True
        self.sampler_restored = restored
        return ck.get("extra"), restored


def broadcast_initial_policy(algo):
    """Rank 0's initial parameters are authoritative (a collective: every rank must call it)."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        theta = flat_params(algo.policy)
        dist.broadcast(theta, 0)
        set_flat_params(algo.policy, theta)


def make_cassie_trpo(n_envs, kind="walk", control_mode="PD", device=0, trajectory=None, seed=1, sync_policy=True, **kw):
    """trpo_cassie.py:12-42 on the batched MI355X environment.  sync_policy=False: NOTHING collective happens in here (the env, its workspaces, the
    policy are local allocations that can fail on one rank alone); the caller agrees on success across ranks first and then calls
    broadcast_initial_policy(algo) (bench.py's TRPO stage)."""
    from .vec_env import CassieVecEnv
    env = CassieVecEnv(n_envs, kind=kind, control_mode=control_mode, n_substeps=10, auto_reset=True, device=device, trajectory=trajectory)
    env.use_torch_stream()
    dev = "cuda:%d" % device
    bufs = env.alloc()
    torch.manual_seed(seed)  # trpo_cassie.py:53 seed=1: every rank builds the same initial policy ...
    obs_w = env.observation_space.shape[0]  # what step() emits (26 for both kinds); the policy is sized from the env, as trpo_cassie.py does through env.spec
    policy = GaussianMLPPolicy(obs_w, env.adim, (32, 32), init_std=2.0).to(dev)
    act_map = NormalizedActions(env.action_space.low, env.action_space.high, dev)
    algo = TRPO(lambda a: env.step(a, bufs), lambda: env.reset(bufs), policy, LinearFeatureBaseline(), n_envs, obs_w, act_map, seed=seed,
                env_reset_masked=lambda m: env.reset(bufs, mask=m), **kw)
    algo.env = env
    if sync_policy:   # ... and rank 0's parameters are authoritative anyway
        broadcast_initial_policy(algo)
    return algo

#!/usr/bin/env python3
"""The TRPO loop's Fisher-vector product kernel alone (not a test): 524 288 samples, the 26-32-32-6 policy, N products -- the workload behind
the PMC figures of the matrix-core kernel (profiles/collect_pmc_trpo.sh).  usage: python tools/prof_fvp.py [products]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cassierl_amd import trpo as T
torch.manual_seed(4)
n, reps = 524288, int(sys.argv[1]) if len(sys.argv) > 1 else 20
pol = T.GaussianMLPPolicy(26, 6, (32, 32), init_std=2.0).cuda()
obs = torch.randn(n, 26, device="cuda")
v = torch.randn(sum(p.numel() for p in pol.parameters()), device="cuda")
F = T.FusedFisher(pol, obs)
for _ in range(3): F.mean_product(v)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(reps): F.mean_product(v)
torch.cuda.synchronize()
print("ms per product (kernel + row sum): %.4f" % ((time.perf_counter() - t0) / reps * 1e3))

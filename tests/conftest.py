import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):   # tools/: planar_proto.py, the executable spec of the kernel math
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # always print the slowest tests: the driver's log of `pytest -m gpu` then says where a slow box spent its time
    if getattr(config.option, "durations", None) is None:
        config.option.durations = 20
        config.option.durations_min = 1.0


@pytest.fixture(params=["g16", "leg", "duo"])
def vec_tier(request):
    """CassieVecEnv with the FIRST physics tier pinned: "g16" = four environments per wavefront (what a small batch gets by the
    size rule), "leg" = the two-lanes-per-environment kernel (`env_step_leg_kernel`, the kernel behind the bench headline, which
    the size rule only selects from 6144 environments up).  The oracle / golden-stream tests take this fixture so that the
    driver's plain `pytest -m gpu` run pins BOTH against the oracle, whatever the batch size of the test.  "duo" (r05) = that tier in its
    64-environments-per-wavefront form (`env_step_duo_kernel`: the kernel behind the headline: above 32 768 environments where it saves whole rounds of the chip, `CassieVecEnv.tier_info()`)."""
    from cassierl_amd.vec_env import CassieVecEnv, LEG_TIER_OFF, LEG_TIER_ON, DUO_TIER_ON, DUO_TIER_OFF
    add = {"leg": LEG_TIER_ON | DUO_TIER_OFF, "duo": LEG_TIER_ON | DUO_TIER_ON, "g16": LEG_TIER_OFF}[request.param]

    def make(*a, flags=0, **k):
        return CassieVecEnv(*a, flags=flags | add, **k)
    make.tier = request.param
    return make


@pytest.fixture(scope="session")
def traj():
    d = np.load(os.path.join(GOLDEN, "traj2d.npz"))
    return {k: d[k] for k in d.files}


@pytest.fixture(scope="session")
def streams():
    d = np.load(os.path.join(GOLDEN, "env_streams.npz"))
    return {k: d[k] for k in d.files}


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle_py
    oracle_py.build()
    return oracle_py


def state_vec(q, v, ws=None, kq=None, kv=None, ctrl=None, qstate=None, time=0.0):
    """88-double resident state record (cassie_vec_layout.h)."""
    s = np.zeros(88)
    s[0:13], s[13:26] = q, v
    if ws is not None:
        s[26:39] = ws
    s[39:52] = q if kq is None else kq
    s[52:65] = v if kv is None else kv
    if qstate is not None:
        s[65:78] = qstate
    if ctrl is not None:
        s[78:84] = ctrl
    s[84] = time
    return s

"""Build guard of the 64-environments-per-wavefront kernels (Weak 3): the same source compiled three times -- as shipped, with every
automatic variable pre-set to a byte pattern (-ftrivial-auto-var-init=pattern) and to zero (=zero) -- must give BIT-IDENTICAL rollouts on the
GPU.  A device-only read of an undefined value, or a register-allocation / spill accident of this 512-register kernel (the r05 experiment whose
device build depended on dead code: DESIGN.md section 5 K1d), shows up here as a difference between the builds.  The guard libraries are built in
the container by __graft_entry__.build() (cassierl_amd.build.build_guards) and travel with the tree; if they are missing they are built here."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rollout(lib, out, extra_flags=0):
    env = dict(os.environ)
    env.pop("CASSIE2D_DUO", None); env.pop("CASSIE2D_LEG", None)
    env["GUARD_EXTRA_FLAGS"] = hex(extra_flags)
    if lib:
        env["CASSIE2D_LIB"] = lib
    else:
        env.pop("CASSIE2D_LIB", None)
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "guard_rollout.py"), out], check=True, env=env, timeout=600)
    d = np.load(out)
    return {k: d[k] for k in d.files}


def _same_rollouts(ref, got, name):
    for k in ref:
        if k.startswith("ms_"):
            continue
        assert np.array_equal(ref[k], got[k], equal_nan=True), (name, k, np.argwhere(ref[k] != got[k])[:5].tolist())


def test_guard_builds_are_bit_identical(tmp_path):
    """shipped == auto-var-init pattern == auto-var-init zero == the r05 dead-code experiment recreated (branch never taken) == that branch taken.
    Every one of these builds went through the ISA guard (cassierl_amd/isa_guard.py); the pattern build of tu_duo as the compiler emits it, which the
    guard flags (a copy of the lane number ahead of an exec restore in env_step_duo_kernel<1>), must show the failure the guard exists for."""
    from cassierl_amd import build as B
    from cassierl_amd import isa_guard as G
    libs = B.build_guards()
    assert set(libs) == {"avi_pattern", "avi_zero", "view", "avi_pattern_raw"} and all(os.path.exists(p) for p in libs.values())
    ref = _rollout(None, str(tmp_path / "shipped.npz"))
    assert ref["stand_tq_cleanup"][0] > 0, "the torque run must include hand-overs to the lower tiers"
    assert ref["walk_pd_done"].all() and not ref["stand_pd_done"].all()
    ms = {"shipped": float(ref["ms_per_65536_env_step"][0])}
    for name in ("avi_pattern", "avi_zero", "view"):
        got = _rollout(libs[name], str(tmp_path / (name + ".npz")))
        _same_rollouts(ref, got, name)
        ms[name] = float(got["ms_per_65536_env_step"][0])
    taken = _rollout(libs["view"], str(tmp_path / "view_taken.npz"), extra_flags=B.DUO_VIEW_FLAG)
    _same_rollouts(ref, taken, "view, branch taken")
    ms["view_taken"] = float(taken["ms_per_65536_env_step"][0])
    print("guard builds, ms per 65 536-env Env.step:", {k: round(v, 4) for k, v in ms.items()})
    assert taken["ms_view_marks"][0] > 0 and got["ms_view_marks"][0] == 0 and ref["ms_view_marks"][0] == 0, "the experiment's branch must run exactly when its flag is set"
    # the known-bad reference
    hits = G.check_object(os.path.join(B.VARDIR, "tu_duo_avi_pattern_raw.o"))
    if hits:
        raw = _rollout(libs["avi_pattern_raw"], str(tmp_path / "raw.npz"))
        wrong = [k for k in ref if not k.startswith("ms_") and not np.array_equal(ref[k], raw[k], equal_nan=True)]
        print("unguarded pattern build (%d exec hole(s): %s): differs from the shipped build in %s" % (len(hits), hits[0][2], wrong))
        assert any(k.startswith("stand_tq") for k in wrong), "the build the ISA guard flags ran correctly: the guard's premise no longer holds"

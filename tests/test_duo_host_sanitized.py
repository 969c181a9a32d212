"""The CPU emulation of the two kernel forms (oracle/leg_host: cassie_leg_core.h + cassie_duo_core.h, the source of the HIP kernels) under
-fsanitize=address,undefined: the slot routing of the 64-environments form's cold storage, its workspace indices, the rows of a group that does
not run, partly empty groups -- an out-of-range slot or a read of something never written shows here, not as a wrong number on the GPU.
Exercised in a child process (the sanitizer runtime has to be loaded before python's own allocations).
The instrumented build of the two template cores takes ~6.5 minutes of g++, so the test only runs when CASSIE_SANITIZE=1 (or the library is
already there); the r05 run is recorded in profiles/r05_sanitized.txt."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

ORACLE = os.path.join(ROOT, "oracle")

DRIVER = r"""
import ctypes as ct, sys
import numpy as np
sys.path.insert(0, %(oracle)r); sys.path.insert(0, %(tests)r); sys.path.insert(0, %(root)r)
import leg_host as LH
import oracle_py as O
from conftest import state_vec
from cassierl_amd import terrain as T
L = ct.CDLL(sys.argv[1])
L.leg_host_ops.restype = ct.c_double
LH._LIB = L
o = O.Oracle()
q, v = o.state()
rng = np.random.default_rng(4)
TQ = np.array([12.0, 12.0, 0.9] * 2)
push = np.array([12.2, -12.2, 0.9, 12.2, -12.2, 0.9])
for hf in (False, True):
    for n in (1, 5):
        pair, duo = LH.LegHostEnv(n, kind="stand", control_mode="Torque", n_substeps=10, auto_reset=False), LH.LegHostEnv(n, duo=True, kind="stand", control_mode="Torque", n_substeps=10, auto_reset=False)
        s0 = np.tile(state_vec(q, v, o.warmstart(), qstate=q), (n, 1))
        s0[:, 2] += rng.uniform(-0.05, 0.05, n)
        for e in (pair, duo):
            if hf:
                e.set_heightfield(T.ramp(nrow=16, ncol=401, size_x=10.0, slope=0.1, x0=0.5), 10.0, 10.0)
            e.set_full_state_host(s0)
        for t in range(100 if not hf else 30):
            a = np.tile(push, (n, 1)) * (1 + 0.2 * rng.uniform(-1, 1, (n, 6))) if t < 80 else np.zeros((n, 6))
            rp, rd = pair.step_host(a), duo.step_host(a)
            stay = pair.pending == 0
            assert np.array_equal(pair.pending, duo.pending)
            sp, sd = pair.get_full_state_host(), duo.get_full_state_host()
            assert np.array_equal(sp[stay], sd[stay], equal_nan=True), (hf, n, t)
            if (~stay).any():
                sp[~stay] = s0[~stay]
                pair.set_full_state_host(sp); duo.set_full_state_host(sp)
print("RESULT ok")
"""


def test_kernel_source_emulation_is_clean_under_asan_and_ubsan(oracle_mod):
    lib = os.path.join(ORACLE, "libleg_host_asan.so")
    if os.environ.get("CASSIE_SANITIZE") == "1":
        subprocess.check_call(["make", "-s", "-C", ORACLE, "libleg_host_asan.so"])
    elif not os.path.exists(lib):
        pytest.skip("instrumented build not present (CASSIE_SANITIZE=1 builds it: ~6.5 min)")
    asan_rt = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1", LD_PRELOAD=asan_rt)
    p = subprocess.run([sys.executable, "-c", DRIVER % dict(oracle=ORACLE, tests=os.path.join(ROOT, "tests"), root=ROOT), os.path.join(ORACLE, "libleg_host_asan.so")],
                       env=env, capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-4000:])
    assert "runtime error" not in p.stderr and "AddressSanitizer" not in p.stderr, p.stderr[-4000:]
    assert "RESULT ok" in p.stdout

"""The oracle is the checker, so it has to be clean itself: its physics (flat floor and height field), OSC and Jacobian paths and the env
layer run under -fsanitize=address,undefined (oracle/Makefile `asan`), and the -O3 -march=native timing build that bench.py's
cpu_baseline leg binds runs the same sequence and agrees with the parity build (r03 routed the CPU legs around an alignment fault of
that build; the cause -- a stack array whose alignment gcc assumed but did not provide -- is fixed in cassie_oracle_ctrl.inc).
Each library is exercised in a child process (the sanitizer runtime has to be loaded before python's own allocations)."""
import json
import os
import subprocess
import sys

import numpy as np

from conftest import ROOT

ORACLE = os.path.join(ROOT, "oracle")

DRIVER = r'''
import ctypes as ct, json, sys
import numpy as np
sys.path.insert(0, %(oracle)r)
import oracle_py as O
L = ct.CDLL(sys.argv[1])
L.orc_create.restype = ct.c_void_p; L.orc_env_create.restype = ct.c_void_p; L.orc_env_oracle.restype = ct.c_void_p
L.orc_energy.restype = ct.c_double; L.orc_env_time.restype = ct.c_double
O._LIB = L
rng = np.random.default_rng(0)
qinit = np.array([0, 0.939, 0, 0.68111815, -1.40730357, 1.62972042, -1.77611107, -0.61968407, 0.68111815, -1.40730357, 1.62972042, -1.77611107, -0.61968407])
o = O.Oracle()
o.reset(qinit, np.zeros(13))
out = {}
for t in range(120):
    a = rng.uniform(-1, 1, 7) * np.array([3, 3, 1, 1, 1, 1, 3.0]); a[3], a[5] = abs(a[3]), abs(a[5])
    o.step_osc(a)
out["osc"] = o.state()[0].tolist()
for t in range(120):
    o.step_jacobian(np.array([0, 150, 0, 0, 150, 0.0]) + rng.uniform(-5, 5, 6))
out["jac"] = o.state()[0].tolist()
for t in range(400):   # robots fall: joint limits, many contacts
    o.step_torque(rng.uniform(-1, 1, 6) * np.array([12, 12, .9] * 2))
for t in range(100):
    o.step_pd(rng.uniform(-1, 1, 6))
out["phys"] = o.state()[0].tolist()
hm = np.tile(0.02 * np.sin(np.linspace(-10, 10, 401) * 3.0), (8, 1))
o2 = O.Oracle(); o2.set_hfield(hm, 10.0, 10.0)
for t in range(300):
    o2.step_torque(rng.uniform(-1, 1, 6) * np.array([12, 12, .9] * 2))
out["terrain"] = o2.state()[0].tolist()
e = O.OracleEnv("stand", "Torque"); e.reset()
for t in range(30):
    ob, r, d = e.step(rng.uniform(-1, 1, 6) * np.array([12, 12, .9] * 2))
    if d: e.reset()
out["env"] = [float(r)] + ob.tolist()
print("RESULT " + json.dumps(out))
'''


def _run(lib, preload=None):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    if preload:
        env["LD_PRELOAD"] = preload
    p = subprocess.run([sys.executable, "-c", DRIVER % dict(oracle=ORACLE), lib], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-4000:])
    assert "runtime error" not in p.stderr and "AddressSanitizer" not in p.stderr, p.stderr[-4000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1]
    return {k: np.array(v) for k, v in json.loads(line[7:]).items()}


def test_oracle_sanitized(oracle_mod):
    subprocess.check_call(["make", "-s", "-B", "-C", ORACLE, "asan"])
    asan_rt = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    san = _run(os.path.join(ORACLE, "liboracle_asan.so"), preload=asan_rt)
    ref = _run(oracle_mod.build())
    for k in ref:   # -O1 against -O2, both with contraction off: same roundings up to libm
        assert np.abs(san[k] - ref[k]).max() < 1e-9, k


def test_oracle_timing_build_runs_the_same_sequence(oracle_mod):
    """liboracle_fast.so (-O3 -march=native, contraction on: bench.py's cpu_baseline leg) survives the OSC / Jacobian / physics / terrain
    / env sequence -- r03's build died in orc_step_osc -- and stays close to the parity build over the short controller horizons."""
    subprocess.check_call(["make", "-s", "-B", "-C", ORACLE, "liboracle_fast.so"])
    fast = _run(os.path.join(ORACLE, "liboracle_fast.so"))
    ref = _run(oracle_mod.build())
    assert np.abs(fast["osc"] - ref["osc"]).max() < 1e-6 and np.abs(fast["jac"] - ref["jac"]).max() < 1e-5
    assert all(np.isfinite(v).all() for v in fast.values())

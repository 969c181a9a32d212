"""CHECKER / CPU-BASELINE INFRASTRUCTURE: ctypes view of oracle/leg_host/leg_host.cpp -- the source of the two-lanes-per-environment HIP kernel
(cassierl_amd/csrc/cassie_leg_core.h) compiled for the CPU with a lane-pair emulation -- behind the subset of the
CassieVecEnv interface the parity tests use, so that the same test bodies run against it without a GPU."""
import ctypes as ct
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
MODES = {"PD": 0, "Torque": 1, "Record": 2}
_LIB = None
_FAST = None
dp = ct.POINTER(ct.c_double)


def _srcs():
    c = os.path.join(ROOT, "cassierl_amd", "csrc")
    return [os.path.join(HERE, "leg_host", "leg_host.cpp")] + [os.path.join(c, f) for f in (
        "cassie_leg_core.h", "cassie_duo_core.h", "cassie2d_planar.h", "cassie2d_legk.h", "cassie_vec_layout.h", "cassie_terrain.h")]


def lib(fast=False):
    """libleg_host.so (parity / op-counting build) or, fast=True, libleg_host_fast.so (bench.py's same-source CPU leg, made on this
    box with -march=native)."""
    global _LIB, _FAST
    if fast:
        if _FAST is None:
            subprocess.check_call(["make", "-s", "-B", "-C", HERE, "libleg_host_fast.so"])
            _FAST = ct.CDLL(os.path.join(HERE, "libleg_host_fast.so"))
        return _FAST
    if _LIB is None:
        import oracle_py
        _LIB = ct.CDLL(oracle_py.make("libleg_host.so", _srcs()))
        _LIB.leg_host_ops.restype = ct.c_double
    return _LIB


def _p(a):
    return a.ctypes.data_as(dp) if a is not None else None


class LegHostEnv:
    def __init__(self, n, kind="stand", control_mode="Torque", n_substeps=10, auto_reset=True, flags=0, duo=False):
        self.duo = duo   # the 64-environments-per-wavefront form of the kernel (cassie_duo_core.h) instead of the two-lanes one
        self.n, self.kind, self.mode, self.n_sub, self.auto_reset, self.flags = n, kind, control_mode, n_substeps, auto_reset, flags
        self.adim = 6
        self.state = np.zeros((n, 88))
        self.traj_q, self.traj_tmax, self.traj_n = None, 0.0, 0
        self.pending = np.zeros(n, dtype=np.int32)
        self.nonfinite = 0

    def set_trajectory(self, time, qpos):
        self.traj_q = np.ascontiguousarray(qpos, dtype=np.float64)
        self.traj_tmax, self.traj_n = float(time[-1]), len(time)

    def set_heightfield(self, heights_m, size_x=10.0, size_y=10.0):
        self.hf = None if heights_m is None else (np.ascontiguousarray(heights_m, dtype=np.float64), float(size_x), float(size_y))

    def set_full_state_host(self, s):
        self.state = np.ascontiguousarray(np.asarray(s, dtype=np.float64).reshape(self.n, 88)).copy()

    def get_full_state_host(self):
        return self.state.copy()

    def _call(self, mode, acts, n_sub, want_obs):
        acts = None if acts is None else np.ascontiguousarray(acts, dtype=np.float64)
        obs = np.zeros((self.n, 26)) if want_obs else None
        rew = np.zeros(self.n) if want_obs else None
        done = np.zeros(self.n, dtype=np.uint8)
        bad = ct.c_int(0)
        if getattr(self, "hf", None) is not None:
            hm, sx, sy = self.hf
            (lib().leg_host_step_duo_hf if self.duo else lib().leg_host_step_hf)(_p(self.state), _p(acts), self.n, acts.shape[1] if acts is not None else 6, MODES[mode], n_sub, self.flags,
                                   0 if self.kind == "walk" else 1, int(self.auto_reset), _p(hm), hm.shape[0], hm.shape[1], ct.c_double(sx), ct.c_double(sy),
                                   _p(obs), _p(rew), done.ctypes.data_as(ct.POINTER(ct.c_ubyte)) if want_obs else None,
                                   self.pending.ctypes.data_as(ct.POINTER(ct.c_int)), ct.byref(bad), 1)
            self.nonfinite += bad.value
            return obs, rew, done.astype(bool)
        (lib().leg_host_step_duo if self.duo else lib().leg_host_step)(_p(self.state), _p(acts), self.n, acts.shape[1] if acts is not None else 6, MODES[mode], n_sub, self.flags,
                            0 if self.kind == "walk" else 1, int(self.auto_reset), _p(self.traj_q), ct.c_double(self.traj_tmax), self.traj_n,
                            _p(obs), _p(rew), done.ctypes.data_as(ct.POINTER(ct.c_ubyte)) if want_obs else None, None,
                            self.pending.ctypes.data_as(ct.POINTER(ct.c_int)), ct.byref(bad), 1)
        self.nonfinite += bad.value
        return obs, rew, done.astype(bool)

    def substep_host(self, mode, acts, n_sub):
        self._call(mode, acts, n_sub, False)

    def step_host(self, acts):
        return self._call(self.mode, acts, self.n_sub, True)

    def close(self):
        pass

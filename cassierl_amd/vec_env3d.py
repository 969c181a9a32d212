"""Batched Cassie3d physics (model/cassie3d_stiff.xml) over include/cassie3d_vec.h -- BASELINE.json configs[4].

The reference has no Cassie3d environment class (only the MJCF), so this mirrors the torque-mode surface of
`Cassie2d` (src/Cassie2d/Cassie2d.cpp:78-94: Reset, Step, GetGeneralState) with a leading n_envs axis.
No CPU fallback: constructing it loads libcassie2d.so and needs a HIP device.
"""
import ctypes as ct

import numpy as np

from . import _lib

NQ, NV, NU, STATE_STRIDE, DEBUG_STRIDE = 21, 20, 10, 80, 1869
OFF = dict(qpos=0, qvel=21, warmstart=41, ctrl=61, time=71, niter=72, nefc=73, overflow=74)
CTRL_RANGE = np.array([4.5, 4.5, 12.2, 12.2, 0.9] * 2)  # cassie3d_stiff.xml:184-195


class Cassie3dVec:
    def __init__(self, n_envs, device=0):
        self.L = _lib.load()
        self.n_envs, self.device = n_envs, device
        h = ct.c_void_p()
        rc = self.L.Cassie3dVecCreate(ct.byref(h), n_envs, device)
        if rc != 0:
            raise RuntimeError("Cassie3dVecCreate failed (%d): no HIP device / allocation failure; there is no CPU path" % rc)
        self.h = h

    def _chk(self, rc):
        if rc != 0:
            raise RuntimeError("libcassie2d error %d: %s" % (rc, self.L.Cassie3dVecLastError(self.h).decode()))

    def close(self):
        if getattr(self, "h", None):
            self.L.Cassie3dVecFree(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        self._chk(self.L.Cassie3dVecSynchronize(self.h))

    def counters(self):
        """env-substeps requested / done by the 64-row kernel / with the 64-row cap leaving contacts out / handed down by the lane-per-leg kernel."""
        out = (ct.c_uint64 * 4)()
        self._chk(self.L.Cassie3dVecGetCounters(self.h, out))
        req, gen, cap, leg = int(out[0]), int(out[1]), int(out[2]), int(out[3])
        return dict(substeps=req, general_kernel_substeps=gen, capped_substeps=cap, general_frac=(gen / req if req else 0.0),
                    leg_handover_substeps=leg, leg_handover_frac=(leg / req if req else 0.0))

    def reset_counters(self):
        self._chk(self.L.Cassie3dVecResetCounters(self.h))

    # ---- device API (torch tensors are only the allocator here)
    def reset(self, qpos=None, qvel=None):
        self._chk(self.L.Cassie3dVecReset(self.h, None if qpos is None else qpos.data_ptr(), None if qvel is None else qvel.data_ptr()))

    def step(self, torques, n_sub=10):
        assert torques.is_cuda and torques.element_size() == 8 and torques.is_contiguous() and torques.shape == (self.n_envs, NU)
        self._chk(self.L.Cassie3dVecStep(self.h, torques.data_ptr(), n_sub))

    def time_steps(self, torques, n_sub, steps):
        ms = ct.c_float()
        self._chk(self.L.Cassie3dVecTimeSteps(self.h, torques.data_ptr(), n_sub, steps, ct.byref(ms)))
        return ms.value

    # ---- host API (tests)
    def step_host(self, torques, n_sub=1):
        a = np.ascontiguousarray(torques, dtype=np.float64).reshape(self.n_envs, NU)
        self._chk(self.L.Cassie3dVecStepHost(self.h, a.ctypes.data, n_sub))

    def get_state_host(self):
        s = np.empty((self.n_envs, STATE_STRIDE))
        self._chk(self.L.Cassie3dVecGetStateHost(self.h, s.ctypes.data))
        return s

    def set_state_host(self, s):
        s = np.ascontiguousarray(s, dtype=np.float64).reshape(self.n_envs, STATE_STRIDE)
        self._chk(self.L.Cassie3dVecSetStateHost(self.h, s.ctypes.data))

    def debug_forward_host(self, torques):
        a = np.ascontiguousarray(torques, dtype=np.float64).reshape(self.n_envs, NU)
        dbg = np.zeros((self.n_envs, DEBUG_STRIDE))
        self._chk(self.L.Cassie3dVecDebugForwardHost(self.h, a.ctypes.data, dbg.ctypes.data))
        return dbg


def state_record(qpos, qvel, warmstart=None, ctrl=None, time=0.0):
    s = np.zeros(STATE_STRIDE)
    s[0:21], s[21:41] = qpos, qvel
    if warmstart is not None:
        s[41:61] = warmstart
    if ctrl is not None:
        s[61:71] = ctrl
    s[71] = time
    return s

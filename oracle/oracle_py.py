"""ctypes binding of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

May be imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg,
never by the cassierl_amd package (the product has no CPU path).
"""
import ctypes as ct
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
NV, NU = 13, 6
dp = ct.POINTER(ct.c_double)


def content_hash(srcs, tag=""):
    import hashlib
    h = hashlib.sha256(tag.encode())
    for f in srcs:
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def make(target, srcs, force=False):
    """`make <target>` in oracle/ when the library is missing or its sources changed -- decided by the CONTENT of the sources (a stamp
    file beside the library), not by file times: a copy of the tree to a GPU box keeps no promise about mtimes."""
    so = os.path.join(HERE, target)
    want = content_hash(srcs + [os.path.join(HERE, "Makefile")], target)
    try:
        fresh = os.path.exists(so) and open(so + ".stamp").read() == want
    except OSError:
        fresh = False
    if force or not fresh:
        import fcntl
        with open(os.path.join(HERE, ".make.lock"), "w") as lock:   # two pytest processes / ranks may get here together
            fcntl.flock(lock, fcntl.LOCK_EX)
            try:
                fresh = (not force) and os.path.exists(so) and open(so + ".stamp").read() == want   # built while this one waited
            except OSError:
                fresh = False
            if not fresh:
                subprocess.check_call(["make", "-s", "-B", "-C", HERE, target])
                with open(so + ".stamp", "w") as f:
                    f.write(want)
    return so


ORACLE_SRCS = [os.path.join(HERE, f) for f in ("cassie_oracle.c", "cassie_oracle_ctrl.inc", "cassie_oracle_env.inc", "cassie_oracle.h",
                                               "cassie2d_model.h")]


def build(force=False):
    return make("liboracle.so", ORACLE_SRCS, force)


_FAST = False


def use_fast_build():
    """bench.py's cpu_baseline leg only: (re)build the oracle with -O3 -march=native on THIS machine and bind it instead of
    the -O2 -ffp-contract=off parity build.  Must be called before the first lib().  Returns False if that build fails."""
    global _FAST
    assert _LIB is None, "use_fast_build() must come before the oracle library is first used"
    try:
        subprocess.check_call(["make", "-s", "-B", "-C", HERE, "liboracle_fast.so"])   # -march=native: always made on the box that runs it
        _FAST = True
    except Exception:
        _FAST = False
    return _FAST


def lib():
    global _LIB
    if _LIB is None:
        L = ct.CDLL(os.path.join(HERE, "liboracle_fast.so") if _FAST else build())
        L.orc_create.restype = ct.c_void_p
        L.orc_env_create.restype = ct.c_void_p
        L.orc_env_oracle.restype = ct.c_void_p
        L.orc_energy.restype = ct.c_double
        L.orc_env_time.restype = ct.c_double
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(dp)


def _vec(x, n=None):
    a = np.ascontiguousarray(np.asarray(x, dtype=np.float64))
    if n is not None:
        assert a.size == n, (a.size, n)
    return a


class Oracle:
    """One Cassie2d instance (counterpart of the reference's `Cassie2d*` handle)."""

    def __init__(self, handle=None):
        self.L = lib()
        self._own = handle is None
        self.h = ct.c_void_p(self.L.orc_create()) if handle is None else ct.c_void_p(handle)

    def __del__(self):
        if getattr(self, "_own", False) and self.h:
            self.L.orc_free(self.h)
            self.h = None

    def reset(self, qpos, qvel):
        self.L.orc_reset(self.h, _p(_vec(qpos, NV)), _p(_vec(qvel, NV)))

    def step_torque(self, u):
        self.L.orc_step_torque(self.h, _p(_vec(u, NU)))

    def step_pd(self, a):
        self.L.orc_step_pd(self.h, _p(_vec(a, NU)))

    def step_jacobian(self, f):
        self.L.orc_step_jacobian(self.h, _p(_vec(f, 6)))

    def step_osc(self, a):
        self.L.orc_step_osc(self.h, _p(_vec(a, 7)))

    def state(self):
        q, v = np.zeros(NV), np.zeros(NV)
        self.L.orc_get_state(self.h, _p(q), _p(v))
        return q, v

    def opstate(self, flags=0):
        s = np.zeros(18)
        self.L.orc_get_opstate(self.h, ct.c_int(flags), _p(s))
        return s

    def forward(self):
        self.L.orc_forward(self.h)

    def set_state_raw(self, qpos, qvel, ws=None):
        w = _vec(ws, NV) if ws is not None else None
        self.L.orc_set_state_raw(self.h, _p(_vec(qpos, NV)), _p(_vec(qvel, NV)), _p(w) if w is not None else None)

    def set_gravity(self, gz):
        self.L.orc_set_gravity(self.h, ct.c_double(gz))

    def set_damping_scale(self, s):
        self.L.orc_set_damping_scale(self.h, ct.c_double(s))

    def set_contact_enabled(self, e):
        self.L.orc_set_contact_enabled(self.h, ct.c_int(int(e)))

    def set_assumptions(self, mask):
        """ORC_ASSUME_* bits (cassie_oracle.h): 1 smoothstep impedance, 2 norm impedance on connect rows, 4 warm start by mj_step only."""
        self.L.orc_set_assumptions(self.h, ct.c_int(mask))

    def set_hfield(self, heights_m, size_x, size_y):
        """Height-field terrain (metres, [nrow, ncol]); None restores the flat floor.  The array is shared, not copied."""
        if heights_m is None:
            self._hf = None
            self.L.orc_set_hfield(self.h, None, 0, 0, ct.c_double(0), ct.c_double(0))
            return
        self._hf = np.ascontiguousarray(heights_m, dtype=np.float64)
        self.L.orc_set_hfield(self.h, _p(self._hf), ct.c_int(self._hf.shape[0]), ct.c_int(self._hf.shape[1]), ct.c_double(size_x), ct.c_double(size_y))

    def contacts(self):
        n = self.ncon
        dist, pos, frame = np.zeros(n), np.zeros((n, 3)), np.zeros((n, 9))
        self.L.orc_get_contacts(self.h, _p(dist), _p(pos), _p(frame))
        return dict(dist=dist, pos=pos, frame=frame)

    def efc_extra(self):
        n = self.nefc
        R, vel, diag, b = np.zeros(n), np.zeros(n), np.zeros(n), np.zeros(n)
        self.L.orc_get_efc_extra(self.h, _p(R), _p(vel), _p(diag), _p(b))
        return dict(R=R, vel=vel, diagApprox=diag, b=b)

    @property
    def nefc(self):
        return self.L.orc_nefc(self.h)

    @property
    def ncon(self):
        return self.L.orc_ncon(self.h)

    @property
    def solver_niter(self):
        return self.L.orc_solver_niter(self.h)

    def mass_matrix(self, qpos, sem=0):
        M = np.zeros((NV, NV))
        self.L.orc_get_mass_matrix(self.h, ct.c_int(sem), _p(_vec(qpos, NV)), _p(M))
        return M

    def bias(self, qpos, qvel, sem=0):
        b = np.zeros(NV)
        self.L.orc_get_bias(self.h, ct.c_int(sem), _p(_vec(qpos, NV)), _p(_vec(qvel, NV)), _p(b))
        return b

    def qacc(self):
        a = np.zeros(NV)
        self.L.orc_get_qacc(self.h, _p(a))
        return a

    def warmstart(self):
        a = np.zeros(NV)
        self.L.orc_get_warmstart(self.h, _p(a))
        return a

    def ctrl(self):
        a = np.zeros(NU)
        self.L.orc_get_ctrl(self.h, _p(a))
        return a

    def efc(self):
        n = self.nefc
        J, f, pos, aref = np.zeros((n, NV)), np.zeros(n), np.zeros(n), np.zeros(n)
        typ = np.zeros(n, dtype=np.int32)
        self.L.orc_get_efc(self.h, _p(J), _p(f), _p(pos), _p(aref), typ.ctypes.data_as(ct.POINTER(ct.c_int)))
        return dict(J=J, force=f, pos=pos, aref=aref, type=typ)

    def energy(self):
        ke, pe = ct.c_double(), ct.c_double()
        tot = self.L.orc_energy(self.h, ct.byref(ke), ct.byref(pe))
        return tot, ke.value, pe.value

    def site_pos(self, qpos, site, sem=0):
        p = np.zeros(3)
        self.L.orc_site_pos(self.h, ct.c_int(sem), _p(_vec(qpos, NV)), ct.c_int(site), _p(p))
        return p

    def model_consts(self, sem=0):
        a2, dw, bw, mi = np.zeros((2, 3)), np.zeros(NV), np.zeros(22), ct.c_double()
        self.L.orc_get_model_consts(self.h, ct.c_int(sem), _p(a2), _p(dw), _p(bw), ct.byref(mi))
        return dict(eq_anchor2=a2, dof_invweight0=dw, body_invweight0_tran=bw, meaninertia=mi.value)

    def dynamic_state(self):
        M, b, Bt = np.zeros((NV, NV)), np.zeros(NV), np.zeros((NV, NU))
        Jc, Jeq, Jd = np.zeros((12, NV)), np.zeros((6, NV)), np.zeros(6)
        self.L.orc_get_dynamic_state(self.h, _p(M), _p(b), _p(Bt), _p(Jc), _p(Jeq), _p(Jd))
        return dict(M=M, bias=b, Bt=Bt, Jc=Jc, Jeq=Jeq, JeqdotQdot=Jd)

    def osc_qp(self):
        x, k = np.zeros(39), np.zeros(4)
        self.L.orc_get_osc_qp(self.h, _p(x), _p(k))
        return x, k


CTRL = {"PD": 0, "Torque": 1, "OSC": 2}


class OracleEnv:
    """Restatement of rllab/envs/cassie2d.py (kind='walk') / cassie_stand2d.py (kind='stand')."""

    def __init__(self, kind="walk", control_mode="PD", flags=0, traj=None):
        self.L = lib()
        self.adim = 7 if control_mode == "OSC" else 6
        self._tq = self._tt = None
        tq = tt = None
        tn = 0
        if traj is not None:
            self._tq = np.ascontiguousarray(traj["qpos"], dtype=np.float64)
            self._tt = np.ascontiguousarray(traj["time"], dtype=np.float64)
            tq, tt, tn = _p(self._tq), _p(self._tt), len(self._tt)
        self.h = ct.c_void_p(self.L.orc_env_create(ct.c_int(0 if kind == "walk" else 1), ct.c_int(CTRL[control_mode]),
                                                   ct.c_int(flags), tq, tt, ct.c_int(tn)))
        self.oracle = Oracle(self.L.orc_env_oracle(self.h))

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_env_free(self.h)
            self.h = None

    def reset(self):
        obs = np.zeros(26)
        self.L.orc_env_reset(self.h, _p(obs))
        return obs

    def step(self, action, n=10):
        obs, r, d = np.zeros(26), ct.c_double(), ct.c_int()
        self.L.orc_env_step(self.h, _p(_vec(action, self.adim)), ct.c_int(n), _p(obs), ct.byref(r), ct.byref(d))
        return obs, r.value, bool(d.value)

    @property
    def time(self):
        return self.L.orc_env_time(self.h)


def envs_step(envs, actions, n_sub=10, auto_reset=True, nthreads=0):
    """OpenMP batch step over a list of OracleEnv (CPU baseline)."""
    L = lib()
    n = len(envs)
    arr = (ct.c_void_p * n)(*[e.h for e in envs])
    actions = np.ascontiguousarray(actions, dtype=np.float64)
    obs, rew, done = np.zeros((n, 26)), np.zeros(n), np.zeros(n, dtype=np.uint8)
    L.orc_envs_step(arr, ct.c_int(n), _p(actions), ct.c_int(actions.shape[1]), ct.c_int(n_sub), ct.c_int(int(auto_reset)),
                    _p(obs), _p(rew), done.ctypes.data_as(ct.POINTER(ct.c_ubyte)), ct.c_int(nthreads))
    return obs, rew, done


# ---------------------------------------------------------------------------------------------------------------------
# Cassie3d (oracle/liboracle3d.so = the same Part 1/2 pipeline compiled with -DORC_CASSIE3D)
NQ3, NV3, NU3 = 21, 20, 10
_LIB3 = None


def build3d(force=False):
    return make("liboracle3d.so", [os.path.join(HERE, f) for f in ("cassie_oracle.c", "cassie_oracle.h", "cassie3d_model.h")], force)


def lib3d():
    global _LIB3
    if _LIB3 is None:
        L = ct.CDLL(build3d())
        L.orc_create.restype = ct.c_void_p
        L.orc_energy.restype = ct.c_double
        _LIB3 = L
    return _LIB3


class Oracle3D:
    """One Cassie3d mechanism: mj_forward / mj_step of model/cassie3d_stiff.xml in torque mode."""

    def __init__(self):
        self.L = lib3d()
        self.h = ct.c_void_p(self.L.orc_create())

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_free(self.h)
            self.h = None

    def reset(self, qpos, qvel):
        self.L.orc_reset(self.h, _p(_vec(qpos, NQ3)), _p(_vec(qvel, NV3)))

    def step_torque(self, u):
        self.L.orc_step_torque(self.h, _p(_vec(u, NU3)))

    def forward(self):
        self.L.orc_forward(self.h)

    def state(self):
        q, v = np.zeros(NQ3), np.zeros(NV3)
        self.L.orc_get_state(self.h, _p(q), _p(v))
        return q, v

    def set_state_raw(self, qpos, qvel, ws=None):
        w = _vec(ws, NV3) if ws is not None else None
        self.L.orc_set_state_raw(self.h, _p(_vec(qpos, NQ3)), _p(_vec(qvel, NV3)), _p(w) if w is not None else None)

    def set_gravity(self, gz):
        self.L.orc_set_gravity(self.h, ct.c_double(gz))

    def set_damping_scale(self, s):
        self.L.orc_set_damping_scale(self.h, ct.c_double(s))

    def set_contact_enabled(self, e):
        self.L.orc_set_contact_enabled(self.h, ct.c_int(int(e)))

    def set_row_cap(self, cap):
        self.L.orc_set_row_cap(self.h, ct.c_int(cap))

    @property
    def nefc(self):
        return self.L.orc_nefc(self.h)

    @property
    def ncon(self):
        return self.L.orc_ncon(self.h)

    @property
    def solver_niter(self):
        return self.L.orc_solver_niter(self.h)

    def mass_matrix(self, qpos):
        M = np.zeros((NV3, NV3))
        self.L.orc_get_mass_matrix(self.h, ct.c_int(0), _p(_vec(qpos, NQ3)), _p(M))
        return M

    def bias(self, qpos, qvel):
        b = np.zeros(NV3)
        self.L.orc_get_bias(self.h, ct.c_int(0), _p(_vec(qpos, NQ3)), _p(_vec(qvel, NV3)), _p(b))
        return b

    def qacc(self):
        a = np.zeros(NV3)
        self.L.orc_get_qacc(self.h, _p(a))
        return a

    def warmstart(self):
        a = np.zeros(NV3)
        self.L.orc_get_warmstart(self.h, _p(a))
        return a

    def efc(self):
        n = self.nefc
        J, f, pos, aref = np.zeros((n, NV3)), np.zeros(n), np.zeros(n), np.zeros(n)
        typ = np.zeros(n, dtype=np.int32)
        self.L.orc_get_efc(self.h, _p(J), _p(f), _p(pos), _p(aref), typ.ctypes.data_as(ct.POINTER(ct.c_int)))
        return J, f, pos, aref, typ

    def energy(self):
        ke, pe = ct.c_double(), ct.c_double()
        e = self.L.orc_energy(self.h, ct.byref(ke), ct.byref(pe))
        return e, ke.value, pe.value

    def site_pos(self, qpos, site):
        p = np.zeros(3)
        self.L.orc_site_pos(self.h, ct.c_int(0), _p(_vec(qpos, NQ3)), ct.c_int(site), _p(p))
        return p

    def model_consts(self):
        a2, dw, bw, mi = np.zeros((2, 3)), np.zeros(NV3), np.zeros(22), ct.c_double()
        self.L.orc_get_model_consts(self.h, ct.c_int(0), _p(a2), _p(dw), _p(bw), ct.byref(mi))
        return a2, dw, bw, mi.value

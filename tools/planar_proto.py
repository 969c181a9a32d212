"""Executable specification of the PLANAR algorithm the HIP kernels implement.

Test infrastructure only (numpy, one environment, plain loops).  It follows the kernel's
formulation step by step -- sagittal-plane FK, subtree-sum mass matrix / bias, fixed
constraint slots (4 connect rows, 8 limit rows, 17 contact pairs), residual-per-row PGS
with 2-row elliptic contacts -- so that (a) the reduction from the oracle's 3-D/3-rows
formulation is validated on CPU (tests/test_planar_proto.py) and (b) the kernel can be
checked stage by stage against it.  Reference semantics: see oracle/cassie_oracle.h.
"""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
NV, NL, NU = 13, 11, 6
NSLOT = 46
SLOT_EQ, SLOT_LIM, SLOT_CON = 0, 4, 12
MINVAL = 1e-15
LIMIT_DOFS = [3, 4, 5, 6, 8, 9, 10, 11]


def load_tables():
    with open(os.path.join(os.path.dirname(HERE), "tests", "golden", "planar_tables.json")) as f:
        return json.load(f)


def rot(c, s, d):
    return np.array([c * d[0] + s * d[1], -s * d[0] + c * d[1]])


def ycross(r):  # y_hat x (rx, rz) in the plane
    return np.array([r[1], -r[0]])


class Planar:
    def __init__(self, tables=None, sem="mj"):
        T = tables or load_tables()
        self.T = T
        P = T["planar"][sem]
        self.links = P["links"]
        self.sites = P["sites"]
        self.spheres = P["spheres"]
        self.eqs = P["eqs"]
        self.qpos0 = np.array(T["qpos0"])
        self.h = T["option"]["timestep"]
        self.g = -T["option"]["gravity"][2]
        self.iterations = T["option"]["iterations"]
        self.tolerance = T["option"]["tolerance"]
        self.meaninertia = T["meaninertia"]
        self.damping = np.array(T["dof"]["damping"])
        self.armature = np.array(T["dof"]["armature"])
        self.parent = [L["parent"] for L in self.links]
        self.ldof = [L["dof"] for L in self.links]
        self.sigma = np.zeros(NV)
        self.dof_link = [-1] * NV
        for li, L in enumerate(self.links):
            self.sigma[L["dof"]] = L["sigma"]
            self.dof_link[L["dof"]] = li
        self.dof_link[0] = self.dof_link[1] = 0
        # subtree(link) and path dofs
        self.sub = [[l for l in range(NL) if self._is_anc(li, l)] for li in range(NL)]
        self.path = []
        for li in range(NL):
            p, dofs = li, []
            while p >= 0:
                dofs.append(self.ldof[p])
                p = self.parent[p]
            self.path.append(sorted(dofs + [0, 1]))

    def _is_anc(self, a, l):
        while l >= 0:
            if l == a:
                return True
            l = self.parent[l]
        return False

    # ------------------------------------------------------------------ kinematics
    def fk(self, q, v=None):
        v = np.zeros(NV) if v is None else v
        th, w = np.zeros(NL), np.zeros(NL)
        o, oa = np.zeros((NL, 2)), np.zeros((NL, 2))
        for li, L in enumerate(self.links):
            d, p = L["dof"], L["parent"]
            if p < 0:
                th[li] = L["sigma"] * (q[d] - self.qpos0[d]); w[li] = L["sigma"] * v[d]
                oa[li] = [0.0, self.g]
            else:
                th[li] = th[p] + L["sigma"] * (q[d] - self.qpos0[d]); w[li] = w[p] + L["sigma"] * v[d]
                r = rot(np.cos(th[p]), np.sin(th[p]), L["off"])
                o[li] = o[p] + r
                oa[li] = oa[p] - w[p] ** 2 * r
        c, s = np.cos(th), np.sin(th)
        com = np.array([o[li] + rot(c[li], s[li], L["com"]) for li, L in enumerate(self.links)])
        ac = np.array([oa[li] - w[li] ** 2 * (com[li] - o[li]) for li in range(NL)])
        base = np.array([q[0] - self.qpos0[0] + self.links[0]["off"][0], q[1] - self.qpos0[1] + self.links[0]["off"][1]])
        return dict(th=th, w=w, c=c, s=s, o=o, com=com, ac=ac, base=base)

    def point(self, k, link, d):  # relative to the pelvis origin
        return k["o"][link] + rot(k["c"][link], k["s"][link], np.asarray(d))

    def jac_point(self, k, link, p):
        J = np.zeros((2, NV))
        for d in self.path[link]:
            if d == 0:
                J[0, d] = 1.0
            elif d == 1:
                J[1, d] = 1.0
            else:
                J[:, d] = self.sigma[d] * ycross(p - k["o"][self.dof_link[d]])
        return J

    def point_vel(self, k, link, p, v):
        return self.jac_point(k, link, p) @ v

    # ------------------------------------------------------------------ M and bias via subtree sums
    def mass_bias(self, k):
        mass = np.array([L["mass"] for L in self.links]); inertia = np.array([L["inertia"] for L in self.links])
        F = mass[:, None] * k["ac"]
        msub, S1, S2, bias = np.zeros(NV), np.zeros((NV, 2)), np.zeros(NV), np.zeros(NV)
        for d in range(NV):
            od = k["o"][self.dof_link[d]]
            for l in self.sub[self.dof_link[d]]:
                r = k["com"][l] - od
                msub[d] += mass[l]; S1[d] += mass[l] * r; S2[d] += mass[l] * (r @ r) + inertia[l]
                if d == 0:
                    bias[d] += F[l][0]
                elif d == 1:
                    bias[d] += F[l][1]
                else:
                    bias[d] += self.sigma[d] * (F[l][0] * r[1] - F[l][1] * r[0])
        M = np.zeros((NV, NV))
        for i in range(NV):
            for j in range(NV):
                li, lj = self.dof_link[i], self.dof_link[j]
                if i < 2 and j < 2:
                    M[i, j] = msub[0] if i == j else 0.0
                    continue
                if i < 2 or j < 2:
                    hd, sl = (j, i) if i < 2 else (i, j)
                    M[i, j] = self.sigma[hd] * (S1[hd][1] if sl == 0 else -S1[hd][0])
                    continue
                if self._is_anc(lj, li):
                    deep, other = i, j
                elif self._is_anc(li, lj):
                    deep, other = j, i
                else:
                    continue
                M[i, j] = self.sigma[i] * self.sigma[j] * (S2[deep] + (k["o"][self.dof_link[deep]] - k["o"][self.dof_link[other]]) @ S1[deep])
        M += np.diag(self.armature)
        return M, bias

    # ------------------------------------------------------------------ constraint slots
    def rows(self, k, q, v):
        T = self.T
        J = np.zeros((NSLOT, NV)); pos = np.zeros(NSLOT); diag = np.zeros(NSLOT); active = np.zeros(NSLOT, bool)
        kind = np.zeros(NSLOT, int)  # 0 eq, 1 limit, 2 contact normal, 3 contact tangent
        for e, E in enumerate(self.eqs):
            p1 = self.point(k, E["link1"], E["d1"]); p2 = self.point(k, E["link2"], E["d2"])
            Jd = self.jac_point(k, E["link1"], p1) - self.jac_point(k, E["link2"], p2)
            for c in range(2):
                s = SLOT_EQ + 2 * e + c
                J[s], pos[s], diag[s], active[s], kind[s] = Jd[c], (p1 - p2)[c], E["invweight"], True, 0
        for n, d in enumerate(LIMIT_DOFS):
            s = SLOT_LIM + n
            kind[s] = 1; diag[s] = T["dof"]["invweight0"][d]
            lo, hi = T["dof"]["range"][d]
            if q[d] - lo < 0:
                active[s], pos[s], J[s, d] = True, q[d] - lo, 1.0
            elif hi - q[d] < 0:
                active[s], pos[s], J[s, d] = True, hi - q[d], -1.0
        for c, S in enumerate(self.spheres):
            sn, st = SLOT_CON + 2 * c, SLOT_CON + 2 * c + 1
            kind[sn], kind[st] = 2, 3
            ctr = self.point(k, S["link"], S["d"])
            dist = k["base"][1] + ctr[1] - S["r"]
            if dist < 0:
                p = np.array([ctr[0], dist / 2 - k["base"][1]])
                Jp = self.jac_point(k, S["link"], p)
                J[sn], J[st] = Jp[1], Jp[0]
                pos[sn] = dist
                diag[sn] = diag[st] = S["invweight"]
                active[sn] = active[st] = True
        return J, pos, diag, active, kind

    @staticmethod
    def impedance(solimp, x):
        if solimp[0] == solimp[1] or solimp[2] <= MINVAL:
            return 0.5 * (solimp[0] + solimp[1])
        x = abs(x / solimp[2])
        if x >= 1:
            return solimp[1]
        if x <= 0:
            return solimp[0]
        y = 2 * x * x if x <= 0.5 else 1 - 2 * (1 - x) ** 2
        return solimp[0] + y * (solimp[1] - solimp[0])

    def row_params(self, J, pos, diag, active, kind, v):
        T = self.T
        R, aref = np.zeros(NSLOT), np.zeros(NSLOT)
        vel = J @ v
        for s in range(NSLOT):
            if not active[s]:
                continue
            if kind[s] == 0:
                solref, solimp = T["eq"]["solref"][s // 2], T["eq"]["solimp"][s // 2]
            elif kind[s] == 1:
                solref, solimp = T["limit"]["solref"], T["limit"]["solimp"]
            else:
                solref, solimp = T["contact"]["solref"], T["contact"]["solimp"]
            tc = max(solref[0], 2 * self.h)
            kk, bb = 1.0 / (solimp[1] ** 2 * tc ** 2 * solref[1] ** 2), 2.0 / (solimp[1] * tc)
            p_imp = pos[s - 1] if kind[s] == 3 else pos[s]  # friction row shares the normal row's regulariser
            imp = self.impedance(solimp, p_imp)
            R[s] = max(MINVAL, (1 - imp) / imp * diag[s])
            imp_own = self.impedance(solimp, pos[s])
            aref[s] = -bb * vel[s] - kk * imp_own * pos[s]
        return R, aref

    # ------------------------------------------------------------------ solver
    def warmstart(self, A, b, R, jar, active, kind):
        mu = self.T["contact"]["mu"]
        f = np.zeros(NSLOT)
        D = np.where(active, 1.0 / np.where(R > 0, R, 1.0), 0.0)
        for s in range(NSLOT):
            if not active[s]:
                continue
            if kind[s] == 0:
                f[s] = -D[s] * jar[s]
            elif kind[s] == 1:
                f[s] = -D[s] * jar[s] if jar[s] < 0 else 0.0
            elif kind[s] == 2:
                N, U1 = jar[s] * mu, jar[s + 1] * mu
                Tn = abs(U1)
                if N >= mu * Tn or (Tn <= 0 and N >= 0):
                    pass
                elif mu * N + Tn <= 0 or (Tn <= 0 and N < 0):
                    f[s], f[s + 1] = -D[s] * jar[s], -D[s + 1] * jar[s + 1]
                else:
                    Dm = D[s] / (mu * mu * (1 + mu * mu)); NmT = N - mu * Tn
                    f[s] = -Dm * NmT * mu
                    f[s + 1] = -f[s] / Tn * U1 * mu
        cost = float(f @ (0.5 * (A @ f) + b))
        if cost > 0:
            f[:] = 0
        return f

    def pgs(self, A, b, f, active, kind):
        mu = self.T["contact"]["mu"]
        f = f.copy()
        res = A @ f + b  # residual per row, maintained incrementally (the kernel keeps row s on lane s)
        scale = 1.0 / (self.meaninertia * NV)
        niter = 0
        for it in range(self.iterations):
            improvement = 0.0
            for s in range(NSLOT):
                if not active[s] or kind[s] == 3:
                    continue
                if kind[s] < 2:
                    old = f[s]
                    new = old - res[s] / A[s, s]
                    if kind[s] == 1 and new < 0:
                        new = 0.0
                    d = new - old
                    change = 0.5 * d * d * A[s, s] + d * res[s]
                    if change > 1e-10:
                        d, change = 0.0, 0.0
                    f[s] = old + d
                    res += A[:, s] * d
                    improvement -= change
                else:
                    t = s + 1
                    rn, rt, on, ot = res[s], res[t], f[s], f[t]
                    Ann, Ant, Att = A[s, s], A[s, t], A[t, t]
                    fn, ft = on, ot
                    if fn < MINVAL:
                        fn = fn - rn / Ann
                        if fn < 0:
                            fn = 0.0
                        ft = 0.0
                    else:
                        denom = fn * (Ann * fn + Ant * ft) + ft * (Ant * fn + Att * ft)
                        if denom >= MINVAL:
                            x = -(fn * rn + ft * rt) / denom
                            if fn + x * fn < 0:
                                x = -1.0
                            fn, ft = fn + x * fn, ft + x * ft
                    if fn >= MINVAL:
                        bc = rt - Att * ot + Ant * (fn - on)
                        x0 = -bc / Att
                        v1 = x0 / mu
                        val = v1 * v1 - fn * fn
                        ft = x0
                        if val >= 1e-10:
                            delta = val * Att * mu * mu / (2 * v1 * v1)
                            if delta >= 1e-10:
                                ft = np.sign(x0) * mu * fn
                    dn, dt = fn - on, ft - ot
                    change = 0.5 * (Ann * dn * dn + 2 * Ant * dn * dt + Att * dt * dt) + dn * rn + dt * rt
                    if change > 1e-10:
                        dn, dt, change = 0.0, 0.0, 0.0
                    f[s], f[t] = on + dn, ot + dt
                    res += A[:, s] * dn + A[:, t] * dt
                    improvement -= change
            niter = it + 1
            if improvement * scale < self.tolerance:
                break
        return f, niter

    # ------------------------------------------------------------------ one mj_step
    def forward(self, q, v, ws, ctrl):
        T = self.T
        k = self.fk(q, v)
        M, bias = self.mass_bias(k)
        act = np.zeros(NV)
        for a in range(NU):
            lo, hi = T["act"]["ctrlrange"][a]
            act[T["act"]["dof"][a]] += T["act"]["gear"][a] * min(max(ctrl[a], lo), hi)
        tau = -self.damping * v - bias + act
        Minv = np.linalg.inv(M)
        qs = Minv @ tau
        J, pos, diag, active, kind = self.rows(k, q, v)
        R, aref = self.row_params(J, pos, diag, active, kind, v)
        A = J @ Minv @ J.T + np.diag(R)
        b = J @ qs - aref
        jar = J @ ws - aref
        f0 = self.warmstart(A, b, R, jar, active, kind)
        f, niter = self.pgs(A, b, f0, active, kind)
        g = tau + J.T @ f
        qacc = Minv @ g
        return dict(M=M, bias=bias, tau=tau, qacc_smooth=qs, J=J, pos=pos, active=active, kind=kind, R=R, aref=aref,
                    A=A, b=b, f0=f0, f=f, niter=niter, qacc=qacc, g=g, k=k)

    def step(self, q, v, ws, ctrl):
        r = self.forward(q, v, ws, ctrl)
        Mh = r["M"] + self.h * np.diag(self.damping)
        qacc_new = np.linalg.solve(Mh, r["g"])
        v2 = v + self.h * qacc_new
        q2 = q + self.h * v2
        return q2, v2, r["qacc"], r

    def pd_ctrl(self, q, v, target):
        j = [3, 4, 6, 8, 9, 11]
        return 10.0 * (np.asarray(target) - q[j]) + 5.0 * (0.0 - v[j])

    # ------------------------------------------------------------------ op-space state (GetOperationalSpaceState)
    def opstate(self, kq, kv, q, v):
        k = self.fk(kq, kv)
        x, xd = [], []
        for sid in (1, 2, 3, 4, 5):
            S = self.sites[sid]
            p = self.point(k, S["link"], S["d"])
            x.append(p + k["base"]); xd.append(self.point_vel(k, S["link"], p, kv))
        s = np.zeros(18)
        for i in range(2):
            s[i] = x[0][i]; s[3 + i] = xd[0][i]
            s[6 + i] = (x[1][i] + x[2][i]) / 2; s[9 + i] = (xd[1][i] + xd[2][i]) / 2
            s[12 + i] = (x[3][i] + x[4][i]) / 2; s[15 + i] = (xd[3][i] + xd[4][i]) / 2
        s[2], s[5] = q[2], v[2]
        return s


# ====================================================================== controllers (planar restatement)
class PlanarControllers:
    """Planar form of DynamicState + Cassie2d::StepJacobian / StepOsc (RBDL-semantics tables), as the HIP
    controller kernels compute them.  y rows/columns of the reference's 3-D matrices are identically zero here and
    are dropped; the OSC QP is solved in a reduced box-constrained form."""

    W_COM, W_STANCE, W_REST, W_F, MU_OSC = 5.0, 10.0, 0.1, 1e-4, 0.5

    def __init__(self, tables=None):
        self.P = Planar(tables, sem="rbdl")
        T = self.P.T
        self.gear = np.array(T["act"]["gear"]); self.adof = T["act"]["dof"]
        self.clo = np.array([r[0] for r in T["act"]["ctrlrange"]]); self.chi = np.array([r[1] for r in T["act"]["ctrlrange"]])

    def dyn(self, q, v):
        P = self.P
        k = P.fk(q, v)
        M, bias = P.mass_bias(k)
        bias = bias + P.damping * v                      # DynamicState.cpp:49-52 (bias -= passive)
        Jeq, jdq = np.zeros((4, NV)), np.zeros(4)
        oa = self._origin_acc(k)
        for e, E in enumerate(P.eqs):
            p1 = P.point(k, E["link1"], E["d1"]); p2 = P.point(k, E["link2"], E["d2"])
            Jeq[2 * e:2 * e + 2] = P.jac_point(k, E["link1"], p1) - P.jac_point(k, E["link2"], p2)
            jdq[2 * e:2 * e + 2] = self._pacc(k, oa, E["link1"], p1) - self._pacc(k, oa, E["link2"], p2)
        Js, acc = np.zeros((10, NV)), np.zeros(10)
        for i, sid in enumerate((1, 2, 3, 4, 5)):
            S = P.sites[sid]
            p = P.point(k, S["link"], S["d"])
            Js[2 * i:2 * i + 2] = P.jac_point(k, S["link"], p)
            acc[2 * i:2 * i + 2] = self._pacc(k, oa, S["link"], p)
        return dict(M=M, bias=bias, Jeq=Jeq, jdq=jdq, Js=Js, sacc=acc, k=k)

    def _origin_acc(self, k):
        """velocity-product acceleration of every link origin WITHOUT gravity (RBDL CalcPointAcceleration, qddot=0)."""
        P = self.P
        oa = np.zeros((NL, 2))
        for li, L in enumerate(P.links):
            p = L["parent"]
            if p >= 0:
                oa[li] = oa[p] - k["w"][p] ** 2 * (k["o"][li] - k["o"][p])
        return oa

    def _pacc(self, k, oa, link, p):
        return oa[link] - k["w"][link] ** 2 * (p - k["o"][link])

    @staticmethod
    def pinv_sym(A, tol):
        w, V = np.linalg.eigh(A)
        inv = np.where(np.abs(w) > tol, 1.0 / np.where(w == 0, 1, w), 0.0)
        return (V * inv) @ V.T

    def projector(self, d):
        Hinv = np.linalg.inv(d["M"])
        JH = d["Jeq"] @ Hinv
        P4 = self.pinv_sym(JH @ d["Jeq"].T, 1e-3)
        Nc = np.eye(NV) - d["Jeq"].T @ P4 @ JH
        gamma = d["Jeq"].T @ P4 @ d["jdq"]
        return Hinv, Nc, gamma

    def Bt(self):
        B = np.zeros((NV, NU))
        for a in range(NU):
            B[self.adof[a], a] = self.gear[a]
        return B

    def ctrl_jacobian(self, q, v, force6):
        d = self.dyn(q, v)
        Hinv, Nc, gamma = self.projector(d)
        # Jc6' f : per foot mean of the two site Jacobians; f = (My, Fx, Fz) -> rows (angular y, linear x, linear z)
        k, P = d["k"], self.P
        Jtf = np.zeros(NV)
        for foot, sids in enumerate(((2, 3), (4, 5))):
            Fx, Fz, My = force6[3 * foot + 0], force6[3 * foot + 1], force6[3 * foot + 2]
            for sid in sids:
                S = P.sites[sid]
                p = P.point(k, S["link"], S["d"])
                J = P.jac_point(k, S["link"], p)
                Jw = np.zeros(NV)
                for dd in P.path[S["link"]]:
                    if dd >= 2:
                        Jw[dd] = P.sigma[dd]  # angular velocity about +y per unit joint rate
                Jtf += 0.5 * (J[0] * Fx + J[1] * Fz + Jw * My)
        NcBt = Nc @ self.Bt()
        U, s, Vt = np.linalg.svd(NcBt, full_matrices=False)
        sinv = np.where(s > 1e-4, 1.0 / s, 0.0)
        pinv = (Vt.T * sinv) @ U.T
        return pinv @ (Nc @ (d["bias"] - Jtf) + gamma)

    def osc_qp(self, q, v, act7):
        d = self.dyn(q, v)
        Hinv, Nc, gamma = self.projector(d)
        mu = self.MU_OSC
        # targets: rows (site1 x,z), (site2..5 x,z), pitch
        A = np.vstack([d["Js"], np.eye(NV)[2:3]])
        adq = np.concatenate([d["sacc"], [0.0]])
        xdd = np.array([act7[0], act7[1], act7[2], act7[3], act7[2], act7[3], act7[4], act7[5], act7[4], act7[5], act7[6]])
        W = np.array([self.W_COM] * 2 + [self.W_STANCE] * 8 + [self.W_REST])
        ce = -Nc @ d["bias"] - gamma
        cols = [Nc @ self.Bt()[:, a] for a in range(NU)]
        for c in range(4):  # contact sites 2..5 = Js rows 2+2c (x), 3+2c (z)
            jx, jz = d["Js"][2 + 2 * c], d["Js"][3 + 2 * c]
            cols.append(Nc @ (mu * jx + jz)); cols.append(Nc @ (-mu * jx + jz))
        PQ = Hinv @ np.array(cols).T                    # 13 x 14 : qdd = PQ z + q0
        q0 = Hinv @ ce
        T = A @ PQ; t0 = A @ q0 + adq - xdd
        G = 2 * T.T @ (W[:, None] * T)
        for c in range(4):
            blk = self.W_F * np.array([[mu * mu + 1, 1 - mu * mu], [1 - mu * mu, mu * mu + 1]])
            G[6 + 2 * c:8 + 2 * c, 6 + 2 * c:8 + 2 * c] += blk
        cvec = 2 * T.T @ (W * t0)
        lo = np.concatenate([self.clo, np.zeros(8)]); hi = np.concatenate([self.chi, np.full(8, np.inf)])
        z, iters = box_qp(G, cvec, lo, hi)
        return z[:6], z, iters, dict(G=G, c=cvec, PQ=PQ, q0=q0)


def box_qp(G, c, lo, hi, z0=None, max_iter=60):
    """min 1/2 z'Gz + c'z, lo <= z <= hi, G SPD.  Primal active-set with masked full-size solves (what the kernel
    does): a working set of variables held at a bound; one change of the working set per iteration.  A full Newton
    step on the working set lands exactly on its minimiser, so optimality is decided from multiplier signs only
    (G is ill-conditioned -- force-regulariser directions have eigenvalues ~5e-5 against 2e7 -- which rules out a
    small-step test; those weak directions do not influence the motor commands)."""
    n = len(c)
    z = np.clip(np.zeros(n) if z0 is None else z0, lo, hi)
    bound = (z <= lo) | (z >= hi)
    for it in range(max_iter):
        g = G @ z + c
        Gm = G.copy(); rhs = -g.copy()
        for i in range(n):
            if bound[i]:
                Gm[i, :] = 0; Gm[:, i] = 0; Gm[i, i] = 1; rhs[i] = 0
        d = np.linalg.solve(Gm, rhs)
        alpha, blk = 1.0, -1
        for i in range(n):
            if bound[i]:
                continue
            if d[i] > 0 and np.isfinite(hi[i]) and (hi[i] - z[i]) < alpha * d[i]:
                alpha, blk = (hi[i] - z[i]) / d[i], i
            elif d[i] < 0 and (lo[i] - z[i]) > alpha * d[i]:
                alpha, blk = (lo[i] - z[i]) / d[i], i
        z = z + alpha * d
        if blk >= 0:
            z[blk] = hi[blk] if d[blk] > 0 else lo[blk]
            bound[blk] = True
            continue
        g = G @ z + c
        viol = np.where(bound, np.where(z <= lo, -g, g), -np.inf)   # > 0: the bound is not wanted
        i = int(np.argmax(viol))
        if viol[i] <= 1e-9:
            return z, it + 1
        bound[i] = False
    return z, max_iter

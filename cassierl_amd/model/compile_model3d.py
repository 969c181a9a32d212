#!/usr/bin/env python3
"""Offline model compiler for cassie3d_stiff.xml (BASELINE.json configs[4]).

The reference has no Cassie3d class -- only the MJCF file and vestigial hooks (xml_parser.h:321-323,
RobotInterface.h:54, DynamicModel.cpp:250-265) -- so what is restated here is MuJoCo's own handling of
that file: a floating base (free joint: 3 world-frame translations + unit quaternion, angular velocity
in the body frame) and 14 hinges.  Run OFFLINE in the container that has /root/reference; outputs are
committed:

  oracle/cassie3d_model.h               raw body/joint/geom tables for the CPU oracle (-DORC_CASSIE3D)
  cassierl_amd/csrc/cassie3d_tables.h   the same mechanism with jointless bodies welded into their
                                        parents, as __constant__ tables for the HIP kernel
  tests/golden/model3d_kat.json         known-answer values from the numpy evaluation below

The free joint is expanded into six dofs: three slides along the world axes (type 0) and three
"free-rot" dofs (type 2) whose axes are the body's own x, y, z axes; their three velocities are the
body-frame angular velocity and their position is the quaternion at qpos[3:7].
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
from compile_model import parse_mjcf, _carr  # noqa: E402

DEFAULT_XML = "/root/reference/model/cassie3d_stiff.xml"

# standing pose used as the reset pose of the 3-D environments: the sagittal angles of Cassie2d.cpp:56-58 /
# cassie2d.py:79-85 on both legs, abduction and yaw at 0, pelvis upright at the same height
QINIT_LEG = [0.0, 0.0, 0.68111815, -1.40730357, 1.62972042, -1.77611107, -0.61968407]
QINIT = np.array([0.0, 0.0, 0.939, 1.0, 0.0, 0.0, 0.0] + QINIT_LEG + QINIT_LEG)


def quat_to_mat(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def rot_axis(axis, ang):
    a = axis / np.linalg.norm(axis)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * (K @ K)


def expand_dofs(mj):
    """joint list -> dof list (free joint = 3 slides + 3 free-rot), with qpos addresses"""
    dofs, jmap, qadr = [], {}, 0
    for j in mj["joints"]:
        jmap[j["name"]] = len(dofs)
        if j["type"] == "free":
            bpos = mj["bodies"][j["body"]]["pos"]
            for k in range(3):
                dofs.append(dict(name=j["name"] + "_t" + "xyz"[k], type=0, body=j["body"], axis=np.eye(3)[k], ref=float(bpos[k]),
                                 range=np.zeros(2), limited=False, damping=0.0, armature=0.0, qadr=qadr + k))
            for k in range(3):
                dofs.append(dict(name=j["name"] + "_r" + "xyz"[k], type=2, body=j["body"], axis=np.eye(3)[k], ref=0.0,
                                 range=np.zeros(2), limited=False, damping=0.0, armature=0.0, qadr=qadr + 3))
            qadr += 7
        else:
            d = dict(j)
            d["type"] = 0 if j["type"] == "slide" else 1
            d["qadr"] = qadr
            dofs.append(d)
            qadr += 1
    return dofs, jmap, qadr


class Tree:
    """numpy evaluation of the mechanism (FK, Jacobians, mass matrix) -- KAT generator and executable spec"""

    def __init__(self, mj):
        self.mj = mj
        self.bodies = mj["bodies"]
        self.dofs, self.jmap, self.nq = expand_dofs(mj)
        self.nv = len(self.dofs)
        self.qpos0 = np.zeros(self.nq)
        for d in self.dofs:
            if d["type"] == 2:
                self.qpos0[d["qadr"]:d["qadr"] + 4] = [1, 0, 0, 0]
            else:
                self.qpos0[d["qadr"]] = d["ref"]
        self.body_dofs = [[k for k, d in enumerate(self.dofs) if d["body"] == b] for b in range(len(self.bodies))]

    def fk(self, q):
        nb = len(self.bodies)
        xpos, xmat = np.zeros((nb, 3)), np.tile(np.eye(3), (nb, 1, 1))
        anchor, axis = np.zeros((self.nv, 3)), np.zeros((self.nv, 3))
        for b in range(1, nb):
            B = self.bodies[b]
            p = B["parent"]
            pos = xpos[p] + xmat[p] @ B["pos"]
            mat = xmat[p] @ B["rot"]
            ks = self.body_dofs[b]
            i = 0
            while i < len(ks):
                k = ks[i]
                d = self.dofs[k]
                if d["type"] == 0:
                    ax = mat @ d["axis"]
                    pos = pos + ax * (q[d["qadr"]] - d["ref"])
                    anchor[k], axis[k] = pos, ax
                    i += 1
                elif d["type"] == 1:
                    ax = mat @ d["axis"]
                    anchor[k], axis[k] = pos, ax
                    mat = rot_axis(ax, q[d["qadr"]] - d["ref"]) @ mat
                    i += 1
                else:
                    quat = q[d["qadr"]:d["qadr"] + 4]
                    mat = mat @ quat_to_mat(quat / np.linalg.norm(quat))
                    for c in range(3):
                        anchor[ks[i + c]], axis[ks[i + c]] = pos, mat[:, c]
                    i += 3
            xpos[b], xmat[b] = pos, mat
        return xpos, xmat, anchor, axis

    def chain(self, b):
        out = []
        while b > 0:
            out = self.body_dofs[b] + out
            b = self.bodies[b]["parent"]
        return out

    def jac(self, b, point, fkres):
        _, _, anchor, axis = fkres
        Jv, Jw = np.zeros((3, self.nv)), np.zeros((3, self.nv))
        for k in self.chain(b):
            if self.dofs[k]["type"] == 0:
                Jv[:, k] = axis[k]
            else:
                Jv[:, k] = np.cross(axis[k], point - anchor[k])
                Jw[:, k] = axis[k]
        return Jv, Jw

    def mass_matrix(self, q):
        fkres = self.fk(q)
        xpos, xmat = fkres[0], fkres[1]
        M = np.diag([d["armature"] for d in self.dofs]).astype(np.float64)
        for b in range(1, len(self.bodies)):
            B = self.bodies[b]
            c = xpos[b] + xmat[b] @ B["ipos"]
            Jv, Jw = self.jac(b, c, fkres)
            Iw = xmat[b] @ B["inertia"] @ xmat[b].T
            M += B["mass"] * Jv.T @ Jv + Jw.T @ Iw @ Jw
        return M

    def constants(self):
        """mj_setConst: invweight0 (free-joint dofs averaged over their translational / rotational triples), meaninertia, anchor2"""
        q0 = self.qpos0
        fkres = self.fk(q0)
        xpos, xmat = fkres[0], fkres[1]
        M0 = self.mass_matrix(q0)
        Minv = np.linalg.inv(M0)
        dw = np.diag(Minv).copy()
        k = 0
        while k < self.nv:
            if self.dofs[k]["type"] == 2:
                dw[k:k + 3] = dw[k:k + 3].mean()
                dw[k - 3:k] = dw[k - 3:k].mean()
                k += 3
            else:
                k += 1
        bw = np.zeros(len(self.bodies))
        for b in range(1, len(self.bodies)):
            c = xpos[b] + xmat[b] @ self.bodies[b]["ipos"]
            Jv, _ = self.jac(b, c, fkres)
            bw[b] = np.trace(Jv @ Minv @ Jv.T) / 3.0
        a2 = []
        for e in self.mj["eqs"]:
            pw = xpos[e["body1"]] + xmat[e["body1"]] @ e["anchor"]
            a2.append(xmat[e["body2"]].T @ (pw - xpos[e["body2"]]))
        return dict(M0=M0, meaninertia=float(np.mean(np.diag(M0))), dof_invweight0=dw, body_invweight0=bw, eq_anchor2=a2)


def weld(mj, tree):
    """Merge every jointless body into its parent (same rigid body): the kernel's link list."""
    nb = len(tree.bodies)
    jointed = [b for b in range(1, nb) if tree.body_dofs[b]]
    fk0 = tree.fk(tree.qpos0)
    xpos, xmat = fk0[0], fk0[1]

    def owner(b):
        while b > 0 and not tree.body_dofs[b]:
            b = tree.bodies[b]["parent"]
        return b

    links = []
    for lb in jointed:
        members = [b for b in range(1, nb) if owner(b) == lb]
        m = sum(tree.bodies[b]["mass"] for b in members)
        R, o = xmat[lb], xpos[lb]
        com = sum(tree.bodies[b]["mass"] * (xpos[b] + xmat[b] @ tree.bodies[b]["ipos"]) for b in members) / m
        I = np.zeros((3, 3))
        for b in members:
            c = xpos[b] + xmat[b] @ tree.bodies[b]["ipos"]
            r = c - com
            I += xmat[b] @ tree.bodies[b]["inertia"] @ xmat[b].T + tree.bodies[b]["mass"] * (np.dot(r, r) * np.eye(3) - np.outer(r, r))
        links.append(dict(body=lb, name=tree.bodies[lb]["name"], members=members, mass=m,
                          ipos=R.T @ (com - o), inertia=R.T @ I @ R))
    lid = {L["body"]: i for i, L in enumerate(links)}
    for L in links:
        p = owner(tree.bodies[L["body"]]["parent"])
        L["parent"] = lid.get(p, -1)
        # frame of this link relative to its parent LINK frame at qpos0 (the intermediate bodies are rigid)
        if p > 0:
            L["pos"] = xmat[p].T @ (xpos[L["body"]] - xpos[p])
            L["rot"] = xmat[p].T @ xmat[L["body"]]
        else:
            L["pos"], L["rot"] = tree.bodies[L["body"]]["pos"].copy(), tree.bodies[L["body"]]["rot"].copy()

    def to_link(b, p_local):
        """point given in body b's frame -> (link index, point in the link frame)"""
        lb = owner(b)
        pw = xpos[b] + xmat[b] @ p_local
        return lid[lb], xmat[lb].T @ (pw - xpos[lb])

    def dir_to_link(b, v_local):
        lb = owner(b)
        return xmat[lb].T @ (xmat[b] @ v_local)

    return links, to_link, dir_to_link


def emit_oracle_header(mj, tree, consts, path):
    nb, nv = len(tree.bodies), tree.nv
    col = [g for g in mj["geoms"] if g["type"] != "plane"]
    o = ["/* GENERATED by cassierl_amd/model/compile_model3d.py from model/cassie3d_stiff.xml -- do not edit.\n"
         " * Raw 3-D tables for the CPU oracle built with -DORC_CASSIE3D (test infrastructure only). */\n"
         "#ifndef CASSIE3D_MODEL_H_\n#define CASSIE3D_MODEL_H_\n"]
    o.append("#define CM_NBODY %d\n#define CM_NV %d\n#define CM_NQ %d\n#define CM_NU %d\n#define CM_NSITE %d\n#define CM_NEQ %d\n#define CM_NGEOM %d\n"
             % (nb, nv, tree.nq, len(mj["acts"]), len(mj["sites"]), len(mj["eqs"]), len(col)))
    op = mj["option"]
    o.append("#define CM_TIMESTEP %.17g\n#define CM_ITERATIONS %d\n#define CM_TOLERANCE %.17g\n#define CM_GRAVITY_Z %.17g\n#define CM_IMPRATIO %.17g\n"
             % (op["timestep"], op["iterations"], op["tolerance"], op["gravity"][2], op["impratio"]))
    B, D = tree.bodies, tree.dofs
    o.append("static const char* const cm_body_name[CM_NBODY] = {%s};\n" % ", ".join('"%s"' % b["name"] for b in B))
    o.append(_carr("cm_body_parent", [b["parent"] for b in B], "%d", "int", 11))
    o.append(_carr("cm_body_pos", [b["pos"] for b in B]))
    o.append(_carr("cm_body_rot", [[b["rot"] for b in B] for _ in range(2)]))
    o.append(_carr("cm_body_mass", [b["mass"] for b in B]))
    o.append(_carr("cm_body_ipos", [b["ipos"] for b in B]))
    o.append(_carr("cm_body_inertia", [b["inertia"] for b in B]))
    o.append(_carr("cm_jnt_type", [d["type"] for d in D], "%d", "int", 20))
    o.append(_carr("cm_jnt_body", [d["body"] for d in D], "%d", "int", 20))
    o.append(_carr("cm_jnt_qadr", [d["qadr"] for d in D], "%d", "int", 20))
    o.append(_carr("cm_jnt_axis", [d["axis"] for d in D]))
    o.append(_carr("cm_jnt_ref", [d["ref"] for d in D]))
    o.append(_carr("cm_qpos0", tree.qpos0))
    o.append(_carr("cm_jnt_limited", [int(d["limited"]) for d in D], "%d", "int", 20))
    o.append(_carr("cm_jnt_range", [d["range"] for d in D]))
    o.append(_carr("cm_dof_damping", [d["damping"] for d in D]))
    o.append(_carr("cm_dof_armature", [d["armature"] for d in D]))
    o.append(_carr("cm_limit_solref", mj["limit_solref"]))
    o.append(_carr("cm_limit_solimp", mj["limit_solimp"]))
    o.append(_carr("cm_geom_type", [0 if g["type"] == "sphere" else 1 for g in col], "%d", "int", 11))
    o.append(_carr("cm_geom_body", [g["body"] for g in col], "%d", "int", 11))
    o.append(_carr("cm_geom_pos", [g["pos"] for g in col]))
    o.append(_carr("cm_geom_axis", [g.get("axis", np.array([0, 0, 1.0])) for g in col]))
    o.append(_carr("cm_geom_halflen", [g.get("halflen", 0.0) for g in col]))
    o.append(_carr("cm_geom_radius", [g["radius"] for g in col]))
    o.append(_carr("cm_contact_solref", col[0]["solref"]))
    o.append(_carr("cm_contact_solimp", col[0]["solimp"]))
    o.append(_carr("cm_contact_friction", col[0]["friction"]))
    o.append(_carr("cm_site_body", [s["body"] for s in mj["sites"]], "%d", "int", 11))
    o.append(_carr("cm_site_pos", [s["pos"] for s in mj["sites"]]))
    o.append(_carr("cm_eq_body1", [e["body1"] for e in mj["eqs"]], "%d", "int"))
    o.append(_carr("cm_eq_body2", [e["body2"] for e in mj["eqs"]], "%d", "int"))
    o.append(_carr("cm_eq_anchor1", [e["anchor"] for e in mj["eqs"]]))
    o.append(_carr("cm_eq_solref", [e["solref"] for e in mj["eqs"]]))
    o.append(_carr("cm_eq_solimp", [e["solimp"] for e in mj["eqs"]]))
    o.append(_carr("cm_act_dof", [a["dof"] for a in mj["acts"]], "%d", "int", 10))
    o.append(_carr("cm_act_gear", [a["gear"] for a in mj["acts"]]))
    o.append(_carr("cm_act_ctrlrange", [a["ctrlrange"] for a in mj["acts"]]))
    o.append(_carr("cm_qpos_init", QINIT))
    o.append("#endif\n")
    with open(path, "w") as f:
        f.write("\n".join(o))


def emit_kernel_header(mj, tree, consts, links, to_link, dir_to_link, path):
    """__constant__ tables for cassie3d_kernels.hip (welded links; one dof group per link)."""
    nl, nv = len(links), tree.nv
    lid = {L["body"]: i for i, L in enumerate(links)}
    D = tree.dofs
    dof_link = [lid[d["body"]] for d in D]
    # ancestors-or-self of each link / links affected by each dof
    anc = []
    for i in range(nl):
        m, a = 0, i
        while a >= 0:
            m |= 1 << a
            a = links[a]["parent"]
        anc.append(m)
    dof_sub = [sum(1 << i for i in range(nl) if (anc[i] >> dof_link[k]) & 1) for k in range(nv)]   # links moved by dof k
    link_dofs = [sum(1 << k for k in range(nv) if (anc[i] >> dof_link[k]) & 1) for i in range(nl)]  # dofs that move link i
    depth = []
    for i in range(nl):
        d, a = 0, links[i]["parent"]
        while a >= 0:
            d, a = d + 1, links[a]["parent"]
        depth.append(d)
    col = [g for g in mj["geoms"] if g["type"] != "plane"]
    sph = []  # collision spheres in MuJoCo contact order (capsule: +axis end, then -axis end)
    for g in col:
        if g["type"] == "sphere":
            l, p = to_link(g["body"], g["pos"])
            sph.append(dict(link=l, pos=p, radius=g["radius"], hint=np.zeros(3), body=g["body"]))
        else:
            for sgn in (1.0, -1.0):
                l, p = to_link(g["body"], g["pos"] + sgn * g["halflen"] * g["axis"])
                sph.append(dict(link=l, pos=p, radius=g["radius"], hint=dir_to_link(g["body"], g["axis"]), body=g["body"]))
    eq = []
    for e, a2 in zip(mj["eqs"], consts["eq_anchor2"]):
        l1, p1 = to_link(e["body1"], e["anchor"])
        l2, p2 = to_link(e["body2"], a2)
        eq.append(dict(l1=l1, p1=p1, l2=l2, p2=p2, invw=consts["body_invweight0"][e["body1"]] + consts["body_invweight0"][e["body2"]],
                       solref=e["solref"], solimp=e["solimp"]))
    lim = [k for k in range(nv) if D[k]["limited"]]
    S = "static __device__ __constant__ "
    o = ["// GENERATED by cassierl_amd/model/compile_model3d.py from model/cassie3d_stiff.xml -- do not edit.\n"
         "// Cassie3d as the HIP kernel sees it: %d links (jointless bodies welded into their parents), %d dofs, %d qpos.\n"
         "#ifndef CASSIE3D_TABLES_H_\n#define CASSIE3D_TABLES_H_\nnamespace cassie3d {\n" % (nl, nv, tree.nq)]
    o.append("constexpr int NL = %d, NV = %d, NQ = %d, NU = %d, NSPH = %d, NEQ = %d, NLIM = %d;\n" % (nl, nv, tree.nq, len(mj["acts"]), len(sph), len(eq), len(lim)))
    op = mj["option"]
    o.append("constexpr double H = %.17g, GRAVITY_Z = %.17g, TOLERANCE = %.17g, MEANINERTIA = %.17g, MU = %.17g;\nconstexpr int ITERATIONS = %d;\n"
             % (op["timestep"], op["gravity"][2], op["tolerance"], consts["meaninertia"], col[0]["friction"][0], op["iterations"]))
    o.append(_carr("c3_link_parent", [L["parent"] for L in links], "%d", "int", 16, S))
    o.append(_carr("c3_link_depth", depth, "%d", "int", 16, S))
    o.append(_carr("c3_link_pos", [L["pos"] for L in links], static=S))
    o.append(_carr("c3_link_rot", [L["rot"] for L in links], static=S))
    o.append(_carr("c3_link_mass", [L["mass"] for L in links], static=S))
    o.append(_carr("c3_link_ipos", [L["ipos"] for L in links], static=S))
    o.append(_carr("c3_link_inertia", [L["inertia"] for L in links], static=S))
    o.append(_carr("c3_link_dofmask", link_dofs, "%d", "int", 8, S))
    o.append(_carr("c3_link_body_invweight", [consts["body_invweight0"][L["body"]] for L in links], static=S))
    o.append(_carr("c3_dof_type", [d["type"] for d in D], "%d", "int", 20, S))
    o.append(_carr("c3_dof_link", dof_link, "%d", "int", 20, S))
    o.append(_carr("c3_dof_qadr", [d["qadr"] for d in D], "%d", "int", 20, S))
    o.append(_carr("c3_dof_axis", [d["axis"] for d in D], static=S))
    o.append(_carr("c3_dof_ref", [d["ref"] for d in D], static=S))
    o.append(_carr("c3_dof_submask", dof_sub, "%d", "int", 8, S))
    o.append(_carr("c3_dof_damping", [d["damping"] for d in D], static=S))
    o.append(_carr("c3_dof_armature", [d["armature"] for d in D], static=S))
    o.append(_carr("c3_dof_invweight", consts["dof_invweight0"], static=S))
    o.append(_carr("c3_dof_act", [next((i for i, a in enumerate(mj["acts"]) if a["dof"] == k), -1) for k in range(nv)], "%d", "int", 20, S))
    o.append(_carr("c3_act_dof", [a["dof"] for a in mj["acts"]], "%d", "int", 10, S))
    o.append(_carr("c3_act_gear", [a["gear"] for a in mj["acts"]], static=S))
    o.append(_carr("c3_act_ctrlrange", [a["ctrlrange"] for a in mj["acts"]], static=S))
    o.append(_carr("c3_lim_dof", lim, "%d", "int", 12, S))
    o.append(_carr("c3_lim_range", [D[k]["range"] for k in lim], static=S))
    o.append(_carr("c3_limit_solref", mj["limit_solref"], static=S))
    o.append(_carr("c3_limit_solimp", mj["limit_solimp"], static=S))
    o.append(_carr("c3_sph_link", [s["link"] for s in sph], "%d", "int", 17, S))
    o.append(_carr("c3_sph_pos", [s["pos"] for s in sph], static=S))
    o.append(_carr("c3_sph_radius", [s["radius"] for s in sph], static=S))
    o.append(_carr("c3_sph_hint", [s["hint"] for s in sph], static=S))
    o.append(_carr("c3_sph_invweight", [consts["body_invweight0"][s["body"]] for s in sph], static=S))
    o.append(_carr("c3_contact_solref", col[0]["solref"], static=S))
    o.append(_carr("c3_contact_solimp", col[0]["solimp"], static=S))
    o.append(_carr("c3_eq_link1", [e["l1"] for e in eq], "%d", "int", 4, S))
    o.append(_carr("c3_eq_link2", [e["l2"] for e in eq], "%d", "int", 4, S))
    o.append(_carr("c3_eq_p1", [e["p1"] for e in eq], static=S))
    o.append(_carr("c3_eq_p2", [e["p2"] for e in eq], static=S))
    o.append(_carr("c3_eq_invweight", [e["invw"] for e in eq], static=S))
    o.append(_carr("c3_eq_solref", [e["solref"] for e in eq], static=S))
    o.append(_carr("c3_eq_solimp", [e["solimp"] for e in eq], static=S))
    o.append(_carr("c3_qpos0", tree.qpos0, static=S))
    o.append(_carr("c3_qpos_init", QINIT, static=S))
    o.append("}  // namespace cassie3d\n#endif\n")
    with open(path, "w") as f:
        f.write("\n".join(o))
    return dict(links=links, dof_link=dof_link, sph=sph, eq=eq, lim=lim)


def main(xml=DEFAULT_XML):
    mj = parse_mjcf(xml)
    tree = Tree(mj)
    for a in mj["acts"]:  # actuator -> dof index after the free-joint expansion
        a["dof"] = tree.jmap[[j["name"] for j in mj["joints"]][a["dof"]]]
    consts = tree.constants()
    links, to_link, dir_to_link = weld(mj, tree)
    emit_oracle_header(mj, tree, consts, os.path.join(REPO, "oracle", "cassie3d_model.h"))
    info = emit_kernel_header(mj, tree, consts, links, to_link, dir_to_link, os.path.join(REPO, "cassierl_amd", "csrc", "cassie3d_tables.h"))
    rng = np.random.default_rng(3)
    qs = []
    for _ in range(3):
        q = QINIT.copy()
        q[:3] += rng.uniform(-0.2, 0.2, 3)
        quat = np.array([1.0, 0, 0, 0]) + rng.uniform(-0.3, 0.3, 4)
        q[3:7] = quat / np.linalg.norm(quat)
        q[7:] += rng.uniform(-0.3, 0.3, tree.nq - 7)
        qs.append(q)
    kat = dict(source="cassie3d_stiff.xml via cassierl_amd/model/compile_model3d.py", nq=tree.nq, nv=tree.nv,
               total_mass=float(sum(b["mass"] for b in tree.bodies)), meaninertia=consts["meaninertia"],
               dof_invweight0=consts["dof_invweight0"].tolist(), body_invweight0=consts["body_invweight0"].tolist(),
               eq_anchor2=[a.tolist() for a in consts["eq_anchor2"]], qpos_init=QINIT.tolist(),
               dof_names=[d["name"] for d in tree.dofs], link_names=[L["name"] for L in links],
               link_mass=[L["mass"] for L in links],
               samples=[dict(qpos=q.tolist(), M=tree.mass_matrix(q).tolist(),
                             site_world={s["name"]: (tree.fk(q)[0][s["body"]] + tree.fk(q)[1][s["body"]] @ s["pos"]).tolist() for s in mj["sites"]})
                        for q in [QINIT] + qs])
    with open(os.path.join(REPO, "tests", "golden", "model3d_kat.json"), "w") as f:
        json.dump(kat, f, indent=1)
    print("nq %d nv %d nu %d  links %d  spheres %d  limited %d  total mass %.4f  meaninertia %.6f" %
          (tree.nq, tree.nv, len(mj["acts"]), len(links), len(info["sph"]), len(info["lim"]), kat["total_mass"], consts["meaninertia"]))
    for i, L in enumerate(links):
        print("%2d %-20s parent %2d mass %.4f members %s" % (i, L["name"], L["parent"], L["mass"], [tree.bodies[b]["name"] for b in L["members"]]))
    fk = tree.fk(QINIT)
    for s in mj["sites"]:
        print(s["name"], np.round(fk[0][s["body"]] + fk[1][s["body"]] @ s["pos"], 5))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else DEFAULT_XML)

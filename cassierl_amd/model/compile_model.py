#!/usr/bin/env python3
"""Offline model compiler: restricted MJCF (cassie2d_stiff.xml) -> constant tables.

This is the init-time counterpart of the reference's model loading
(src/xml_parser.h:104-363 + src/DynamicModel.cpp:23-235, and MuJoCo's own
``mj_loadXML`` at src/Cassie2d/Cassie2d.cpp:47).  It is run OFFLINE, in the
container that has /root/reference, and its outputs are committed:

  oracle/cassie2d_model.h              3-D body/joint/geom tables for the CPU oracle
  cassierl_amd/csrc/cassie2d_planar.h  sagittal-plane (x,z,pitch) reduction for the HIP kernels
  tests/golden/model_kat.json          known-answer values

Two "semantics" are emitted because the reference holds TWO models of the same
robot:
  * ``mj``   -- MuJoCo compile semantics: body frame = xyaxes (x normalised, y
               orthogonalised), hinge angle measured from ``ref``.  Used by the
               physics step (mj_step).
  * ``rbdl`` -- DynamicModel::LoadModel semantics (DynamicModel.cpp:84-111): a body
               whose last joint has |ref| >= 1e-3 gets an IDENTITY frame and the
               joint angle is absolute, i.e. the frame is Rz(ref) exactly instead
               of the 5-digit xyaxes; the loop-closure anchor posB is re-derived
               from that model at q = ref (DynamicModel.cpp:152-168).  Used by the
               controllers and by GetOperationalSpaceState.
    Deviation (documented in DESIGN.md): LoadModel normalises the two xyaxes
    vectors separately and does not re-orthogonalise them (DynamicModel.cpp:87-96),
    which for left/right_achilles_rod gives RBDL a slightly sheared "rotation".
    We orthogonalise (as MuJoCo does); the affected quantities only involve the
    rod's local x axis, which is identical in both treatments.
"""
import json
import os
import re
import sys
import xml.etree.ElementTree as ET

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
DEFAULT_XML = "/root/reference/model/cassie2d_stiff.xml"

DEG = np.pi / 180.0


# --------------------------------------------------------------------------- parsing
def _floats(s):
    return np.array([float(x) for x in s.split()], dtype=np.float64)


def _frame_from_xyaxes(v):
    x = v[:3] / np.linalg.norm(v[:3])
    y = v[3:] - x * np.dot(x, v[3:])
    y = y / np.linalg.norm(y)
    z = np.cross(x, y)
    return np.stack([x, y, z], axis=1)  # columns = local axes in parent coords


def parse_mjcf(path):
    # the file has "<!---...--->" comments, which strict XML parsers reject: strip comments first
    text = re.sub(r"<!--.*?-->", "", open(path).read(), flags=re.S)
    root = ET.fromstring(text)
    comp = root.find("compiler")
    assert comp.get("angle") == "degree" and comp.get("inertiafromgeom") == "false"
    opt = root.find("option")
    option = dict(
        timestep=float(opt.get("timestep")),
        iterations=int(opt.get("iterations")),
        solver=opt.get("solver"),
        cone=opt.get("cone"),
        gravity=_floats(opt.get("gravity")),
        tolerance=1e-8,  # MuJoCo default (not set in the XML)
        impratio=1.0,
    )
    dflt = root.find("default")
    gd = dflt.find("geom")
    geom_default = dict(
        contype=int(gd.get("contype")), conaffinity=int(gd.get("conaffinity")),
        condim=int(gd.get("condim")), solref=_floats(gd.get("solref")),
        solimp=_floats(gd.get("solimp")), friction=_floats(gd.get("friction")))
    joint_limited_default = dflt.find("joint").get("limited") == "true"

    bodies = [dict(name="world", parent=-1, pos=np.zeros(3), rot=np.eye(3), mass=0.0,
                   ipos=np.zeros(3), inertia=np.zeros((3, 3)), has_xyaxes=False)]
    joints, geoms, sites = [], [], []

    def add_geom(g, body_id):
        if g.get("type") == "mesh":
            return  # visual only (contype=conaffinity=0 from the default class)
        d = dict(geom_default)
        d.update(type=g.get("type"), body=body_id, name=g.get("name", ""))
        for k in ("contype", "conaffinity", "condim"):
            if g.get(k) is not None:
                d[k] = int(g.get(k))
        if d["type"] == "plane":
            d["pos"] = _floats(g.get("pos"))
        elif d["type"] == "sphere":
            d["radius"] = float(g.get("size"))
            d["pos"] = _floats(g.get("pos"))
        elif d["type"] == "capsule":
            d["radius"] = float(g.get("size"))
            ft = _floats(g.get("fromto"))
            d["from"], d["to"] = ft[:3], ft[3:]
            d["pos"] = 0.5 * (ft[:3] + ft[3:])
            d["axis"] = (ft[3:] - ft[:3]) / np.linalg.norm(ft[3:] - ft[:3])
            d["halflen"] = 0.5 * np.linalg.norm(ft[3:] - ft[:3])
        else:
            raise ValueError(d["type"])
        geoms.append(d)

    def walk(elem, parent_id):
        for child in elem:
            if child.tag == "geom" and parent_id == 0:
                add_geom(child, 0)
        for b in elem.findall("body"):
            bid = len(bodies)
            rot = np.eye(3)
            if b.get("xyaxes") is not None:
                rot = _frame_from_xyaxes(_floats(b.get("xyaxes")))
            ine = b.find("inertial")
            fi = _floats(ine.get("fullinertia"))
            I = np.array([[fi[0], fi[3], fi[4]], [fi[3], fi[1], fi[5]], [fi[4], fi[5], fi[2]]])
            bodies.append(dict(name=b.get("name"), parent=parent_id, pos=_floats(b.get("pos")),
                               rot=rot, mass=float(ine.get("mass")), ipos=_floats(ine.get("pos")),
                               inertia=I, has_xyaxes=b.get("xyaxes") is not None))
            for j in b.findall("joint"):
                jt = j.get("type")
                axis = _floats(j.get("axis")) if j.get("axis") else np.array([0.0, 0.0, 1.0])
                limited = joint_limited_default
                if j.get("limited") is not None:
                    limited = j.get("limited") == "true"
                ref = float(j.get("ref", "0"))
                rng = _floats(j.get("range")) if j.get("range") else np.zeros(2)
                if jt == "hinge":
                    ref, rng = ref * DEG, rng * DEG
                joints.append(dict(name=j.get("name"), type=jt, body=bid, axis=axis / np.linalg.norm(axis),
                                   ref=ref, range=rng, limited=limited,
                                   damping=float(j.get("damping", "0")),
                                   armature=float(j.get("armature", "0"))))
            for g in b.findall("geom"):
                add_geom(g, bid)
            for s in b.findall("site"):
                sites.append(dict(name=s.get("name"), body=bid, pos=_floats(s.get("pos"))))
            walk(b, bid)

    walk(root.find("worldbody"), 0)
    names = [b["name"] for b in bodies]
    eqs = []
    for c in root.find("equality").findall("connect"):
        eqs.append(dict(body1=names.index(c.get("body1")), body2=names.index(c.get("body2")),
                        anchor=_floats(c.get("anchor")), solref=_floats(c.get("solref")),
                        solimp=np.array([0.9, 0.95, 0.001])))  # MuJoCo global default
    jn = [j["name"] for j in joints]
    acts = []
    for m in root.find("actuator").findall("motor"):
        acts.append(dict(name=m.get("name"), dof=jn.index(m.get("joint")), gear=float(m.get("gear")),
                         ctrlrange=_floats(m.get("ctrlrange"))))
    return dict(option=option, bodies=bodies, joints=joints, geoms=geoms, sites=sites, eqs=eqs, acts=acts,
                limit_solref=np.array([0.02, 1.0]), limit_solimp=np.array([0.9, 0.95, 0.001]))


# --------------------------------------------------------------------------- 3-D kinematics (compile time only)
def _rot_axis(axis, ang):
    a = axis / np.linalg.norm(axis)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * (K @ K)


class Model3D:
    """Kinematic tree with slide/hinge joints, evaluated the way mj_kinematics does."""

    def __init__(self, mj, semantics):
        self.mj = mj
        self.semantics = semantics
        self.bodies = [dict(b) for b in mj["bodies"]]
        self.joints = mj["joints"]
        self.nv = len(self.joints)
        self.qpos0 = np.array([j["ref"] for j in self.joints])
        if semantics == "rbdl":
            # DynamicModel.cpp:84-103: identity frame when the body's last joint has |ref| >= 1e-3
            # (ref still in DEGREES there).  Joint angle is then absolute => frame = Rz(ref) exactly.
            for bid, b in enumerate(self.bodies):
                js = [j for j in self.joints if j["body"] == bid]
                if js and abs(js[-1]["ref"] / DEG) >= 1e-3:
                    assert js[-1]["type"] == "hinge"
                    b["rot"] = _rot_axis(js[-1]["axis"], js[-1]["ref"])
        self.body_joints = [[k for k, j in enumerate(self.joints) if j["body"] == bid]
                            for bid in range(len(self.bodies))]

    def fk(self, q):
        nb = len(self.bodies)
        xpos = np.zeros((nb, 3))
        xmat = np.tile(np.eye(3), (nb, 1, 1))
        anchor = np.zeros((self.nv, 3))
        axis = np.zeros((self.nv, 3))
        for bid in range(1, nb):
            b = self.bodies[bid]
            p = b["parent"]
            pos = xpos[p] + xmat[p] @ b["pos"]
            mat = xmat[p] @ b["rot"]
            for k in self.body_joints[bid]:
                j = self.joints[k]
                ax = mat @ j["axis"]
                if j["type"] == "slide":
                    pos = pos + ax * (q[k] - self.qpos0[k])
                    anchor[k] = pos
                else:
                    anchor[k] = pos  # joint pos = 0 0 0 in this model
                    mat = _rot_axis(ax, q[k] - self.qpos0[k]) @ mat
                axis[k] = ax
            xpos[bid], xmat[bid] = pos, mat
        return xpos, xmat, anchor, axis

    def chain(self, bid):
        dofs = []
        while bid > 0:
            dofs = self.body_joints[bid] + dofs
            bid = self.bodies[bid]["parent"]
        return dofs

    def jac(self, q, bid, point_world, fkres=None):
        xpos, xmat, anchor, axis = fkres if fkres else self.fk(q)
        Jv, Jw = np.zeros((3, self.nv)), np.zeros((3, self.nv))
        for k in self.chain(bid):
            if self.joints[k]["type"] == "slide":
                Jv[:, k] = axis[k]
            else:
                Jv[:, k] = np.cross(axis[k], point_world - anchor[k])
                Jw[:, k] = axis[k]
        return Jv, Jw

    def mass_matrix(self, q):
        fkres = self.fk(q)
        xpos, xmat = fkres[0], fkres[1]
        M = np.diag([j["armature"] for j in self.joints]).astype(np.float64)
        for bid in range(1, len(self.bodies)):
            b = self.bodies[bid]
            c = xpos[bid] + xmat[bid] @ b["ipos"]
            Jv, Jw = self.jac(q, bid, c, fkres)
            Iw = xmat[bid] @ b["inertia"] @ xmat[bid].T
            M += b["mass"] * Jv.T @ Jv + Jw.T @ Iw @ Jw
        return M


def compile_constants(m3):
    """qpos0-time constants MuJoCo's compiler derives (set0): eq anchor2, invweight0, meaninertia."""
    mj = m3.mj
    q0 = m3.qpos0
    fkres = m3.fk(q0)
    xpos, xmat = fkres[0], fkres[1]
    M0 = m3.mass_matrix(q0)
    Minv = np.linalg.inv(M0)
    out = dict(meaninertia=float(np.mean(np.diag(M0))), dof_invweight0=np.diag(Minv).copy())
    inv_t, inv_r = np.zeros(len(m3.bodies)), np.zeros(len(m3.bodies))
    for bid in range(1, len(m3.bodies)):
        b = m3.bodies[bid]
        c = xpos[bid] + xmat[bid] @ b["ipos"]
        Jv, Jw = m3.jac(q0, bid, c, fkres)
        inv_t[bid] = np.trace(Jv @ Minv @ Jv.T) / 3.0
        inv_r[bid] = np.trace(Jw @ Minv @ Jw.T) / 3.0
    out["body_invweight0_tran"], out["body_invweight0_rot"] = inv_t, inv_r
    anchors2 = []
    for e in mj["eqs"]:
        pw = xpos[e["body1"]] + xmat[e["body1"]] @ e["anchor"]
        anchors2.append(xmat[e["body2"]].T @ (pw - xpos[e["body2"]]))
    out["eq_anchor2"] = anchors2
    out["M0"] = M0
    return out


# --------------------------------------------------------------------------- planar reduction
LINK_ROOT_BODIES = ["pelvis", "left_thigh", "left_knee", "left_tarsus", "left_toe", "left_achilles_rod",
                    "right_thigh", "right_knee", "right_tarsus", "right_toe", "right_achilles_rod"]
LINK_PARENT = [-1, 0, 1, 2, 3, 1, 0, 6, 7, 8, 6]
LINK_DOF = [2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12]  # the hinge dof that rotates each link (pelvis: pitch)


def planar_reduce(m3, consts):
    """Project the 3-D model at qpos0 onto the world x-z plane.

    Every link is a rigid planar body: origin o (its hinge anchor), absolute
    rotation theta about +y measured from the qpos0 pose.  All offsets below
    are world-axis (x,z) vectors AT qpos0, so at run time
        point(q) = o_link(q) + Ry(theta_link) * d0,   Ry(t)(dx,dz) = (c dx + s dz, -s dx + c dz).
    """
    mj = m3.mj
    names = [b["name"] for b in m3.bodies]
    q0 = m3.qpos0
    xpos, xmat, anchor, axis = m3.fk(q0)
    nb = len(m3.bodies)
    # body -> link (nearest ancestor-or-self that is a link root)
    roots = [names.index(n) for n in LINK_ROOT_BODIES]
    body_link = [-1] * nb
    for bid in range(1, nb):
        b = bid
        while b not in roots:
            b = m3.bodies[b]["parent"]
        body_link[bid] = roots.index(b)
    xz = lambda v: np.array([v[0], v[2]])
    # all hinge axes must be +-y, slides x and z, nothing may leave the plane
    sigma = np.zeros(m3.nv)
    for k, j in enumerate(m3.joints):
        if j["type"] == "hinge":
            assert abs(abs(axis[k][1]) - 1.0) < 1e-12, (j["name"], axis[k])
            sigma[k] = np.sign(axis[k][1])
    assert np.allclose(axis[0], [1, 0, 0]) and np.allclose(axis[1], [0, 0, 1])
    links = []
    for li, rb in enumerate(roots):
        o = xpos[rb]
        members = [b for b in range(1, nb) if body_link[b] == li]
        m = sum(m3.bodies[b]["mass"] for b in members)
        com = sum(m3.bodies[b]["mass"] * (xpos[b] + xmat[b] @ m3.bodies[b]["ipos"]) for b in members) / m
        Iyy = 0.0
        for b in members:
            cb = xpos[b] + xmat[b] @ m3.bodies[b]["ipos"]
            Iw = xmat[b] @ m3.bodies[b]["inertia"] @ xmat[b].T
            d = xz(cb - com)
            Iyy += Iw[1, 1] + m3.bodies[b]["mass"] * float(d @ d)
        par = LINK_PARENT[li]
        off = xz(o - xpos[roots[par]]) if par >= 0 else xz(o)
        links.append(dict(name=LINK_ROOT_BODIES[li], parent=par, dof=LINK_DOF[li], sigma=float(sigma[LINK_DOF[li]]),
                          off=off, mass=m, com=xz(com - o), inertia=Iyy, members=[names[b] for b in members]))

    def on_link(bid, p_world):
        li = body_link[bid]
        return li, xz(p_world - xpos[roots[li]])

    sites = []
    for s in mj["sites"]:
        li, d = on_link(s["body"], xpos[s["body"]] + xmat[s["body"]] @ s["pos"])
        sites.append(dict(name=s["name"], link=li, d=d))
    # collision spheres in MuJoCo contact order: body id ascending, within a capsule
    # the +axis ("to") end first, then the "from" end (mjc_PlaneCapsule).
    spheres = []
    for g in mj["geoms"]:
        if g["type"] == "plane":
            continue
        bid = g["body"]
        tran = consts["body_invweight0_tran"][bid]
        if g["type"] == "sphere":
            pw = xpos[bid] + xmat[bid] @ g["pos"]
            li, d = on_link(bid, pw)
            spheres.append(dict(geom=names[bid] + ":sphere", link=li, d=d, r=g["radius"], invweight=tran, y=float(pw[1])))
        else:
            for end, tag in ((g["to"], "to"), (g["from"], "from")):
                pw = xpos[bid] + xmat[bid] @ end
                li, d = on_link(bid, pw)
                spheres.append(dict(geom=names[bid] + ":capsule:" + tag, link=li, d=d, r=g["radius"], invweight=tran, y=float(pw[1])))
    eqs = []
    for e, a2 in zip(mj["eqs"], consts["eq_anchor2"]):
        l1, d1 = on_link(e["body1"], xpos[e["body1"]] + xmat[e["body1"]] @ e["anchor"])
        l2, d2 = on_link(e["body2"], xpos[e["body2"]] + xmat[e["body2"]] @ a2)
        eqs.append(dict(link1=l1, d1=d1, link2=l2, d2=d2,
                        invweight=consts["body_invweight0_tran"][e["body1"]] + consts["body_invweight0_tran"][e["body2"]]))
    return dict(links=links, sites=sites, spheres=spheres, eqs=eqs, sigma=sigma)


def planar_fk(pl, m3, q):
    """Reference implementation of the run-time planar FK (used to validate the reduction)."""
    nl = len(pl["links"])
    theta, o = np.zeros(nl), np.zeros((nl, 2))
    ry = lambda t, d: np.array([np.cos(t) * d[0] + np.sin(t) * d[1], -np.sin(t) * d[0] + np.cos(t) * d[1]])
    for li, L in enumerate(pl["links"]):
        k = L["dof"]
        if L["parent"] < 0:
            theta[li] = L["sigma"] * (q[k] - m3.qpos0[k])
            o[li] = L["off"] + np.array([q[0] - m3.qpos0[0], q[1] - m3.qpos0[1]])
        else:
            p = L["parent"]
            theta[li] = theta[p] + L["sigma"] * (q[k] - m3.qpos0[k])
            o[li] = o[p] + ry(theta[p], L["off"])
    return theta, o, ry


# --------------------------------------------------------------------------- emit
def _carr(name, a, fmt="%.17g", ctype="double", per_line=4, static="static const "):
    a = np.asarray(a)
    flat = a.reshape(-1)
    dims = "".join("[%d]" % d for d in a.shape)
    lines = []
    for i in range(0, len(flat), per_line):
        lines.append("  " + ", ".join(fmt % v for v in flat[i:i + per_line]))
    return "%s%s %s%s = {\n%s\n};\n" % (static, ctype, name, dims, ",\n".join(lines))


def emit_oracle_header(mj, m3s, consts, path):
    m3 = m3s["mj"]
    nb, nv = len(m3.bodies), m3.nv
    col = [g for g in mj["geoms"] if g["type"] != "plane"]
    plane = [g for g in mj["geoms"] if g["type"] == "plane"][0]
    o = []
    o.append("/* GENERATED by cassierl_amd/model/compile_model.py from model/cassie2d_stiff.xml -- do not edit.\n"
             " * 3-D tables for the CPU oracle (test infrastructure only).  Index 0 of the semantics axis\n"
             " * is MuJoCo compile semantics, 1 is DynamicModel::LoadModel (RBDL) semantics. */\n"
             "#ifndef CASSIE2D_MODEL_H_\n#define CASSIE2D_MODEL_H_\n")
    o.append("#define CM_NBODY %d\n#define CM_NV %d\n#define CM_NU %d\n#define CM_NSITE %d\n#define CM_NEQ %d\n"
             "#define CM_NGEOM %d\n" % (nb, nv, len(mj["acts"]), len(mj["sites"]), len(mj["eqs"]), len(col)))
    op = mj["option"]
    o.append("#define CM_TIMESTEP %.17g\n#define CM_ITERATIONS %d\n#define CM_TOLERANCE %.17g\n#define CM_GRAVITY_Z %.17g\n"
             "#define CM_IMPRATIO %.17g\n" % (op["timestep"], op["iterations"], op["tolerance"], op["gravity"][2], op["impratio"]))
    o.append("static const char* const cm_body_name[CM_NBODY] = {%s};\n" % ", ".join('"%s"' % b["name"] for b in m3.bodies))
    o.append(_carr("cm_body_parent", [b["parent"] for b in m3.bodies], "%d", "int", 11))
    o.append(_carr("cm_body_pos", [b["pos"] for b in m3.bodies]))
    o.append(_carr("cm_body_rot", [[m3s[s].bodies[i]["rot"] for i in range(nb)] for s in ("mj", "rbdl")]))
    o.append(_carr("cm_body_mass", [b["mass"] for b in m3.bodies]))
    o.append(_carr("cm_body_ipos", [b["ipos"] for b in m3.bodies]))
    o.append(_carr("cm_body_inertia", [b["inertia"] for b in m3.bodies]))
    o.append(_carr("cm_jnt_type", [0 if j["type"] == "slide" else 1 for j in m3.joints], "%d", "int", 13))
    o.append(_carr("cm_jnt_body", [j["body"] for j in m3.joints], "%d", "int", 13))
    o.append(_carr("cm_jnt_axis", [j["axis"] for j in m3.joints]))
    o.append(_carr("cm_jnt_ref", [j["ref"] for j in m3.joints]))
    o.append(_carr("cm_jnt_limited", [int(j["limited"]) for j in m3.joints], "%d", "int", 13))
    o.append(_carr("cm_jnt_range", [j["range"] for j in m3.joints]))
    o.append(_carr("cm_dof_damping", [j["damping"] for j in m3.joints]))
    o.append(_carr("cm_dof_armature", [j["armature"] for j in m3.joints]))
    o.append(_carr("cm_limit_solref", mj["limit_solref"]))
    o.append(_carr("cm_limit_solimp", mj["limit_solimp"]))
    o.append("/* collision geoms (all collide with the floor plane z=%g only; SURVEY.md R13) */\n" % plane["pos"][2])
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
    o.append(_carr("cm_act_dof", [a["dof"] for a in mj["acts"]], "%d", "int", 6))
    o.append(_carr("cm_act_gear", [a["gear"] for a in mj["acts"]]))
    o.append(_carr("cm_act_ctrlrange", [a["ctrlrange"] for a in mj["acts"]]))
    o.append("/* Compile-time constants as derived by THIS script (KAT cross-check; the oracle\n"
             " * re-derives them itself at init from the raw tables above). */\n")
    o.append(_carr("cm_kat_eq_anchor2", [consts[s]["eq_anchor2"] for s in ("mj", "rbdl")]))
    o.append(_carr("cm_kat_dof_invweight0", consts["mj"]["dof_invweight0"]))
    o.append(_carr("cm_kat_body_invweight0_tran", consts["mj"]["body_invweight0_tran"]))
    o.append("#define CM_KAT_MEANINERTIA %.17g\n" % consts["mj"]["meaninertia"])
    o.append("#endif\n")
    with open(path, "w") as f:
        f.write("\n".join(o))


def emit_planar_header(mj, m3s, consts, planars, path):
    m3 = m3s["mj"]
    o = []
    o.append("/* GENERATED by cassierl_amd/model/compile_model.py from model/cassie2d_stiff.xml -- do not edit.\n"
             " * Sagittal-plane reduction of Cassie2d for the HIP kernels.  11 planar links, 13 dof.\n"
             " * [2] axes are the model semantics: 0 = MuJoCo (physics), 1 = RBDL/LoadModel (controllers, op-space state).\n"
             " * All offsets are world-axis (x,z) vectors at qpos0; see planar_reduce() for the convention. */\n"
             "#ifndef CASSIE2D_PLANAR_H_\n#define CASSIE2D_PLANAR_H_\n")
    pl = planars["mj"]
    nl = len(pl["links"])
    op = mj["option"]
    o.append("#define CP_NLINK %d\n#define CP_NV %d\n#define CP_NU %d\n#define CP_NSITE %d\n#define CP_NSPHERE %d\n#define CP_NEQ %d\n"
             % (nl, m3.nv, len(mj["acts"]), len(pl["sites"]), len(pl["spheres"]), len(pl["eqs"])))
    o.append("#define CP_TIMESTEP %.17g\n#define CP_ITERATIONS %d\n#define CP_TOLERANCE %.17g\n#define CP_GRAVITY %.17g\n"
             % (op["timestep"], op["iterations"], op["tolerance"], -op["gravity"][2]))
    o.append("#define CP_MEANINERTIA %.17g\n" % consts["mj"]["meaninertia"])
    D = "static __device__ __constant__ const "
    ca = lambda n, a, fmt="%.17g", ct="double", pl_=4: _carr(n, a, fmt, ct, pl_, static=D)
    o.append(ca("cp_link_parent", [L["parent"] for L in pl["links"]], "%d", "int", 11))
    o.append(ca("cp_link_dof", [L["dof"] for L in pl["links"]], "%d", "int", 11))
    o.append(ca("cp_link_sigma", [L["sigma"] for L in pl["links"]]))
    o.append(ca("cp_link_off", [[L["off"] for L in planars[s]["links"]] for s in ("mj", "rbdl")]))
    o.append(ca("cp_link_mass", [L["mass"] for L in pl["links"]]))
    o.append(ca("cp_link_com", [[L["com"] for L in planars[s]["links"]] for s in ("mj", "rbdl")]))
    o.append(ca("cp_link_inertia", [[L["inertia"] for L in planars[s]["links"]] for s in ("mj", "rbdl")]))
    o.append(ca("cp_qpos0", m3.qpos0))
    o.append(ca("cp_dof_damping", [j["damping"] for j in m3.joints]))
    o.append(ca("cp_dof_armature", [j["armature"] for j in m3.joints]))
    o.append(ca("cp_dof_invweight0", consts["mj"]["dof_invweight0"]))
    o.append(ca("cp_jnt_limited", [int(j["limited"]) for j in m3.joints], "%d", "int", 13))
    o.append(ca("cp_jnt_range", [j["range"] for j in m3.joints]))
    o.append(ca("cp_limit_solref", mj["limit_solref"]))
    o.append(ca("cp_limit_solimp", mj["limit_solimp"]))
    o.append(ca("cp_site_link", [s["link"] for s in pl["sites"]], "%d", "int", 11))
    o.append(ca("cp_site_d", [[s["d"] for s in planars[sm]["sites"]] for sm in ("mj", "rbdl")]))
    o.append("/* collision spheres vs the floor plane z=0, in MuJoCo contact order */\n")
    o.append(ca("cp_sph_link", [s["link"] for s in pl["spheres"]], "%d", "int", 17))
    o.append(ca("cp_sph_d", [s["d"] for s in pl["spheres"]]))
    o.append(ca("cp_sph_r", [s["r"] for s in pl["spheres"]]))
    o.append("/* world y of each sphere centre (constant: the mechanism moves in the x-z plane); only the height-field lookup reads it */\n")
    o.append(ca("cp_sph_y", [s["y"] for s in pl["spheres"]]))
    o.append(ca("cp_sph_invweight", [s["invweight"] for s in pl["spheres"]]))
    col = [g for g in mj["geoms"] if g["type"] != "plane"][0]
    o.append(ca("cp_contact_solref", col["solref"]))
    o.append(ca("cp_contact_solimp", col["solimp"]))
    o.append("#define CP_CONTACT_MU %.17g\n" % col["friction"][0])
    o.append(ca("cp_eq_link1", [e["link1"] for e in pl["eqs"]], "%d", "int"))
    o.append(ca("cp_eq_link2", [e["link2"] for e in pl["eqs"]], "%d", "int"))
    o.append(ca("cp_eq_d1", [[e["d1"] for e in planars[s]["eqs"]] for s in ("mj", "rbdl")]))
    o.append(ca("cp_eq_d2", [[e["d2"] for e in planars[s]["eqs"]] for s in ("mj", "rbdl")]))
    o.append(ca("cp_eq_invweight", [e["invweight"] for e in pl["eqs"]]))
    o.append(ca("cp_eq_solref", [e["solref"] for e in mj["eqs"]]))
    o.append(ca("cp_eq_solimp", [e["solimp"] for e in mj["eqs"]]))
    o.append(ca("cp_act_dof", [a["dof"] for a in mj["acts"]], "%d", "int", 6))
    o.append(ca("cp_act_gear", [a["gear"] for a in mj["acts"]]))
    o.append(ca("cp_act_ctrlrange", [a["ctrlrange"] for a in mj["acts"]]))
    # ---- lane tables for the wave-per-environment kernels
    parent = [L["parent"] for L in pl["links"]]

    def is_anc(a, l):
        while l >= 0:
            if l == a:
                return True
            l = parent[l]
        return False
    dof_link = [0, 0] + [None] * (m3.nv - 2)
    for li, L in enumerate(pl["links"]):
        dof_link[L["dof"]] = li
    o.append("/* link lanes: bit k set <=> link k is an ancestor-or-self of this link */\n")
    o.append(ca("cp_link_ancmask", [sum(1 << k for k in range(nl) if is_anc(k, li)) for li in range(nl)], "%d", "int", 11))
    o.append("/* dof lanes: owning link, bit l set <=> link l is in the subtree moved by this dof */\n")
    o.append(ca("cp_dof_link", dof_link, "%d", "int", 13))
    o.append(ca("cp_dof_submask", [sum(1 << l for l in range(nl) if is_anc(dof_link[d], l)) for d in range(m3.nv)], "%d", "int", 13))
    o.append(ca("cp_dof_sigma", [0.0, 0.0] + [pl["links"][dof_link[d]]["sigma"] for d in range(2, m3.nv)]))
    rel = []
    for r in range(m3.nv):
        code = 0
        for c in range(m3.nv):
            if r < 2 or c < 2:
                continue
            if is_anc(dof_link[c], dof_link[r]):
                code |= 1 << (2 * c)      # column's link is ancestor-or-self of the row's: deep = row
            elif is_anc(dof_link[r], dof_link[c]):
                code |= 2 << (2 * c)      # column's link is a strict descendant: deep = column
        rel.append(code)
    o.append("/* 2 bits per column: 1 = column dof is ancestor-or-self of the row dof, 2 = strict descendant, 0 = unrelated */\n")
    o.append(ca("cp_dof_rel", rel, "%d", "int", 13))
    act_of = [-1] * m3.nv
    for a, A in enumerate(mj["acts"]):
        act_of[A["dof"]] = a
    o.append(ca("cp_dof_act", act_of, "%d", "int", 13))
    # compact 8-entry Jacobian layout: [0..2] base dofs, [3..7] the five dofs of the row's leg
    def pathmask8(li):
        if li == 0:
            return 0b111
        leg0 = 3 if li <= 5 else 8
        mk = 0b111
        p = li
        while p > 0:
            mk |= 1 << (3 + pl["links"][p]["dof"] - leg0)
            p = parent[p]
        return mk
    o.append(ca("cp_link_pathmask8", [pathmask8(li) for li in range(nl)], "%d", "int", 11))
    LIMIT_DOFS = [d for d in range(m3.nv) if m3.joints[d]["limited"]]
    assert LIMIT_DOFS == [3, 4, 5, 6, 8, 9, 10, 11]
    nslot = 4 + 8 + 2 * len(pl["spheres"])
    kind, leg, l1, l2, comp, ldof = [], [], [], [], [], []
    d1 = {s_: [] for s_ in ("mj", "rbdl")}
    d2 = {s_: [] for s_ in ("mj", "rbdl")}
    rad, invw = [], []
    for sl in range(nslot):
        if sl < 4:
            e, c = sl // 2, sl % 2
            kind.append(0); leg.append(e); comp.append(c); ldof.append(-1)
            l1.append(pl["eqs"][e]["link1"]); l2.append(pl["eqs"][e]["link2"])
            for s_ in ("mj", "rbdl"):
                d1[s_].append(planars[s_]["eqs"][e]["d1"]); d2[s_].append(planars[s_]["eqs"][e]["d2"])
            rad.append(0.0); invw.append(pl["eqs"][e]["invweight"])
        elif sl < 12:
            d = LIMIT_DOFS[sl - 4]
            kind.append(1); leg.append((sl - 4) // 4); comp.append(0); ldof.append(d)
            l1.append(0); l2.append(0)
            for s_ in ("mj", "rbdl"):
                d1[s_].append([0.0, 0.0]); d2[s_].append([0.0, 0.0])
            rad.append(0.0); invw.append(consts["mj"]["dof_invweight0"][d])
        else:
            c = (sl - 12) // 2
            S = pl["spheres"][c]
            kind.append(2 + (sl - 12) % 2); leg.append(0 if S["link"] <= 5 else 1); comp.append(1 - (sl - 12) % 2)
            ldof.append(-1); l1.append(S["link"]); l2.append(0)
            for s_ in ("mj", "rbdl"):
                d1[s_].append(S["d"]); d2[s_].append([0.0, 0.0])
            rad.append(S["r"]); invw.append(S["invweight"])
    o.append("/* Cassie2dEnv.reset pose (rllab/envs/cassie2d.py:79-85) and Cassie2d ctor pose (Cassie2d.cpp:56-58) */\n")
    o.append(ca("cp_env_qinit", [0.0, 0.939, 0.0, 0.68111815, -1.40730357, 1.62972042, -1.77611107, -0.61968407,
                                 0.68111815, -1.40730357, 1.62972042, -1.77611107, -0.61968407]))
    o.append(ca("cp_ctor_qinit", [0.0, 0.939, 0.0, 0.68111815, -1.40730357, 1.62972042, -1.77611107, -0.61968407,
                                  0.68111815, -1.40730353, 1.62972043, -1.77611107, -0.61968402]))
    o.append("/* fixed constraint slots: 0-3 connect rows (Lx,Lz,Rx,Rz), 4-11 joint limits, 12.. contact (normal,tangent) pairs */\n")
    o.append("#define CP_NSLOT %d\n" % nslot)
    o.append(ca("cp_slot_kind", kind, "%d", "int", 23))
    o.append(ca("cp_slot_leg", leg, "%d", "int", 23))
    o.append(ca("cp_slot_comp", comp, "%d", "int", 23))
    o.append(ca("cp_slot_dof", ldof, "%d", "int", 23))
    o.append(ca("cp_slot_link1", l1, "%d", "int", 23))
    o.append(ca("cp_slot_link2", l2, "%d", "int", 23))
    o.append(ca("cp_slot_d1", [d1["mj"], d1["rbdl"]]))
    o.append(ca("cp_slot_d2", [d2["mj"], d2["rbdl"]]))
    o.append(ca("cp_slot_radius", rad))
    o.append(ca("cp_slot_invweight", invw))
    o.append("#endif\n")
    with open(path, "w") as f:
        f.write("\n".join(o))


def main(xml=DEFAULT_XML):
    mj = parse_mjcf(xml)
    m3s = {s: Model3D(mj, s) for s in ("mj", "rbdl")}
    consts = {s: compile_constants(m3s[s]) for s in ("mj", "rbdl")}
    planars = {s: planar_reduce(m3s[s], consts[s]) for s in ("mj", "rbdl")}
    # ---- validate the planar reduction against 3-D FK at random configurations
    rng = np.random.default_rng(0)
    for s in ("mj", "rbdl"):
        m3, pl = m3s[s], planars[s]
        for _ in range(20):
            q = m3.qpos0 + rng.uniform(-0.7, 0.7, m3.nv)
            xpos, xmat, _, _ = m3.fk(q)
            theta, o, ry = planar_fk(pl, m3, q)
            for st in pl["sites"]:
                src = [x for x in mj["sites"] if x["name"] == st["name"]][0]
                pw = xpos[src["body"]] + xmat[src["body"]] @ src["pos"]
                pp = o[st["link"]] + ry(theta[st["link"]], st["d"])
                assert np.allclose([pw[0], pw[2]], pp, atol=1e-13), (s, st["name"], pw, pp)
    # ---- KATs
    m3 = m3s["mj"]
    names = [b["name"] for b in m3.bodies]
    qinit = np.array([0.0, 0.939, 0.0, 0.68111815, -1.40730357, 1.62972042, -1.77611107, -0.61968407,
                      0.68111815, -1.40730353, 1.62972043, -1.77611107, -0.61968402])  # Cassie2d.cpp:56-58
    kat = dict(source="cassie2d_stiff.xml via cassierl_amd/model/compile_model.py",
               total_mass=float(sum(b["mass"] for b in m3.bodies)),
               meaninertia=consts["mj"]["meaninertia"],
               qpos0=m3.qpos0.tolist(),
               eq_anchor2={s: [a.tolist() for a in consts[s]["eq_anchor2"]] for s in ("mj", "rbdl")},
               dof_invweight0=consts["mj"]["dof_invweight0"].tolist(),
               body_invweight0_tran={names[i]: float(v) for i, v in enumerate(consts["mj"]["body_invweight0_tran"])},
               M0_diag=np.diag(consts["mj"]["M0"]).tolist(), qpos_init=qinit.tolist())
    for s in ("mj", "rbdl"):
        xpos, xmat, _, _ = m3s[s].fk(qinit)
        kat["site_world_at_qinit_" + s] = {st["name"]: (xpos[st["body"]] + xmat[st["body"]] @ st["pos"]).tolist()
                                           for st in mj["sites"]}
        errs = []
        for e, a2 in zip(mj["eqs"], consts[s]["eq_anchor2"]):
            p1 = xpos[e["body1"]] + xmat[e["body1"]] @ e["anchor"]
            p2 = xpos[e["body2"]] + xmat[e["body2"]] @ a2
            errs.append((p1 - p2).tolist())
        kat["closure_error_at_qinit_" + s] = errs
        kat["M_at_qinit_" + s] = m3s[s].mass_matrix(qinit).tolist()
    emit_oracle_header(mj, m3s, consts, os.path.join(REPO, "oracle", "cassie2d_model.h"))
    emit_planar_header(mj, m3s, consts, planars, os.path.join(REPO, "cassierl_amd", "csrc", "cassie2d_planar.h"))
    os.makedirs(os.path.join(REPO, "tests", "golden"), exist_ok=True)
    with open(os.path.join(REPO, "tests", "golden", "model_kat.json"), "w") as f:
        json.dump(kat, f, indent=1)
    # planar tables as JSON (consumed by tools/planar_proto.py, the executable spec of the kernel math)
    def _j(x):
        if isinstance(x, dict):
            return {k: _j(v) for k, v in x.items()}
        if isinstance(x, (list, tuple)):
            return [_j(v) for v in x]
        if isinstance(x, np.ndarray):
            return x.tolist()
        if isinstance(x, (np.floating, np.integer)):
            return x.item()
        return x
    ptab = dict(planar={s: _j(planars[s]) for s in ("mj", "rbdl")}, qpos0=m3.qpos0.tolist(),
                option=_j({k: v for k, v in mj["option"].items()}), meaninertia=consts["mj"]["meaninertia"],
                dof=dict(damping=[j["damping"] for j in m3.joints], armature=[j["armature"] for j in m3.joints],
                         limited=[int(j["limited"]) for j in m3.joints], range=[j["range"].tolist() for j in m3.joints],
                         invweight0=consts["mj"]["dof_invweight0"].tolist()),
                limit=dict(solref=mj["limit_solref"].tolist(), solimp=mj["limit_solimp"].tolist()),
                contact=dict(solref=[g for g in mj["geoms"] if g["type"] != "plane"][0]["solref"].tolist(),
                             solimp=[g for g in mj["geoms"] if g["type"] != "plane"][0]["solimp"].tolist(),
                             mu=float([g for g in mj["geoms"] if g["type"] != "plane"][0]["friction"][0])),
                eq=dict(solref=[e["solref"].tolist() for e in mj["eqs"]], solimp=[e["solimp"].tolist() for e in mj["eqs"]]),
                act=dict(dof=[a["dof"] for a in mj["acts"]], gear=[a["gear"] for a in mj["acts"]],
                         ctrlrange=[a["ctrlrange"].tolist() for a in mj["acts"]]))
    with open(os.path.join(REPO, "tests", "golden", "planar_tables.json"), "w") as f:
        json.dump(ptab, f, indent=1)
    print("total mass %.4f  meaninertia %.6f" % (kat["total_mass"], kat["meaninertia"]))
    print("anchor2 mj  ", consts["mj"]["eq_anchor2"][0])
    print("anchor2 rbdl", consts["rbdl"]["eq_anchor2"][0])
    print("closure@qinit mj  ", kat["closure_error_at_qinit_mj"][0])
    print("closure@qinit rbdl", kat["closure_error_at_qinit_rbdl"][0])
    print("sites@qinit", {k: np.round(v, 5).tolist() for k, v in kat["site_world_at_qinit_mj"].items()})
    for L in planars["mj"]["links"]:
        print("%-20s m=%.4f I=%.6f com=%s off=%s  %s" % (L["name"], L["mass"], L["inertia"], np.round(L["com"], 5),
                                                       np.round(L["off"], 5), L["members"]))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else DEFAULT_XML)

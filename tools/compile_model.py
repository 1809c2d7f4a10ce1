#!/usr/bin/env python3
"""MJCF-subset model compiler for the rodent walker  (build-container tool).

Reads the reference's MJCF asset *where it lies* (never copied into this repo):
    /root/reference/track_mjx/environment/walker/assets/rodent/rodent.xml
applies the two spec edits the reference walker applies before compiling
    - torque-actuator rewrite            (reference: walker/rodent.py:70-78)
    - dm_scale_spec(rescale_factor)      (reference: walker/spec_utils.py:19-52)
and restates the parts of MuJoCo 3.3.2's model compiler the hot path depends on
(defaults resolution, frame resolution, inertia-from-geoms, kinematic-tree index
tables, qpos0 constants body_invweight0 / dof_invweight0 / stat.meaninertia,
static plane-vs-paw collision pair list in MJX's contact order).

Output: track_mjx_amd/assets/rodent_model.tmjx  (binary blob, float64/int32)
        track_mjx_amd/assets/rodent_model_dump.txt (human-readable review copy)

mujoco itself is not importable in the build container, so this is a
restatement from the published MuJoCo semantics ("parity unpinned").
All maths is float64 numpy; consumers narrow to fp32 like mjx.put_model does.
"""
from __future__ import annotations

import argparse
import sys
import xml.etree.ElementTree as ET
from collections import OrderedDict
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
from track_mjx_amd import blob  # noqa: E402

DEFAULT_XML = "/root/reference/track_mjx/environment/walker/assets/rodent/rodent.xml"

GEOM_PLANE, GEOM_SPHERE, GEOM_CAPSULE, GEOM_ELLIPSOID, GEOM_CYLINDER, GEOM_BOX = 0, 2, 3, 4, 5, 6
GEOM_TYPES = {"plane": 0, "hfield": 1, "sphere": 2, "capsule": 3, "ellipsoid": 4,
              "cylinder": 5, "box": 6, "mesh": 7}
JNT_FREE, JNT_BALL, JNT_SLIDE, JNT_HINGE = 0, 1, 2, 3


# --------------------------------------------------------------------------- math
def quat_mul(a, b):
    return np.array([
        a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3],
        a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
        a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1],
        a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0]])


def quat_to_mat(q):
    w, x, y, z = q
    return np.array([
        [w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y)],
        [2 * (x * y + w * z), w * w - x * x + y * y - z * z, 2 * (y * z - w * x)],
        [2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z]])


def mat_to_quat(R):
    # robust conversion, w >= 0
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q = np.array([0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s])
    elif R[0, 0] > R[1, 1] and R[0, 0] > R[2, 2]:
        s = np.sqrt(1.0 + R[0, 0] - R[1, 1] - R[2, 2]) * 2
        q = np.array([(R[2, 1] - R[1, 2]) / s, 0.25 * s, (R[0, 1] + R[1, 0]) / s, (R[0, 2] + R[2, 0]) / s])
    elif R[1, 1] > R[2, 2]:
        s = np.sqrt(1.0 + R[1, 1] - R[0, 0] - R[2, 2]) * 2
        q = np.array([(R[0, 2] - R[2, 0]) / s, (R[0, 1] + R[1, 0]) / s, 0.25 * s, (R[1, 2] + R[2, 1]) / s])
    else:
        s = np.sqrt(1.0 + R[2, 2] - R[0, 0] - R[1, 1]) * 2
        q = np.array([(R[1, 0] - R[0, 1]) / s, (R[0, 2] + R[2, 0]) / s, (R[1, 2] + R[2, 1]) / s, 0.25 * s])
    q = q / np.linalg.norm(q)
    return q if q[0] >= 0 else -q


def axis_angle_quat(axis, angle):
    axis = np.asarray(axis, float)
    return np.concatenate([[np.cos(angle / 2)], axis * np.sin(angle / 2)])


def euler_to_quat(e, seq="xyz"):
    """MuJoCo eulerseq semantics: lower case = intrinsic (q = q * r), upper = extrinsic."""
    q = np.array([1.0, 0, 0, 0])
    for ang, ch in zip(e, seq):
        ax = np.zeros(3)
        ax["xyz".index(ch.lower())] = 1.0
        r = axis_angle_quat(ax, ang)
        q = quat_mul(q, r) if ch.islower() else quat_mul(r, q)
    return q / np.linalg.norm(q)


def zaxis_to_quat(z):
    """Minimal rotation taking (0,0,1) to z (MuJoCo mjuu_z2quat)."""
    z = np.asarray(z, float)
    z = z / np.linalg.norm(z)
    ax = np.cross([0, 0, 1.0], z)
    s = np.linalg.norm(ax)
    if s < 1e-10:
        return np.array([1.0, 0, 0, 0]) if z[2] > 0 else np.array([0.0, 1, 0, 0])
    ang = np.arctan2(s, z[2])
    return axis_angle_quat(ax / s, ang)


def fvec(s):
    return np.array([float(x) for x in s.split()], dtype=float)


# --------------------------------------------------------------------------- parse
class Body:
    def __init__(self, name, parent):
        self.name, self.parent = name, parent
        self.pos = np.zeros(3)
        self.quat = np.array([1.0, 0, 0, 0])
        self.joints, self.geoms, self.sites, self.children = [], [], [], []
        self.id = -1


def parse_defaults(root):
    """class name -> {tag -> attrs}; nested classes inherit from their parent class."""
    classes = {}

    def rec(node, parent_attrs, name):
        attrs = {k: dict(v) for k, v in parent_attrs.items()}
        for ch in node:
            if ch.tag == "default":
                continue
            attrs.setdefault(ch.tag, {}).update(ch.attrib)
        classes[name] = attrs
        for ch in node:
            if ch.tag == "default":
                rec(ch, attrs, ch.attrib["class"])

    top = [d for d in root.findall("default")]
    base = {}
    for d in top:
        rec(d, base, d.attrib.get("class", "main"))
        base = classes[d.attrib.get("class", "main")]
    if "main" not in classes:
        classes["main"] = {}
    return classes


def resolve(el, tag, classes, childclass):
    cls = el.attrib.get("class", childclass or "main")
    out = dict(classes.get(cls, {}).get(tag, {}))
    out.update({k: v for k, v in el.attrib.items() if k != "class"})
    return out


def orientation(attrs, eulerseq="xyz"):
    if "quat" in attrs:
        q = fvec(attrs["quat"])
        return q / np.linalg.norm(q)
    if "euler" in attrs:
        return euler_to_quat(fvec(attrs["euler"]), eulerseq)
    if "zaxis" in attrs:
        return zaxis_to_quat(fvec(attrs["zaxis"]))
    if "axisangle" in attrs or "xyaxes" in attrs:
        raise NotImplementedError("axisangle/xyaxes orientation not used by the rodent model's physics elements")
    return np.array([1.0, 0, 0, 0])


def parse_body(el, parent, classes, childclass, bodies):
    b = Body(el.attrib.get("name", f"body{len(bodies)}"), parent)
    b.id = len(bodies)
    bodies.append(b)
    if "pos" in el.attrib:
        b.pos = fvec(el.attrib["pos"])
    b.quat = orientation(el.attrib)
    cc = el.attrib.get("childclass", childclass)
    for ch in el:
        if ch.tag in ("joint", "freejoint"):
            a = resolve(ch, "joint", classes, cc) if ch.tag == "joint" else dict(ch.attrib)
            if ch.tag == "freejoint":
                a["type"] = "free"
            b.joints.append(a)
        elif ch.tag == "geom":
            b.geoms.append(resolve(ch, "geom", classes, cc))
        elif ch.tag == "site":
            b.sites.append(resolve(ch, "site", classes, cc))
    for ch in el:
        if ch.tag == "body":
            b.children.append(parse_body(ch, b, classes, cc, bodies))
    return b


# --------------------------------------------------------------------------- geoms
def geom_size3(g):
    s = fvec(g.get("size", "0"))
    out = np.zeros(3)
    out[:min(3, len(s))] = s[:3]
    return out


def geom_volume_inertia(gtype, size):
    """Volume and unit-density principal inertia of a primitive (MuJoCo user_objects.cc)."""
    if gtype == GEOM_SPHERE:
        r = size[0]
        v = 4.0 / 3.0 * np.pi * r ** 3
        i = 2.0 / 5.0 * v * r * r
        return v, np.array([i, i, i])
    if gtype == GEOM_CAPSULE:
        r, hh = size[0], size[1]
        h = 2 * hh
        v = np.pi * (r * r * h + 4.0 / 3.0 * r ** 3)
        sphere_mass = v * 4 * r / (4 * r + 3 * h)
        cyl_mass = v - sphere_mass
        ix = cyl_mass * (3 * r * r + h * h) / 12.0
        iz = cyl_mass * r * r / 2.0
        si = 2.0 * sphere_mass * r * r / 5.0
        ix += si + sphere_mass * h * (3 * r + 2 * h) / 8.0
        iz += si
        return v, np.array([ix, ix, iz])
    if gtype == GEOM_CYLINDER:
        r, hh = size[0], size[1]
        h = 2 * hh
        v = np.pi * r * r * h
        ix = v * (3 * r * r + h * h) / 12.0
        return v, np.array([ix, ix, v * r * r / 2.0])
    if gtype == GEOM_ELLIPSOID:
        a, b, c = size
        v = 4.0 / 3.0 * np.pi * a * b * c
        return v, v / 5.0 * np.array([b * b + c * c, a * a + c * c, a * a + b * b])
    if gtype == GEOM_BOX:
        a, b, c = size
        v = 8 * a * b * c
        return v, v / 3.0 * np.array([b * b + c * c, a * a + c * c, a * a + b * b])
    if gtype == GEOM_PLANE:
        return 0.0, np.zeros(3)
    raise NotImplementedError(gtype)


# --------------------------------------------------------------------------- compile
def compile_model(xml_path, torque_actuators=True, rescale_factor=0.9):
    root = ET.parse(xml_path).getroot()
    comp = {}
    for c in root.findall("compiler"):
        comp.update(c.attrib)
    assert comp.get("angle", "degree") == "radian", "rodent.xml declares angle=radian"
    classes = parse_defaults(root)

    bodies = []
    world = Body("world", None)
    world.id = 0
    bodies.append(world)
    wb = root.find("worldbody")
    for ch in wb:
        if ch.tag == "geom":
            world.geoms.append(resolve(ch, "geom", classes, None))
    for ch in wb:
        if ch.tag == "body":
            world.children.append(parse_body(ch, world, classes, None, bodies))
    nbody = len(bodies)
    name2body = {b.name: b for b in bodies}

    # ---- spec edit (b): dm_scale_spec — every body below "walker": body.pos, geom.size, geom.pos
    s = float(rescale_factor)
    if s != 1.0:
        def scale_rec(parent):
            for b in parent.children:
                b.pos = b.pos * s
                for g in b.geoms:
                    g["_scale"] = s
                scale_rec(b)
        scale_rec(name2body["walker"])

    # ---- joints / dofs
    jnt_type, jnt_bodyid, jnt_qposadr, jnt_dofadr, jnt_pos, jnt_axis = [], [], [], [], [], []
    jnt_range, jnt_stiffness, jnt_springref, jnt_ref, jnt_limited, jnt_names = [], [], [], [], [], []
    jnt_solref, jnt_solimp, jnt_margin = [], [], []
    dof_bodyid, dof_jntid, dof_damping, dof_armature = [], [], [], []
    body_jntadr, body_jntnum, body_dofadr, body_dofnum = [], [], [], []
    nq = nv = 0
    for b in bodies:
        body_jntadr.append(len(jnt_type) if b.joints else -1)
        body_jntnum.append(len(b.joints))
        body_dofadr.append(nv if b.joints else -1)
        nd0 = nv
        for a in b.joints:
            t = {"free": JNT_FREE, "ball": JNT_BALL, "slide": JNT_SLIDE, "hinge": JNT_HINGE}[a.get("type", "hinge")]
            assert t in (JNT_FREE, JNT_HINGE), "rodent model has only free + hinge joints"
            jid = len(jnt_type)
            jnt_names.append(a.get("name", f"jnt{jid}"))
            jnt_type.append(t)
            jnt_bodyid.append(b.id)
            jnt_qposadr.append(nq)
            jnt_dofadr.append(nv)
            jnt_pos.append(fvec(a["pos"]) if "pos" in a else np.zeros(3))
            ax = fvec(a["axis"]) if "axis" in a else np.array([0, 0, 1.0])
            jnt_axis.append(ax / np.linalg.norm(ax))
            rng = fvec(a["range"]) if "range" in a else np.zeros(2)
            jnt_range.append(rng)
            lim = a.get("limited", "auto")
            jnt_limited.append(1 if (lim == "true" or (lim == "auto" and "range" in a and comp.get("autolimits", "true") == "true")) and t == JNT_HINGE else 0)
            jnt_stiffness.append(float(a.get("stiffness", 0)))
            jnt_springref.append(float(a.get("springref", 0)))
            jnt_ref.append(float(a.get("ref", 0)))
            jnt_solref.append(fvec(a.get("solreflimit", "0.02 1")))
            si = fvec(a.get("solimplimit", "0.9 0.95 0.001 0.5 2"))
            full = np.array([0.9, 0.95, 0.001, 0.5, 2.0])
            full[:len(si)] = si
            jnt_solimp.append(full)
            jnt_margin.append(float(a.get("margin", 0)))
            ndof = 6 if t == JNT_FREE else 1
            assert float(a.get("frictionloss", 0)) == 0.0
            for _ in range(ndof):
                dof_bodyid.append(b.id)
                dof_jntid.append(jid)
                dof_damping.append(0.0 if t == JNT_FREE else float(a.get("damping", 0)))
                dof_armature.append(0.0 if t == JNT_FREE else float(a.get("armature", 0)))
            nq += 7 if t == JNT_FREE else 1
            nv += ndof
        body_dofnum.append(nv - nd0)
    njnt = len(jnt_type)
    jnt_type = np.array(jnt_type)
    jnt_name2id = {n: i for i, n in enumerate(jnt_names)}

    body_parentid = np.array([b.parent.id if b.parent else 0 for b in bodies])
    # root id: topmost ancestor below world
    body_rootid = np.zeros(nbody, int)
    for b in bodies[1:]:
        body_rootid[b.id] = b.id if b.parent.id == 0 else body_rootid[b.parent.id]
    # dof_parentid: previous dof in the same body, else last dof of nearest ancestor with dofs
    dof_parentid = np.full(nv, -1)
    body_lastdof = np.full(nbody, -1)
    for b in bodies:
        anc = body_lastdof[b.parent.id] if b.parent else -1
        last = anc
        if b.joints:
            for d in range(body_dofadr[b.id], body_dofadr[b.id] + body_dofnum[b.id]):
                dof_parentid[d] = last
                last = d
        body_lastdof[b.id] = last

    # ---- geoms
    geoms = []
    for b in bodies:
        for g in b.geoms:
            gt = GEOM_TYPES[g.get("type", "sphere")]
            assert "fromto" not in g and "mass" not in g and gt != 7
            size = geom_size3(g)
            pos = fvec(g["pos"]) if "pos" in g else np.zeros(3)
            sc = g.get("_scale", 1.0)
            size, pos = size * sc, pos * sc
            geoms.append(dict(
                name=g.get("name", ""), body=b.id, type=gt, size=size, pos=pos, quat=orientation(g),
                density=float(g.get("density", 1000.0)),
                contype=int(g.get("contype", 1)), conaffinity=int(g.get("conaffinity", 1)),
                condim=int(g.get("condim", 3)), priority=int(g.get("priority", 0)),
                friction=fvec(g.get("friction", "1 0.005 0.0001")),
                solref=fvec(g.get("solref", "0.02 1")), solimp=g.get("solimp", "0.9 0.95 0.001 0.5 2"),
                margin=float(g.get("margin", 0)), gap=float(g.get("gap", 0)),
                solmix=float(g.get("solmix", 1))))
    ngeom = len(geoms)

    # ---- inertia from geoms (inertiafromgeom=auto, no <inertial> elements in this model)
    body_mass = np.zeros(nbody)
    body_ipos = np.zeros((nbody, 3))
    body_iquat = np.tile([1.0, 0, 0, 0], (nbody, 1))
    body_inertia = np.zeros((nbody, 3))
    for b in bodies:
        gs = [g for g in geoms if g["body"] == b.id]
        ms, coms = [], []
        for g in gs:
            vol, _ = geom_volume_inertia(g["type"], g["size"])
            ms.append(vol * g["density"])
            coms.append(g["pos"])
        M = float(np.sum(ms)) if ms else 0.0
        if M <= 0:
            continue
        com = np.sum([m * c for m, c in zip(ms, coms)], axis=0) / M
        I = np.zeros((3, 3))
        for g, m in zip(gs, ms):
            vol, ip = geom_volume_inertia(g["type"], g["size"])
            R = quat_to_mat(g["quat"])
            I += R @ np.diag(ip * g["density"]) @ R.T
            d = g["pos"] - com
            I += m * (d @ d * np.eye(3) - np.outer(d, d))
        w, V = np.linalg.eigh(I)
        order = np.argsort(-w)  # MuJoCo's eig3 returns eigenvalues in decreasing order
        w, V = w[order], V[:, order]
        if np.linalg.det(V) < 0:
            V[:, 2] = -V[:, 2]
        body_mass[b.id], body_ipos[b.id] = M, com
        body_inertia[b.id], body_iquat[b.id] = w, mat_to_quat(V)
    def weld_mass(b):
        # mass of the body plus everything rigidly welded below it (MuJoCo accepts a massless
        # moving body when its welded children carry mass: walker -> torso)
        return body_mass[b.id] + sum(weld_mass(c) for c in b.children if not c.joints)
    for b in bodies:
        if b.joints and weld_mass(b) <= 1e-15:
            raise ValueError(f"moving body {b.name} has no mass")

    # ---- tendons (fixed) and actuators
    tendon_names, ten_coef = [], []
    for t in root.find("tendon") if root.find("tendon") is not None else []:
        assert t.tag == "fixed"
        a = resolve(t, "tendon", classes, None)
        assert a.get("limited", "auto") == "false", "tendon limits would add constraint rows"
        assert float(a.get("stiffness", 0)) == 0 and float(a.get("damping", 0)) == 0 and float(a.get("frictionloss", 0)) == 0
        coef = np.zeros(nv)
        for j in t.findall("joint"):
            coef[jnt_dofadr[jnt_name2id[j.attrib["joint"]]]] = float(j.attrib["coef"])
        tendon_names.append(a["name"])
        ten_coef.append(coef)
    act_names, act_moment, act_gain, act_tau, act_ctrlrange = [], [], [], [], []
    for ael in root.find("actuator"):
        assert ael.tag == "general"
        a = resolve(ael, "general", classes, None)
        assert a.get("dyntype") == "filter" and a.get("forcelimited", "false") == "false"
        assert a.get("ctrllimited") == "true"
        gear = float(a.get("gear", "1").split()[0])
        gain = float(a["gainprm"].split()[0])
        bias = fvec(a.get("biasprm", "0 0 0"))
        biastype = a.get("biastype", "none")
        if torque_actuators:
            # spec edit (a): gainprm[0] = forcerange[1]; biastype none; biasprm 0
            fr = fvec(a["forcerange"])
            gain, biastype, bias = fr[1], "none", np.zeros(3)
        assert biastype == "none", "non-torque actuator mode is not compiled (reference config uses torque_actuators=True)"
        gear *= s * s  # dm_scale_spec: gear *= scale^2
        if "joint" in a:
            mom = np.zeros(nv)
            mom[jnt_dofadr[jnt_name2id[a["joint"]]]] = gear
        else:
            mom = gear * ten_coef[tendon_names.index(a["tendon"])]
        act_names.append(a["name"])
        act_moment.append(mom)
        act_gain.append(gain)
        act_tau.append(float(a["dynprm"].split()[0]))
        act_ctrlrange.append(fvec(a["ctrlrange"]))
    nu = len(act_names)

    # ---- options
    opt = {}
    for o in root.findall("option"):
        opt.update(o.attrib)
    gravity = fvec(opt.get("gravity", "0 0 -9.81"))

    m = dict(
        nbody=nbody, njnt=njnt, nq=nq, nv=nv, nu=nu, ngeom=ngeom,
        body_names=[b.name for b in bodies], jnt_names=jnt_names, act_names=act_names,
        geom_names=[g["name"] for g in geoms],
        body_parentid=body_parentid, body_rootid=body_rootid,
        body_pos=np.array([b.pos for b in bodies]), body_quat=np.array([b.quat for b in bodies]),
        body_mass=body_mass, body_ipos=body_ipos, body_iquat=body_iquat, body_inertia=body_inertia,
        body_jntadr=np.array(body_jntadr), body_jntnum=np.array(body_jntnum),
        body_dofadr=np.array(body_dofadr), body_dofnum=np.array(body_dofnum),
        jnt_type=jnt_type, jnt_bodyid=np.array(jnt_bodyid), jnt_qposadr=np.array(jnt_qposadr),
        jnt_dofadr=np.array(jnt_dofadr), jnt_pos=np.array(jnt_pos), jnt_axis=np.array(jnt_axis),
        jnt_range=np.array(jnt_range), jnt_limited=np.array(jnt_limited),
        jnt_stiffness=np.array(jnt_stiffness), jnt_springref=np.array(jnt_springref), jnt_ref=np.array(jnt_ref),
        jnt_solref=np.array(jnt_solref), jnt_solimp=np.array(jnt_solimp), jnt_margin=np.array(jnt_margin),
        dof_bodyid=np.array(dof_bodyid), dof_jntid=np.array(dof_jntid), dof_parentid=dof_parentid,
        dof_damping=np.array(dof_damping), dof_armature=np.array(dof_armature),
        act_moment=np.array(act_moment), act_gain=np.array(act_gain), act_tau=np.array(act_tau),
        act_ctrlrange=np.array(act_ctrlrange),
        gravity=gravity, geoms=geoms,
    )
    # qpos0 / qpos_spring
    qpos0 = np.zeros(nq)
    qpos_spring = np.zeros(nq)
    for j in range(njnt):
        a = jnt_qposadr[j]
        if jnt_type[j] == JNT_FREE:
            b = bodies[jnt_bodyid[j]]
            qpos0[a:a + 3], qpos0[a + 3:a + 7] = b.pos, b.quat
            qpos_spring[a:a + 7] = qpos0[a:a + 7]
        else:
            qpos0[a] = jnt_ref[j]
            qpos_spring[a] = jnt_springref[j]
    m["qpos0"], m["qpos_spring"] = qpos0, qpos_spring

    _collision_pairs(m)
    _set_const(m)
    return m


def _collision_pairs(m):
    """Static broadphase in MJX's order (mjx/_src/collision_driver.py _geom_pairs/_geom_groups,
    from memory of 3.3.x): body pairs (b1<=b2) ascending, geoms ascending, filtered by
    contype/conaffinity; grouped by (type1,type2) in first-seen order, condim ascending."""
    geoms = m["geoms"]
    by_body = {}
    for gi, g in enumerate(geoms):
        by_body.setdefault(g["body"], []).append(gi)
    pairs = []
    for b1 in range(m["nbody"]):
        for b2 in range(b1, m["nbody"]):
            for g1 in by_body.get(b1, []):
                for g2 in by_body.get(b2, []):
                    if b1 == b2 and g2 <= g1:
                        continue
                    A, B = geoms[g1], geoms[g2]
                    mask = (A["contype"] & B["conaffinity"]) or (B["contype"] & A["conaffinity"])
                    if not mask:
                        continue
                    # parent-child / same-body filtering (world-welded bodies are exempt, as in MuJoCo)
                    if b1 == b2:
                        continue
                    if A["type"] > B["type"]:
                        g1_, g2_ = g2, g1
                    else:
                        g1_, g2_ = g1, g2
                    pairs.append((g1_, g2_))
    groups = OrderedDict()
    for g1, g2 in pairs:
        key = (geoms[g1]["type"], geoms[g2]["type"])
        groups.setdefault(key, []).append((g1, g2))
    ncon_of = {(GEOM_PLANE, GEOM_CAPSULE): 2, (GEOM_PLANE, GEOM_ELLIPSOID): 1, (GEOM_PLANE, GEOM_SPHERE): 1}
    con_geom1, con_geom2, con_sub, con_type = [], [], [], []
    for key, lst in groups.items():
        if key not in ncon_of:
            raise NotImplementedError(f"collider {key} not on the rodent hot path")
        for g1, g2 in lst:
            for k in range(ncon_of[key]):
                con_geom1.append(g1)
                con_geom2.append(g2)
                con_sub.append(k)
                con_type.append(key[1])
    m["con_geom1"], m["con_geom2"] = np.array(con_geom1), np.array(con_geom2)
    m["con_sub"], m["con_type"] = np.array(con_sub), np.array(con_type)
    # contact parameter mixing (priority decides here: paw priority 1 > floor 0)
    fr, sr, si, cd = [], [], [], []
    for g1, g2 in zip(con_geom1, con_geom2):
        A, B = geoms[g1], geoms[g2]
        assert A["priority"] != B["priority"], "equal-priority mixing not needed by this model"
        W = A if A["priority"] > B["priority"] else B
        assert W["condim"] == 3 and max(A["margin"], B["margin"]) == 0 and max(A["gap"], B["gap"]) == 0
        fr.append(W["friction"])
        sr.append(W["solref"])
        full = np.array([0.9, 0.95, 0.001, 0.5, 2.0])
        v = fvec(W["solimp"])
        full[:len(v)] = v
        si.append(full)
        cd.append(W["condim"])
    m["con_friction"], m["con_solref"], m["con_solimp"] = np.array(fr), np.array(sr), np.array(si)
    m["ncon"] = len(con_geom1)


def fk(m, qpos):
    """Forward kinematics in float64 (used for qpos0 constants and for synthetic clips)."""
    nb = m["nbody"]
    xpos, xquat = np.zeros((nb, 3)), np.tile([1.0, 0, 0, 0], (nb, 1))
    xanchor, xaxis = np.zeros((m["njnt"], 3)), np.zeros((m["njnt"], 3))
    for b in range(1, nb):
        p = m["body_parentid"][b]
        R = quat_to_mat(xquat[p])
        pos = xpos[p] + R @ m["body_pos"][b]
        quat = quat_mul(xquat[p], m["body_quat"][b])
        for j in range(m["body_jntadr"][b], m["body_jntadr"][b] + m["body_jntnum"][b]) if m["body_jntnum"][b] else []:
            a = m["jnt_qposadr"][j]
            if m["jnt_type"][j] == JNT_FREE:
                pos = qpos[a:a + 3].copy()
                quat = qpos[a + 3:a + 7] / np.linalg.norm(qpos[a + 3:a + 7])
                xanchor[j], xaxis[j] = pos, [0, 0, 1]
            else:
                Rq = quat_to_mat(quat)
                xanchor[j] = Rq @ m["jnt_pos"][j] + pos
                xaxis[j] = Rq @ m["jnt_axis"][j]
                quat = quat_mul(quat, axis_angle_quat(m["jnt_axis"][j], qpos[a] - m["qpos0"][a]))
                pos = xanchor[j] - quat_to_mat(quat) @ m["jnt_pos"][j]
        xpos[b], xquat[b] = pos, quat / np.linalg.norm(quat)
    return xpos, xquat, xanchor, xaxis


def mass_matrix(m, qpos):
    """Dense joint-space inertia by the Jacobian method (independent of the CRB used by the hot path)."""
    nv, nb = m["nv"], m["nbody"]
    xpos, xquat, xanchor, xaxis = fk(m, qpos)
    M = np.diag(m["dof_armature"]).astype(float)
    jacs = {}
    for b in range(1, nb):
        R = quat_to_mat(xquat[b])
        com = xpos[b] + R @ m["body_ipos"][b]
        jp, jr = np.zeros((3, nv)), np.zeros((3, nv))
        d = m["body_dofadr"][b] + m["body_dofnum"][b] - 1 if m["body_dofnum"][b] else -1
        if d < 0:
            # nearest ancestor dof
            a = m["body_parentid"][b]
            while a > 0 and m["body_dofnum"][a] == 0:
                a = m["body_parentid"][a]
            d = m["body_dofadr"][a] + m["body_dofnum"][a] - 1 if a > 0 else -1
        while d >= 0:
            j = m["dof_jntid"][d]
            if m["jnt_type"][j] == JNT_FREE:
                k = d - m["jnt_dofadr"][j]
                if k < 3:
                    jp[k, d] = 1.0
                else:
                    ax = quat_to_mat(xquat[m["jnt_bodyid"][j]])[:, k - 3]
                    jr[:, d] = ax
                    jp[:, d] = np.cross(ax, com - xpos[m["jnt_bodyid"][j]])
            else:
                jr[:, d] = xaxis[j]
                jp[:, d] = np.cross(xaxis[j], com - xanchor[j])
            d = m["dof_parentid"][d]
        jacs[b] = (jp, jr)
        if m["body_mass"][b] > 0:
            Ri = R @ quat_to_mat(m["body_iquat"][b])
            Iw = Ri @ np.diag(m["body_inertia"][b]) @ Ri.T
            M += m["body_mass"][b] * jp.T @ jp + jr.T @ Iw @ jr
    return M, jacs


def _set_const(m):
    """qpos0-dependent constants (MuJoCo engine_setconst.c set0): dof_invweight0, body_invweight0, meaninertia."""
    M, jacs = mass_matrix(m, m["qpos0"])
    Minv = np.linalg.inv(M)
    nv = m["nv"]
    dinv = np.diag(Minv).copy()
    for j in range(m["njnt"]):
        if m["jnt_type"][j] == JNT_FREE:
            a = m["jnt_dofadr"][j]
            dinv[a:a + 3] = dinv[a:a + 3].mean()
            dinv[a + 3:a + 6] = dinv[a + 3:a + 6].mean()
    binv = np.zeros((m["nbody"], 2))
    for b, (jp, jr) in jacs.items():
        if not jp.any() and not jr.any():
            continue
        binv[b, 0] = np.trace(jp @ Minv @ jp.T) / 3
        binv[b, 1] = np.trace(jr @ Minv @ jr.T) / 3
    m["dof_invweight0"], m["body_invweight0"] = dinv, binv
    m["meaninertia"] = float(np.trace(M) / nv)
    m["M0"] = M


# --------------------------------------------------------------------------- emit
def to_blob(m):
    e = OrderedDict()
    e["dims"] = np.array([m["nbody"], m["njnt"], m["nq"], m["nv"], m["nu"], m["ncon"]], dtype=np.int32)
    for k in ("body_parentid", "body_rootid", "body_jntadr", "body_jntnum", "body_dofadr", "body_dofnum",
              "jnt_type", "jnt_bodyid", "jnt_qposadr", "jnt_dofadr", "jnt_limited",
              "dof_bodyid", "dof_jntid", "dof_parentid"):
        e[k] = np.asarray(m[k], dtype=np.int32)
    for k in ("body_pos", "body_quat", "body_mass", "body_ipos", "body_iquat", "body_inertia",
              "jnt_pos", "jnt_axis", "jnt_range", "jnt_stiffness", "jnt_solref", "jnt_solimp", "jnt_margin",
              "qpos0", "qpos_spring", "dof_damping", "dof_armature", "dof_invweight0", "body_invweight0",
              "act_moment", "act_gain", "act_tau", "act_ctrlrange", "gravity"):
        e[k] = np.asarray(m[k], dtype=np.float64).ravel()
    e["meaninertia"] = np.array([m["meaninertia"]])
    g = m["geoms"]
    g1, g2 = m["con_geom1"], m["con_geom2"]
    e["con_geom1"] = g1.astype(np.int32)
    e["con_geom2"] = g2.astype(np.int32)
    e["con_sub"] = m["con_sub"].astype(np.int32)
    e["con_type"] = m["con_type"].astype(np.int32)
    e["con_body2"] = np.array([g[i]["body"] for i in g2], dtype=np.int32)
    e["con_body1"] = np.array([g[i]["body"] for i in g1], dtype=np.int32)
    e["con_friction"] = m["con_friction"].ravel()
    e["con_solref"] = m["con_solref"].ravel()
    e["con_solimp"] = m["con_solimp"].ravel()
    # geometry of the two sides of each contact slot (plane: body-local frame; paw geom: body-local frame)
    for side, idx in (("g1", g1), ("g2", g2)):
        e[f"con_{side}_pos"] = np.array([g[i]["pos"] for i in idx]).ravel()
        e[f"con_{side}_quat"] = np.array([g[i]["quat"] for i in idx]).ravel()
        e[f"con_{side}_size"] = np.array([g[i]["size"] for i in idx]).ravel()
    return e


def dump_text(m, path):
    np.set_printoptions(precision=10, linewidth=160, suppress=False)
    with open(path, "w") as f:
        f.write("# compiled rodent model (tools/compile_model.py) - review copy of the blob\n")
        f.write(f"nbody={m['nbody']} njnt={m['njnt']} nq={m['nq']} nv={m['nv']} nu={m['nu']} ngeom={m['ngeom']} ncon={m['ncon']}\n")
        f.write(f"total_mass={m['body_mass'].sum():.10g} meaninertia={m['meaninertia']:.10g}\n\n# bodies: id name parent root mass ipos inertia invweight0\n")
        for i, n in enumerate(m["body_names"]):
            f.write(f"{i:3d} {n:22s} p={m['body_parentid'][i]:2d} r={m['body_rootid'][i]:2d} pos={m['body_pos'][i]} "
                    f"mass={m['body_mass'][i]:.8g} ipos={m['body_ipos'][i]} inertia={m['body_inertia'][i]} invw={m['body_invweight0'][i]}\n")
        f.write("\n# joints: id name type body qposadr dofadr axis pos range stiffness springref damping armature invweight0\n")
        for j, n in enumerate(m["jnt_names"]):
            d = m["jnt_dofadr"][j]
            f.write(f"{j:3d} {n:28s} t={m['jnt_type'][j]} b={m['jnt_bodyid'][j]:2d} q={m['jnt_qposadr'][j]:2d} d={d:2d} "
                    f"axis={m['jnt_axis'][j]} pos={m['jnt_pos'][j]} range={m['jnt_range'][j]} k={m['jnt_stiffness'][j]} "
                    f"sref={m['jnt_springref'][j]} damp={m['dof_damping'][d]} arm={m['dof_armature'][d]} invw={m['dof_invweight0'][d]:.8g} "
                    f"dofparent={m['dof_parentid'][d]}\n")
        f.write("\n# actuators: id name gain tau nonzero-moment\n")
        for a, n in enumerate(m["act_names"]):
            nz = np.nonzero(m["act_moment"][a])[0]
            f.write(f"{a:3d} {n:22s} gain={m['act_gain'][a]:.6g} tau={m['act_tau'][a]} moment={{" +
                    ", ".join(f"{d}:{m['act_moment'][a][d]:.8g}" for d in nz) + "}\n")
        f.write("\n# contact slots (MJX order): slot geom1 geom2 sub body2\n")
        for c in range(m["ncon"]):
            g2 = m["geoms"][m["con_geom2"][c]]
            f.write(f"{c:3d} g1={m['con_geom1'][c]:3d} g2={m['con_geom2'][c]:3d} ({g2['name']}) sub={m['con_sub'][c]} type={m['con_type'][c]} "
                    f"body={g2['body']} size={g2['size']} friction={m['con_friction'][c]}\n")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--xml", default=DEFAULT_XML)
    ap.add_argument("--out", default=str(REPO / "track_mjx_amd" / "assets" / "rodent_model.tmjx"))
    ap.add_argument("--rescale", type=float, default=0.9)
    ap.add_argument("--no-torque", action="store_true")
    args = ap.parse_args()
    m = compile_model(args.xml, torque_actuators=not args.no_torque, rescale_factor=args.rescale)
    e = to_blob(m)
    # names travel as a text side file (ids for the walker's name -> id lookups)
    blob.save(args.out, e)
    names = Path(args.out).with_suffix(".names.txt")
    with open(names, "w") as f:
        for kind, lst in (("body", m["body_names"]), ("joint", m["jnt_names"]), ("actuator", m["act_names"])):
            for i, n in enumerate(lst):
                f.write(f"{kind} {i} {n}\n")
    dump_text(m, Path(args.out).with_name("rodent_model_dump.txt"))
    print(f"wrote {args.out}: nbody={m['nbody']} nq={m['nq']} nv={m['nv']} nu={m['nu']} ncon={m['ncon']} "
          f"mass={m['body_mass'].sum():.6f} meaninertia={m['meaninertia']:.6g}")


if __name__ == "__main__":
    main()

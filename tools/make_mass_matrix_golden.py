#!/usr/bin/env python3
"""Joint-space mass matrices of the ten in-scope robots, computed STRAIGHT FROM THE URDF FILES by an implementation that shares
nothing with the product's model pipeline (mppiisaac/backend/urdf_compile.py -> assets/compiled/*.json -> Scene.to_c ->
mppi_model_t -> pack_model): its own XML walk, its own rpy / transform algebra, its own convex-hull mass properties, and a
different algorithm - M(q) = sum_links m Jv^T Jv + Jw^T I Jw from point Jacobians of every link's centre of mass, no spatial
algebra, no body merging, no z-framing.  tests/test_mass_matrix_golden.py inverts the oracle's articulated-body algorithm column
by column on the SAME joint positions and compares: a wrong inertia, axis, joint frame or unit anywhere between the URDF and the
model blob both the oracle and the kernels consume would show here ("what's weak" 1: the shared-packer blind spot).

Runs only where /root/reference exists (this container); the output tests/golden/mass_matrices.json is data (joint names,
positions, matrices).  Modelling conventions restated, not imported: links without <inertial> get the mass properties of their
collision geometry at 1000 kg/m^3 (meshes: convex hull), SURVEY.md B / D.    Usage: python tools/make_mass_matrix_golden.py"""
import json
import math
import os
import struct
import sys
import xml.etree.ElementTree as ET

import numpy as np
from scipy.spatial import ConvexHull

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("MPPI_REFERENCE", "/root/reference")
RHO = 1000.0
URDFS = ["point_robot.urdf", "panda_isaac/robots/franka_panda_stick.urdf", "panda_isaac/robots/franka_panda_gripper.urdf",
         "panda_isaac/robots/franka_panda.urdf", "boxer/boxer.urdf", "heijn/heijn.urdf", "jackal/jackal.urdf", "albert/albert.urdf",
         "omni_panda/omniPandaWithGripper.urdf", "anymal_c/urdf/anymal.urdf"]


def rot_rpy(r, p, y):
    """URDF fixed-axis roll-pitch-yaw: R = Rz(y) Ry(p) Rx(r)"""
    cr, sr, cp, sp, cy, sy = math.cos(r), math.sin(r), math.cos(p), math.sin(p), math.cos(y), math.sin(y)
    Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def rot_axis(axis, angle):
    """Rodrigues"""
    a = np.asarray(axis, float)
    a = a / np.linalg.norm(a)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + math.sin(angle) * K + (1 - math.cos(angle)) * (K @ K)


def frame_of(elem):
    o = elem.find("origin") if elem is not None else None
    if o is None:
        return np.eye(3), np.zeros(3)
    xyz = [float(v) for v in (o.get("xyz") or "0 0 0").split()]
    rpy = [float(v) for v in (o.get("rpy") or "0 0 0").split()]
    return rot_rpy(*rpy), np.array(xyz)


def mesh_points(path):
    if path.lower().endswith(".obj"):
        return np.array([[float(v) for v in ln.split()[1:4]] for ln in open(path, errors="ignore") if ln.startswith("v ")])
    raw = open(path, "rb").read()
    n = struct.unpack_from("<I", raw, 80)[0]
    if 84 + 50 * n == len(raw):
        return np.array([struct.unpack_from("<9f", raw, 84 + 50 * i + 12) for i in range(n)]).reshape(-1, 3)
    return np.array([[float(v) for v in ln.split()[1:4]] for ln in raw.decode("ascii", "ignore").splitlines() if ln.strip().startswith("vertex")])


def find_mesh(name, urdf):
    d = os.path.dirname(os.path.abspath(urdf))
    if name.startswith("package://"):
        rel = name[len("package://"):]
        for _ in range(7):
            for cand in (os.path.join(d, rel), os.path.join(os.path.dirname(d), rel)):
                if os.path.exists(cand):
                    return cand
            d = os.path.dirname(d)
        raise FileNotFoundError(name)
    return os.path.join(d, name)


def hull_properties(pts):
    """mass, centre of mass, inertia about the centre of mass of the convex hull at RHO: tetrahedra from an INTERIOR point (the
    vertex mean) to the hull triangles; a tetrahedron with edge matrix A (columns = its three edges from the apex) has volume
    |det A| / 6, centroid apex + (sum of edges) / 4 and covariance about its apex |det A| * A C0 A^T with the canonical
    C0 = (1/120) (1 + I)  (Tonon 2004)"""
    hull = ConvexHull(pts)
    apex = pts[hull.vertices].mean(0)
    C0 = (np.ones((3, 3)) + np.eye(3)) / 120.0
    vol, moment, cov = 0.0, np.zeros(3), np.zeros((3, 3))
    for tri in hull.simplices:
        A = (pts[tri] - apex).T
        d = abs(np.linalg.det(A))
        v = d / 6.0
        c = A.sum(1) / 4.0                      # centroid relative to the apex
        vol += v
        moment += v * c
        cov += d * (A @ C0 @ A.T)               # second moment about the apex
    com_rel = moment / vol
    cov_c = cov - vol * np.outer(com_rel, com_rel)
    I = RHO * (np.trace(cov_c) * np.eye(3) - cov_c)
    return RHO * vol, apex + com_rel, I


def link_inertial(link, urdf):
    """(mass, com in the link frame, inertia about the com in link axes) of one URDF link"""
    ine = link.find("inertial")
    if ine is not None:
        R, p = frame_of(ine)
        m = float(ine.find("mass").get("value"))
        t = ine.find("inertia")
        g = lambda k: float(t.get(k, 0.0))
        I = np.array([[g("ixx"), g("ixy"), g("ixz")], [g("ixy"), g("iyy"), g("iyz")], [g("ixz"), g("iyz"), g("izz")]])
        return m, p, R @ I @ R.T
    parts = []
    for col in link.findall("collision"):
        R, p = frame_of(col)
        geo = col.find("geometry")
        if geo is None:
            continue
        g = list(geo)[0]
        if g.tag == "box":
            x, y, z = (float(v) for v in g.get("size").split())
            m = RHO * x * y * z
            parts.append((m, p, R @ (m / 12.0 * np.diag([y * y + z * z, x * x + z * z, x * x + y * y])) @ R.T))
        elif g.tag == "sphere":
            r = float(g.get("radius"))
            m = RHO * 4.0 / 3.0 * math.pi * r ** 3
            parts.append((m, p, 0.4 * m * r * r * np.eye(3)))
        elif g.tag == "cylinder":
            r, L = float(g.get("radius")), float(g.get("length"))
            m = RHO * math.pi * r * r * L
            a = m * (3 * r * r + L * L) / 12.0
            parts.append((m, p, R @ np.diag([a, a, 0.5 * m * r * r]) @ R.T))
        elif g.tag == "mesh":
            s = np.array([float(v) for v in g.get("scale").split()]) if g.get("scale") else np.ones(3)
            m, c, I = hull_properties(mesh_points(find_mesh(g.get("filename"), urdf)) * s)
            parts.append((m, R @ c + p, R @ I @ R.T))
    if not parts:
        return 0.0, np.zeros(3), np.zeros((3, 3))
    M = sum(m for m, _, _ in parts)
    com = sum(m * c for m, c, _ in parts) / M
    I = np.zeros((3, 3))
    for m, c, Ic in parts:          # parallel axes to the common centre of mass
        d = c - com
        I += Ic + m * (d @ d * np.eye(3) - np.outer(d, d))
    return M, com, I


def mass_matrix(urdf, q_of_joint):
    root = ET.parse(urdf).getroot()
    links = {l.get("name"): l for l in root.findall("link")}
    joints = root.findall("joint")
    child_names = {j.find("child").get("link") for j in joints}
    parent_names = {j.find("parent").get("link") for j in joints}
    tops = [n for n in links if n not in child_names and n in parent_names] or [n for n in links if n not in child_names]
    assert len(tops) == 1, tops
    moving = [j.get("name") for j in joints if j.get("type") in ("revolute", "continuous", "prismatic")]
    n = len(moving)
    M = np.zeros((n, n))

    def walk(name, R, p, chain):
        """R, p: link frame in the world; chain: [(dof index, kind, world axis, world point on the axis)] of the joints above"""
        m, c, I = link_inertial(links[name], urdf)
        if m > 0 or np.any(I):     # (albert's mmrobot_link8: mass 0, inertia 0.3 - a pure rotational inertia)
            cw, Iw = R @ c + p, R @ I @ R.T
            Jv, Jw = np.zeros((3, n)), np.zeros((3, n))
            for i, kind, ax, o in chain:
                if kind == "prismatic":
                    Jv[:, i] = ax
                else:
                    Jv[:, i], Jw[:, i] = np.cross(ax, cw - o), ax
            M[:] += m * Jv.T @ Jv + Jw.T @ Iw @ Jw
        for j in joints:
            if j.find("parent").get("link") != name:
                continue
            Rj, pj = frame_of(j)
            Rw, pw = R @ Rj, R @ pj + p
            kind = j.get("type")
            if kind == "fixed":
                walk(j.find("child").get("link"), Rw, pw, chain)
                continue
            ax_el = j.find("axis")
            a = np.array([float(v) for v in ax_el.get("xyz").split()]) if ax_el is not None else np.array([1.0, 0.0, 0.0])
            a = a / np.linalg.norm(a)
            qi = q_of_joint[j.get("name")]
            i = moving.index(j.get("name"))
            if kind == "prismatic":
                walk(j.find("child").get("link"), Rw, pw + Rw @ a * qi, chain + [(i, kind, Rw @ a, pw)])
            else:
                walk(j.find("child").get("link"), Rw @ rot_axis(a, qi), pw, chain + [(i, "revolute", Rw @ a, pw)])
    walk(tops[0], np.eye(3), np.zeros(3), [])
    return moving, M


def joint_ranges(urdf):
    out = {}
    for j in ET.parse(urdf).getroot().findall("joint"):
        if j.get("type") in ("revolute", "continuous", "prismatic"):
            lim = j.find("limit")
            lo = float(lim.get("lower", -1.0)) if lim is not None and j.get("type") != "continuous" else -1.5
            hi = float(lim.get("upper", 1.0)) if lim is not None and j.get("type") != "continuous" else 1.5
            if not hi > lo:
                lo, hi = -1.0, 1.0
            out[j.get("name")] = (lo, hi)
    return out


if __name__ == "__main__":
    rng = np.random.default_rng(20260928)
    out = {"generated_by": "tools/make_mass_matrix_golden.py (independent of mppiisaac/backend/urdf_compile.py)", "density": RHO, "robots": []}
    for rel in URDFS:
        path = os.path.join(REF, "assets", "urdf", rel)
        if not os.path.exists(path):
            print("skip (missing)", rel)
            continue
        ranges = joint_ranges(path)
        cases = []
        for _ in range(3):
            q = {k: float(rng.uniform(lo + 0.1 * (hi - lo), hi - 0.1 * (hi - lo))) for k, (lo, hi) in ranges.items()}
            names, M = mass_matrix(path, q)
            cases.append({"q": [q[n] for n in names], "M": M.tolist()})
        out["robots"].append({"urdf_file": rel, "joints": names, "cases": cases})
        print(f"{rel}: {len(names)} dof, diag(M) of case 0 = {np.round(np.diag(np.asarray(cases[0]['M'])), 4).tolist()}")
    with open(os.path.join(ROOT, "tests", "golden", "mass_matrices.json"), "w") as f:
        json.dump(out, f)
    print("wrote tests/golden/mass_matrices.json")

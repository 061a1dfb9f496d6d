"""URDF -> flat "compiled model" for the HIP rollout kernels and the CPU oracle.

This replaces what Isaac Gym's asset importer does for the reference
(`gym.load_asset`, reference mppiisaac/utils/isaacgym_utils.py:14-29): it reads the
kinematic tree, joint frames/axes/limits and link inertias, and - like Isaac Gym with
its default `AssetOptions.density = 1000` - derives the inertia of links that carry no
`<inertial>` tag from the convex hull of their collision geometry.

The output is a plain dict (JSON-serialisable, a few KB per robot) with

* ``links``   every URDF link in depth-first order (the rigid-body order used for
              ``rigid_body_state`` rows, reference isaacgym_wrapper.py:193-195), each with
              the moving body it is rigidly attached to and the fixed offset from it;
* ``bodies``  the *moving* bodies (one per non-fixed joint, DFS order == DOF order):
              parent body, joint type/axis, tree transform, limits, composite rigid
              inertia of the link plus everything welded to it;
* ``base``    composite inertia of the root link cluster (used when the base floats).

Conventions: a transform (R, p) maps child-frame coordinates to parent-frame
coordinates, x_parent = R @ x_child + p.  Rotations from URDF rpy are fixed-axis
XYZ (R = Rz(y) Ry(p) Rx(r)).  Inertias are stored about the *body-frame origin*,
expressed in body-frame axes, as (m, m*c, I_o[xx,xy,xz,yy,yz,zz]).

The compiler runs where the URDF assets live; the GPU box only ever sees the JSON.
"""
from __future__ import annotations

import json
import math
import os
import struct
import xml.etree.ElementTree as ET
from typing import List, Optional, Tuple

import numpy as np

DENSITY = 1000.0  # kg/m^3, Isaac Gym AssetOptions.density default


# ----------------------------------------------------------------------------- math helpers
def rpy_to_R(rpy) -> np.ndarray:
    r, p, y = (float(v) for v in rpy)
    cr, sr, cp, sp, cy, sy = math.cos(r), math.sin(r), math.cos(p), math.sin(p), math.cos(y), math.sin(y)
    Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def skew(v) -> np.ndarray:
    x, y, z = v
    return np.array([[0, -z, y], [z, 0, -x], [-y, x, 0]], dtype=float)


def _floats(s: Optional[str], n: int, default: float = 0.0) -> List[float]:
    if s is None:
        return [default] * n
    vals = [float(t) for t in s.split()]
    assert len(vals) == n, f"expected {n} floats in '{s}'"
    return vals


class Inertia:
    """Rigid-body inertia about the frame origin: mass m, first moment h = m*c, I_o (3x3)."""

    def __init__(self, m=0.0, h=None, Io=None):
        self.m = float(m)
        self.h = np.zeros(3) if h is None else np.asarray(h, float)
        self.Io = np.zeros((3, 3)) if Io is None else np.asarray(Io, float)

    @staticmethod
    def from_com(m: float, c, Ic) -> "Inertia":
        c = np.asarray(c, float)
        Ic = np.asarray(Ic, float)
        Io = Ic + m * (np.dot(c, c) * np.eye(3) - np.outer(c, c))
        return Inertia(m, m * c, Io)

    def transformed(self, R: np.ndarray, p: np.ndarray) -> "Inertia":
        """Express this inertia (given in a child frame) in the parent frame, x_p = R x_c + p."""
        if self.m == 0.0 and not self.Io.any():
            return Inertia()
        c = self.h / self.m if self.m > 0 else np.zeros(3)
        Ic = self.Io - self.m * (np.dot(c, c) * np.eye(3) - np.outer(c, c))
        return Inertia.from_com(self.m, R @ c + p, R @ Ic @ R.T)

    def __add__(self, o: "Inertia") -> "Inertia":
        return Inertia(self.m + o.m, self.h + o.h, self.Io + o.Io)

    def to_dict(self) -> dict:
        I = self.Io
        return {
            "mass": self.m,
            "h": self.h.tolist(),
            "Io": [I[0, 0], I[0, 1], I[0, 2], I[1, 1], I[1, 2], I[2, 2]],
        }


# ----------------------------------------------------------------------------- mesh readers
def read_obj_vertices(path: str) -> np.ndarray:
    verts = []
    with open(path, "r", errors="ignore") as f:
        for line in f:
            if line.startswith("v "):
                t = line.split()
                verts.append([float(t[1]), float(t[2]), float(t[3])])
    return np.asarray(verts, float)


def read_stl_vertices(path: str) -> np.ndarray:
    with open(path, "rb") as f:
        data = f.read()
    ntri = struct.unpack_from("<I", data, 80)[0]
    if 84 + 50 * ntri == len(data):  # binary STL
        out = np.empty((ntri * 3, 3), float)
        for i in range(ntri):
            vals = struct.unpack_from("<12f", data, 84 + 50 * i)
            out[3 * i:3 * i + 3] = np.asarray(vals[3:12]).reshape(3, 3)
        return out
    verts = []  # ascii STL
    for line in data.decode("ascii", errors="ignore").splitlines():
        t = line.split()
        if len(t) == 4 and t[0] == "vertex":
            verts.append([float(t[1]), float(t[2]), float(t[3])])
    return np.asarray(verts, float)


def read_mesh_vertices(path: str) -> np.ndarray:
    ext = os.path.splitext(path)[1].lower()
    if ext == ".obj":
        return read_obj_vertices(path)
    if ext == ".stl":
        return read_stl_vertices(path)
    raise NotImplementedError(f"mesh format {ext} ({path})")


def hull_mass_properties(verts: np.ndarray, density: float = DENSITY) -> Tuple[float, np.ndarray, np.ndarray]:
    """Mass, centre of mass and inertia about the COM of the convex hull of `verts`.

    Signed tetrahedra from the origin over the outward-oriented hull facets:
    V = det/6, int x dV = det/24 * s, int x x^T dV = det/120 * (sum v v^T + s s^T), s = a+b+c.
    """
    from scipy.spatial import ConvexHull

    hull = ConvexHull(verts)
    vol = 0.0
    first = np.zeros(3)
    second = np.zeros((3, 3))
    for simplex, eq in zip(hull.simplices, hull.equations):
        a, b, c = verts[simplex[0]], verts[simplex[1]], verts[simplex[2]]
        if np.dot(np.cross(b - a, c - a), eq[:3]) < 0:  # orient outward
            b, c = c, b
        det = float(np.dot(a, np.cross(b, c)))
        s = a + b + c
        vol += det / 6.0
        first += det / 24.0 * s
        second += det / 120.0 * (np.outer(a, a) + np.outer(b, b) + np.outer(c, c) + np.outer(s, s))
    com = first / vol
    C_o = second  # second moment about origin
    I_o = np.trace(C_o) * np.eye(3) - C_o
    m = density * vol
    I_c = density * I_o - m * (np.dot(com, com) * np.eye(3) - np.outer(com, com))
    return m, com, I_c


# ----------------------------------------------------------------------------- URDF parsing
def _origin(elem) -> Tuple[np.ndarray, np.ndarray]:
    o = elem.find("origin") if elem is not None else None
    if o is None:
        return np.eye(3), np.zeros(3)
    return rpy_to_R(_floats(o.get("rpy"), 3)), np.asarray(_floats(o.get("xyz"), 3))


def _resolve_mesh(filename: str, urdf_path: str) -> str:
    urdf_dir = os.path.dirname(os.path.abspath(urdf_path))
    if filename.startswith("package://"):
        rel = filename[len("package://"):]
        # package root = the directory named like the first path component, searched upward
        d = urdf_dir
        for _ in range(6):
            cand = os.path.join(d, rel)
            if os.path.exists(cand):
                return cand
            cand = os.path.join(os.path.dirname(d), rel)
            if os.path.exists(cand):
                return cand
            d = os.path.dirname(d)
        raise FileNotFoundError(filename)
    return os.path.join(urdf_dir, filename)


def _collision_shapes(link_elem, urdf_path: str) -> List[dict]:
    """Collision primitives of a link; meshes are reduced to their AABB box
    while their hull mass properties are kept for inertia derivation."""
    shapes = []
    for col in link_elem.findall("collision"):
        R, p = _origin(col)
        geom = col.find("geometry")
        if geom is None:
            continue
        g = list(geom)[0]
        if g.tag == "box":
            sz = _floats(g.get("size"), 3)
            shapes.append({"type": "box", "size": sz, "R": R.tolist(), "p": p.tolist()})
        elif g.tag == "sphere":
            shapes.append({"type": "sphere", "radius": float(g.get("radius")), "R": R.tolist(), "p": p.tolist()})
        elif g.tag == "cylinder":
            shapes.append({"type": "cylinder", "radius": float(g.get("radius")), "length": float(g.get("length")),
                           "R": R.tolist(), "p": p.tolist()})
        elif g.tag == "mesh":
            scale = _floats(g.get("scale"), 3, 1.0) if g.get("scale") else [1.0, 1.0, 1.0]
            verts = read_mesh_vertices(_resolve_mesh(g.get("filename"), urdf_path)) * np.asarray(scale)
            lo, hi = verts.min(0), verts.max(0)
            m, com, Ic = hull_mass_properties(verts)
            shapes.append({"type": "mesh", "aabb_min": lo.tolist(), "aabb_max": hi.tolist(),
                           "hull_mass": m, "hull_com": com.tolist(), "hull_Ic": Ic.tolist(),
                           "R": R.tolist(), "p": p.tolist()})
    return shapes


def _shape_inertia(shape: dict) -> Inertia:
    """Inertia of one collision shape at DENSITY, in the link frame."""
    R, p = np.asarray(shape["R"]), np.asarray(shape["p"])
    t = shape["type"]
    if t == "box":
        x, y, z = shape["size"]
        m = DENSITY * x * y * z
        Ic = m / 12.0 * np.diag([y * y + z * z, x * x + z * z, x * x + y * y])
        loc = Inertia.from_com(m, np.zeros(3), Ic)
    elif t == "sphere":
        r = shape["radius"]
        m = DENSITY * 4.0 / 3.0 * math.pi * r ** 3
        loc = Inertia.from_com(m, np.zeros(3), 0.4 * m * r * r * np.eye(3))
    elif t == "cylinder":  # axis z
        r, L = shape["radius"], shape["length"]
        m = DENSITY * math.pi * r * r * L
        ixx = m * (3 * r * r + L * L) / 12.0
        loc = Inertia.from_com(m, np.zeros(3), np.diag([ixx, ixx, 0.5 * m * r * r]))
    elif t == "mesh":
        loc = Inertia.from_com(shape["hull_mass"], shape["hull_com"], np.asarray(shape["hull_Ic"]))
    else:
        raise NotImplementedError(t)
    return loc.transformed(R, p)


def _link_inertia(link_elem, shapes: List[dict]) -> Tuple[Inertia, str]:
    inertial = link_elem.find("inertial")
    if inertial is not None:
        R, p = _origin(inertial)
        m = float(inertial.find("mass").get("value"))
        it = inertial.find("inertia")
        g = lambda k: float(it.get(k, 0.0))
        Ic = np.array([[g("ixx"), g("ixy"), g("ixz")], [g("ixy"), g("iyy"), g("iyz")], [g("ixz"), g("iyz"), g("izz")]])
        return Inertia.from_com(m, p, R @ Ic @ R.T), "urdf"
    tot = Inertia()
    for s in shapes:
        tot = tot + _shape_inertia(s)
    return tot, ("hull@1000" if shapes else "massless")


def compile_urdf(urdf_path: str, name: Optional[str] = None) -> dict:
    root = ET.parse(urdf_path).getroot()
    link_elems = {l.get("name"): l for l in root.findall("link")}
    joints = []
    for j in root.findall("joint"):
        R, p = _origin(j)
        ax = j.find("axis")
        axis = np.asarray(_floats(ax.get("xyz"), 3)) if ax is not None else np.array([1.0, 0.0, 0.0])
        lim = j.find("limit")
        joints.append({
            "name": j.get("name"), "type": j.get("type"),
            "parent": j.find("parent").get("link"), "child": j.find("child").get("link"),
            "R": R, "p": p, "axis": axis,
            "effort": float(lim.get("effort", 0.0)) if lim is not None else 0.0,
            "velocity": float(lim.get("velocity", 0.0)) if lim is not None else 0.0,
            "lower": float(lim.get("lower", 0.0)) if lim is not None else 0.0,
            "upper": float(lim.get("upper", 0.0)) if lim is not None else 0.0,
        })
    children = {j["child"] for j in joints}
    roots = [n for n in link_elems if n not in children]
    # a dangling link (e.g. panda_link8 in franka_panda_gripper.urdf) is a second root: ignore
    # roots that have no outgoing joint unless they are the only one.
    parents = {j["parent"] for j in joints}
    roots = [r for r in roots if r in parents] or roots
    assert len(roots) == 1, f"expected one root link, got {roots}"
    root_link = roots[0]

    links: List[dict] = []
    bodies: List[dict] = []
    base_inertia = Inertia()

    def visit(link_name: str, parent_link_idx: int, body_idx: int, R_bl: np.ndarray, p_bl: np.ndarray):
        """body_idx = moving body this link is welded to (-1 = base); (R_bl,p_bl) link->body."""
        nonlocal base_inertia
        elem = link_elems[link_name]
        shapes = _collision_shapes(elem, urdf_path)
        inertia, src = _link_inertia(elem, shapes)
        link_idx = len(links)
        links.append({
            "name": link_name, "parent_link": parent_link_idx, "body": body_idx,
            "R": R_bl.tolist(), "p": p_bl.tolist(),
            "inertia_source": src, "own_inertia": inertia.to_dict(), "collision": shapes,
        })
        in_body = inertia.transformed(R_bl, p_bl)
        if body_idx < 0:
            base_inertia = base_inertia + in_body
        else:
            bodies[body_idx]["_inertia"] = bodies[body_idx]["_inertia"] + in_body
        for j in joints:
            if j["parent"] != link_name:
                continue
            # joint frame (== child link frame at q=0) expressed in the current body frame
            R_bj = R_bl @ j["R"]
            p_bj = R_bl @ j["p"] + p_bl
            if j["type"] == "fixed":
                visit(j["child"], link_idx, body_idx, R_bj, p_bj)
            elif j["type"] in ("revolute", "continuous", "prismatic"):
                axis = j["axis"] / np.linalg.norm(j["axis"])
                limited = j["type"] != "continuous"
                new_idx = len(bodies)
                bodies.append({
                    "name": j["child"], "joint": j["name"], "parent": body_idx,
                    "jtype": "prismatic" if j["type"] == "prismatic" else "revolute",
                    "axis": axis.tolist(), "R_tree": R_bj.tolist(), "p_tree": p_bj.tolist(),
                    "limited": bool(limited), "lower": j["lower"], "upper": j["upper"],
                    "effort": j["effort"], "velocity": j["velocity"],
                    "_inertia": Inertia(),
                })
                visit(j["child"], link_idx, new_idx, np.eye(3), np.zeros(3))
            else:
                raise NotImplementedError(f"joint type {j['type']}")

    visit(root_link, -1, -1, np.eye(3), np.zeros(3))
    for b in bodies:
        b["inertia"] = b.pop("_inertia").to_dict()
    return {
        "format": "mppi-hip-model/1",
        "name": name or root.get("name"),
        "source": os.path.relpath(urdf_path, start=os.path.join(os.path.dirname(urdf_path), "..")),
        "root_link": root_link,
        "links": links,
        "bodies": bodies,
        "base": {"inertia": base_inertia.to_dict()},
    }


def prune_links(model: dict, keep=()) -> dict:
    """Drop the links nobody can observe - no collision geometry, not named in `keep` - from the reported rigid bodies of a
    compiled model (their inertia is already merged into the bodies).  Isaac Gym reports every URDF link as a rigid body; the
    78 links of the ANYmal URDF (camera frames, shells, ...) are three times MPPI_MAX_LINKS.  Objectives address links by name,
    so the rows that remain keep their meaning; only their count differs from the reference's tensor."""
    links = model["links"]
    kept = [i for i, l in enumerate(links) if i == 0 or l["collision"] or l["name"] in keep]
    new_index = {old: new for new, old in enumerate(kept)}

    def kept_ancestor(i):
        i = links[i]["parent_link"]
        while i >= 0 and i not in new_index:
            i = links[i]["parent_link"]
        return new_index.get(i, -1)
    out = dict(model)
    out["links"] = [dict(links[i], parent_link=kept_ancestor(i)) for i in kept]
    out["pruned_links"] = len(links) - len(kept)
    return out


def save_model(model: dict, path: str) -> None:
    def rnd(o):
        if isinstance(o, float):
            return float(f"{o:.12g}")
        if isinstance(o, list):
            return [rnd(v) for v in o]
        if isinstance(o, dict):
            return {k: rnd(v) for k, v in o.items()}
        return o
    with open(path, "w") as f:
        json.dump(rnd(model), f, indent=1)
        f.write("\n")


def load_model(path: str) -> dict:
    with open(path) as f:
        return json.load(f)

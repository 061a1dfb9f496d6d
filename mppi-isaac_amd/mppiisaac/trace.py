"""Tracing a reference-style `Objective.compute_cost(sim)` into a cost program.

The reference's users write their stage cost as torch code over the sim getters (reference examples/*/planner.py:
`sim.get_actor_link_by_name(...)[:, 0:3] - sim.get_actor_position_by_name(...)[:, 0:3]`, `torch.linalg.norm(..., axis=1)`,
`torch.sum(torch.abs(forces[:, 0:2]), axis=1)`, ... called once per rollout step, mppi_isaac.py:67-69).  Run eagerly that is ~70
torch launches per control iteration (generic mode).  Every example objective is a weighted sum of the nine
measurements of `mppiisaac.objectives`, so here the Objective is called ONCE on a symbolic sim whose getters return `Sym` proxies;
the arithmetic the Objective performs on them is recorded as linear forms over a few kinds of atoms, and `match()` maps the result
to a `Term` list - the cost program the rollout kernels evaluate in registers (`objectives.compile_program`).

Anything the tracer does not understand raises `TraceError` with the reason: the planner then runs the Objective in generic mode and
says why (MPPIisaacPlanner._bind_objective).  A traced program is validated against the eager Objective on the first command and
every 64th (planner/mppi.py), so an Objective whose Python side changes (an attribute other than `.weights`) is noticed.

What is understood (the grammar of the reference's eleven planners and the point-robot benchmark cost):
  getters      get_actor_link_by_name [K, 13], get_actor_position_by_name / _velocity_ / _orientation_ [K, 3 | 3 | 4],
               get_actor_contact_forces_by_name [K, 3], get_dof_state [K, 2 n]; keyword or positional arguments
  indexing     x[:, a:b(:s)], x[:, i], x[:, -2:] on those and on their results
  arithmetic   + - between proxies and with numbers / constant tensors / lists, * / by numbers, * and / between proxies
               (dot products and the cosine of the push-align term), ** 2, unary -
  torch        torch.linalg.norm / torch.norm (axis | dim = 1), torch.sum (axis | dim = 1), torch.abs, torch.square, torch.clamp(min=0)
  rotations    mppiisaac.utils.conversions.quaternion_to_yaw, quaternion_to_matrix + matrix_to_euler_angles(..., "ZYX") - the
               stand-ins of pytorch3d.transforms, which the image lacks (pytorch3d's own functions take the Sym apart with
               torch.unbind and are reported as untraceable)."""
import math
from typing import Dict, List, Tuple

import torch

from mppiisaac.objectives import Term


class TraceError(Exception):
    """the Objective does something the tracer cannot express as a cost program (the message says what)"""


# ---- linear forms over atoms ----------------------------------------------------------------------------------------------------
# An atom is a hashable tuple:
#   ("col", src, j)            column j of a raw sim answer; src = ("link", actor, link) | ("pos", a) | ("vel", a) | ("ori", a) |
#                              ("cf", actor, link) | ("dof",)
#   ("norm", vec) ("abs", lin) ("clamp0", lin) ("mul", lin, lin) ("div", lin, lin) ("yaw", actor) ("euler", src, i)
# where `lin` is the canonical key of a Lin and `vec` a tuple of such keys.
class Lin(object):
    """c0 + sum_i c_i * atom_i: one value per env"""
    __slots__ = ("t", "c")

    def __init__(self, terms: Dict[tuple, float] = None, const: float = 0.0):
        self.t = {k: v for k, v in (terms or {}).items() if v != 0.0}
        self.c = float(const)

    def key(self):
        return (tuple(sorted(self.t.items(), key=repr)), self.c)

    def scaled(self, s: float):
        return Lin({k: v * s for k, v in self.t.items()}, self.c * s)

    def plus(self, o: "Lin", sign: float = 1.0):
        t = dict(self.t)
        for k, v in o.t.items():
            t[k] = t.get(k, 0.0) + sign * v
        return Lin(t, self.c + sign * o.c)

    def single(self):
        """(atom, coefficient) when the form is exactly one atom without a constant, else None"""
        if self.c == 0.0 and len(self.t) == 1:
            return next(iter(self.t.items()))
        return None


def _atom(a: tuple) -> Lin:
    return Lin({a: 1.0})


def _num(x):
    """python / numpy / 0-d tensor number -> float, else None"""
    if isinstance(x, bool):
        return None
    if isinstance(x, (int, float)):
        return float(x)
    if isinstance(x, torch.Tensor) and x.dim() == 0:
        return float(x)
    try:
        import numpy as np
        if isinstance(x, np.generic):
            return float(x)
    except Exception:
        pass
    return None


def _const_row(x, n: int):
    """a constant the Objective adds to an n-column proxy: list / tuple / tensor of n numbers (or [1, n]) -> list of floats"""
    if isinstance(x, torch.Tensor):
        x = x.detach().cpu().reshape(-1).tolist()
    elif isinstance(x, (list, tuple)):
        x = [float(v) for v in x]
    else:
        try:
            import numpy as np
            if isinstance(x, np.ndarray):
                x = [float(v) for v in x.reshape(-1)]
            else:
                return None
        except Exception:
            return None
    return x if len(x) == n else None


class Sym(object):
    """stand-in for a [K, n] (cols is a list of Lin) or [K] (cols is one Lin, scalar = True) tensor of the rollout envs"""
    __array_priority__ = 1000
    _mppi_sym = True   # (what mppiisaac.utils.conversions looks for)

    def __init__(self, cols, scalar=False, rot_of=None):
        self.cols: List[Lin] = cols
        self.scalar = scalar
        self.rot_of = rot_of   # quaternion_to_matrix(link quaternion): the link's source, cols unused

    # -- shape protocol the Objectives may touch
    @property
    def shape(self):
        return (Sym.K,) if self.scalar else (Sym.K, len(self.cols))
    K = 1

    def dim(self):
        return 1 if self.scalar else 2

    # -- indexing
    def __getitem__(self, idx):
        if self.rot_of is not None:
            raise TraceError("indexing into a rotation matrix")
        if self.scalar:
            raise TraceError("indexing a per-env scalar")
        if not (isinstance(idx, tuple) and len(idx) == 2 and idx[0] == slice(None)):
            raise TraceError(f"indexing other than x[:, columns] ({idx!r})")
        c = idx[1]
        if isinstance(c, slice):
            return Sym(self.cols[c])
        if isinstance(c, int):
            if not -len(self.cols) <= c < len(self.cols):
                raise TraceError("column index out of range")
            return Sym([self.cols[c]], scalar=True)
        if isinstance(c, (list, tuple)) and all(isinstance(j, int) for j in c):
            return Sym([self.cols[j] for j in c])
        raise TraceError(f"column selector {c!r}")

    # -- arithmetic
    def _zip(self, o, what):
        """the other operand as a list of Lin matching self's columns (numbers and constants broadcast)"""
        n = len(self.cols)
        if isinstance(o, Sym):
            if o.rot_of is not None or self.rot_of is not None:
                raise TraceError(f"{what} with a rotation matrix")
            if len(o.cols) == n and o.scalar == self.scalar:
                return o.cols
            if o.scalar and not self.scalar:
                raise TraceError(f"{what} of a [K, n] value with a [K] value (broadcast over columns)")
            if len(o.cols) == 1 and not o.scalar:
                return o.cols * n
            raise TraceError(f"{what} of values with {n} and {len(o.cols)} columns")
        v = _num(o)
        if v is not None:
            return [Lin(const=v)] * n
        row = _const_row(o, n)
        if row is not None:
            return [Lin(const=v) for v in row]
        raise TraceError(f"{what} with a {type(o).__name__} the tracer cannot read as a constant of {n} columns")

    def _wrap(self, cols, like=None):
        return Sym(cols, scalar=self.scalar if like is None else like)

    def __add__(self, o):
        return self._wrap([a.plus(b) for a, b in zip(self.cols, self._zip(o, "addition"))])
    __radd__ = __add__

    def __sub__(self, o):
        return self._wrap([a.plus(b, -1.0) for a, b in zip(self.cols, self._zip(o, "subtraction"))])

    def __rsub__(self, o):
        return self._wrap([b.plus(a, -1.0) for a, b in zip(self.cols, self._zip(o, "subtraction"))])

    def __neg__(self):
        return self._wrap([a.scaled(-1.0) for a in self.cols])

    def __mul__(self, o):
        v = _num(o)
        if v is not None:
            return self._wrap([a.scaled(v) for a in self.cols])
        out = []
        for a, b in zip(self.cols, self._zip(o, "multiplication")):
            if not b.t:
                out.append(a.scaled(b.c))
            elif not a.t:
                out.append(b.scaled(a.c))
            else:
                ka, kb = sorted((a.key(), b.key()), key=repr)
                out.append(_atom(("mul", ka, kb)))
        return self._wrap(out)
    __rmul__ = __mul__

    def __truediv__(self, o):
        v = _num(o)
        if v is not None:
            if v == 0.0:
                raise TraceError("division by zero")
            return self._wrap([a.scaled(1.0 / v) for a in self.cols])
        out = []
        for a, b in zip(self.cols, self._zip(o, "division")):
            if not b.t:
                out.append(a.scaled(1.0 / b.c))
            else:
                out.append(_atom(("div", a.key(), b.key())))
        return self._wrap(out)

    def __rtruediv__(self, o):
        raise TraceError("a constant divided by a per-env value")

    def __pow__(self, p):
        if _num(p) == 2.0:
            return self * self
        raise TraceError(f"power {p!r} (only ** 2)")

    def __abs__(self):
        return _abs(self)

    # tensor methods the planners use
    def abs(self):
        return _abs(self)

    def square(self):
        return self * self

    def sum(self, dim=None, axis=None, **kw):
        return _sum(self, dim if dim is not None else axis)

    def norm(self, p=2, dim=None, **kw):
        return _norm(self, dim)

    def clamp(self, min=None, max=None):
        return _clamp(self, min, max)

    def __bool__(self):
        raise TraceError("a branch on a per-env value (data-dependent control flow)")

    def __iter__(self):
        raise TraceError("iteration over a per-env value")

    def __getattr__(self, name):
        raise TraceError(f"tensor attribute / method '.{name}' on a per-env value")

    # -- torch.* functions called on a Sym
    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        name = getattr(func, "__name__", str(func))
        h = _TORCH.get(func)
        if h is None:
            raise TraceError(f"torch function '{name}' on a per-env value")
        return h(*args, **kwargs)


def _axis(kw_dim, kw_axis, pos):
    d = kw_dim if kw_dim is not None else (kw_axis if kw_axis is not None else pos)
    if d not in (1, -1):
        raise TraceError(f"reduction over dimension {d!r} (only the columns, dim = 1)")


def _need_vec(x, what):
    if not isinstance(x, Sym) or x.scalar or x.rot_of is not None:
        raise TraceError(f"{what} of something that is not a [K, n] per-env value")


def _norm(x, dim=None, axis=None, ord=None, p=None, **kw):
    _need_vec(x, "norm")
    if ord not in (None, 2) or p not in (None, 2, "fro"):
        raise TraceError("a norm other than the Euclidean one")
    _axis(dim, axis, None)
    return Sym([_atom(("norm", tuple(c.key() for c in x.cols)))], scalar=True)


def _sum(x, dim=None, axis=None, **kw):
    _need_vec(x, "sum")
    _axis(dim, axis, None)
    tot = Lin()
    for c in x.cols:
        tot = tot.plus(c)
    return Sym([tot], scalar=True)


def _abs(x):
    if not isinstance(x, Sym) or x.rot_of is not None:
        raise TraceError("abs of something that is not a per-env value")
    return Sym([_atom(("abs", c.key())) if c.t else Lin(const=abs(c.c)) for c in x.cols], scalar=x.scalar)


def _clamp(x, min=None, max=None):
    if not isinstance(x, Sym) or x.rot_of is not None or max is not None or _num(min) != 0.0:
        raise TraceError("clamp other than clamp(x, min=0)")
    return Sym([_atom(("clamp0", c.key())) for c in x.cols], scalar=x.scalar)


_TORCH = {
    torch.linalg.norm: lambda x, ord=None, dim=None, axis=None, **kw: _norm(x, dim=dim, axis=axis, ord=ord),
    torch.norm: lambda x, p=2, dim=None, **kw: _norm(x, dim=dim, p=p),
    torch.sum: lambda x, dim=None, axis=None, **kw: _sum(x, dim=dim, axis=axis),
    torch.abs: _abs,
    torch.square: lambda x: x * x,
    torch.clamp: lambda x, min=None, max=None: _clamp(x, min, max),
    torch.add: lambda a, b: a + b,
    torch.sub: lambda a, b: a - b,
    torch.mul: lambda a, b: a * b,
    torch.div: lambda a, b: a / b,
    torch.neg: lambda a: -a,
}


# ---- the rotation helpers of mppiisaac.utils.conversions, on proxies ------------------------------------------------------------
def _quat_source(q: Sym, what: str):
    """the link / actor whose quaternion (4 columns, xyzw, untouched) `q` is"""
    if not isinstance(q, Sym) or q.scalar or len(q.cols) != 4:
        raise TraceError(f"{what} of something that is not a [K, 4] quaternion")
    src = None
    for j, c in enumerate(q.cols):
        s = c.single()
        if s is None or s[1] != 1.0 or s[0][0] != "col":
            raise TraceError(f"{what} of a quaternion that was modified")
        _, sj, col = s[0]
        first = 3 if sj[0] == "link" else 0
        if sj[0] not in ("link", "ori") or col != first + j or (src is not None and sj != src):
            raise TraceError(f"{what} of columns that are not one body's quaternion")
        src = sj
    return src


def quaternion_to_yaw(q: Sym):
    src = _quat_source(q, "quaternion_to_yaw")
    return Sym([_atom(("yaw", src))], scalar=True)


def quaternion_to_matrix(q: Sym):
    return Sym([], rot_of=_quat_source(q, "quaternion_to_matrix"))


def matrix_to_euler_angles(R: Sym, convention: str):
    if not isinstance(R, Sym) or R.rot_of is None:
        raise TraceError("matrix_to_euler_angles of something that is not quaternion_to_matrix(...)")
    if convention != "ZYX":
        raise TraceError(f"Euler convention {convention!r} (only 'ZYX')")
    return Sym([_atom(("euler", R.rot_of, i)) for i in range(3)])


# ---- the symbolic sim ------------------------------------------------------------------------------------------------------------
class TraceSim(object):
    """what `compute_cost` sees while it is traced: the getters of the reference's IsaacGymWrapper (isaacgym_wrapper.py:299-356)
    returning proxies; anything else of the sim is refused"""

    def __init__(self, sim):
        self._sim = sim
        self.num_envs = sim.num_envs
        self.device = sim.device
        self.env_cfg = sim.env_cfg
        Sym.K = sim.num_envs

    def _rows(self, src, n):
        return Sym([_atom(("col", src, j)) for j in range(n)])

    def get_actor_link_by_name(self, actor_name, link_name):
        self._sim.scene.rigid_body_index(actor_name, link_name)   # (unknown names fail here as they do on the real sim)
        return self._rows(("link", actor_name, link_name), 13)

    def get_actor_position_by_name(self, name):
        self._sim.scene.actor_index(name)
        return self._rows(("pos", name), 3)

    def get_actor_velocity_by_name(self, name):
        self._sim.scene.actor_index(name)
        return self._rows(("vel", name), 3)

    def get_actor_orientation_by_name(self, name):
        self._sim.scene.actor_index(name)
        return self._rows(("ori", name), 4)

    def get_actor_contact_forces_by_name(self, actor_name, link_name):
        self._sim.scene.rigid_body_index(actor_name, link_name)
        return self._rows(("cf", actor_name, link_name), 3)

    def get_dof_state(self):
        return self._rows(("dof",), 2 * self._sim.scene.n_dof)

    def __getattr__(self, name):
        raise TraceError(f"sim.{name} (only the by-name getters and get_dof_state are traceable)")


# ---- matching the traced cost to the measurements of mppiisaac.objectives ------------------------------------------------------
def _point(src, cols):
    """operand of a Term for position columns `cols` (0, 1[, 2]) of `src`, or None"""
    if list(cols) != list(range(len(cols))):
        return None
    if src[0] == "link":
        return ("link", src[1], src[2])
    if src[0] == "pos":
        return ("actor", src[1])
    return None


def _col(lin_key):
    """(src, col, coef) when the form is coef * one raw column, else None"""
    terms, c = lin_key
    if c == 0.0 and len(terms) == 1 and terms[0][0][0] == "col":
        return terms[0][0][1], terms[0][0][2], terms[0][1]
    return None


def _diff(vec):
    """a tuple of Lin keys as the difference of two points: -> (a, b, n) operands with vec = +-(a - b), or None.  b may be a constant."""
    n = len(vec)
    if n not in (2, 3):
        return None
    A, B, sgn = [], [], None
    for terms, c in vec:
        cols = [(k[1], k[2], v) for k, v in terms if k[0] == "col"]
        if len(cols) != len(terms):
            return None
        if len(cols) == 2 and c == 0.0:
            (s0, j0, v0), (s1, j1, v1) = cols
            if {v0, v1} != {1.0, -1.0}:
                return None
            pos, neg = ((s0, j0), (s1, j1)) if v0 > 0 else ((s1, j1), (s0, j0))
            A.append(pos); B.append(neg)
        elif len(cols) == 1 and abs(cols[0][2]) == 1.0:
            s = cols[0][2]
            if sgn is not None and s != sgn:
                return None
            sgn = s
            A.append((cols[0][0], cols[0][1])); B.append(("const", -c * s))
        else:
            return None

    def operand(P):
        if all(p[0] == "const" for p in P):
            return tuple(p[1] for p in P) + (0.0,) * (3 - len(P))
        if any(p[0] == "const" for p in P) or len({p[0] for p in P}) != 1:
            return None
        src = P[0][0]
        if src == ("dof",) and [p[1] for p in P] == [0, 2]:
            return ("dof_xy",)
        return _point(src, [p[1] for p in P])
    a, b = operand(A), operand(B)
    if a is None or b is None:
        return None
    if isinstance(a, tuple) and a and isinstance(a[0], float):   # (constant first: swap - the distance is symmetric)
        a, b = b, a
    return a, b, n


def match(cost: Sym, n_dof: int) -> List[Term]:
    """the traced per-env cost -> Term list (fixed weights), or TraceError naming what has no counterpart"""
    if not isinstance(cost, Sym):
        raise TraceError(f"compute_cost returned a {type(cost).__name__}, not a value computed from the sim")
    if not cost.scalar:
        if len(cost.cols) != 1:
            raise TraceError("compute_cost returned a [K, n] value, not one cost per env")
    lin = cost.cols[0]
    terms: List[Term] = []
    const_expected = 0.0
    cf_abs: Dict[tuple, Dict[int, float]] = {}
    dof_sq: Dict[tuple, Dict[int, Tuple[float, float]]] = {}
    for atom, w in lin.t.items():
        kind = atom[0]
        if kind == "norm":
            vec = atom[1]
            eul = [k for k in vec if len(k[0]) == 1 and k[1] == 0.0 and k[0][0][0][0] == "euler" and k[0][0][1] == 1.0]
            if len(eul) == len(vec) == 2 and [k[0][0][0][2] for k in vec] == [0, 1] and vec[0][0][0][0][1] == vec[1][0][0][0][1]:
                src = vec[0][0][0][0][1]
                if src[0] != "link":
                    raise TraceError("the tilt of an actor's root orientation (only links)")
                terms.append(Term(w, "tilt", (("link", src[1], src[2]),)))
                continue
            vel = [_col(k) for k in vec]
            if all(v is not None and v[0][0] == "vel" and v[2] == 1.0 for v in vel) and [v[1] for v in vel] == list(range(len(vel))) and len({v[0] for v in vel}) == 1:
                terms.append(Term(w, "speed", (("actor", vel[0][0][1]), len(vel))))
                continue
            d = _diff(vec)
            if d is None:
                raise TraceError("the norm of something other than a point difference, a velocity or two Euler angles")
            terms.append(Term(w, "dist", d))
        elif kind == "abs":
            inner_terms, c = atom[1]
            one = _col(atom[1])
            if one is not None and one[0][0] == "cf":
                cf_abs.setdefault((one[0][1], one[0][2], w), {})[one[1]] = abs(one[2])
                continue
            if len(inner_terms) == 1 and inner_terms[0][0][0] == "yaw" and abs(inner_terms[0][1]) == 1.0:
                src = inner_terms[0][0][1]
                if src[0] != "ori":
                    raise TraceError("the yaw of a link (only actors' root orientations)")
                terms.append(Term(w, "yaw_abs", (("actor", src[1]), -c * inner_terms[0][1])))
                continue
            zs = [(k[1], k[2], v) for k, v in inner_terms if k[0] == "col"]
            if len(zs) == len(inner_terms) and all(j == 2 and s[0] in ("link", "pos") for s, j, v in zs):
                if len(zs) == 1 and abs(zs[0][2]) == 1.0:
                    terms.append(Term(w, "abs_dz", (_point(zs[0][0], [0, 1, 2]), -c * zs[0][2])))
                    continue
                if len(zs) == 2 and c == 0.0 and {zs[0][2], zs[1][2]} == {1.0, -1.0}:
                    terms.append(Term(w, "abs_dz", (_point(zs[0][0], [0, 1, 2]), _point(zs[1][0], [0, 1, 2]))))
                    continue
            raise TraceError("abs() of something other than a contact-force component, a yaw difference or a height difference")
        elif kind == "clamp0":
            inner_terms, c = atom[1]
            one = _col((inner_terms, 0.0))
            if one is None or one[1] != 2 or one[2] != -1.0 or one[0][0] not in ("link", "pos"):
                raise TraceError("clamp(min=0) of something other than (height - z of a link or actor)")
            terms.append(Term(w, "below", (_point(one[0], [0, 1, 2]), c)))
        elif kind == "mul":
            if atom[1] != atom[2]:
                raise TraceError("a product of two different per-env values outside a push-align cosine")
            inner_terms, c = atom[1]
            one = _col((inner_terms, 0.0))
            if one is None or one[0] != ("dof",) or one[2] != 1.0:
                raise TraceError("a square of something other than (DOF value - reference)")
            which, idx = ("pos" if one[1] % 2 == 0 else "vel"), one[1] // 2
            dof_sq.setdefault((which, w), {})[idx] = -c
        elif kind == "div":
            # (a - b) . (c - b) / (|a - b| |c - b|), planar: the push-align cosine; its "+ 1" arrives as a constant w
            num_terms, nc = atom[1]
            den = atom[2]
            ok = nc == 0.0 and len(num_terms) == 2 and all(k[0] == "mul" and v == 1.0 for k, v in num_terms)
            dv = den[0][0][0] if (den[1] == 0.0 and len(den[0]) == 1 and den[0][0][1] == 1.0) else None
            if not ok or dv is None or dv[0] != "mul":
                raise TraceError("a quotient other than the push-align cosine dot(a - b, c - b) / (|a - b| |c - b|)")
            norms = []
            for k in (dv[1], dv[2]):
                if not (k[1] == 0.0 and len(k[0]) == 1 and k[0][0][1] == 1.0 and k[0][0][0][0] == "norm"):
                    raise TraceError("the denominator of a quotient is not a product of two norms")
                norms.append(k[0][0][0][1])
            # the two factors of the dot product, component by component
            fx = [(num_terms[i][0][1], num_terms[i][0][2]) for i in range(2)]
            found = None
            for V1, V2 in ((norms[0], norms[1]), (norms[1], norms[0])):
                if len(V1) != 2 or len(V2) != 2:
                    continue
                want = {tuple(sorted((V1[i], V2[i]), key=repr)) for i in range(2)}
                if want == {tuple(sorted(f, key=repr)) for f in fx}:
                    found = (V1, V2)
            if found is None:
                raise TraceError("the numerator of a quotient is not the dot product of the two normed vectors")
            d1, d2 = _diff(found[0]), _diff(found[1])
            if d1 is None or d2 is None:
                raise TraceError("the vectors of a push-align cosine are not point differences")
            # vectors from a common point b: d1 = +-(a - b), d2 = +-(c - b); the signs must agree for the cosine to be what align computes
            common = [p for p in d1[:2] if p in d2[:2]]
            if len(common) != 1:
                raise TraceError("the vectors of a push-align cosine do not share a point")
            b = common[0]
            a = d1[0] if d1[1] == b else d1[1]
            c3 = d2[0] if d2[1] == b else d2[1]
            if _orientation(found[0], b) * _orientation(found[1], b) < 0:
                raise TraceError("a cosine between rays of opposite sense (a - b against b - c): not what `align` computes")
            terms.append(Term(w, "align", (a, b, c3)))
            const_expected += w
        elif kind == "col":
            raise TraceError("a raw state component in the cost without a norm / abs / square around it")
        else:
            raise TraceError(f"the measurement '{kind}' has no counterpart in the cost programs")
    for (actor_name, link_name, w), comps in cf_abs.items():
        n = len(comps)
        if sorted(comps) != list(range(n)) or any(v != 1.0 for v in comps.values()):
            raise TraceError("contact-force components that are not the first n of a body, each once")
        terms.append(Term(w, "force_l1", (actor_name, link_name, n)))
    for (which, w), comps in dof_sq.items():
        idx = sorted(comps)
        if idx != list(range(idx[0], idx[-1] + 1)):
            raise TraceError("squares of DOF values that are not a contiguous range")
        ref = [comps[i] for i in idx]
        lo, hi = idx[0], idx[-1] + 1
        if len(ref) > 8 and any(r != 0.0 for r in ref):
            raise TraceError("a DOF reference of more than eight values")
        terms.append(Term(w, "dof_sq", (which, lo, hi if hi < n_dof else 0, ref if any(r != 0.0 for r in ref) else None)))
    if not math.isclose(lin.c, const_expected, rel_tol=1e-9, abs_tol=1e-12):
        raise TraceError(f"a constant offset of {lin.c - const_expected:g} in the cost")
    if not terms:
        raise TraceError("the cost does not depend on the sim")
    return terms


def _orientation(vec, b):
    """+1 when the planar vector `vec` (tuple of Lin keys) is (x - b), -1 when it is (b - x)"""
    terms, c = vec[0]
    for k, v in terms:
        src, j = k[1], k[2]
        p = _point(src, [0, 1, 2]) if src[0] in ("link", "pos") else (("dof_xy",) if src == ("dof",) else None)
        if p == b:
            return -1.0 if v > 0 else 1.0
    return -1.0 if c > 0 else 1.0   # (b is a constant point: x - b carries -b)


def trace_objective(objective, sim) -> List[Term]:
    """run `objective.compute_cost` once on the symbolic sim -> Term list with the weights the Objective applied (numbers)"""
    tsim = TraceSim(sim)
    try:
        return match(objective.compute_cost(tsim), sim.scene.n_dof)
    except TraceError:
        raise
    except Exception as e:   # whatever the Objective's own code raises on proxies
        raise TraceError(f"{type(e).__name__}: {e}") from e

"""Stage-cost Objectives as small declarative cost programs.

The reference ships one hand-written `Objective.compute_cost(sim)` per example
(reference examples/*/planner.py, benchmarks/point_robot/mppi_planner/mppi_planner_wrapper.py).  All of
them are weighted sums of a handful of geometric measurements on what the simulator reports, so here an
Objective is DATA - a list of `Term(weight, op, operands)` - and there is exactly one evaluator:

  * `evaluate(terms, weights, sim)` runs the program with torch through the reference's sim getter protocol
    (`get_actor_link_by_name`, `get_actor_position_by_name`, ...): the generic Objective mode, and what the
    golden fixtures generated from the reference's own planners are replayed through (tests/golden/);
  * `fused_spec(sim)` hands the same program to the rollout kernel as an `mppi_cost_t` when it is one of the
    shapes the kernel evaluates in-line (MPPI_COST_*), so the cost never leaves the GPU registers.

The Objective contract of the reference is kept: `compute_cost(sim) -> [K]`, `reset()`, mutable `.weights`.
`graph_safe = True`: compute_cost is a pure tensor program of sim tensors and `.weights`, so the planner may
capture the generic horizon into a HIP graph (planner/mppi.py:_replay_horizon).

Operands (where a 3-vector comes from):  ("link", actor, link) rigid-body position; ("actor", name) root position;
("dof_xy",) the first two DOF positions; a plain (x, y, z) tuple is a constant.

Ops (each yields one value per env):
  dist(a, b, n)            Euclidean distance of the first n components of a and b
  tilt(link)               size of the first two "ZYX" Euler angles of the link quaternion, fed xyzw into a real-first
                           conversion exactly as the reference's planners do (examples/panda/planner.py:30-32)
  yaw_abs(actor, ref)      |yaw of the actor's root quaternion - ref|
  align(a, b, c)           1 + cos of the planar angle at b between the rays to a and to c (0 when a pushes b towards c)
  force_l1(actor, link, n) sum |F| over the first n components of the body's net contact force
  speed(actor, n)          norm of the first n components of the actor's root linear velocity
  dof_sq(which, lo, hi, ref)  sum of squares of DOF positions/velocities lo..hi-1 (minus ref)
  abs_dz(a, b)             |a.z - b.z|  (b may be a float height)
  below(a, h)              max(h - a.z, 0)
"""
from typing import NamedTuple, Sequence, Tuple, Union

import torch

from mppiisaac.backend import capi
from mppiisaac.utils.conversions import matrix_to_euler_angles, quaternion_to_matrix, quaternion_to_yaw


class Term(NamedTuple):
    weight: Union[str, float]   # key into Objective.weights, or a fixed factor
    op: str
    args: Tuple


def link(actor: str, name: str):
    return ("link", actor, name)


def actor(name: str):
    return ("actor", name)


def _point(sim, src, like=None):
    """operand -> [K, 3] positions (a constant operand: [1, 3] in the dtype / on the device of `like`)"""
    kind = src[0]
    if kind == "link":
        return sim.get_actor_link_by_name(src[1], src[2])[:, 0:3]
    if kind == "actor":
        return sim.get_actor_position_by_name(src[1])[:, 0:3]
    if kind == "dof_xy":
        dof = sim.get_dof_state()
        return torch.stack((dof[:, 0], dof[:, 2], torch.zeros_like(dof[:, 0])), dim=1)
    if like.dtype == torch.float32:
        return _const_tensor(tuple(float(v) for v in src), like.device).view(1, 3)
    return torch.as_tensor([float(v) for v in src], dtype=like.dtype, device=like.device).view(1, 3)


_CONSTS = {}


def _const_tensor(values: tuple, device) -> torch.Tensor:
    """fp32 constant on `device`, created once (no host-to-device copy inside compute_cost: the horizon stays graph-capturable)"""
    key = (values, str(device))
    if key not in _CONSTS:
        _CONSTS[key] = torch.tensor(values, dtype=torch.float32, device=device)
    return _CONSTS[key]


def _norm(v):
    return torch.sqrt(torch.sum(v * v, dim=1))


def _measure(sim, op: str, a: Tuple):
    if op == "dist":
        n = a[2]
        first = _point(sim, a[0])  # (by convention the first operand is never a constant)
        return _norm(first[:, :n] - _point(sim, a[1], like=first)[:, :n])
    if op == "tilt":
        quat = sim.get_actor_link_by_name(a[0][1], a[0][2])[:, 3:7]
        return _norm(matrix_to_euler_angles(quaternion_to_matrix(quat), "ZYX")[:, 0:2])
    if op == "yaw_abs":
        return torch.abs(quaternion_to_yaw(sim.get_actor_orientation_by_name(a[0][1])) - a[1])
    if op == "align":
        to_a = (_point(sim, a[0]) - _point(sim, a[1]))[:, :2]
        to_c = (_point(sim, a[2]) - _point(sim, a[1]))[:, :2]
        return torch.sum(to_a * to_c, dim=1) / (_norm(to_a) * _norm(to_c)) + 1
    if op == "force_l1":
        return torch.sum(torch.abs(sim.get_actor_contact_forces_by_name(a[0], a[1])[:, : a[2]]), dim=1)
    if op == "speed":
        return _norm(sim.get_actor_velocity_by_name(a[0][1])[:, : a[1]])
    if op == "dof_sq":
        which, lo, hi, ref = a
        x = sim.get_dof_state()[:, (0 if which == "pos" else 1)::2]
        hi = x.shape[1] + hi if hi <= 0 else hi
        lo = x.shape[1] + lo if lo < 0 else lo
        x = x[:, lo:hi]
        if ref is not None:
            x = x - _const_tensor(tuple(float(v) for v in ref), x.device)  # (fp32 constants, as the reference's planners hold them)
        return torch.sum(x * x, dim=1)
    if op == "abs_dz":
        z = _point(sim, a[0])[:, 2]
        return torch.abs(z - (a[1] if isinstance(a[1], (int, float)) else _point(sim, a[1])[:, 2]))
    if op == "below":
        return torch.clamp(a[1] - _point(sim, a[0])[:, 2], min=0)
    raise ValueError(f"unknown cost op '{op}'")


def evaluate(terms: Sequence[Term], weights: dict, sim) -> torch.Tensor:
    """sum_i weight_i * measure_i(sim) -> [K]"""
    total = None
    for t in terms:
        w = weights[t.weight] if isinstance(t.weight, str) else t.weight
        v = w * _measure(sim, t.op, t.args)
        total = v if total is None else total + v
    return total


_OPS = {"dist": capi.OP_DIST, "tilt": capi.OP_TILT, "yaw_abs": capi.OP_YAW_ABS, "align": capi.OP_ALIGN, "force_l1": capi.OP_FORCE_L1,
        "speed": capi.OP_SPEED, "dof_sq": capi.OP_DOF_SQ, "abs_dz": capi.OP_ABS_DZ, "below": capi.OP_BELOW}


def compile_program(terms: Sequence[Term], weights: dict, scene) -> capi.Cost:
    """Term list -> mppi_cost_t of kind MPPI_COST_PROGRAM for `scene` (Scene: name -> index lookups)."""
    if len(terms) > capi.MAX_TERMS:
        raise ValueError(f"{len(terms)} terms exceed MPPI_MAX_TERMS = {capi.MAX_TERMS}")
    c = capi.Cost()
    c.kind, c.n_terms = capi.COST_PROGRAM, len(terms)

    def operand(t, slot, src):
        if src[0] == "link":
            t.src[slot], t.idx[slot] = capi.SRC_RB, scene.rigid_body_index(src[1], src[2])
        elif src[0] == "actor":
            t.src[slot], t.idx[slot] = capi.SRC_ACTOR, scene.actor_index(src[1])
        elif src[0] == "dof_xy":
            t.src[slot] = capi.SRC_DOF_XY
        else:
            if slot != 1:
                raise ValueError("a constant point can only be the second operand of a term")
            t.src[slot] = capi.SRC_CONST
            for j in range(3):
                t.p[j] = float(src[j])

    for i, term in enumerate(terms):
        t, a = c.terms[i], term.args
        t.op = _OPS[term.op]
        t.w = float(weights[term.weight] if isinstance(term.weight, str) else term.weight)
        if term.op == "dist":
            operand(t, 0, a[0]); operand(t, 1, a[1]); t.n = int(a[2])
        elif term.op == "tilt":
            operand(t, 0, a[0])
            if t.src[0] != capi.SRC_RB:
                raise ValueError("tilt needs a link operand")
        elif term.op == "yaw_abs":
            operand(t, 0, a[0]); t.p[3] = float(a[1])
        elif term.op == "align":
            operand(t, 0, a[0]); operand(t, 1, a[1]); operand(t, 2, a[2])
        elif term.op == "force_l1":
            t.src[0], t.idx[0], t.n = capi.SRC_RB, scene.rigid_body_index(a[0], a[1]), int(a[2])
        elif term.op == "speed":
            operand(t, 0, a[0]); t.n = int(a[1])
            if t.src[0] != capi.SRC_ACTOR:
                raise ValueError("speed needs an actor operand")
        elif term.op == "dof_sq":
            which, lo, hi, ref = a
            n = scene.n_dof
            t.n = 0 if which == "pos" else 1
            t.idx[0], t.idx[1] = (n + lo if lo < 0 else lo), (n + hi if hi <= 0 else hi)
            ref = list(ref or [])
            if len(ref) > 8 or (ref and len(ref) != t.idx[1] - t.idx[0]):
                raise ValueError("dof_sq: the reference needs one value per DOF of the range (at most 8)")
            t.idx[2] = len(ref)
            for j, v in enumerate(ref):
                t.p[j] = float(v)
        elif term.op == "abs_dz":
            operand(t, 0, a[0])
            operand(t, 1, (0.0, 0.0, float(a[1])) if isinstance(a[1], (int, float)) else a[1])
        elif term.op == "below":
            operand(t, 0, a[0]); t.p[3] = float(a[1])
    return c


class ProgramObjective(object):
    """Objective contract of the reference (compute_cost / reset / weights) over a term list."""
    graph_safe = True
    WEIGHTS: dict = {}

    def __init__(self, cfg=None):
        self.weights = dict(self.WEIGHTS)
        self.reset()

    def reset(self):
        pass

    def terms(self) -> Sequence[Term]:
        raise NotImplementedError

    def compute_cost(self, sim):
        return evaluate(self.terms(), self.weights, sim)

    def program_spec(self, sim) -> capi.Cost:
        """the term list as an MPPI_COST_PROGRAM (include/mppi_hip.h): names resolved to rigid-body / actor indices of the
        sim's scene, weights taken from `.weights` - evaluated inside the rollout kernel like the in-line kinds"""
        return compile_program(self.terms(), self.weights, sim.scene)

    def fused_spec(self, sim) -> capi.Cost:
        return self.program_spec(sim)

    # shared by the fused specs: weight table -> mppi_cost_t.w in the order the kernel reads it
    def _spec(self, kind: int, order: Sequence[str]) -> capi.Cost:
        c = capi.Cost()
        c.kind = kind
        for i, k in enumerate(order):
            c.w[i] = float(self.weights[k])
        return c


class ReachTiltObjective(ProgramObjective):
    """end-effector to goal + keep the end effector level: the reach objective of the arm examples (reference
    examples/panda/planner.py:10-40, examples/panda_effort/planner.py, examples/albert/planner.py differ in the actor,
    the link and the weights only)."""
    WEIGHTS = {"robot_to_goal": 1.0, "robot_ori": 0.5}

    def __init__(self, cfg=None, actor="panda", link="panda_ee_tip", goal="goal"):
        self.actor, self.link, self.goal = actor, link, goal
        super().__init__(cfg)

    def terms(self):
        ee = link(self.actor, self.link)
        return [Term("robot_to_goal", "dist", (ee, actor(self.goal), 3)), Term("robot_ori", "tilt", (ee,))]

    def fused_spec(self, sim) -> capi.Cost:
        c = self._spec(capi.COST_PANDA_REACH, ("robot_to_goal", "robot_ori"))
        c.link[0] = sim.scene.rigid_body_index(self.actor, self.link)
        c.actor[0] = sim.scene.actor_index(self.goal)
        return c


class PandaReachObjective(ReachTiltObjective):
    """reference examples/panda/planner.py (panda_stick arm, link panda_ee_tip)"""


class PandaEffortReachObjective(ReachTiltObjective):
    """reference examples/panda_effort/planner.py (effort-driven panda, link panda_link7)"""

    def __init__(self, cfg=None, actor="panda", link="panda_link7", goal="goal"):
        super().__init__(cfg, actor, link, goal)


class AlbertReachObjective(ReachTiltObjective):
    """reference examples/albert/planner.py (diff-drive base + arm, link mmrobot_link7)"""
    WEIGHTS = {"robot_to_goal": 4.0, "robot_ori": 0.5}

    def __init__(self, cfg=None, actor="albert", link="mmrobot_link7", goal="goal"):
        super().__init__(cfg, actor, link, goal)


class PointReachObjective(ProgramObjective):
    """navigation term of reference benchmarks/point_robot/mppi_planner/mppi_planner_wrapper.py:10-35: planar distance
    of the base (DOF positions 0 and 1) to the goal; `goal` is an actor name or a fixed [x, y]."""

    def __init__(self, cfg=None, goal="goal", w_nav=2.0):
        self.goal = goal
        super().__init__(cfg)
        self.weights = {"w_nav": w_nav}

    def terms(self):
        target = actor(self.goal) if isinstance(self.goal, str) else (float(self.goal[0]), float(self.goal[1]), 0.0)
        return [Term("w_nav", "dist", (("dof_xy",), target, 2))]

    def fused_spec(self, sim) -> capi.Cost:
        c = self._spec(capi.COST_POINT_REACH, ("w_nav",))
        if isinstance(self.goal, str):
            c.actor[0] = sim.scene.actor_index(self.goal)
        else:
            c.actor[0] = -1
            c.w[1], c.w[2] = float(self.goal[0]), float(self.goal[1])
        return c


class PlanarPushObjective(ProgramObjective):
    """non-prehensile pushing with a mobile base: approach the block, bring it to the goal position and yaw, stay behind
    it, keep off the obstacles (reference examples/boxer_push/planner.py:9-67, examples/heijn_push/planner.py)."""
    WEIGHTS = {"robot_to_block": 0.1, "block_to_goal": 2.0, "block_to_goal_ort": 3.0, "push_align": 0.6,
               "collision": 100, "velocity": 0.0}

    def __init__(self, cfg=None, robot="boxer", link="ee_link", block="block", goal="goal",
                 obstacles=("paper_obst1", "paper_obst2")):
        self.goal_yaw = 0.0
        self.robot, self.link, self.block, self.goal, self.obstacles = robot, link, block, goal, tuple(obstacles)
        super().__init__(cfg)

    def terms(self):
        pusher, blk, tgt = link(self.robot, self.link), actor(self.block), actor(self.goal)
        out = [Term("robot_to_block", "dist", (pusher, blk, 2)), Term("block_to_goal", "dist", (tgt, blk, 2)),
               Term("block_to_goal_ort", "yaw_abs", (blk, self.goal_yaw)), Term("push_align", "align", (pusher, blk, tgt)),
               Term("velocity", "speed", (blk, 2))]
        return out + [Term("collision", "force_l1", (o, "box", 2)) for o in self.obstacles]

    def fused_spec(self, sim) -> capi.Cost:
        c = self._spec(capi.COST_BOXER_PUSH, ("robot_to_block", "block_to_goal", "block_to_goal_ort", "push_align", "velocity", "collision"))
        c.link[0] = sim.scene.rigid_body_index(self.robot, self.link)
        c.link[1] = sim.scene.rigid_body_index(self.obstacles[0], "box")
        c.link[2] = sim.scene.rigid_body_index(self.obstacles[1], "box")
        c.actor[0] = sim.scene.actor_index(self.block)
        c.actor[1] = sim.scene.actor_index(self.goal)
        c.w[6] = float(self.goal_yaw)
        return c


class BoxerPushObjective(PlanarPushObjective):
    """reference examples/boxer_push/planner.py"""


class HeijnPushObjective(PlanarPushObjective):
    """reference examples/heijn_push/planner.py (holonomic base, link front_link, softer collision weight)"""
    WEIGHTS = {"robot_to_block": 0.2, "block_to_goal": 2.0, "block_to_goal_ort": 3.0, "push_align": 0.6,
               "collision": 10, "velocity": 0.0}

    def __init__(self, cfg=None, robot="heijn", link="front_link", **kw):
        super().__init__(cfg, robot=robot, link=link, **kw)


class BaseReachObjective(ProgramObjective):
    """drive a mobile base to the goal without leaning on the wall (reference examples/boxer_reach/planner.py,
    examples/heijn_reach/planner.py: unit weights, no weight table)"""

    def __init__(self, cfg=None, robot="boxer", link="ee_link", goal="goal", wall="wall"):
        self.robot, self.link, self.goal, self.wall = robot, link, goal, wall
        super().__init__(cfg)

    def terms(self):
        return [Term(1.0, "dist", (actor(self.goal), link(self.robot, self.link), 2)), Term(1.0, "force_l1", (self.wall, "box", 3))]


class BoxerReachObjective(BaseReachObjective):
    """reference examples/boxer_reach/planner.py"""


class HeijnReachObjective(BaseReachObjective):
    """reference examples/heijn_reach/planner.py"""

    def __init__(self, cfg=None, robot="heijn", link="front_link", **kw):
        super().__init__(cfg, robot=robot, link=link, **kw)


class PandaPickObjective(ProgramObjective):
    """reach the block, bring it to the goal, stay off the table, keep the hand level
    (reference examples/panda_pick/planner.py:9-53)"""
    WEIGHTS = {"robot_to_block": 40.0, "block_to_goal": 10.0, "collision": 26.0, "robot_ori": 2.0}

    def __init__(self, cfg=None, robot="panda", link="panda_ee", block="panda_pick_block", goal="goal", table="table"):
        self.robot, self.link, self.block, self.goal, self.table = robot, link, block, goal, table
        super().__init__(cfg)

    def reset(self):  # attributes the reference's planner keeps (never read by its cost)
        self.prev_block_to_goal_dist = 1
        self.prev_robot_to_block_dist = 1

    def terms(self):
        hand, blk = link(self.robot, self.link), actor(self.block)
        return [Term("robot_to_block", "dist", (hand, blk, 3)), Term("block_to_goal", "dist", (blk, actor(self.goal), 3)),
                Term("collision", "force_l1", (self.table, "box", 3)), Term("robot_ori", "tilt", (hand,))]

    def fused_spec(self, sim) -> capi.Cost:
        c = self._spec(capi.COST_PANDA_PICK, ("robot_to_block", "block_to_goal", "collision", "robot_ori"))
        c.link[0] = sim.scene.rigid_body_index(self.robot, self.link)
        c.link[1] = sim.scene.rigid_body_index(self.table, "box")
        c.actor[0] = sim.scene.actor_index(self.block)
        c.actor[1] = sim.scene.actor_index(self.goal)
        return c


class OmniPandaPickObjective(PandaPickObjective):
    """mobile manipulator pick: the panda_pick terms plus base / arm speed, a comfortable arm pose and gripper opening,
    and a floor for the hand (reference examples/omni_panda_pick/planner.py; DOF order: 3 base, 7 arm, 2 fingers)"""
    WEIGHTS = {"robot_to_block": 10.0, "block_to_goal": 4.0, "collision": 0.1, "robot_ori": 1.0, "base_vel": 2.0,
               "arm_vel": 0.1, "comfy_gripper_state": 200.0, "comfy_arm_pose": 0.1, "height_cost": 10000.0}
    COMFY_GRIPPER = (0.025, 0.025)
    COMFY_ARM = (-1.57, -0.94, 0.0, -2.8, 0.0, 1.8675, 0.75)
    MIN_HAND_HEIGHT = 0.12

    def __init__(self, cfg=None, robot="omnipanda", link="panda_hand", **kw):
        super().__init__(cfg, robot=robot, link=link, **kw)

    def terms(self):
        hand = link(self.robot, self.link)
        return super().terms() + [
            Term("base_vel", "dof_sq", ("vel", 0, 3, None)), Term("arm_vel", "dof_sq", ("vel", 3, 10, None)),
            Term("comfy_gripper_state", "dof_sq", ("pos", -2, 0, self.COMFY_GRIPPER)),
            Term("comfy_arm_pose", "dof_sq", ("pos", 3, 10, self.COMFY_ARM)),
            Term("height_cost", "below", (hand, self.MIN_HAND_HEIGHT))]

    def fused_spec(self, sim) -> capi.Cost:   # (the parent's in-line PANDA_PICK kind does not know the extra terms)
        return self.program_spec(sim)


class PandaStickPushObjective(ProgramObjective):
    """push a block over a table with the stick of the panda_stick arm (reference examples/panda_stick_push/planner.py)"""
    WEIGHTS = {"robot_to_block": 5.0, "block_to_goal": 25.0, "collision": 0.0, "robot_ori": 5.0, "block_height": 20.0,
               "push_align": 45.0}

    def __init__(self, cfg=None, robot="panda", link="panda_ee_tip", block="panda_push_block", goal="goal", table="table"):
        self.robot, self.link, self.block, self.goal, self.table = robot, link, block, goal, table
        super().__init__(cfg)

    def reset(self):
        self.prev_block_to_goal_dist = 1
        self.prev_robot_to_block_dist = 1

    def terms(self):
        tip, blk, tgt = link(self.robot, self.link), actor(self.block), actor(self.goal)
        return [Term("robot_to_block", "dist", (tip, blk, 3)), Term("block_to_goal", "dist", (tgt, blk, 3)),
                Term("collision", "force_l1", (self.table, "box", 3)), Term("robot_ori", "tilt", (tip,)),
                Term("block_height", "abs_dz", (tip, blk)), Term("push_align", "align", (tip, blk, tgt))]


class AnymalWalkObjective(ProgramObjective):
    """quadruped: trunk to the goal while trunk and knees keep their nominal heights (reference examples/anymal/planner.py)"""
    WEIGHTS = {"robot_to_goal": 1.0, "robot_off_ground": 5.0, "knees_off_ground": 5.0}
    TRUNK_HEIGHT, KNEE_HEIGHT = 0.65, 0.35
    TRUNK_LINKS = ("base", "face_front", "face_rear")
    KNEE_LINKS = ("LF_KFE", "LH_KFE", "RH_KFE", "RF_KFE")

    def __init__(self, cfg=None, robot="anymal", goal="goal"):
        self.robot, self.goal = robot, goal
        super().__init__(cfg)

    def reset(self):
        self.prev_block_to_goal_dist = 1
        self.prev_robot_to_block_dist = 1

    def terms(self):
        out = [Term("robot_to_goal", "dist", (link(self.robot, "base"), actor(self.goal), 3))]
        out += [Term("robot_off_ground", "abs_dz", (link(self.robot, n), self.TRUNK_HEIGHT)) for n in self.TRUNK_LINKS]
        return out + [Term("knees_off_ground", "abs_dz", (link(self.robot, n), self.KNEE_HEIGHT)) for n in self.KNEE_LINKS]


class MultiJackalObjective(ProgramObjective):
    """several moving-base robots in one env (reference conf/mppi/multi-jackal.yaml; no planner of the reference uses it): every
    robot drives to its own target - the first to the goal actor, the others to fixed points"""
    WEIGHTS = {"robot_to_goal": 1.0}

    def __init__(self, cfg=None, robots=("jackal_a", "jackal_b"), goal="goal", targets=((-1.0, -3.0, 0.0),)):
        self.robots, self.goal, self.targets = tuple(robots), goal, tuple(targets)
        super().__init__(cfg)

    def terms(self):
        out = [Term("robot_to_goal", "dist", (actor(self.robots[0]), actor(self.goal), 2))]
        return out + [Term("robot_to_goal", "dist", (actor(r), t, 2)) for r, t in zip(self.robots[1:], self.targets)]



def specialise_program(terms: Sequence[Term], weights: dict, scene) -> capi.Cost:
    """Term list -> mppi_cost_t: one of the IN-LINE kinds when the list has exactly that shape (the four workloads of BASELINE.json:
    the rollout kernels evaluate those without the interpreter, and a contact-free scene keeps its octet-layout kernel), otherwise
    the MPPI_COST_PROGRAM of `compile_program`.  Used for traced Objectives (mppiisaac/trace.py), whose Term lists carry numbers
    as weights."""
    def w_of(t):
        return float(weights[t.weight] if isinstance(t.weight, str) else t.weight)

    by_op = {}
    for t in terms:
        by_op.setdefault(t.op, []).append(t)
    ops = {k: len(v) for k, v in by_op.items()}
    is_link = lambda s: isinstance(s, tuple) and len(s) == 3 and s[0] == "link"
    is_actor = lambda s: isinstance(s, tuple) and len(s) == 2 and s[0] == "actor"

    def plain(kind, ws):
        c = capi.Cost()
        c.kind = kind
        for i, v in enumerate(ws):
            c.w[i] = float(v)
        return c

    def pair(t, n):
        """(a, b) of a dist term over n components, or None"""
        return (t.args[0], t.args[1]) if t.args[2] == n else None
    # PANDA_REACH: w0 |link - actor|_3 + w1 tilt(link)
    if ops == {"dist": 1, "tilt": 1} or ops == {"dist": 1}:
        d = by_op["dist"][0]
        p = pair(d, 3)
        if p is not None:
            lk = next((s for s in p if is_link(s)), None)
            ac = next((s for s in p if is_actor(s)), None)
            tl = by_op.get("tilt", [None])[0]
            if lk is not None and ac is not None and (tl is None or tl.args[0] == lk) and lk[1] in [scene.env_cfg[i].name for i in scene.robot_ids]:
                c = plain(capi.COST_PANDA_REACH, (w_of(d), w_of(tl) if tl is not None else 0.0))
                c.link[0] = scene.rigid_body_index(lk[1], lk[2])
                c.actor[0] = scene.actor_index(ac[1])
                return c
        p = pair(d, 2)
        if p is not None and ops == {"dist": 1} and ("dof_xy",) in p:   # POINT_REACH: w |dof_xy - goal|_2
            other = p[1] if p[0] == ("dof_xy",) else p[0]
            c = plain(capi.COST_POINT_REACH, (w_of(d),))
            if is_actor(other):
                c.actor[0] = scene.actor_index(other[1])
                return c
            if isinstance(other, tuple) and len(other) == 3 and all(isinstance(v, float) for v in other):
                c.actor[0] = -1
                c.w[1], c.w[2] = other[0], other[1]
                return c
    # PANDA_PICK: w0 |hand - block|_3 + w1 |block - goal|_3 + w2 |F_table|_1 + w3 tilt(hand)
    if ops == {"dist": 2, "force_l1": 1, "tilt": 1}:
        tl, fo = by_op["tilt"][0], by_op["force_l1"][0]
        hand = tl.args[0]
        d_hb = next((t for t in by_op["dist"] if hand in t.args[:2] and t.args[2] == 3), None)
        d_bg = next((t for t in by_op["dist"] if t is not d_hb and t.args[2] == 3 and is_actor(t.args[0]) and is_actor(t.args[1])), None)
        if d_hb is not None and d_bg is not None and fo.args[2] == 3:
            blk = d_hb.args[1] if d_hb.args[0] == hand else d_hb.args[0]
            if is_actor(blk) and blk in d_bg.args[:2]:
                goal = d_bg.args[1] if d_bg.args[0] == blk else d_bg.args[0]
                c = plain(capi.COST_PANDA_PICK, (w_of(d_hb), w_of(d_bg), w_of(fo), w_of(tl)))
                c.link[0] = scene.rigid_body_index(hand[1], hand[2])
                c.link[1] = scene.rigid_body_index(fo.args[0], fo.args[1])
                c.actor[0], c.actor[1] = scene.actor_index(blk[1]), scene.actor_index(goal[1])
                return c
    # BOXER_PUSH: pusher -> block, block -> goal (planar), |yaw - ref|, align, planar speed, |F_xy| of two obstacles (one weight)
    if ops == {"dist": 2, "yaw_abs": 1, "align": 1, "force_l1": 2} or ops == {"dist": 2, "yaw_abs": 1, "align": 1, "force_l1": 2, "speed": 1}:
        al, ya = by_op["align"][0], by_op["yaw_abs"][0]
        pusher, blk, goal = al.args
        f1, f2 = by_op["force_l1"]
        sp = by_op.get("speed", [None])[0]
        d_rb = next((t for t in by_op["dist"] if t.args[2] == 2 and set(t.args[:2]) == {pusher, blk}), None)
        d_bg = next((t for t in by_op["dist"] if t.args[2] == 2 and set(t.args[:2]) == {goal, blk}), None)
        if (d_rb is not None and d_bg is not None and d_rb is not d_bg and is_link(pusher) and is_actor(blk) and is_actor(goal) and ya.args[0] == blk
                and f1.args[2] == f2.args[2] == 2 and w_of(f1) == w_of(f2) and (sp is None or (sp.args[0] == blk and sp.args[1] == 2))):
            c = plain(capi.COST_BOXER_PUSH, (w_of(d_rb), w_of(d_bg), w_of(ya), w_of(al), w_of(sp) if sp is not None else 0.0, w_of(f1), float(ya.args[1])))
            c.link[0] = scene.rigid_body_index(pusher[1], pusher[2])
            c.link[1], c.link[2] = scene.rigid_body_index(f1.args[0], f1.args[1]), scene.rigid_body_index(f2.args[0], f2.args[1])
            c.actor[0], c.actor[1] = scene.actor_index(blk[1]), scene.actor_index(goal[1])
            return c
    return compile_program(terms, weights, scene)

"""Batched rigid-body simulator with the reference's `IsaacGymWrapper` surface
(reference mppiisaac/planner/isaacgym_wrapper.py:83-774), backed by libmppi_hip.so.

What is kept from the reference: the constructor signature, `env_cfg` (list of ActorWrapper),
`num_envs`, `device`, the four state tensors `_dof_state [K,2n]`, `_root_state [K,A,13]`,
`_rigid_body_state [K,B,13]`, `_net_contact_force [K,B,3]` (same layouts, :186-199), the name-based
getters used by Objectives (:298-356), `apply_robot_cmd` (:524-572), `step` (:639-655),
`reset_robot_state` (:574-619), `save/reset_root_state` (:662-675), `visualize_link_buffer`.
What is different: the state of truth lives in sample-minor device buffers inside the HIP
library; the reference-layout tensors are materialised after each `step()`.  Viewer, keyboard
listeners and line drawing are graphics and out of scope.
"""
from __future__ import annotations

import contextlib
import ctypes
import logging
import math
from dataclasses import dataclass, field
from enum import Enum
from typing import List, Optional

import numpy as np
import torch

from mppiisaac.backend import capi


@dataclass
class IsaacGymConfig(object):
    # reference isaacgym_wrapper.py:10-18; only dt and substeps act on the HIP simulator
    dt: float = 0.05
    substeps: int = 2
    use_gpu_pipeline: bool = True
    num_client_threads: int = 0
    viewer: bool = False
    num_obstacles: int = 10
    spacing: float = 6.0


class SupportedActorTypes(Enum):
    Axis = 1
    Robot = 2
    Sphere = 3
    Box = 4


@dataclass
class ActorWrapper:
    # field names and defaults: reference isaacgym_wrapper.py:49-77 (pinned by tests/golden/actor_cfgs.json)
    type: SupportedActorTypes
    name: str
    dof_mode: str = "velocity"
    init_pos: List[float] = field(default_factory=lambda: [0, 0, 0])
    init_ori: List[float] = field(default_factory=lambda: [0, 0, 0, 1])
    size: List[float] = field(default_factory=lambda: [0.1, 0.1, 0.1])
    mass: float = 1.0
    color: List[float] = field(default_factory=lambda: [1.0, 1.0, 1.0])
    fixed: bool = False
    collision: bool = True
    friction: float = 1.0
    handle: Optional[int] = None
    flip_visual: bool = False
    urdf_file: str = None
    visualize_link: str = None
    gravity: bool = True
    differential_drive: bool = False
    init_joint_pose: List[float] = None
    wheel_radius: Optional[float] = None
    wheel_base: Optional[float] = None
    wheel_count: Optional[float] = None
    left_wheel_joints: Optional[List[str]] = None
    right_wheel_joints: Optional[List[str]] = None
    caster_links: Optional[List[str]] = None
    noise_sigma_size: Optional[List[float]] = None
    noise_percentage_mass: float = 0.0
    noise_percentage_friction: float = 0.0


DRIVE_GAINS = {"velocity": (capi.DRIVE_VELOCITY, 600.0, 0.0), "effort": (capi.DRIVE_EFFORT, 10.0, 0.0),
               "position": (capi.DRIVE_POSITION, 0.0, 80.0)}  # (mode, damping, stiffness): reference :491-507
GRAVITY = (0.0, 0.0, -9.8)  # reference :29
_REPORTED_DROPS = set()


def diff_drive_ik(actor: ActorWrapper, u: torch.Tensor):
    """(v, omega) -> (left, right) wheel velocity; reference IsaacGymWrapper._ik :510-522."""
    r, L = actor.wheel_radius, actor.wheel_base
    left = (u[:, 0] / r) - ((L * u[:, 1]) / (2 * r))
    right = (u[:, 0] / r) + ((L * u[:, 1]) / (2 * r))
    return left, right


def interleave_dof_state(q, qdot, n_dof: int) -> np.ndarray:
    """[q0, qd0, q1, qd1, ...] as float32 - the dof_state row reset_robot_state builds (reference :606-612)."""
    dof = np.zeros(2 * n_dof, np.float32)
    dof[0::2] = np.asarray(q, np.float32)[:n_dof]
    dof[1::2] = np.asarray(qdot, np.float32)[:n_dof]
    return dof


def _quat_to_R(q) -> np.ndarray:
    x, y, z, w = (float(v) for v in q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def merge_robots(env_cfg: List["ActorWrapper"], robots: List[int], models: List[dict]):
    """Several robots in one env (reference isaacgym_wrapper.py:101-106,220-236,534-559,574-612: commands, joint states and
    initial joint poses of the robots are concatenated in env order).  Fixed-base robots are independent trees hanging off the
    world, so together they are ONE articulated forest: the compiled models are concatenated, every later robot's root joints
    and base-welded links carry its base pose relative to the first robot's.  Moving-base robots (multi-jackal) form a forest
    with one floating base per tree (mppi_hip.h, ABI 7).  -> (merged model, actor index per link)"""
    moving = [not env_cfg[i].fixed for i in robots]
    if any(moving) and not all(moving):
        raise NotImplementedError("several robots per env: the robots of an env are either all fixed or all moving")
    for i in robots:
        a = env_cfg[i]
        if a.dof_mode != env_cfg[robots[0]].dof_mode:
            raise ValueError("All robots must have the same dof_mode")        # (reference :541-542)
        if bool(a.gravity) != bool(env_cfg[robots[0]].gravity):
            # the kernels apply gravity to the whole forest from ONE flag (DevModel.gravity_on, the first robot's); the reference
            # sets disable_gravity per actor (isaacgym_utils.py:17-18) - refused rather than silently different
            raise NotImplementedError("several robots per env: the robots of an env must agree on `gravity`")
    if robots != list(range(robots[0], robots[0] + len(robots))):
        raise NotImplementedError("several robots per env: list the robot actors next to each other (their rigid-body rows form one block)")
    if all(moving) and len(robots) > 1 + capi.MAX_EXTRA_BASES:
        raise ValueError(f"several robots per env: at most {1 + capi.MAX_EXTRA_BASES} moving bases (MPPI_MAX_EXTRA_BASES)")
    if sum(len(m["links"]) for m in models) > capi.MAX_LINKS:
        # more URDF links than reported rigid-body rows exist (two jackals: 2 x 14): report the links somebody can observe - base
        # link, collision geometry, visualize_link - as for the ANYmal's 78 (urdf_compile.prune_links; links are addressed by name)
        from mppiisaac.backend.urdf_compile import prune_links
        models = [prune_links(m, keep=[env_cfg[i].visualize_link] if env_cfg[i].visualize_link else ()) for i, m in zip(robots, models)]
        logging.getLogger("mppiisaac").warning("several robots per env: %d links without collision geometry are not reported as rigid bodies "
                                               "(MPPI_MAX_LINKS = %d)", sum(m["pruned_links"] for m in models), capi.MAX_LINKS)
    first = env_cfg[robots[0]]
    R0, p0 = _quat_to_R(first.init_ori), np.asarray(first.init_pos, float)
    merged = {"format": models[0]["format"], "name": "+".join(m["name"] for m in models), "root_link": models[0]["root_link"],
              "links": [], "bodies": [], "base": models[0]["base"]}
    # MOVING bases (ABI 7; reference conf/mppi/multi-jackal.yaml): every robot keeps its own root row and its own 6x6 base system;
    # a root body's parent and a base link's body index name the base they hang off as -1 - r.  No relative poses: the bases move
    merged["bases"] = [dict(actor=i, inertia=m["base"]["inertia"], own_inertia=m["links"][0]["own_inertia"]) for i, m in zip(robots, models)] \
        if all(moving) else None
    owner = []
    for r, (i, m) in enumerate(zip(robots, models)):
        a = env_cfg[i]
        R_rel = R0.T @ _quat_to_R(a.init_ori)
        p_rel = R0.T @ (np.asarray(a.init_pos, float) - p0)
        boff, loff = len(merged["bodies"]), len(merged["links"])
        for b in m["bodies"]:
            nb = dict(b)
            if b["parent"] < 0:
                if all(moving):
                    nb["parent"] = -1 - r
                else:
                    nb["R_tree"] = (R_rel @ np.asarray(b["R_tree"])).tolist()
                    nb["p_tree"] = (R_rel @ np.asarray(b["p_tree"]) + p_rel).tolist()
            else:
                nb["parent"] = b["parent"] + boff
            merged["bodies"].append(nb)
        for l in m["links"]:
            nl = dict(l)
            nl["parent_link"] = l["parent_link"] + loff if l["parent_link"] >= 0 else -1
            if l["body"] < 0:
                if all(moving):
                    nl["body"] = -1 - r
                else:
                    nl["R"] = (R_rel @ np.asarray(l["R"])).tolist()
                    nl["p"] = (R_rel @ np.asarray(l["p"]) + p_rel).tolist()
            else:
                nl["body"] = l["body"] + boff
            merged["links"].append(nl)
            owner.append(i)
    return merged, owner


class Scene:
    """Host description of one env: actors + compiled robot model -> C-ABI mppi_model_t."""

    def __init__(self, env_cfg: List[ActorWrapper], cfg: IsaacGymConfig, robot_model):
        robots = [i for i, a in enumerate(env_cfg) if a.type == "robot"]
        if not robots:
            raise NotImplementedError("an env needs a robot actor")
        if len(env_cfg) > capi.MAX_ACTORS:
            raise ValueError("too many actors")
        self.env_cfg = env_cfg
        self.cfg = cfg
        self.robot_ids = robots
        self.robot_idx = robots[0]
        self.robot = env_cfg[self.robot_idx]
        if len(robots) > 1:   # robot_model: one compiled model per robot actor, in env order
            if not isinstance(robot_model, (list, tuple)) or len(robot_model) != len(robots):
                raise ValueError("several robots per env: pass one compiled model per robot actor")
            robot_model, self.link_owner = merge_robots(env_cfg, robots, list(robot_model))
        else:
            robot_model = robot_model[0] if isinstance(robot_model, (list, tuple)) else robot_model
            self.link_owner = [self.robot_idx] * len(robot_model["links"])
        self.robot_model = robot_model
        self.dof_names = [b["joint"] for b in robot_model["bodies"]]
        self.n_dof = len(self.dof_names)
        self.link_names = [l["name"] for l in robot_model["links"]]
        # rigid-body rows: actors in env order; a robot contributes its links, box/sphere one body
        self.first_rb, self.rb_names = [], []
        for i, a in enumerate(env_cfg):
            self.first_rb.append(len(self.rb_names))
            if a.type == "robot":
                self.rb_names += [(a.name, n) for n, o in zip(self.link_names, self.link_owner) if o == i]
            else:
                self.rb_names.append((a.name, a.type))  # primitive bodies are named "box"/"sphere"
        self.n_rb = len(self.rb_names)
        self.cmd_terms, self.nu = self._command_map()
        self._report_deviations()
        self.shapes, self.pairs = self._contact_scene()
        self.randomize_seed = -1  # >= 0: per-sample size/mass/friction draws of the noisy box/sphere actors

    def _command_map(self):
        """apply_robot_cmd's scatter (reference :524-559) as <=2 (column, coefficient) terms per DOF.  Several robots: their DOFs
        follow one another and take the next commands in turn - a diff-drive robot its own (v, yaw rate) pair.  (The reference's
        loop hands EVERY diff-drive robot the first two commands and writes through actor-local DOF indices, :545-559: with two
        jackals the second one's columns stay zero.  Restated as what `u_desired_idx += 2` says it means.)"""
        idx, terms = 0, []
        owner_of_dof = [self._actor_of_body(b) for b in range(self.n_dof)]
        for ai in self.robot_ids:
            a = self.env_cfg[ai]
            base = idx
            if a.differential_drive:
                if not a.left_wheel_joints or not a.right_wheel_joints:
                    # e.g. the shipped conf/actors/jackal.yaml: the reference evaluates `name in None` here and raises
                    # TypeError (isaacgym_wrapper.py:552-555); fail with a message instead
                    raise ValueError(f"actor '{a.name}': differential_drive needs left_wheel_joints and right_wheel_joints")
                idx += 2
            for b, name in enumerate(self.dof_names):
                if owner_of_dof[b] != ai:
                    continue
                if a.differential_drive and name in (a.left_wheel_joints or []):
                    terms.append(((base, 1.0 / a.wheel_radius), (base + 1, -a.wheel_base / (2 * a.wheel_radius))))
                elif a.differential_drive and name in (a.right_wheel_joints or []):
                    terms.append(((base, 1.0 / a.wheel_radius), (base + 1, a.wheel_base / (2 * a.wheel_radius))))
                else:
                    terms.append(((idx, 1.0), (0, 0.0)))
                    idx += 1
        return terms, idx

    def substeps(self) -> int:
        """integration steps per control interval: cfg.substeps, multiplied up in contact scenes until a step is no longer than
        MAX_CONTACT_SUBSTEP (see there)"""
        n = int(self.cfg.substeps)
        h = float(self.cfg.dt) / n
        if self.pairs and self.MAX_CONTACT_SUBSTEP > 0 and h > self.MAX_CONTACT_SUBSTEP * (1 + 1e-9):
            k = int(math.ceil(h / self.MAX_CONTACT_SUBSTEP - 1e-9))
            if "contact-substep" not in _REPORTED_DROPS:
                _REPORTED_DROPS.add("contact-substep")
                logging.getLogger("mppiisaac").warning(
                    "known deviation from the reference: contact scene with dt / substeps = %.3f s: integrated with %d x %d substeps of %.4f s "
                    "(the penalty contact carries a body within |g| h^2 / alpha of the surface: %.0f mm at the configured step, %.1f mm now)",
                    h, n, k, h / k, 1e3 * abs(GRAVITY[2]) * h * h / self.CONTACT_ALPHA, 1e3 * abs(GRAVITY[2]) * (h / k) ** 2 / self.CONTACT_ALPHA)
            n *= k
        return n

    def _report_deviations(self):
        """behaviour that differs from what the reference's code literally does ("Known deviations"): said once
        per process and kind, like the pruned links of merge_robots"""
        notes = []
        if sum(1 for i in self.robot_ids if self.env_cfg[i].differential_drive) > 1:
            notes.append(("multi-diff-drive", "several differential-drive robots in one env: every robot takes its OWN (v, yaw rate) pair of the command; the "
                                              "reference hands u[:, :2] to each of them (isaacgym_wrapper.py:545-559)"))
        if self.robot.dof_mode == "position":
            notes.append(("position-mode", "dof_mode 'position': the command overwrites q (qd <- 0) at every step and is then HELD by a stiffness drive "
                                           "(kp = 80); the reference writes a half-size tensor with set_dof_state_tensor and sets no position target "
                                           "(isaacgym_wrapper.py:501-504,571-572)"))
        for key, text in notes:
            if key not in _REPORTED_DROPS:
                _REPORTED_DROPS.add(key)
                logging.getLogger("mppiisaac").warning("known deviation from the reference: %s", text)

    def _actor_of_body(self, body: int) -> int:
        """robot actor that owns moving body `body` (the owner of the links welded to it)"""
        for l, o in zip(self.robot_model["links"], self.link_owner):
            if l["body"] == body:
                return o
        return self.robot_idx

    # contact parameters of the penalty model (build-normative; no reference counterpart)
    CONTACT_ALPHA, CONTACT_BETA, FRICTION_BETA, GROUND_FRICTION = 0.8, 0.8, 1.0, 1.0
    # depth over which the damper and the implicit spring term of a contact ramp in (None: the static sag |g| h^2 / alpha of a
    # body resting on its contact patch - 7.7 mm at h = 25 ms -, so a body at rest sees the full law; 0: no ramp)
    CONTACT_RAMP_DEPTH = None
    # wheel / caster discs against the boxes of other actors (round 5; False restores the ground-only wheels of rounds 1-4: an A/B
    # switch for measurements - MPPI_WHEEL_BOX_PAIRS=0 in the environment does the same)
    import os as _os
    WHEEL_BOX_PAIRS = _os.environ.get("MPPI_WHEEL_BOX_PAIRS", "1") != "0"
    # round 5: the robots of one env meet each other (the boxes and spheres of their moving links: chassis against chassis); two
    # dynamic boxes get ONE normal per pair from a separating-axis test (mppi_model_t.contact_flags).  Switches for
    # measurements and for the known-answer tests of the laws before round 5
    ROBOT_ROBOT_PAIRS = _os.environ.get("MPPI_ROBOT_ROBOT_PAIRS", "1") != "0"
    BOX_PAIR_NORMAL = _os.environ.get("MPPI_BOX_PAIR_NORMAL", "1") != "0"
    # round 6: a free actor of at most 0.25 kg that the robot outweighs 100 times is held IMPLICITLY by the links that touch it
    # (mppi_model_t.contact_flags bit 1, DESIGN.md 3 "light bodies"; reference examples/panda_pick: the 1-gram block).  True restores the
    # explicit law of rounds 1-5 - a gripper closes through the block - for measurements and the tests that show the difference
    EXPLICIT_LIGHT = _os.environ.get("MPPI_EXPLICIT_LIGHT", "0") == "1"
    # The penalty contact's stiffness is tied to the integration step, k = alpha m / h^2 (what an explicit step of length h can carry):
    # a body at rest sags |g| h^2 / alpha into what it rests on - 7.7 mm at the 25 ms of conf/isaacgym/normal.yaml, 12 CENTIMETRES at
    # the 100 ms of conf/isaacgym/push.yaml (dt 0.1, substeps 1: fine for PhysX's implicit solver; here the block of heijn_push sank
    # into the floor until the robot's bumper passed over it).  Contact scenes therefore never integrate with a longer step than
    # this: the configured substeps are multiplied up (push.yaml: 1 -> 4), dt - what one control interval is - stays.  0 = off.
    MAX_CONTACT_SUBSTEP = float(_os.environ.get("MPPI_MAX_CONTACT_SUBSTEP", "0.025"))

    def _contact_scene(self):
        """Collision primitives and candidate pairs of one env.

        Robot links contribute their URDF <collision> geometry (meshes as their AABB box, thin cylinders
        as discs); box/sphere actors their own shape (reference isaacgym_utils.py:26-52).  Pairs follow the
        reference's collision filter: same env only, both actors `collision: true`
        (isaacgym_wrapper.py:441), no self-collision of one robot; the robots of an env meet each other (boxes and spheres of their
        moving links); wheels / casters meet the ground and the boxes and spheres of OTHER actors, not other wheels or other robots
        (those candidates are listed in `dropped_pairs`)."""
        shapes = []
        self.dropped_pairs = []  # (shape, shape) candidates the contact model leaves out on purpose
        self.dropped_pair_shapes = []  # the same candidates as indices into the shape list (tests measure their clearance)
        # (a scene in which no candidate pair survives the filter below - e.g. a fixed-base arm and a collision-free goal -
        # is contact-free: the contact-free kernels run it.  A FIXED-base robot still moves its links into static geometry:
        # heijn_reach's base against the wall, an arm against a fixed obstacle sphere)
        for ai, a in enumerate(self.env_cfg):
            if not a.collision:
                continue
            if a.type == "robot":
                casters = set(a.caster_links or [])
                for li, l in enumerate(self.robot_model["links"]):
                    if self.link_owner[li] != ai:
                        continue
                    Rl, pl = np.asarray(l["R"]), np.asarray(l["p"])
                    for c in l["collision"]:
                        Rc, pc = np.asarray(c["R"]), np.asarray(c["p"])
                        t = c["type"]
                        if t == "box":
                            kind, size = capi.SHAPE_BOX, [0.5 * v for v in c["size"]]
                        elif t == "sphere":
                            kind, size = capi.SHAPE_SPHERE, [c["radius"], 0.0, 0.0]
                        elif t == "cylinder" and (c["length"] < 0.25 * c["radius"] or self._is_wheel(l, Rl @ Rc)):
                            kind, size = capi.SHAPE_DISC, [c["radius"], 0.0, 0.0]      # wheel / caster: rim contact
                        elif t == "cylinder":
                            kind, size = capi.SHAPE_BOX, [c["radius"], c["radius"], 0.5 * c["length"]]
                        elif t == "mesh":
                            lo, hi = np.asarray(c["aabb_min"]), np.asarray(c["aabb_max"])
                            kind, size = capi.SHAPE_BOX, list(0.5 * (hi - lo))
                            pc = pc + Rc @ (0.5 * (hi + lo))
                        else:
                            continue
                        # (several robots form one articulated forest: their link shapes all belong to it; `owner` says whose
                        # they are - the links of ONE robot do not meet, those of different robots do: pair filter below)
                        shapes.append(dict(actor=self.robot_idx, body=l["body"], type=kind, rb=self.first_rb[self.robot_idx] + li, size=size,
                                           R=Rl @ Rc, p=Rl @ pc + pl, friction=0.0 if l["name"] in casters else a.friction,
                                           fixed=bool(a.fixed), link=l["name"], R_in_link=Rc, p_in_link=pc, owner=ai))
            elif a.type == "box":
                shapes.append(dict(actor=ai, body=-1, type=capi.SHAPE_BOX, rb=self.first_rb[ai], size=[0.5 * v for v in a.size],
                                   R=np.eye(3), p=np.zeros(3), friction=a.friction, fixed=bool(a.fixed), link="box",
                                   R_in_link=np.eye(3), p_in_link=np.zeros(3)))
            elif a.type == "sphere":
                shapes.append(dict(actor=ai, body=-1, type=capi.SHAPE_SPHERE, rb=self.first_rb[ai], size=[a.size[0], 0.0, 0.0],
                                   R=np.eye(3), p=np.zeros(3), friction=a.friction, fixed=bool(a.fixed), link="sphere",
                                   R_in_link=np.eye(3), p_in_link=np.zeros(3)))
        pairs = []
        for i, si in enumerate(shapes):
            robot_i = self.env_cfg[si["actor"]].type == "robot"
            # a robot link welded to a FIXED base never reacts: it only matters as an obstacle
            static_i = si["fixed"] and (not robot_i or si["body"] < 0)
            if not si["fixed"]:
                pairs.append((i, -1))  # ground plane
            for j in range(i + 1, len(shapes)):
                sj = shapes[j]
                robot_j = self.env_cfg[sj["actor"]].type == "robot"
                static_j = sj["fixed"] and (not robot_j or sj["body"] < 0)
                if static_i and static_j:
                    continue
                kinds = {si["type"], sj["type"]}
                if si["actor"] == sj["actor"]:
                    # the links of ONE robot never meet (the reference's self-collision filter, isaacgym_wrapper.py:441).  The
                    # robots of an env do (round 5; one collision group per env) - moving bases and the moving links of fixed-base
                    # robots alike: their boxes and spheres against each other - chassis against chassis.  The WHEELS of one robot
                    # against another robot are left out and listed: every further pair between the same two bodies adds a nominal
                    # stiffness of its own to the explicit law ("several robots per env"; the jackal's tyres stand
                    # 2 cm proud of its chassis: two jackals meet that much later than their tyres would)
                    if not (robot_i and si.get("owner") != sj.get("owner")):
                        continue
                    if capi.SHAPE_DISC in kinds or not self.ROBOT_ROBOT_PAIRS:
                        self.dropped_pairs.append((f'{self.env_cfg[si["owner"]].name}:{si["link"]}', f'{self.env_cfg[sj["owner"]].name}:{sj["link"]}'))
                        self.dropped_pair_shapes.append((i, j))
                        continue
                names = (f'{self.env_cfg[si.get("owner", si["actor"])].name}:{si["link"]}', f'{self.env_cfg[sj.get("owner", sj["actor"])].name}:{sj["link"]}')
                if kinds == {capi.SHAPE_DISC}:
                    # round 5: wheels and casters meet the boxes and spheres of other actors (block, obstacles, walls, table, the
                    # obstacle spheres of the benchmark adapters) as well as the ground; what is still left out - a wheel against
                    # another wheel - is listed here and logged
                    self.dropped_pairs.append(names)
                    self.dropped_pair_shapes.append((i, j))
                    continue
                if capi.SHAPE_DISC in kinds and not self.WHEEL_BOX_PAIRS:
                    self.dropped_pairs.append(names)
                    self.dropped_pair_shapes.append((i, j))
                    continue
                pairs.append((i, j))
        self.all_shapes = shapes  # (kept when the scene turns out contact-free, for the clearance tests)
        if not pairs:
            return [], []  # nothing can come into contact with anything that reacts
        if len(shapes) > capi.MAX_SHAPES or len(pairs) > capi.MAX_PAIRS:
            raise ValueError(f"contact scene too large: {len(shapes)} shapes, {len(pairs)} pairs")
        if self.dropped_pairs:
            key = tuple(self.dropped_pairs)
            if key not in _REPORTED_DROPS:  # once per distinct scene and process
                _REPORTED_DROPS.add(key)
                logging.getLogger("mppiisaac").warning(
                    "contact model: %d wheel/caster pairs are not tested (a wheel against another wheel or against another robot%s; e.g. %s / %s); "
                    "see Scene.dropped_pairs", len(key), "" if self.WHEEL_BOX_PAIRS else ", and - MPPI_WHEEL_BOX_PAIRS=0 - against boxes and spheres", *key[0])
        return shapes, pairs

    def _is_wheel(self, link: dict, R_shape: np.ndarray) -> bool:
        """a cylinder that spins about its own axis on one of the actor's declared wheel joints"""
        if link["body"] < 0:
            return False
        b = self.robot_model["bodies"][link["body"]]
        owner = self.env_cfg[self._actor_of_body(link["body"])]
        wheels = (owner.left_wheel_joints or []) + (owner.right_wheel_joints or [])
        return b["joint"] in wheels and abs(float(np.dot(R_shape[:, 2], np.asarray(b["axis"])))) > 0.99

    def viz_link_index(self) -> int:
        """link index (within the robot) of ActorWrapper.visualize_link, -1 if unset."""
        v = self.robot.visualize_link
        # e.g. conf/actors/omnipanda_effort.yaml names a link its URDF does not have (Isaac Gym then reports -1)
        return self.link_names.index(v) if v and v in self.link_names else -1

    def actor_index(self, name: str) -> int:
        return [a.name for a in self.env_cfg].index(name)

    def rigid_body_index(self, actor_name: str, link_name: str) -> int:
        return self.rb_names.index((actor_name, link_name))

    def initial_state(self):
        """dof_state [2n] (interleaved q, qdot) and root_state [A,13] of the initial pose
        (reference :219-236 and reset_to_initial_poses :238-246)."""
        dof = np.zeros(2 * self.n_dof, np.float32)
        off = 0
        for i in self.robot_ids:   # (the reference keeps only the LAST robot's pose here, :220-233 - a defect; each robot gets its own)
            a = self.env_cfg[i]
            n = self._n_dof_of(i)
            if a.init_joint_pose:
                pose = np.asarray(a.init_joint_pose, np.float32)[:2 * n]
                dof[2 * off:2 * off + len(pose)] = pose
            off += n
        root = np.zeros((len(self.env_cfg), 13), np.float32)
        for i, a in enumerate(self.env_cfg):
            root[i, 0:3] = a.init_pos
            root[i, 3:7] = a.init_ori
        return dof, root

    def _n_dof_of(self, actor_idx: int) -> int:
        """DOFs of one robot actor of a multi-robot env (bodies whose links that actor owns)"""
        bodies = {l["body"] for l, o in zip(self.robot_model["links"], self.link_owner) if o == actor_idx and l["body"] >= 0}
        return len(bodies)

    def to_c(self) -> capi.Model:
        m = capi.Model()
        m.abi_version = capi.ABI_VERSION
        m.n_actors = len(self.env_cfg)
        kinds = {"robot": capi.ACTOR_ROBOT, "box": capi.ACTOR_BOX, "sphere": capi.ACTOR_SPHERE}
        for i, a in enumerate(self.env_cfg):
            ca = m.actors[i]
            if a.type not in kinds:
                raise NotImplementedError(f"actor asset of type {a.type} is not yet implemented!")
            ca.type = kinds[a.type]
            ca.fixed, ca.collision, ca.gravity = int(a.fixed), int(a.collision), int(a.gravity)
            size = list(a.size) + [0.0] * (3 - len(a.size))
            for j in range(3):
                ca.size[j] = float(size[j])
            ca.mass, ca.friction = float(a.mass), float(a.friction)
            if a.type != "robot":
                ns = list(a.noise_sigma_size or []) + [0.0] * 3
                for j in range(3):
                    ca.noise_sigma_size[j] = float(ns[j])
                ca.noise_percentage_mass = float(a.noise_percentage_mass)
                ca.noise_percentage_friction = float(a.noise_percentage_friction)
            ca.first_rb = self.first_rb[i]
            ca.n_rb = sum(1 for o in self.link_owner if o == i) if a.type == "robot" else 1
        m.robot_actor = self.robot_idx
        rm = self.robot_model
        if len(rm["bodies"]) > capi.MAX_BODIES or len(rm["links"]) > capi.MAX_LINKS:
            raise ValueError("robot exceeds MPPI_MAX_BODIES / MPPI_MAX_LINKS")
        m.n_bodies = len(rm["bodies"])
        for i, b in enumerate(rm["bodies"]):
            cb = m.bodies[i]
            cb.parent = b["parent"]
            cb.jtype = capi.JOINT_PRISMATIC if b["jtype"] == "prismatic" else capi.JOINT_REVOLUTE
            for j in range(3):
                cb.axis[j] = b["axis"][j]
                cb.p_tree[j] = b["p_tree"][j]
                cb.h[j] = b["inertia"]["h"][j]
            for j in range(9):
                cb.R_tree[j] = b["R_tree"][j // 3][j % 3]
            cb.mass = b["inertia"]["mass"]
            for j in range(6):
                cb.Io[j] = b["inertia"]["Io"][j]
            cb.limited = int(b["limited"])
            cb.lower, cb.upper, cb.effort, cb.velocity = b["lower"], b["upper"], b["effort"], b["velocity"]
            (c0, k0), (c1, k1) = self.cmd_terms[i]
            m.cmd_col[i][0], m.cmd_col[i][1] = c0, c1
            m.cmd_coef[i][0], m.cmd_coef[i][1] = k0, k1
        m.n_links = len(rm["links"])
        m.n_rb = self.n_rb
        for i, l in enumerate(rm["links"]):
            cl = m.links[i]
            cl.body = l["body"]
            for j in range(9):
                cl.R[j] = l["R"][j // 3][j % 3]
            for j in range(3):
                cl.p[j] = l["p"][j]
        # the reference overwrites the mass of rigid body 0 of every actor with ActorWrapper.mass
        # (isaacgym_wrapper.py:450-456); the inertia tensor is left as imported.  Only matters for floating bases.
        def base_inertia(bi, own, actor):
            dm = float(actor.mass) - own["mass"]
            c0 = np.asarray(own["h"]) / own["mass"] if own["mass"] > 0 else np.zeros(3)
            return bi["mass"] + dm, [bi["h"][j] + dm * c0[j] for j in range(3)], list(bi["Io"])
        bases = rm.get("bases") or [dict(actor=self.robot_idx, inertia=rm["base"]["inertia"], own_inertia=rm["links"][0]["own_inertia"])]
        m.base_mass, bh, bI = base_inertia(bases[0]["inertia"], bases[0]["own_inertia"], self.env_cfg[bases[0]["actor"]])
        for j in range(3):
            m.base_h[j] = bh[j]
        for j in range(6):
            m.base_Io[j] = bI[j]
        m.n_extra_bases = len(bases) - 1            # several moving-base robots (ABI 7): base r > 0
        for r, b in enumerate(bases[1:]):
            m.extra_base_actor[r] = b["actor"]
            m.extra_base_mass[r], bh, bI = base_inertia(b["inertia"], b["own_inertia"], self.env_cfg[b["actor"]])
            for j in range(3):
                m.extra_base_h[r][j] = bh[j]
            for j in range(6):
                m.extra_base_Io[r][j] = bI[j]
        m.n_shapes, m.n_pairs = len(self.shapes), len(self.pairs)
        for i, sh in enumerate(self.shapes):
            cs = m.shapes[i]
            cs.actor, cs.body, cs.type, cs.rb, cs.friction = sh["actor"], sh["body"], sh["type"], sh["rb"], float(sh["friction"])
            for j in range(3):
                cs.size[j] = float(sh["size"][j])
                cs.p[j] = float(sh["p"][j])
            for j in range(9):
                cs.R[j] = float(sh["R"][j // 3][j % 3])
        for i, (a, b) in enumerate(self.pairs):
            m.pairs[i].a, m.pairs[i].b = a, b
        m.ground_friction = self.GROUND_FRICTION
        m.contact_alpha, m.contact_beta, m.friction_beta = self.CONTACT_ALPHA, self.CONTACT_BETA, self.FRICTION_BETA
        substeps = self.substeps()
        hsub = float(self.cfg.dt) / substeps
        m.contact_flags = (0 if self.BOX_PAIR_NORMAL else capi.CONTACT_POINT_NORMALS) | (capi.CONTACT_EXPLICIT_LIGHT if self.EXPLICIT_LIGHT else 0)
        m.contact_ramp_depth = (abs(GRAVITY[2]) * hsub * hsub / self.CONTACT_ALPHA) if self.CONTACT_RAMP_DEPTH is None else float(self.CONTACT_RAMP_DEPTH)
        m.randomize_seed = int(self.randomize_seed)
        if self.robot.dof_mode not in DRIVE_GAINS:
            raise ValueError("Invalid dof_mode")
        m.drive_mode, m.drive_kd, m.drive_kp = DRIVE_GAINS[self.robot.dof_mode]
        m.substeps = substeps
        m.dt = float(self.cfg.dt)
        for j in range(3):
            m.gravity[j] = GRAVITY[j]
        m.nu = self.nu
        return m


def _dev_ptr(t: Optional[torch.Tensor]):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


class IsaacGymWrapper:
    def __init__(self, cfg: IsaacGymConfig, actors: List[str], init_positions: List[List[float]] = None,
                 num_envs: int = 1, viewer: bool = False, device: str = "cuda:0", interactive_goal=True,
                 mppi_config=None, randomize_seed: Optional[int] = None):
        from mppiisaac.utils.isaacgym_utils import load_actor_cfgs, load_asset

        if viewer or getattr(cfg, "viewer", False):
            raise NotImplementedError("the Isaac Gym viewer is graphics and out of scope of the HIP backend")
        if not str(device).startswith(("cuda", "hip")):
            raise capi.MppiHipError(
                f"device '{device}': this backend runs on an AMD GPU only (no CPU pipeline, no fallback)")
        self.env_cfg = load_actor_cfgs(actors)
        self.device = str(device).replace("hip", "cuda")
        robots = [a for a in self.env_cfg if a.type == "robot"]
        if init_positions is not None:
            assert len(robots) == len(init_positions)
            for init_pos, actor_cfg in zip(init_positions, robots):
                actor_cfg.init_pos = list(init_pos)
        for i, a in enumerate(self.env_cfg):
            a.handle = i
        self.cfg = cfg
        self.interactive_goal = interactive_goal
        self.num_envs = int(num_envs)
        self.viewer = None
        self.saved_root_state = None
        self._mppi_config = mppi_config
        self._mppi_config_factory = mppi_config  # kept to rebuild the C config when the actor list changes
        self.generation = 0
        self.scene = Scene(self.env_cfg, cfg, [load_asset(r) for r in robots])
        # The reference draws a different size/mass/friction for every env of noisy box actors (unseeded
        # np.random, :430-475).  Here the K rollout envs draw from a seeded hash of the global sample id; a
        # single env (the K=1 "world") keeps the nominal values unless a seed is passed explicitly.
        self._randomize_seed = randomize_seed if randomize_seed is not None else (0 if self.num_envs > 1 else -1)
        self.scene.randomize_seed = self._randomize_seed
        self.start_sim()

    # ------------------------------------------------------------------ lifetime
    def start_sim(self):
        self._lib = capi.load_library()
        sc = self.scene
        self._c_model = sc.to_c()
        if callable(self._mppi_config):  # built once the scene (viz link, nu) is known
            self._mppi_config = self._mppi_config(sc)
        if self._mppi_config is None:  # plain simulator (e.g. the K=1 "world"): minimal MPPI block
            c = capi.Config()
            c.abi_version = capi.ABI_VERSION
            c.num_samples = c.k_total = self.num_envs
            c.horizon, c.nu, c.n_knots, c.sampling = 1, sc.nu, 1, capi.SAMPLE_EXTERNAL
            c.lambda_, c.rollout_var_discount = 1.0, 1.0
            for j in range(sc.nu):
                c.u_min[j], c.u_max[j], c.noise_sigma_diag[j] = -1e30, 1e30, 1.0
            self._mppi_config = c
        if self._mppi_config.num_samples != self.num_envs:
            raise ValueError("mppi_config.num_samples must equal num_envs")
        if self._mppi_config.nu != sc.nu:
            raise ValueError(f"control dimension mismatch: noise_sigma is {self._mppi_config.nu}x{self._mppi_config.nu}, "
                             f"the actors take {sc.nu} commands")
        dev_index = torch.device(self.device).index or 0
        torch.cuda.set_device(dev_index)
        ctx = ctypes.c_void_p()
        capi.check(self._lib, self._lib.mppi_create(ctypes.byref(self._c_model), ctypes.byref(self._mppi_config),
                                                    dev_index, ctypes.byref(ctx)))
        self._ctx = ctx
        capi.check(self._lib, self._lib.mppi_set_stream(ctx, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
        K, A, B, n = self.num_envs, len(self.env_cfg), sc.n_rb, sc.n_dof
        f32 = dict(dtype=torch.float32, device=self.device)
        # the four gym state tensors (reference :186-199) are refreshed LAZILY: the state of truth lives in the library's
        # sample-minor buffers, and a fused-mode planner never reads these copies - the `_dof_state` ... properties below
        # materialise them on first access after a change (one kernel), so the per-command path of
        # compute_action_tensor is set_state -> rollout with no [K, ...] tensor written
        self._state_t = {"dof": torch.zeros((K, 2 * n), **f32), "root": torch.zeros((K, A, 13), **f32),
                         "rb": torch.zeros((K, B, 13), **f32), "cf": torch.zeros((K, B, 3), **f32)}
        self._state_t_own = self._state_t
        self._state_ptrs = tuple(_dev_ptr(self._state_t[k]) for k in ("dof", "root", "rb", "cf"))
        self._stale, self._needs_reset = True, False
        self._visualize_link_present = sc.viz_link_index() >= 0
        self.visualize_link_buffer = []
        if self._visualize_link_present:
            self.robot_rigid_body_viz_idx = sc.rigid_body_index(sc.robot.name, sc.robot.visualize_link)
        self.robot_indices = torch.tensor([i for i, a in enumerate(self.env_cfg) if a.type == "robot"], device=self.device)
        self.obstacle_indices = torch.tensor(
            [i for i, a in enumerate(self.env_cfg) if (a.type in ["sphere", "box"] and a.name != "dummy")], device=self.device)
        self._pending_cmd = None
        self._pending_host = None
        # The K = 1 "world" (reference examples/<x>/world.py) hands its state to the planner as torch.save blobs every control
        # iteration: the library mirrors dof / root into mapped host memory right behind the kernel that materialises them, and
        # `torch_to_bytes(sim._dof_state)` takes its payload from there - no device-to-host copy, no synchronise (utils/transport.py)
        self._mirror = None
        if self.num_envs == 1:
            from mppiisaac.utils.transport import register_host_mirror
            self._mirror = {"dof": np.zeros(2 * n, np.float32), "root": np.zeros(13 * A, np.float32), "fresh": False, "versions": None}
            # (the registry is module-level: it holds the wrapper through a WEAK reference, or `del world` would never reach __del__
            # and the HIP context, device buffers and pinned host blocks of every K = 1 wrapper would live as long as the process)
            import weakref
            me = weakref.ref(self)

            def mirrored(which, me=me):
                w = me()
                return None if w is None else w._mirrored(which)
            register_host_mirror(self._state_t["dof"], lambda: mirrored("dof"))
            register_host_mirror(self._state_t["root"], lambda: mirrored("root"))
        self.reset_to_initial_poses()

    def stop_sim(self):
        if getattr(self, "_mirror", None) is not None:
            from mppiisaac.utils.transport import unregister_host_mirror
            for k in ("dof", "root"):
                unregister_host_mirror(self._state_t[k])
            self._mirror = None
        if getattr(self, "_ctx", None):
            self._lib.mppi_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.stop_sim()
        except Exception:
            pass

    # ------------------------------------------------------------------ state in / out
    def _push_single_state(self, dof: np.ndarray, root: np.ndarray):
        """one env state -> x0 of the HIP context, broadcast to all K envs."""
        dof = np.ascontiguousarray(dof, np.float32).reshape(-1)
        root = np.ascontiguousarray(root, np.float32).reshape(-1)
        capi.check(self._lib, self._lib.mppi_set_state(self._ctx, capi.fptr(dof), capi.fptr(root)))
        if self.num_envs == 1:   # the K=1 world is also stepped through the C-ABI directly (closed loops on the device)
            capi.check(self._lib, self._lib.mppi_sim_reset(self._ctx))
        else:                    # K rollout envs: broadcast x0 only when somebody steps or reads them
            self._needs_reset = True
        self._stale = True

    def _reset_envs_if_needed(self):
        if self._needs_reset:
            capi.check(self._lib, self._lib.mppi_sim_reset(self._ctx))
            self._needs_reset = False

    def _materialise(self):
        self._reset_envs_if_needed()
        t = self._state_t
        m = self._mirror
        if m is not None and t is self._state_t_own:   # K = 1 world: the same kernel mirrors dof / root into mapped host memory
            capi.check(self._lib, self._lib.mppi_sim_materialise_mirror(self._ctx, *self._state_ptrs))
            m["fresh"], m["versions"] = False, (t["dof"]._version, t["root"]._version)
        else:
            capi.check(self._lib, self._lib.mppi_sim_materialise(self._ctx, _dev_ptr(t["dof"]), _dev_ptr(t["root"]), _dev_ptr(t["rb"]), _dev_ptr(t["cf"])))
        self._stale = False

    def _mirrored(self, key):
        """host copy of the K = 1 world's dof / root state tensor as the last materialise left it (None: the tensor has been
        written to since - the caller copies it from the device instead)"""
        m, t = self._mirror, self._state_t_own
        if m is None or m["versions"] != (t["dof"]._version, t["root"]._version):
            return None
        if not m["fresh"]:
            capi.check(self._lib, self._lib.mppi_mirror_wait(self._ctx, capi.fptr(m["dof"]), capi.fptr(m["root"])))
            m["fresh"] = True
        return m[key]

    def _fresh(self, key):
        if self._stale:
            self._materialise()
        if self._view_lazy is not None:   # horizon view of the planner: a state tensor is produced when somebody reads it
            self._view_lazy(key)
        return self._state_t[key]

    _view_lazy = None
    _view_link = None

    @contextlib.contextmanager
    def _horizon_view(self, tensors: dict, n_rows: int, lazy=None, link=None):
        """the four state tensors replaced by [H*K, ...] blocks (row block t = the envs after horizon step t) and num_envs by
        H*K, for ONE compute_cost call over a whole horizon (planner/mppi.py: _horizon_batched).  `lazy(key)`: called before a
        tensor is handed out - the planner materialises only the tensors an Objective actually reads; `link(rb_index)`: dense
        [n_rows, 13] rows of ONE rigid body (or None) - what get_actor_link_by_name hands out instead of a slice of the whole
        rigid-body tensor"""
        saved = (self._state_t, self.num_envs, self._stale)
        self._state_t, self.num_envs, self._stale, self._view_lazy, self._view_link = tensors, int(n_rows), False, lazy, link
        try:
            yield self
        finally:
            self._state_t, self.num_envs, self._stale, self._view_lazy, self._view_link = saved[0], saved[1], True, None, None

    _dof_state = property(lambda self: self._fresh("dof"))
    _root_state = property(lambda self: self._fresh("root"))
    _rigid_body_state = property(lambda self: self._fresh("rb"))
    _net_contact_force = property(lambda self: self._fresh("cf"))

    def _rigid_body_rows(self, rb_index: int):
        """[num_envs, 13] rows of one rigid body: a slice of the rigid-body tensor, or - over the planner's horizon view - the dense
        rows of just that body (the whole [H*K, B, 13] tensor is then never produced)"""
        if self._view_link is not None:
            rows = self._view_link(rb_index)
            if rows is not None:
                return rows
        return self._rigid_body_state[:, rb_index, :]

    @property
    def visualize_link_pos(self):
        return self._rigid_body_rows(self.robot_rigid_body_viz_idx)[:, 0:3]

    def reset_to_initial_poses(self):
        dof, root = self.scene.initial_state()
        self._push_single_state(dof, root)

    def set_state_from_env0(self, dof_state: torch.Tensor, root_state: torch.Tensor):
        """reset_rollout_sim path (reference mppi_isaac.py:87-99): a [1,2n] / [1,A,13] world state is
        broadcast to every env.  Only the single state crosses to the device."""
        if isinstance(dof_state, torch.Tensor):
            dof_state, root_state = dof_state.detach().cpu().numpy(), root_state.detach().cpu().numpy()
        self._push_single_state(dof_state.reshape(-1)[: 2 * self.scene.n_dof], root_state.reshape(-1)[: 13 * len(self.env_cfg)])

    def reset_robot_state(self, q, qdot):
        """reference :574-619 (urdfenvs compatibility): q, qdot lists -> interleaved DOF state in all envs."""
        root = (self.saved_root_state if self.saved_root_state is not None else self._root_state)[0].cpu().numpy().copy()
        robot = self.scene.robot
        if robot.differential_drive:
            # q = (x, y, yaw, <arm joints>): the base pose goes to the root state, wheels to 0.  The reference's
            # branch (:596-604 -> set_state_tensor_by_pos_vel :677-693) writes a misspelled attribute and raises
            # (SURVEY.md C); this is its intended behaviour.
            pos, vel = list(q[:3]), list(qdot[:3])
            yaw = float(pos[2])
            ri = self.scene.robot_idx
            root[ri, 0:2] = pos[:2]
            root[ri, 3:7] = [0.0, 0.0, np.sin(yaw / 2), np.cos(yaw / 2)]
            root[ri, 7:10] = [vel[0], vel[1], 0.0]
            root[ri, 10:13] = [0.0, 0.0, vel[2]]
            n_wheels = int(robot.wheel_count)
            q = list(q[3:]) + [0.0] * n_wheels
            qdot = list(qdot[3:]) + [0.0] * n_wheels
        dof = interleave_dof_state(q, qdot, self.scene.n_dof)
        self._push_single_state(dof, root)

    def set_state_tensor_by_pos_vel(self, handle, pos, vel):
        """planar pose (x, y, yaw) and velocity of one actor into its root row (reference :677-693, which writes a
        misspelled attribute and raises; this is its intended behaviour)"""
        yaw = float(pos[2])
        root = self._root_state[0].cpu().numpy().copy()
        i = self._as_index(handle)
        root[i, 0:2] = [float(pos[0]), float(pos[1])]
        root[i, 3:7] = [0.0, 0.0, np.sin(yaw / 2), np.cos(yaw / 2)]
        root[i, 7:10] = [float(v) for v in vel][:3]
        self._push_single_state(self._dof_state[0].cpu().numpy(), root)

    def _ik(self, actor, u):
        """(v, omega) -> (left, right) wheel velocities (reference :510-522)"""
        return diff_drive_ik(actor, u)

    def save_root_state(self):
        self.saved_root_state = self._root_state.clone()

    def get_saved_root_state(self):
        return self.saved_root_state

    def reset_root_state(self):
        if self._visualize_link_present:
            self.visualize_link_buffer = []
        if self.saved_root_state is not None:
            self._push_single_state(self._dof_state[0].cpu().numpy(), self.saved_root_state[0].cpu().numpy())

    # ------------------------------------------------------------------ stepping
    def apply_robot_cmd(self, u_desired: torch.Tensor):
        """Latch the command for the next step().  [nu] applies to all envs, [K,nu] per env.  The
        scatter to DOF targets incl. the diff-drive map happens in the step kernel (reference :524-572)."""
        if not (isinstance(u_desired, torch.Tensor) and u_desired.is_cuda):
            # ONE command that lives on the host (what the world loop of the reference's examples applies: the planner's action as it
            # came off the wire, examples/<x>/world.py:42): it stays there - step() hands it to the step kernel through the
            # context's mapped host block instead of a host-to-device copy
            h = np.ascontiguousarray(u_desired.detach().numpy() if isinstance(u_desired, torch.Tensor) else u_desired, dtype=np.float32)
            if h.size == self.scene.nu and h.ndim <= 2:
                self._pending_host, self._pending_cmd = h.reshape(-1), None
                return
        u = torch.as_tensor(u_desired, dtype=torch.float32, device=self.device)
        if u.dim() == 1:
            u = u.unsqueeze(0)
        if u.shape[-1] != self.scene.nu:
            raise ValueError(f"command has {u.shape[-1]} entries, expected {self.scene.nu}")
        self._pending_cmd, self._pending_host = u.contiguous(), None

    # per-DOF targets set directly (reference :402-406; used by examples/*/tuning.py and examples/anymal/world.py).  The step
    # kernels take COMMANDS (nu columns, scattered by the command map), so these are available where one command drives
    # one DOF with unit gain - every robot but the differential-drive bases.
    def _latch_dof_targets(self, u, mode: str):
        if self.scene.robot.dof_mode != mode:
            raise ValueError(f"the robot is driven in '{self.scene.robot.dof_mode}' mode, not '{mode}'")
        if self.scene.nu != self.scene.n_dof or any(t[0][1] != 1.0 or t[1][1] != 0.0 for t in self.scene.cmd_terms):
            raise NotImplementedError("per-DOF targets bypass the differential-drive command map, which the step kernels apply; "
                                      "use apply_robot_cmd((v, omega, ...))")
        self.apply_robot_cmd(u)

    def set_dof_velocity_target_tensor(self, u):
        self._latch_dof_targets(u, "velocity")

    def set_dof_actuation_force_tensor(self, u):
        self._latch_dof_targets(u, "effort")

    def step(self):
        if self._pending_host is not None:   # one command for every env, still on the host
            self._reset_envs_if_needed()
            capi.check(self._lib, self._lib.mppi_sim_step_host(self._ctx, capi.fptr(self._pending_host)))
            self._stale = True
            if self._visualize_link_present:
                self.visualize_link_buffer.append(self.visualize_link_pos.clone())
            return
        u = self._pending_cmd
        if u is None:
            u = torch.zeros((1, self.scene.nu), dtype=torch.float32, device=self.device)
        shared = 1 if u.shape[0] == 1 and self.num_envs != 1 else 0
        if not shared and u.shape[0] != self.num_envs:
            raise ValueError("command batch does not match num_envs")
        self._reset_envs_if_needed()
        capi.check(self._lib, self._lib.mppi_sim_step(self._ctx, _dev_ptr(u), shared))
        self._stale = True
        if self._visualize_link_present:
            self.visualize_link_buffer.append(self.visualize_link_pos.clone())

    # ------------------------------------------------------------------ getters (reference :268-356)
    @property
    def substeps_integrated(self) -> int:
        """integration steps per control interval actually taken (Scene.substeps: cfg.substeps, multiplied up in contact scenes so that
        a step is at most MAX_CONTACT_SUBSTEP) - what bench.py reports next to the configured value"""
        return int(self.scene.substeps())

    @property
    def num_robots(self):
        return len(self.robot_indices)

    @property
    def robot_positions(self):
        return torch.index_select(self._root_state, 1, self.robot_indices)[:, :, 0:3]

    @property
    def robot_velocities(self):
        return torch.index_select(self._root_state, 1, self.robot_indices)[:, :, 7:10]

    @property
    def obstacle_positions(self):
        return torch.index_select(self._root_state, 1, self.obstacle_indices)[:, :, 0:3]

    @property
    def ostacle_velocities(self):  # (sic: the reference's spelling, isaacgym_wrapper.py:288)
        return torch.index_select(self._root_state, 1, self.obstacle_indices)[:, :, 7:10]

    obstacle_velocities = ostacle_velocities

    def _get_actor_index_by_name(self, name: str):
        return self.scene.actor_index(name)

    def _get_actor_index_by_robot_index(self, robot_idx: int):
        return int(self.robot_indices[robot_idx])

    # by-index getters of the reference (:297-330); indices may be ints or 0-d / 1-element tensors as there
    @staticmethod
    def _as_index(idx) -> int:
        return int(torch.as_tensor(idx).reshape(-1)[0])

    def get_actor_position_by_actor_index(self, actor_idx):
        return self._root_state[:, self._as_index(actor_idx), 0:3]

    def get_actor_position_by_robot_index(self, robot_idx: int):
        return self.get_actor_position_by_actor_index(self._get_actor_index_by_robot_index(robot_idx))

    def get_actor_velocity_by_actor_index(self, idx):
        return self._root_state[:, self._as_index(idx), 7:10]

    def get_actor_velocity_by_robot_index(self, robot_idx: int):
        return self.get_actor_velocity_by_actor_index(self._get_actor_index_by_robot_index(robot_idx))

    def get_actor_orientation_by_actor_index(self, idx):
        return self._root_state[:, self._as_index(idx), 3:7]

    def get_actor_orientation_by_robot_index(self, robot_idx: int):
        return self.get_actor_orientation_by_actor_index(self._get_actor_index_by_robot_index(robot_idx))

    def get_rigid_body_by_rigid_body_index(self, rigid_body_idx):
        return self._rigid_body_rows(self._as_index(rigid_body_idx))

    def get_actor_position_by_name(self, name: str):
        return self._root_state[:, self.scene.actor_index(name), 0:3]

    def get_actor_velocity_by_name(self, name: str):
        return self._root_state[:, self.scene.actor_index(name), 7:10]

    def get_actor_orientation_by_name(self, name: str):
        return self._root_state[:, self.scene.actor_index(name), 3:7]

    def get_actor_link_by_name(self, actor_name: str, link_name: str):
        return self._rigid_body_rows(self.scene.rigid_body_index(actor_name, link_name))

    def get_actor_contact_forces_by_name(self, actor_name: str, link_name: str):
        return self._net_contact_force[:, self.scene.rigid_body_index(actor_name, link_name)]

    def get_dof_state(self):
        return self._dof_state

    # ------------------------------------------------------------------ env mutation (reference :423-427, :695-758)
    def _restart(self):
        """reference add_to_envs / obstacle-size change: stop_sim(); start_sim() (there broken by a misspelled
        attribute).  The HIP context is rebuilt for the new actor list; `generation` lets owners of
        the old context (the MPPI driver) notice."""
        from mppiisaac.utils.isaacgym_utils import load_asset
        keep_root = self._root_state[0].clone() if hasattr(self, "_state_t") else None
        keep_dof = self._dof_state[0].clone() if hasattr(self, "_state_t") else None
        n_old = keep_root.shape[0] if keep_root is not None else 0
        # the nominal control sequence lives in the context that is about to be destroyed: carry it over here
        # (nobody may touch the old handle afterwards - in the reference the mppi object simply survives)
        keep_U = None
        if getattr(self, "_ctx", None) and not callable(self._mppi_config) and self._mppi_config is not None:
            keep_U = np.zeros((self._mppi_config.horizon, self._mppi_config.nu), np.float32)
            capi.check(self._lib, self._lib.mppi_get_nominal(self._ctx, capi.fptr(keep_U)))
        self.stop_sim()
        for i, a in enumerate(self.env_cfg):
            a.handle = i
        robots = [a for a in self.env_cfg if a.type == "robot"]
        self.scene = Scene(self.env_cfg, self.cfg, [load_asset(r) for r in robots])
        self.scene.randomize_seed = self._randomize_seed
        self._mppi_config = self._mppi_config_factory
        self.generation += 1
        self.start_sim()
        if keep_U is not None and keep_U.shape == (self._mppi_config.horizon, self._mppi_config.nu):
            capi.check(self._lib, self._lib.mppi_set_nominal(self._ctx, capi.fptr(keep_U)))
        if keep_root is not None:  # carry the state of the actors that already existed
            root = self._root_state[0].clone()
            root[:n_old] = keep_root
            self._push_single_state(keep_dof.cpu().numpy(), root.cpu().numpy())

    def add_to_envs(self, additions):
        for a in additions:
            self.env_cfg.append(ActorWrapper(**a))
        self._restart()

    def update_root_state_tensor_by_obstacles(self, obstacles):
        """obstacles: dict name -> {position, velocity, size[, type]} (urdfenvs FullSensor layout, reference
        :695-742).  Obstacle i is the fixed sphere actor 'sphere<i>'; unknown ones are added (simulator restart),
        a changed radius restarts too, otherwise only the root rows move."""
        changed = False
        updates = []
        for i, obst in enumerate(list(obstacles.values())):
            name = f"sphere{i}"
            size = list(np.atleast_1d(obst["size"]).astype(float))
            idx = [k for k, a in enumerate(self.env_cfg) if a.name == name]
            if not idx:
                self.env_cfg.append(ActorWrapper(**{"type": "sphere", "name": name, "handle": None, "size": size, "fixed": True,
                                                    "init_pos": [float(v) for v in obst["position"]]}))
                changed = True
                continue
            if not all(a == b for a, b in zip(size, self.env_cfg[idx[0]].size)):
                self.env_cfg[idx[0]].size = size
                changed = True
            updates.append((idx[0], obst))
        if changed:
            self._restart()
        root = self._root_state[0].clone()
        for k, obst in updates:
            root[k] = torch.tensor([*obst["position"], 0, 0, 0, 1, *obst["velocity"], 0, 0, 0], dtype=torch.float32, device=self.device)
        self._push_single_state(self._dof_state[0].cpu().numpy(), root.cpu().numpy())
        return changed

    def update_root_state_tensor_by_obstacles_tensor(self, obst_tensor):
        """reference :744-758: each row of obst_tensor [n,13] overwrites the root state of the first
        non-robot, non-fixed actor."""
        idx = [k for k, a in enumerate(self.env_cfg) if (a.type != "robot" and not a.fixed)]
        if not idx:
            raise ValueError("no free (non-robot, non-fixed) actor to place")
        root = self._root_state[0].clone()
        for o in torch.as_tensor(obst_tensor, dtype=torch.float32, device=self.device).reshape(-1, 13):
            root[idx[0]] = o
        self._push_single_state(self._dof_state[0].cpu().numpy(), root.cpu().numpy())

    # setters of the reference (:359-397): the root row of one actor in every env; takes effect for the next rollout / step
    def _set_root_columns(self, actor_idx, lo: int, hi: int, value) -> None:
        if len(self.scene.robot_ids) > 1 and self._as_index(actor_idx) in self.scene.robot_ids and self.scene.robot.fixed:
            raise NotImplementedError("the fixed bases of several robots in one env are part of the compiled forest: place them "
                                      "with initial_actor_positions")
        root = self._root_state[0].clone()
        root[self._as_index(actor_idx), lo:hi] = torch.as_tensor(value, dtype=torch.float32, device=self.device).reshape(-1)[: hi - lo]
        self._push_single_state(self._dof_state[0].cpu().numpy(), root.cpu().numpy())

    def set_actor_position_by_actor_index(self, position, actor_idx) -> None:
        self._set_root_columns(actor_idx, 0, 3, position)

    def set_actor_position_by_name(self, position, name: str) -> None:
        """Move an actor (e.g. the goal) in every env; takes effect for the next rollout/step."""
        self.set_actor_position_by_actor_index(position, self.scene.actor_index(name))

    def set_actor_position_by_robot_index(self, position, robot_idx) -> None:
        self.set_actor_position_by_actor_index(position, self._get_actor_index_by_robot_index(robot_idx))

    def set_actor_velocity_by_actor_index(self, velocity, actor_idx) -> None:
        self._set_root_columns(actor_idx, 7, 10, velocity)

    def set_actor_velocity_by_name(self, velocity, name: str) -> None:
        self.set_actor_velocity_by_actor_index(velocity, self.scene.actor_index(name))

    def set_actor_velocity_by_robot_index(self, velocity, robot_idx) -> None:
        self.set_actor_velocity_by_actor_index(velocity, self._get_actor_index_by_robot_index(robot_idx))

    def set_root_state_tensor_by_actor_idx(self, state_tensor, idx) -> None:
        """whole 13-float root row (pos, quat xyzw, linvel, angvel) of one actor (reference :662-667)"""
        self._set_root_columns(idx, 0, 13, state_tensor)

    def set_actor_dof_state(self, state) -> None:
        """`dof_mode: position` path of the reference (:399-400): overwrite the interleaved (q, qd) DOF state"""
        st = torch.as_tensor(state, dtype=torch.float32).reshape(-1)[: 2 * self.scene.n_dof]
        self._push_single_state(st.cpu().numpy(), self._root_state[0].cpu().numpy())

    def draw_lines(self, lines) -> None:
        """viewer call of the reference's world scripts: there is no viewer here; accepted and ignored"""

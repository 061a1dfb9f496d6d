#!/usr/bin/env python3
"""Generate tests/golden/*.json by importing the reference's pure-Python boundary logic with
`isaacgym` mocked (the reference's own docs/source/conf.py:1-6 does the same).  Runs ONLY in the
build container where /root/reference exists; the vectors (data, not source) are committed.
Covers SURVEY.md section 8c items 1-6."""
import json, os, sys, io
from unittest.mock import MagicMock
import numpy as np
import torch

REF = os.environ.get("MPPI_REFERENCE", "/root/reference")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")
sys.modules["isaacgym"] = MagicMock()
sys.path.insert(0, REF)
from mppiisaac.planner.isaacgym_wrapper import ActorWrapper, IsaacGymWrapper, IsaacGymConfig  # noqa
from mppiisaac.utils.isaacgym_utils import load_actor_cfgs  # noqa
from mppiisaac.utils.conversions import quaternion_to_yaw  # noqa
from mppiisaac.utils.transport import torch_to_bytes, bytes_to_torch  # noqa
import dataclasses

def dump(name, obj):
    with open(os.path.join(OUT, name), "w") as f:
        json.dump(obj, f, indent=1, sort_keys=True)
        f.write("\n")
    print("wrote", name)

# 1. actor YAML -> ActorWrapper field dicts (all conf/actors/*.yaml)
names = sorted(os.path.splitext(f)[0] for f in os.listdir(os.path.join(REF, "conf", "actors")))
actors = {}
for n in names:
    try:
        a = load_actor_cfgs([n])[0]
        actors[n] = dataclasses.asdict(a)
    except Exception as e:  # record failures too
        actors[n] = {"__error__": type(e).__name__}
dump("actor_cfgs.json", actors)
dump("isaacgym_config_defaults.json", dataclasses.asdict(IsaacGymConfig()))

# a wrapper instance without running __init__ (no simulator needed for the boundary logic)
def fake_wrapper(actor_names, dof_dicts, K):
    w = IsaacGymWrapper.__new__(IsaacGymWrapper)
    w.env_cfg = load_actor_cfgs(actor_names)
    w.device = "cpu"; w.num_envs = K
    w.envs = [object()]
    robots = [a for a in w.env_cfg if a.type == "robot"]
    for i, a in enumerate(robots):
        a.handle = i
    gym = MagicMock()
    gym.get_actor_dof_count = lambda env, h: len(dof_dicts[h])
    gym.get_actor_dof_dict = lambda env, h: dof_dicts[h]
    w._gym = gym
    ndof = sum(len(d) for d in dof_dicts)
    w._dof_state = torch.zeros(K, 2 * ndof)
    w._root_state = torch.zeros(K, len(w.env_cfg), 13)
    w._sim = None
    captured = {}
    w.set_dof_velocity_target_tensor = lambda u: captured.__setitem__("velocity", u.clone())
    w.set_dof_actuation_force_tensor = lambda u: captured.__setitem__("effort", u.clone())
    w.set_actor_dof_state = lambda u: captured.__setitem__("position", u.clone())
    return w, captured

# 2. diff-drive _ik grid
boxer = load_actor_cfgs(["boxer"])[0]
w = IsaacGymWrapper.__new__(IsaacGymWrapper)
grid = [[v, om] for v in (-1.2, -0.3, 0.0, 0.2, 1.2) for om in (-3.5, -1.0, 0.0, 1.0, 3.5)]
u = torch.tensor(grid, dtype=torch.float32)
l, r = w._ik(boxer, u)
dump("diff_drive_ik.json", {"wheel_radius": boxer.wheel_radius, "wheel_base": boxer.wheel_base,
                            "u": grid, "left": l.tolist(), "right": r.tolist()})

# 3. apply_robot_cmd scatter
torch.manual_seed(0)
cases = {}
w, cap = fake_wrapper(["boxer", "block", "goal"], [{"wheel_right_joint": 0, "wheel_left_joint": 1}], 4)
u = torch.tensor([[0.2, 0.0], [0.0, 1.0], [1.2, -3.5], [-0.3, 0.7]])
w.apply_robot_cmd(u)
cases["boxer"] = {"actors": ["boxer", "block", "goal"], "dof_names": ["wheel_right_joint", "wheel_left_joint"],
                  "u": u.tolist(), "mode": list(cap.keys())[0], "dof_cmd": list(cap.values())[0].tolist()}
pj = {f"panda_joint{i+1}": i for i in range(7)}
w, cap = fake_wrapper(["panda_stick", "goal"], [pj], 3)
u = torch.randn(3, 7)
w.apply_robot_cmd(u)
cases["panda_stick"] = {"actors": ["panda_stick", "goal"], "dof_names": list(pj), "u": u.tolist(),
                        "mode": list(cap.keys())[0], "dof_cmd": list(cap.values())[0].tolist()}
pg = dict(pj); pg["panda_finger_joint1"] = 7; pg["panda_finger_joint2"] = 8
w, cap = fake_wrapper(["panda_gripper", "goal"], [pg], 3)
u = torch.randn(3, 9)
w.apply_robot_cmd(u)
cases["panda_gripper"] = {"actors": ["panda_gripper", "goal"], "dof_names": list(pg), "u": u.tolist(),
                          "mode": list(cap.keys())[0], "dof_cmd": list(cap.values())[0].tolist()}
pr = {"mobile_joint_x": 0, "mobile_joint_y": 1, "mobile_joint_theta": 2}
w, cap = fake_wrapper(["point_robot", "goal"], [pr], 2)
u = torch.randn(7)[:3]  # 1-D input gets unsqueezed
w.apply_robot_cmd(u)
cases["point_robot_1d"] = {"actors": ["point_robot", "goal"], "dof_names": list(pr), "u": u.tolist(),
                           "mode": list(cap.keys())[0], "dof_cmd": list(cap.values())[0].tolist()}
dump("apply_robot_cmd.json", cases)

# 4. reset_robot_state interleave (non diff-drive) + diff-drive branch behaviour
w, cap = fake_wrapper(["panda_stick", "goal"], [pj], 3)
q = [0.1 * i for i in range(7)]; qd = [-0.01 * i for i in range(7)]
w.reset_robot_state(q, qd)
rrs = {"panda_stick": {"q": q, "qdot": qd, "K": 3, "dof_state": cap["position"].tolist()}}
w, cap = fake_wrapper(["boxer", "goal"], [{"wheel_right_joint": 0, "wheel_left_joint": 1}], 2)
try:
    w.reset_robot_state([1.0, 2.0, 0.5], [0.1, 0.2, 0.3])
    rrs["boxer"] = {"raised": None}
except Exception as e:
    rrs["boxer"] = {"raised": type(e).__name__}
yaw = 0.5
rrs["boxer"]["intended_quat_xyzw_for_yaw_0.5"] = [0.0, 0.0, float(np.sin(yaw / 2)), float(np.cos(yaw / 2))]
dump("reset_robot_state.json", rrs)

# 5. quaternion_to_yaw
g = torch.Generator().manual_seed(1)
quat = torch.randn(16, 4, generator=g)
quat = quat / quat.norm(dim=1, keepdim=True)
quat = torch.cat([quat, torch.tensor([[0, 0, 0.3827, 0.9239], [0, 0, 0, 1.0]])])
dump("quaternion_to_yaw.json", {"quat_xyzw": quat.tolist(), "yaw": quaternion_to_yaw(quat).tolist()})

# 6. transport round trip
t = torch.arange(6, dtype=torch.float32).reshape(2, 3)
b = torch_to_bytes(t)
dump("transport.json", {"tensor": t.tolist(), "nbytes": len(b), "magic": list(b[:4]),
                        "roundtrip": bytes_to_torch(b).tolist()})

# 6b. MPPI parameter files: the VALUES of every reference conf/mppi/*.yaml (data; tools/make_conf.py restates them)
import glob as _glob
import yaml as _yaml
mppi_cfgs = {}
for path in sorted(_glob.glob(os.path.join(REF, "conf", "mppi", "*.yaml"))):
    raw = _yaml.safe_load(open(path))
    raw.pop("defaults", None)
    mppi_cfgs[os.path.splitext(os.path.basename(path))[0]] = raw
dump("mppi_cfgs.json", mppi_cfgs)

# 7. Objective.compute_cost of the reference's example planners on seeded random simulator states.
#    hydra / zerorpc / mppi_torch are mocked (absent, unused by compute_cost).  pytorch3d is absent too: the two
#    functions the panda objectives call are served by scipy.spatial.transform (an independent implementation of
#    the same conventions: real-first quaternion -> matrix, intrinsic "ZYX" angles); those cases are labelled.
import importlib.util
import types
from scipy.spatial.transform import Rotation

for name in ("hydra", "hydra.core", "hydra.core.config_store", "omegaconf", "zerorpc", "mppi_torch", "mppi_torch.mppi"):
    sys.modules.setdefault(name, MagicMock())

def _q2m(q):
    a = q.detach().double().numpy()
    return torch.tensor(Rotation.from_quat(np.concatenate([a[:, 1:4], a[:, 0:1]], 1)).as_matrix())

def _m2e(M, convention):
    return torch.tensor(Rotation.from_matrix(M.detach().double().numpy()).as_euler(convention))

p3d = types.ModuleType("pytorch3d")
p3d.transforms = types.ModuleType("pytorch3d.transforms")
p3d.transforms.quaternion_to_matrix = _q2m
p3d.transforms.matrix_to_euler_angles = _m2e
sys.modules["pytorch3d"] = p3d
sys.modules["pytorch3d.transforms"] = p3d.transforms

class RecordingSim:
    """answers every getter with a seeded random tensor of the reference's shape and remembers what it returned"""
    def __init__(self, K, seed):
        self.K, self.g, self.calls = K, torch.Generator().manual_seed(seed), {}
    def _rand(self, n):
        return torch.randn(self.K, n, generator=self.g, dtype=torch.float64)
    def _answer(self, key, make):
        if key not in self.calls:
            self.calls[key] = make()
        return self.calls[key]
    def get_actor_link_by_name(self, actor_name, link_name):
        def make():
            x = self._rand(13)
            x[:, 3:7] = x[:, 3:7] / x[:, 3:7].norm(dim=1, keepdim=True)
            return x
        return self._answer(f"link:{actor_name}:{link_name}", make)
    def get_actor_position_by_name(self, name):
        return self._answer(f"position:{name}", lambda: self._rand(3))
    def get_actor_velocity_by_name(self, name):
        return self._answer(f"velocity:{name}", lambda: self._rand(3))
    def get_actor_orientation_by_name(self, name):
        def make():
            x = self._rand(4)
            return x / x.norm(dim=1, keepdim=True)
        return self._answer(f"orientation:{name}", make)
    def get_actor_contact_forces_by_name(self, actor_name, link_name):
        # contact is sparse: zero rows in half of the samples
        return self._answer(f"contact:{actor_name}:{link_name}", lambda: self._rand(3) * (self._rand(1) > 0))
    def get_dof_state(self):
        # interleaved (q, qdot) of a 12-DOF robot (omnipanda: 3 base + 7 arm + 2 fingers)
        return self._answer("dof_state", lambda: self._rand(24))

obj_cases = {}
class _Cfg:  # the one thing an example Objective reads from its config (omni_panda_pick: cfg.mppi.device)
    class mppi:
        device = "cpu"

P3D = ["pytorch3d.transforms -> scipy"]
for case, rel, stand_ins in (("panda", "examples/panda/planner.py", P3D),
                             ("boxer_push", "examples/boxer_push/planner.py", []),
                             ("panda_pick", "examples/panda_pick/planner.py", P3D),
                             ("boxer_reach", "examples/boxer_reach/planner.py", []),
                             ("heijn_reach", "examples/heijn_reach/planner.py", []),
                             ("heijn_push", "examples/heijn_push/planner.py", []),
                             ("albert", "examples/albert/planner.py", P3D),
                             ("omni_panda_pick", "examples/omni_panda_pick/planner.py", P3D),
                             ("panda_effort", "examples/panda_effort/planner.py", P3D),
                             ("panda_stick_push", "examples/panda_stick_push/planner.py", P3D),
                             ("anymal", "examples/anymal/planner.py", [])):
    spec = importlib.util.spec_from_file_location(f"ref_example_{case}", os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    obj = mod.Objective(_Cfg)
    sim = RecordingSim(K=24, seed=11 + len(obj_cases))
    cost = obj.compute_cost(sim)
    obj_cases[case] = {"source": rel, "stand_ins": stand_ins, "weights": {k: float(v) for k, v in getattr(obj, "weights", {}).items()},
                       "goal_yaw": float(getattr(obj, "goal_yaw", 0.0)),
                       "inputs": {k: v.tolist() for k, v in sim.calls.items()}, "cost": cost.tolist()}
dump("objective_costs.json", obj_cases)

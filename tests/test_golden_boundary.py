"""Boundary logic against the golden vectors captured from the mock-imported reference
(tools/make_golden.py).  CPU only."""
import dataclasses
import json
import os

import numpy as np
import pytest
import torch

from mppiisaac.planner.isaacgym_wrapper import (ActorWrapper, IsaacGymConfig, Scene, diff_drive_ik, interleave_dof_state)
from mppiisaac.utils.conversions import quaternion_to_yaw
from mppiisaac.utils.isaacgym_utils import load_actor_cfgs, load_asset
from mppiisaac.utils.transport import bytes_to_torch, torch_to_bytes

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def gold(name):
    with open(os.path.join(GOLD, name)) as f:
        return json.load(f)


def test_actor_cfgs_match_reference_loader():
    g = gold("actor_cfgs.json")
    assert len(g) == 24
    for name, fields in g.items():
        ours = dataclasses.asdict(load_actor_cfgs([name])[0])
        assert ours == fields, name


def test_isaacgym_config_defaults():
    assert dataclasses.asdict(IsaacGymConfig()) == gold("isaacgym_config_defaults.json")


def test_diff_drive_ik_bitwise():
    g = gold("diff_drive_ik.json")
    boxer = load_actor_cfgs(["boxer"])[0]
    assert boxer.wheel_radius == g["wheel_radius"] and boxer.wheel_base == g["wheel_base"]
    l, r = diff_drive_ik(boxer, torch.tensor(g["u"], dtype=torch.float32))
    assert l.tolist() == g["left"] and r.tolist() == g["right"]  # same float32 expression -> bit-exact


@pytest.mark.parametrize("case", ["boxer", "panda_stick", "panda_gripper", "point_robot_1d"])
def test_apply_robot_cmd_scatter(case, oracle32, oracle64):
    g = gold("apply_robot_cmd.json")[case]
    env_cfg = load_actor_cfgs(g["actors"])
    robot = [a for a in env_cfg if a.type == "robot"][0]
    scene = Scene(env_cfg, IsaacGymConfig(), load_asset(robot))
    assert scene.dof_names == g["dof_names"]  # DOF order = URDF depth-first joint order
    assert g["mode"] == robot.dof_mode
    u = np.atleast_2d(np.asarray(g["u"], np.float32))
    want = np.atleast_2d(np.asarray(g["dof_cmd"], np.float32))
    m = scene.to_c()
    assert m.nu == u.shape[1]
    for row, w in zip(u, want):
        got64 = oracle64.cmd_map(m, row)
        # the reference evaluates (v/r) -+ (L*w)/(2r) in float32; the folded linear map differs by rounding only
        np.testing.assert_allclose(got64, w, rtol=2e-6, atol=1e-6)
        np.testing.assert_allclose(oracle32.cmd_map(m, row), w, rtol=2e-6, atol=1e-6)
    if case == "point_robot_1d":  # 1-D command broadcast to every env (reference :525-526)
        assert want.shape[0] == 2 and np.array_equal(want[0], want[1])


def test_reset_robot_state_interleave():
    g = gold("reset_robot_state.json")
    p = g["panda_stick"]
    row = interleave_dof_state(p["q"], p["qdot"], 7)
    want = np.asarray(p["dof_state"], np.float32)
    assert want.shape == (p["K"], 14)
    for k in range(p["K"]):
        assert np.array_equal(row, want[k])
    # the reference's differential-drive branch raises (defect catalogue); ours refuses loudly too
    assert g["boxer"]["raised"] == "AttributeError"


def test_quaternion_to_yaw():
    g = gold("quaternion_to_yaw.json")
    got = quaternion_to_yaw(torch.tensor(g["quat_xyzw"], dtype=torch.float32))
    np.testing.assert_allclose(got.numpy(), np.asarray(g["yaw"], np.float32), rtol=0, atol=1e-6)


def test_transport_roundtrip():
    g = gold("transport.json")
    t = torch.tensor(g["tensor"], dtype=torch.float32)
    b = torch_to_bytes(t)
    assert list(b[:4]) == g["magic"]  # torch.save zip container
    assert bytes_to_torch(b).tolist() == g["roundtrip"]


# ---- stage costs against the reference's own example Objectives (tools/make_golden.py section 7) ----
class ReplaySim:
    """feeds an Objective the recorded simulator answers of the golden case"""
    device = "cpu"

    def __init__(self, inputs):
        self.inputs = {k: torch.tensor(v, dtype=torch.float64) for k, v in inputs.items()}

    def get_actor_link_by_name(self, actor_name, link_name):
        return self.inputs[f"link:{actor_name}:{link_name}"]

    def get_actor_position_by_name(self, name):
        return self.inputs[f"position:{name}"]

    def get_actor_velocity_by_name(self, name):
        return self.inputs[f"velocity:{name}"]

    def get_actor_orientation_by_name(self, name):
        return self.inputs[f"orientation:{name}"]

    def get_actor_contact_forces_by_name(self, actor_name, link_name):
        return self.inputs[f"contact:{actor_name}:{link_name}"]

    def get_dof_state(self):
        return self.inputs["dof_state"]


OBJECTIVES = {"panda": "PandaReachObjective", "boxer_push": "BoxerPushObjective", "panda_pick": "PandaPickObjective",
              "boxer_reach": "BoxerReachObjective", "heijn_reach": "HeijnReachObjective", "heijn_push": "HeijnPushObjective",
              "albert": "AlbertReachObjective", "omni_panda_pick": "OmniPandaPickObjective", "panda_effort": "PandaEffortReachObjective",
              "panda_stick_push": "PandaStickPushObjective", "anymal": "AnymalWalkObjective"}
FUSED = ("panda", "boxer_push", "panda_pick")  # objectives the rollout kernel evaluates in-line (mppi_cost_t)


@pytest.mark.parametrize("case", sorted(OBJECTIVES))
def test_objective_compute_cost_matches_reference_objective(case):
    """generic-mode Objectives (mppiisaac/objectives.py) == the reference example's Objective.compute_cost on the same
    simulator answers; also the default weights"""
    import mppiisaac.objectives as objectives
    g = gold("objective_costs.json")[case]
    obj = getattr(objectives, OBJECTIVES[case])(None)
    assert {k: float(v) for k, v in obj.weights.items()} == g["weights"]
    got = obj.compute_cost(ReplaySim(g["inputs"]))
    np.testing.assert_allclose(got.numpy(), np.array(g["cost"]), rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("case", sorted(FUSED))
def test_oracle_stage_cost_matches_reference_objective(case, oracle64):
    """the oracle's C stage cost (what the fused HIP cost is checked against) == the reference Objective on the same
    rigid-body rows, actor rows and contact forces"""
    from mppiisaac.backend import capi
    from scenes import panda_reach
    g = gold("objective_costs.json")[case]
    inp = {k: np.array(v) for k, v in g["inputs"].items()}
    _, model, *_ = panda_reach()  # (orc_cost does not read the model)
    c = capi.Cost()
    w = g["weights"]
    K = len(g["cost"])
    rb, root, cf = np.zeros((K, 3, 13)), np.zeros((K, 2, 13)), np.zeros((K, 3, 3))
    if case == "panda":
        c.kind = capi.COST_PANDA_REACH
        rb[:, 0] = inp["link:panda:panda_ee_tip"]
        root[:, 0, 0:3] = inp["position:goal"]
        c.link[0], c.actor[0] = 0, 0
        c.w[0], c.w[1] = w["robot_to_goal"], w["robot_ori"]
    elif case == "boxer_push":
        c.kind = capi.COST_BOXER_PUSH
        rb[:, 0] = inp["link:boxer:ee_link"]
        root[:, 0, 0:3], root[:, 0, 3:7], root[:, 0, 7:10] = inp["position:block"], inp["orientation:block"], inp["velocity:block"]
        root[:, 1, 0:3] = inp["position:goal"]
        cf[:, 1], cf[:, 2] = inp["contact:paper_obst1:box"], inp["contact:paper_obst2:box"]
        c.link[0], c.link[1], c.link[2], c.actor[0], c.actor[1] = 0, 1, 2, 0, 1
        for i, k in enumerate(("robot_to_block", "block_to_goal", "block_to_goal_ort", "push_align", "velocity", "collision")):
            c.w[i] = w[k]
        c.w[6] = g["goal_yaw"]
    else:
        c.kind = capi.COST_PANDA_PICK
        rb[:, 0] = inp["link:panda:panda_ee"]
        root[:, 0, 0:3], root[:, 1, 0:3] = inp["position:panda_pick_block"], inp["position:goal"]
        cf[:, 1] = inp["contact:table:box"]
        c.link[0], c.link[1], c.actor[0], c.actor[1] = 0, 1, 0, 1
        for i, k in enumerate(("robot_to_block", "block_to_goal", "collision", "robot_ori")):
            c.w[i] = w[k]
    q = np.zeros(16)
    got = [oracle64.cost(model, c, root[k], q, q, rb[k], cf[k]) for k in range(K)]
    np.testing.assert_allclose(got, g["cost"], rtol=1e-6, atol=1e-6)  # (weights travel as fp32 in mppi_cost_t)


EXAMPLE_SCENES = {   # actors of the reference's example configs (examples/<x>/*.yaml), for name -> index resolution
    "panda": ["panda_stick", "goal"], "boxer_push": ["boxer", "block", "paper_obst1", "paper_obst2", "goal"],
    "panda_pick": ["panda_gripper", "xaxis", "yaxis", "panda_pick_block", "table", "goal"], "boxer_reach": ["boxer", "wall", "goal"],
    "heijn_reach": ["heijn", "wall", "goal"], "heijn_push": ["heijn", "block", "paper_obst1", "paper_obst2", "goal"],
    "albert": ["albert", "goal"], "omni_panda_pick": ["omnipanda_effort", "xaxis", "yaxis", "block2", "table2", "goal"],
    "panda_effort": ["panda_effort", "goal"], "panda_stick_push": ["panda_stick", "xaxis", "yaxis", "panda_push_block", "table", "goal"],
    "anymal": ["anymal", "goal"]}


@pytest.mark.parametrize("case", sorted(EXAMPLE_SCENES))
def test_oracle_cost_program_matches_reference_objective(case, oracle64):
    """MPPI_COST_PROGRAM: the term list of each Objective compiled for its example scene (names -> rigid-body / actor indices)
    and evaluated by the oracle's interpreter (what the in-kernel interpreter is checked against) == the reference planner's
    compute_cost on the same simulator answers (tests/golden/objective_costs.json)"""
    import mppiisaac.objectives as objectives
    from scenes import build_scene
    g = gold("objective_costs.json")[case]
    scene = build_scene(EXAMPLE_SCENES[case], [[0.0, 0.0, 0.05]])
    m = scene.to_c()
    obj = getattr(objectives, OBJECTIVES[case])(None)

    class Sim:
        pass
    sim = Sim()
    sim.scene = scene
    spec = obj.program_spec(sim)
    inp = {k: np.array(v) for k, v in g["inputs"].items()}
    got = []
    for k in range(len(g["cost"])):
        rb, root, cf = np.zeros((m.n_rb, 13)), np.zeros((m.n_actors, 13)), np.zeros((m.n_rb, 3))
        q, qd = np.zeros(16), np.zeros(16)
        for key, val in inp.items():
            kind, *names = key.split(":")
            if kind == "link":
                rb[scene.rigid_body_index(*names)] = val[k]
            elif kind == "contact":
                cf[scene.rigid_body_index(*names)] = val[k]
            elif kind == "dof_state":
                q[:scene.n_dof], qd[:scene.n_dof] = val[k][0::2][:scene.n_dof], val[k][1::2][:scene.n_dof]
            else:
                col = {"position": slice(0, 3), "orientation": slice(3, 7), "velocity": slice(7, 10)}[kind]
                root[scene.actor_index(names[0]), col] = val[k]
        got.append(oracle64.cost(m, spec, root, q, qd, rb, cf))
    np.testing.assert_allclose(got, g["cost"], rtol=1e-6, atol=1e-6)   # (weights and constants travel through the C struct as doubles; fp32 refs)

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

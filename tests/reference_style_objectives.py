"""Eleven Objectives written the way a user of the reference writes theirs - `compute_cost(sim)` in torch over the by-name getters, a
`weights` dict, nothing declared for this backend (no cost program, no fused_spec) - one per example of the reference
(examples/<name>/planner.py: the same measurements with the same weights, restated here from the term lists of mppiisaac/objectives.py).
The tracer (mppiisaac/trace.py) has to turn each of them into a cost program that reproduces tests/golden/objective_costs.json.
pytorch3d's two rotation helpers come from mppiisaac.utils.conversions (pytorch3d is not in the image)."""
import torch

from mppiisaac.utils.conversions import matrix_to_euler_angles, quaternion_to_matrix, quaternion_to_yaw


def _tilt(rows):
    return torch.linalg.norm(matrix_to_euler_angles(quaternion_to_matrix(rows[:, 3:7]), "ZYX")[:, 0:2], axis=1)


class _Base(object):
    weights = {}

    def __init__(self, cfg=None):
        self.weights = dict(type(self).weights)
        self.reset()

    def reset(self):
        pass


class ArmReach(_Base):
    weights = {"robot_to_goal": 1.0, "robot_ori": 0.5}
    actor, link = "panda", "panda_ee_tip"

    def compute_cost(self, sim):
        tip = sim.get_actor_link_by_name(self.actor, self.link)
        target = sim.get_actor_position_by_name("goal")
        d = torch.linalg.norm(tip[:, 0:3] - target[:, 0:3], axis=1)
        return self.weights["robot_to_goal"] * d + self.weights["robot_ori"] * _tilt(tip)


class EffortArmReach(ArmReach):
    link = "panda_link7"


class AlbertReach(ArmReach):
    weights = {"robot_to_goal": 4.0, "robot_ori": 0.5}
    actor, link = "albert", "mmrobot_link7"


class PlanarPush(_Base):
    weights = {"robot_to_block": 0.1, "block_to_goal": 2.0, "block_to_goal_ort": 3.0, "push_align": 0.6, "collision": 100, "velocity": 0.0}
    robot, link = "boxer", "ee_link"
    goal_yaw = 0.0

    def compute_cost(self, sim):
        pusher = sim.get_actor_link_by_name(actor_name=self.robot, link_name=self.link)
        blk = sim.get_actor_position_by_name("block")
        to_block = pusher[:, 0:2] - blk[:, 0:2]
        to_goal = sim.get_actor_position_by_name("goal")[:, 0:2] - blk[:, 0:2]
        d_rb = torch.linalg.norm(to_block, axis=1)
        d_bg = torch.linalg.norm(to_goal, axis=1)
        yaw_err = torch.abs(quaternion_to_yaw(sim.get_actor_orientation_by_name("block")) - self.goal_yaw)
        behind = torch.sum(to_block * to_goal, 1) / (d_rb * d_bg) + 1
        hits = sum(torch.sum(torch.abs(sim.get_actor_contact_forces_by_name(actor_name=o, link_name="box")[:, 0:2]), axis=1)
                   for o in ("paper_obst1", "paper_obst2"))
        speed = torch.linalg.norm(sim.get_actor_velocity_by_name("block")[:, 0:2], axis=1)
        w = self.weights
        return (w["robot_to_block"] * d_rb + w["block_to_goal"] * d_bg + w["block_to_goal_ort"] * yaw_err + w["push_align"] * behind
                + w["velocity"] * speed + w["collision"] * hits)


class HeijnPush(PlanarPush):
    weights = {"robot_to_block": 0.2, "block_to_goal": 2.0, "block_to_goal_ort": 3.0, "push_align": 0.6, "collision": 10, "velocity": 0.0}
    robot, link = "heijn", "front_link"


class BaseReach(_Base):
    robot, link = "boxer", "ee_link"

    def compute_cost(self, sim):
        here = sim.get_actor_link_by_name(actor_name=self.robot, link_name=self.link)
        there = sim.get_actor_position_by_name("goal")
        push = sim.get_actor_contact_forces_by_name("wall", "box")
        return torch.linalg.norm(there[:, 0:2] - here[:, 0:2], axis=1) + torch.sum(torch.abs(push[:, 0:3]), axis=1)


class HeijnReach(BaseReach):
    robot, link = "heijn", "front_link"


class Pick(_Base):
    weights = {"robot_to_block": 40.0, "block_to_goal": 10.0, "collision": 26.0, "robot_ori": 2.0}
    robot, link = "panda", "panda_ee"

    def pick_terms(self, sim):
        hand = sim.get_actor_link_by_name(self.robot, self.link)
        blk = sim.get_actor_position_by_name("panda_pick_block")
        w = self.weights
        return hand, (w["robot_to_block"] * torch.linalg.norm(hand[:, 0:3] - blk[:, 0:3], axis=1)
                      + w["block_to_goal"] * torch.linalg.norm(blk[:, 0:3] - sim.get_actor_position_by_name("goal")[:, 0:3], axis=1)
                      + w["collision"] * torch.sum(torch.abs(sim.get_actor_contact_forces_by_name("table", "box")[:, 0:3]), axis=1)
                      + w["robot_ori"] * _tilt(hand))

    def compute_cost(self, sim):
        return self.pick_terms(sim)[1]


class OmniPick(Pick):
    weights = {"robot_to_block": 10.0, "block_to_goal": 4.0, "collision": 0.1, "robot_ori": 1.0, "base_vel": 2.0, "arm_vel": 0.1,
               "comfy_gripper_state": 200.0, "comfy_arm_pose": 0.1, "height_cost": 10000.0}
    robot, link = "omnipanda", "panda_hand"

    def reset(self):
        self.rest_fingers = torch.tensor([0.025, 0.025])
        self.rest_arm = torch.tensor([-1.57, -0.94, 0.0, -2.8, 0.0, 1.8675, 0.75])

    def compute_cost(self, sim):
        hand, cost = self.pick_terms(sim)
        state = sim.get_dof_state()
        pos, vel = state[:, 0::2], state[:, 1::2]
        w = self.weights
        return (cost + w["base_vel"] * torch.sum(torch.square(vel[:, 0:3]), dim=1) + w["arm_vel"] * torch.sum(torch.square(vel[:, 3:10]), dim=1)
                + w["comfy_gripper_state"] * torch.sum(torch.square(pos[:, -2:] - self.rest_fingers), dim=1)
                + w["comfy_arm_pose"] * torch.sum(torch.square(pos[:, 3:10] - self.rest_arm), dim=1)
                + w["height_cost"] * torch.clamp(0.12 - hand[:, 2], min=0))


class StickPush(_Base):
    weights = {"robot_to_block": 5.0, "block_to_goal": 25.0, "collision": 0.0, "robot_ori": 5.0, "block_height": 20.0, "push_align": 45.0}

    def compute_cost(self, sim):
        tip = sim.get_actor_link_by_name("panda", "panda_ee_tip")
        blk = sim.get_actor_position_by_name("panda_push_block")
        to_block = tip[:, 0:3] - blk[:, 0:3]
        to_goal = sim.get_actor_position_by_name("goal")[:, 0:3] - blk[:, 0:3]
        flat = torch.sum(to_block[:, 0:2] * to_goal[:, 0:2], 1) / (torch.linalg.norm(to_block[:, :2], axis=1) * torch.linalg.norm(to_goal[:, :2], axis=1)) + 1
        w = self.weights
        return (w["robot_to_block"] * torch.linalg.norm(to_block, axis=1) + w["block_to_goal"] * torch.linalg.norm(to_goal, axis=1)
                + w["collision"] * torch.sum(torch.abs(sim.get_actor_contact_forces_by_name("table", "box")[:, 0:3]), axis=1)
                + w["robot_ori"] * _tilt(tip) + w["block_height"] * torch.abs(tip[:, 2] - blk[:, 2]) + w["push_align"] * flat)


class AnymalWalk(_Base):
    weights = {"robot_to_goal": 1.0, "robot_off_ground": 5.0, "knees_off_ground": 5.0}

    def compute_cost(self, sim):
        trunk = sim.get_actor_link_by_name("anymal", "base")
        d = torch.linalg.norm(trunk[:, 0:3] - sim.get_actor_position_by_name("goal")[:, 0:3], axis=1)
        body = sum(torch.abs(sim.get_actor_link_by_name("anymal", n)[:, 2] - 0.65) for n in ("base", "face_front", "face_rear"))
        knees = sum(torch.abs(sim.get_actor_link_by_name("anymal", n)[:, 2] - 0.35) for n in ("LF_KFE", "LH_KFE", "RH_KFE", "RF_KFE"))
        return self.weights["robot_to_goal"] * d + self.weights["robot_off_ground"] * body + self.weights["knees_off_ground"] * knees


CASES = {"panda": ArmReach, "panda_effort": EffortArmReach, "albert": AlbertReach, "boxer_push": PlanarPush, "heijn_push": HeijnPush,
         "boxer_reach": BaseReach, "heijn_reach": HeijnReach, "panda_pick": Pick, "omni_panda_pick": OmniPick, "panda_stick_push": StickPush,
         "anymal": AnymalWalk}

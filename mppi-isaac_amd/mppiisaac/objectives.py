"""Stage-cost Objectives for the BASELINE workloads, honouring the reference's Objective contract
(`compute_cost(sim) -> [K]`, `reset()`, mutable `.weights`; reference examples/*/planner.py) and
additionally declaring `fused_spec(sim)` so the planner can evaluate the same cost inside the
persistent rollout kernel.  `compute_cost` is the reference's torch code path (generic mode);
tests check fused == generic."""
import torch

from mppiisaac.backend import capi
from mppiisaac.utils.conversions import matrix_to_euler_angles, quaternion_to_matrix, quaternion_to_yaw


class PandaReachObjective(object):
    """reference examples/panda/planner.py:10-40 (incl. the xyzw-into-wxyz quirk, kept as is)."""

    def __init__(self, cfg=None, actor="panda", link="panda_ee_tip", goal="goal"):
        self.weights = {"robot_to_goal": 1.0, "robot_ori": 0.5}
        self.actor, self.link, self.goal = actor, link, goal
        self.reset()

    def reset(self):
        pass

    def compute_cost(self, sim):
        r_pos = sim.get_actor_link_by_name(self.actor, self.link)
        goal_pos = sim.get_actor_position_by_name(self.goal)
        robot_to_goal = r_pos[:, 0:3] - goal_pos[:, 0:3]
        robot_to_goal_dist = torch.linalg.norm(robot_to_goal, axis=1)
        robot_rpy = matrix_to_euler_angles(quaternion_to_matrix(r_pos[:, 3:7]), "ZYX")[:, 0:2]
        robot_rpy_dist = torch.linalg.norm(robot_rpy, axis=1)
        return self.weights["robot_to_goal"] * robot_to_goal_dist + self.weights["robot_ori"] * robot_rpy_dist

    def fused_spec(self, sim) -> capi.Cost:
        c = capi.Cost()
        c.kind = capi.COST_PANDA_REACH
        c.link[0] = sim.scene.rigid_body_index(self.actor, self.link)
        c.actor[0] = sim.scene.actor_index(self.goal)
        c.w[0], c.w[1] = self.weights["robot_to_goal"], self.weights["robot_ori"]
        return c


class PointReachObjective(object):
    """Navigation term of reference benchmarks/point_robot/mppi_planner/mppi_planner_wrapper.py:10-35:
    w_nav * || (x, y) - goal ||, x/y = DOF positions 0 and 1.  `goal` is either an actor name or [x, y]."""

    def __init__(self, cfg=None, goal="goal", w_nav=2.0):
        self.weights = {"w_nav": w_nav}
        self.goal = goal
        self.reset()

    def reset(self):
        pass

    def _goal_xy(self, sim):
        if isinstance(self.goal, str):
            return sim.get_actor_position_by_name(self.goal)[:, 0:2]
        return torch.tensor(self.goal, dtype=torch.float32, device=sim.device).view(1, 2)

    def compute_cost(self, sim):
        dof_state = sim.get_dof_state()
        pos = torch.cat((dof_state[:, 0].unsqueeze(1), dof_state[:, 2].unsqueeze(1)), 1)
        return self.weights["w_nav"] * torch.linalg.norm(pos - self._goal_xy(sim), axis=1)

    def fused_spec(self, sim) -> capi.Cost:
        c = capi.Cost()
        c.kind = capi.COST_POINT_REACH
        c.w[0] = self.weights["w_nav"]
        if isinstance(self.goal, str):
            c.actor[0] = sim.scene.actor_index(self.goal)
        else:
            c.actor[0] = -1
            c.w[1], c.w[2] = float(self.goal[0]), float(self.goal[1])
        return c


class BoxerPushObjective(object):
    """reference examples/boxer_push/planner.py:9-67: push a block to a goal pose with a differential-drive
    base while avoiding contact with two obstacles."""

    def __init__(self, cfg=None, robot="boxer", link="ee_link", block="block", goal="goal",
                 obstacles=("paper_obst1", "paper_obst2")):
        self.weights = {"robot_to_block": 0.1, "block_to_goal": 2.0, "block_to_goal_ort": 3.0, "push_align": 0.6,
                        "collision": 100, "velocity": 0.0}
        self.goal_yaw = 0.0
        self.robot, self.link, self.block, self.goal, self.obstacles = robot, link, block, goal, tuple(obstacles)

    def reset(self):
        pass

    def compute_cost(self, sim):
        r_pos = sim.get_actor_link_by_name(actor_name=self.robot, link_name=self.link)
        block_pos = sim.get_actor_position_by_name(self.block)
        block_vel = sim.get_actor_velocity_by_name(self.block)
        block_ort = sim.get_actor_orientation_by_name(self.block)
        block_goal = sim.get_actor_position_by_name(self.goal)
        robot_to_block = r_pos[:, 0:2] - block_pos[:, 0:2]
        block_to_goal = block_goal[:, 0:2] - block_pos[:, 0:2]
        block_yaws = quaternion_to_yaw(block_ort)
        robot_to_block_dist = torch.linalg.norm(robot_to_block[:, 0:2], axis=1)
        block_to_pos_dist = torch.linalg.norm(block_to_goal, axis=1)
        block_to_ort_dist = torch.abs(block_yaws - self.goal_yaw)
        push_align = torch.sum(robot_to_block[:, 0:2] * block_to_goal, 1) / (robot_to_block_dist * block_to_pos_dist) + 1
        obst1_forces = sim.get_actor_contact_forces_by_name(actor_name=self.obstacles[0], link_name="box")
        obst2_forces = sim.get_actor_contact_forces_by_name(actor_name=self.obstacles[1], link_name="box")
        coll = torch.sum(torch.abs(obst1_forces[:, 0:2]), axis=1) + torch.sum(torch.abs(obst2_forces[:, 0:2]), axis=1)
        vel = torch.linalg.norm(block_vel[:, 0:2], axis=1)
        w = self.weights
        return (w["robot_to_block"] * robot_to_block_dist + w["block_to_goal"] * block_to_pos_dist
                + w["block_to_goal_ort"] * block_to_ort_dist + w["push_align"] * push_align
                + w["velocity"] * vel + w["collision"] * coll)

    def fused_spec(self, sim) -> capi.Cost:
        c = capi.Cost()
        c.kind = capi.COST_BOXER_PUSH
        c.link[0] = sim.scene.rigid_body_index(self.robot, self.link)
        c.link[1] = sim.scene.rigid_body_index(self.obstacles[0], "box")
        c.link[2] = sim.scene.rigid_body_index(self.obstacles[1], "box")
        c.actor[0] = sim.scene.actor_index(self.block)
        c.actor[1] = sim.scene.actor_index(self.goal)
        w = self.weights
        for i, k in enumerate(("robot_to_block", "block_to_goal", "block_to_goal_ort", "push_align", "velocity", "collision")):
            c.w[i] = float(w[k])
        c.w[6] = float(self.goal_yaw)
        return c


class PandaPickObjective(object):
    """reference examples/panda_pick/planner.py:9-53: reach the block, bring it to the goal, stay off the table."""

    def __init__(self, cfg=None, robot="panda", link="panda_ee", block="panda_pick_block", goal="goal", table="table"):
        self.weights = {"robot_to_block": 40.0, "block_to_goal": 10.0, "collision": 26.0, "robot_ori": 2.0}
        self.robot, self.link, self.block, self.goal, self.table = robot, link, block, goal, table
        self.reset()

    def reset(self):
        self.prev_block_to_goal_dist = 1
        self.prev_robot_to_block_dist = 1

    def compute_cost(self, sim):
        r_pos = sim.get_actor_link_by_name(self.robot, self.link)
        block_pos = sim.get_actor_position_by_name(self.block)
        goal_pos = sim.get_actor_position_by_name(self.goal)
        table_forces = sim.get_actor_contact_forces_by_name(self.table, "box")
        robot_to_block_dist = torch.linalg.norm(r_pos[:, 0:3] - block_pos[:, 0:3], axis=1)
        block_to_goal_dist = torch.linalg.norm(block_pos[:, 0:3] - goal_pos[:, 0:3], axis=1)
        robot_rpy = matrix_to_euler_angles(quaternion_to_matrix(r_pos[:, 3:7]), "ZYX")[:, 0:2]
        robot_rpy_dist = torch.linalg.norm(robot_rpy, axis=1)
        forces = torch.sum(torch.abs(table_forces[:, 0:3]), axis=1)
        w = self.weights
        self.prev_block_to_goal_dist = block_to_goal_dist
        return (w["robot_to_block"] * robot_to_block_dist + w["block_to_goal"] * block_to_goal_dist
                + w["collision"] * forces + w["robot_ori"] * robot_rpy_dist)

    def fused_spec(self, sim) -> capi.Cost:
        c = capi.Cost()
        c.kind = capi.COST_PANDA_PICK
        c.link[0] = sim.scene.rigid_body_index(self.robot, self.link)
        c.link[1] = sim.scene.rigid_body_index(self.table, "box")
        c.actor[0] = sim.scene.actor_index(self.block)
        c.actor[1] = sim.scene.actor_index(self.goal)
        for i, k in enumerate(("robot_to_block", "block_to_goal", "collision", "robot_ori")):
            c.w[i] = float(self.weights[k])
        return c

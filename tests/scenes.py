"""Shared builders for the BASELINE workloads: scene + MPPI config + cost spec."""
import numpy as np

from mppiisaac.backend import capi
from mppiisaac.planner.isaacgym_wrapper import IsaacGymConfig, Scene
from mppiisaac.planner.mppi import make_config
from mppiisaac.utils.config_store import load_config
from mppiisaac.utils.isaacgym_utils import load_actor_cfgs, load_asset


def build_scene(actors, init_positions=None, isaacgym="normal", overrides=None, robot_overrides=None):
    env_cfg = load_actor_cfgs(actors)
    robots = [a for a in env_cfg if a.type == "robot"]
    for k, v in (robot_overrides or {}).items():   # (every robot of the env: the multi-robot scenes repeat one robot type)
        for r in robots:
            setattr(r, k, v)
    if init_positions:
        for p, a in zip(init_positions, robots):
            a.init_pos = list(p)
    cfg = load_config({"defaults": [{"isaacgym": isaacgym}]}).isaacgym
    for k, v in (overrides or {}).items():
        setattr(cfg, k, v)
    return Scene(env_cfg, cfg, [load_asset(r) for r in robots])


def panda_reach(K=64, H=20, goal=(0.5, -0.4, 0.3), **mppi_over):
    """BASELINE config 3: panda_stick + goal, conf/mppi/panda.yaml with K/H overridden."""
    scene = build_scene(["panda_stick", "goal"], [[0.0, 0.0, 0.0]])
    ex = load_config({"defaults": [{"mppi": "panda"}, {"isaacgym": "normal"}]})
    ex.mppi.num_samples, ex.mppi.horizon = K, H
    for k, v in mppi_over.items():
        setattr(ex.mppi, k, v)
    cfg = make_config(ex.mppi, viz_link=scene.viz_link_index())
    dof, root = scene.initial_state()
    root[scene.actor_index("goal"), 0:3] = goal
    cost = capi.Cost()
    cost.kind = capi.COST_PANDA_REACH
    cost.link[0] = scene.rigid_body_index("panda", "panda_ee_tip")
    cost.actor[0] = scene.actor_index("goal")
    cost.w[0], cost.w[1] = 1.0, 0.5
    return scene, scene.to_c(), cfg, cost, dof, root


def point_reach(K=64, H=10, goal=(4.5, 0.2), **mppi_over):
    """BASELINE configs 1/2: point_robot + goal, conf/mppi/pointbot.yaml with K/H overridden."""
    scene = build_scene(["point_robot", "goal"], [[0.0, 0.0, 0.05]])
    ex = load_config({"defaults": [{"mppi": "pointbot"}, {"isaacgym": "normal"}]})
    ex.mppi.num_samples, ex.mppi.horizon, ex.mppi.use_priors = K, H, False
    for k, v in mppi_over.items():
        setattr(ex.mppi, k, v)
    cfg = make_config(ex.mppi, viz_link=-1)
    dof, root = scene.initial_state()
    dof[0] = 0.1
    root[scene.actor_index("goal"), 0:2] = goal
    cost = capi.Cost()
    cost.kind = capi.COST_POINT_REACH
    cost.actor[0] = scene.actor_index("goal")
    cost.w[0] = 2.0
    return scene, scene.to_c(), cfg, cost, dof, root


def boxer_push(K=64, H=12, **mppi_over):
    """BASELINE config 4: boxer + block + two obstacles + goal (reference examples/boxer_push/config_boxer_push.yaml:9-10),
    conf/mppi/boxer_push.yaml with K/H overridden; size/mass/friction noise off."""
    scene = build_scene(["boxer", "block", "paper_obst1", "paper_obst2", "goal"], [[0.0, 2.5, 0.05]])
    ex = load_config({"defaults": [{"mppi": "boxer_push"}, {"isaacgym": "normal"}]})
    ex.mppi.num_samples, ex.mppi.horizon = K, H
    for k, v in mppi_over.items():
        setattr(ex.mppi, k, v)
    cfg = make_config(ex.mppi, viz_link=scene.viz_link_index())
    dof, root = scene.initial_state()
    cost = capi.Cost()
    cost.kind = capi.COST_BOXER_PUSH
    cost.link[0] = scene.rigid_body_index("boxer", "ee_link")
    cost.link[1] = scene.rigid_body_index("paper_obst1", "box")
    cost.link[2] = scene.rigid_body_index("paper_obst2", "box")
    cost.actor[0], cost.actor[1] = scene.actor_index("block"), scene.actor_index("goal")
    for i, w in enumerate((0.1, 2.0, 3.0, 0.6, 0.0, 100.0, 0.0)):
        cost.w[i] = w
    return scene, scene.to_c(), cfg, cost, dof, root


def panda_pick(K=64, H=12, **mppi_over):
    """BASELINE config 5 scene: gripper panda + axes + pick block + table + goal (reference
    examples/panda_pick/panda_pick.yaml:6), conf/mppi/panda_pick.yaml with K/H overridden."""
    scene = build_scene(["panda_gripper", "xaxis", "yaxis", "panda_pick_block", "table", "goal"], [[0.0, 0.0, 0.0]])
    ex = load_config({"defaults": [{"mppi": "panda_pick"}, {"isaacgym": "normal"}]})
    ex.mppi.num_samples, ex.mppi.horizon = K, H
    for k, v in mppi_over.items():
        setattr(ex.mppi, k, v)
    cfg = make_config(ex.mppi, viz_link=scene.viz_link_index())
    dof, root = scene.initial_state()
    cost = capi.Cost()
    cost.kind = capi.COST_PANDA_PICK
    cost.link[0] = scene.rigid_body_index("panda", "panda_ee")
    cost.link[1] = scene.rigid_body_index("table", "box")
    cost.actor[0], cost.actor[1] = scene.actor_index("panda_pick_block"), scene.actor_index("goal")
    for i, w in enumerate((40.0, 10.0, 26.0, 2.0)):
        cost.w[i] = w
    return scene, scene.to_c(), cfg, cost, dof, root

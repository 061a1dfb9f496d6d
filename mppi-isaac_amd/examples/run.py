"""Every example scene of the reference (examples/<name>/{config,planner,world}.py) on the HIP backend, table-driven:
the example's actors and conf groups (reference examples/<name>/*.yaml), its Objective (mppiisaac/objectives.py: the same
cost as the example's planner.py, as a cost program that runs inside the rollout kernel), an MPPIisaacPlanner, a K=1
IsaacGymWrapper as the world, state and action exchanged as torch.save blobs (reference world.py / planner.py).

    python run.py boxer_push                # planner and world in one process
    python run.py panda_pick --serve        # planner process (reference: zerorpc.Server(...).bind("tcp://0.0.0.0:4242"))
    python run.py panda_pick --connect      # world process   (reference: zerorpc.Client().connect("tcp://127.0.0.1:4242"))
    python run.py --list
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import torch  # noqa: E402

import mppiisaac.objectives as objectives  # noqa: E402
from mppiisaac.planner.isaacgym_wrapper import IsaacGymWrapper  # noqa: E402
from mppiisaac.planner.mppi_isaac import MPPIisaacPlanner  # noqa: E402
from mppiisaac.utils import rpc as zerorpc  # noqa: E402
from mppiisaac.utils.config_store import load_config  # noqa: E402
from mppiisaac.utils.transport import bytes_to_torch, torch_to_bytes  # noqa: E402

# name -> the example's config (reference examples/<name>/*.yaml: defaults, actors, initial_actor_positions, nx) + Objective
EXAMPLES = {
    "panda": dict(mppi="panda", isaacgym="normal", actors=["panda_stick", "goal"], init=[[0.0, 0.0, 0.0]], nx=14, objective="PandaReachObjective",
                  goal=[0.5, -0.4, 0.3]),
    "panda_effort": dict(mppi="panda_effort", isaacgym="normal", actors=["panda_effort", "goal"], init=[[0.0, 0.0, 0.0]], nx=14,
                         objective="PandaEffortReachObjective", goal=[0.5, -0.4, 0.3]),
    "boxer_reach": dict(mppi="boxer_reach", isaacgym="normal", actors=["boxer", "wall", "goal"], init=[[0.0, 0.0, 0.05]], nx=4, objective="BoxerReachObjective"),
    "boxer_push": dict(mppi="boxer_push", isaacgym="normal", actors=["boxer", "block", "paper_obst1", "paper_obst2", "goal"], init=[[0.0, 2.5, 0.05]], nx=4,
                       objective="BoxerPushObjective"),
    "heijn_reach": dict(mppi="heijn_reach", isaacgym="normal", actors=["heijn", "wall", "goal"], init=[[0.0, 0.0, 0.05]], nx=6, objective="HeijnReachObjective"),
    "heijn_push": dict(mppi="heijn_push", isaacgym="push", actors=["heijn", "block", "paper_obst1", "paper_obst2", "goal"], init=[[0.0, 1.5, 0.05]], nx=6,
                       objective="HeijnPushObjective"),
    "albert": dict(mppi="albert", isaacgym="normal", actors=["albert", "goal"], init=[[0.0, 0.0, 0.05]], nx=18, objective="AlbertReachObjective"),
    "panda_pick": dict(mppi="panda_pick", isaacgym="normal", actors=["panda_gripper", "xaxis", "yaxis", "panda_pick_block", "table", "goal"], init=[[0.0, 0.0, 0.0]],
                       nx=18, objective="PandaPickObjective"),
    "omni_panda_pick": dict(mppi="omnipanda_effort", isaacgym="pick", actors=["omnipanda_effort", "xaxis", "yaxis", "block2", "table2", "goal"],
                            init=[[1.0, 2.0, 0.0]], nx=24, objective="OmniPandaPickObjective"),
    "panda_stick_push": dict(mppi="panda_stick_push", isaacgym="normal", actors=["panda_stick", "xaxis", "yaxis", "panda_push_block", "table", "goal"],
                             init=[[0.0, 0.0, 0.0]], nx=14, objective="PandaStickPushObjective"),
    # reference examples/anymal: its conf/mppi/anymal.yaml ships with `noise_sigma` commented out (the planner cannot be built
    # from it there); the runner supplies one - joint velocity noise of 1 (rad/s)^2 on the twelve leg joints
    "anymal": dict(mppi="anymal", isaacgym="push", actors=["anymal", "goal"], init=[[0.0, 2.0, 1.2]], nx=24, objective="AnymalWalkObjective",
                   goal=[2.0, 2.0, 0.5], overrides={"noise_sigma": [[1.0 if i == j else 0.0 for j in range(12)] for i in range(12)]}),
    # reference conf/mppi/multi-jackal.yaml (two jackals, nu = 4; no example script of the reference uses it): every robot drives to
    # its own target.  conf/actors/jackal_a.yaml / jackal_b.yaml are named copies of jackal.yaml with the wheel joints filled in
    "multi_jackal": dict(mppi="multi-jackal", isaacgym="normal", actors=["jackal_a", "jackal_b", "goal"], init=[[0.0, 0.0, 0.1], [0.5, -2.0, 0.1]], nx=8,
                         objective="MultiJackalObjective", goal=[2.0, 1.0, 0.1]),
}


def config(name, **overrides):
    e = EXAMPLES[name]
    overrides = {**e.get("overrides", {}), **overrides}
    return load_config({"defaults": [{"mppi": e["mppi"]}, {"isaacgym": e["isaacgym"]}], "actors": e["actors"], "initial_actor_positions": e["init"],
                        "nx": e["nx"]}, overrides={f"mppi.{k}": v for k, v in overrides.items()})


def make_planner(name, cfg):
    planner = MPPIisaacPlanner(cfg, getattr(objectives, EXAMPLES[name]["objective"])(cfg), prior=None)
    if EXAMPLES[name].get("goal"):
        planner.sim.set_actor_position_by_name(EXAMPLES[name]["goal"], "goal")
    return planner


def make_world(name, cfg):
    sim = IsaacGymWrapper(cfg.isaacgym, actors=cfg.actors, init_positions=cfg.initial_actor_positions, num_envs=1)
    if EXAMPLES[name].get("goal"):
        sim.set_actor_position_by_name(EXAMPLES[name]["goal"], "goal")
    return sim


def run_world(name, cfg, planner, steps, report=True, hook=None):
    """the loop of the reference's world.py: world state -> planner (bytes) -> action -> apply + step.  Returns the stage cost of
    the world's state (the example's own Objective evaluated on the K=1 world) before and after."""
    sim = make_world(name, cfg)
    objective = getattr(objectives, EXAMPLES[name]["objective"])(cfg)
    first = float(objective.compute_cost(sim)[0])
    t0 = time.perf_counter()
    for i in range(steps):
        action = bytes_to_torch(planner.compute_action_tensor(torch_to_bytes(sim._dof_state), torch_to_bytes(sim._root_state)))
        sim.apply_robot_cmd(action.to(sim.device).reshape(1, -1))
        sim.step()
        if hook is not None:
            hook(i, sim)
        if report and i % 25 == 0:
            print(f"step {i:4d}  stage cost of the world state = {float(objective.compute_cost(sim)[0]):.4f}")
    rate = steps / (time.perf_counter() - t0)
    last = float(objective.compute_cost(sim)[0])
    if report:
        print(f"{name}: {rate:.0f} control iterations per second through the bytes API; stage cost {first:.4f} -> {last:.4f}")
    return first, last, rate


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("example", nargs="?", default="panda", choices=sorted(EXAMPLES))
    ap.add_argument("--list", action="store_true")
    ap.add_argument("--serve", action="store_true")
    ap.add_argument("--connect", action="store_true")
    ap.add_argument("--endpoint", default="tcp://127.0.0.1:4242")
    ap.add_argument("--steps", type=int, default=200)
    a = ap.parse_args()
    if a.list:
        for k, e in EXAMPLES.items():
            print(f"{k:18s} actors {e['actors']}  conf/mppi/{e['mppi']}.yaml  {e['objective']}")
        sys.exit(0)
    cfg = config(a.example)
    if a.serve:
        server = zerorpc.Server(make_planner(a.example, cfg))
        server.bind(a.endpoint)
        server.run()
    elif a.connect:
        client = zerorpc.Client()
        client.connect(a.endpoint)
        run_world(a.example, cfg, client, a.steps)
    else:
        run_world(a.example, cfg, make_planner(a.example, cfg), a.steps)

"""Panda reach with the HIP backend, written the way the reference's examples are (examples/panda/planner.py +
world.py): an Objective with compute_cost(sim), an MPPIisaacPlanner around it, a K=1 IsaacGymWrapper as the world,
state and action exchanged as torch.save blobs.

    python panda_reach.py                 # planner and world in one process
    python panda_reach.py --serve         # planner process  (reference: zerorpc.Server(...).bind("tcp://0.0.0.0:4242"))
    python panda_reach.py --connect       # world process    (reference: zerorpc.Client().connect("tcp://127.0.0.1:4242"))
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import torch  # noqa: E402

from mppiisaac.objectives import PandaReachObjective  # noqa: E402  (its compute_cost is examples/panda/planner.py:22-40)
from mppiisaac.planner.isaacgym_wrapper import IsaacGymWrapper  # noqa: E402
from mppiisaac.planner.mppi_isaac import MPPIisaacPlanner  # noqa: E402
from mppiisaac.utils import rpc as zerorpc  # noqa: E402
from mppiisaac.utils.config_store import load_config  # noqa: E402
from mppiisaac.utils.transport import bytes_to_torch, torch_to_bytes  # noqa: E402

GOAL = [0.5, -0.4, 0.3]


def config():
    return load_config({"defaults": [{"mppi": "panda"}, {"isaacgym": "normal"}], "actors": ["panda_stick", "goal"],
                       "initial_actor_positions": [[0.0, 0.0, 0.0]], "nx": 14})


def make_planner(cfg):
    planner = MPPIisaacPlanner(cfg, PandaReachObjective(cfg), prior=None)
    planner.sim.set_actor_position_by_name(GOAL, "goal")
    return planner


def run_world(cfg, planner, steps):
    sim = IsaacGymWrapper(cfg.isaacgym, actors=cfg.actors, init_positions=cfg.initial_actor_positions, num_envs=1)
    sim.set_actor_position_by_name(GOAL, "goal")
    ee = sim.scene.rigid_body_index("panda", "panda_ee_tip")
    t0 = time.perf_counter()
    for i in range(steps):
        action = bytes_to_torch(planner.compute_action_tensor(torch_to_bytes(sim._dof_state), torch_to_bytes(sim._root_state)))
        sim.apply_robot_cmd(action.to(sim.device).reshape(1, -1))
        sim.step()
        if i % 20 == 0:
            d = torch.linalg.norm(sim._rigid_body_state[0, ee, 0:3].cpu() - torch.tensor(GOAL))
            print(f"step {i:4d}  |ee - goal| = {float(d):.3f} m")
    print(f"{steps / (time.perf_counter() - t0):.0f} control iterations per second through the bytes API")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--serve", action="store_true")
    ap.add_argument("--connect", action="store_true")
    ap.add_argument("--endpoint", default="tcp://127.0.0.1:4242")
    ap.add_argument("--steps", type=int, default=200)
    a = ap.parse_args()
    cfg = config()
    if a.serve:
        server = zerorpc.Server(make_planner(cfg))
        server.bind(a.endpoint)
        server.run()
    elif a.connect:
        client = zerorpc.Client()
        client.connect(a.endpoint)
        run_world(cfg, client, a.steps)
    else:
        run_world(cfg, make_planner(cfg), a.steps)

"""Control-loop rate of the GENERIC Objective mode (an unmodified reference-style Objective: Python compute_cost(sim)
per horizon step) next to the fused mode, panda reach K=4096 H=20.  Experiment, not the benchmark."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(__file__), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mppi-isaac_amd"))
import torch
from mppiisaac.objectives import PandaReachObjective
from mppiisaac.planner.isaacgym_wrapper import IsaacGymWrapper
from mppiisaac.planner.mppi_isaac import MPPIisaacPlanner
from mppiisaac.utils.config_store import load_config
from mppiisaac.utils.transport import torch_to_bytes, bytes_to_torch


class GenericReach:
    """reference examples/panda/planner.py:22-40 verbatim in spirit: only compute_cost(sim)"""
    def __init__(self, cfg):
        self.inner = PandaReachObjective(cfg)
    def reset(self):
        pass
    def compute_cost(self, sim):
        return self.inner.compute_cost(sim)


def rate(objective_cls, n=60):
    cfg = load_config({"defaults": [{"mppi": "panda"}, {"isaacgym": "normal"}], "actors": ["panda_stick", "goal"],
                       "initial_actor_positions": [[0.0, 0.0, 0.0]], "nx": 14},
                      overrides={"mppi.num_samples": 4096, "mppi.horizon": 20})
    planner = MPPIisaacPlanner(cfg, objective_cls(cfg))
    world = IsaacGymWrapper(cfg.isaacgym, actors=cfg.actors, init_positions=cfg.initial_actor_positions, num_envs=1)
    for sim in (planner.sim, world):
        sim.set_actor_position_by_name([0.5, -0.4, 0.3], "goal")
    def it():
        a = bytes_to_torch(planner.compute_action_tensor(torch_to_bytes(world._dof_state), torch_to_bytes(world._root_state)))
        world.apply_robot_cmd(a.to(world.device).reshape(1, -1))
        world.step()
    for _ in range(5): it()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): it()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / n
    print(f"{objective_cls.__name__:24s} {1/dt:8.1f} Hz  ({1e3*dt:.2f} ms / iteration, bytes API + world step through Python)", flush=True)

rate(PandaReachObjective)
rate(GenericReach)

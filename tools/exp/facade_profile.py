"""cProfile of the reference-shaped closed loop (compute_action_tensor bytes API + K=1 world through Python)."""
import cProfile, os, pstats, sys
ROOT = os.path.join(os.path.dirname(__file__), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mppi-isaac_amd"))
import torch
from mppiisaac.objectives import PandaReachObjective
from mppiisaac.planner.isaacgym_wrapper import IsaacGymWrapper
from mppiisaac.planner.mppi_isaac import MPPIisaacPlanner
from mppiisaac.utils.config_store import load_config
from mppiisaac.utils.transport import torch_to_bytes, bytes_to_torch
cfg = load_config({"defaults": [{"mppi": "panda"}, {"isaacgym": "normal"}], "actors": ["panda_stick", "goal"],
                   "initial_actor_positions": [[0.0, 0.0, 0.0]], "nx": 14}, overrides={"mppi.num_samples": 4096, "mppi.horizon": 20})
planner = MPPIisaacPlanner(cfg, PandaReachObjective(cfg))
world = IsaacGymWrapper(cfg.isaacgym, actors=cfg.actors, init_positions=cfg.initial_actor_positions, num_envs=1)
for sim in (planner.sim, world): sim.set_actor_position_by_name([0.5, -0.4, 0.3], "goal")
def it():
    a = bytes_to_torch(planner.compute_action_tensor(torch_to_bytes(world._dof_state), torch_to_bytes(world._root_state)))
    world.apply_robot_cmd(a.to(world.device).reshape(1, -1))
    world.step()
import time
for _ in range(20): it()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(500): it()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 500
print(f"bytes-API closed loop (compute_action_tensor + python world step): {1 / dt:.0f} Hz, {dt * 1e3:.3f} ms/iteration", flush=True)
pr = cProfile.Profile(); pr.enable()
for _ in range(300): it()
torch.cuda.synchronize(); pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(28)

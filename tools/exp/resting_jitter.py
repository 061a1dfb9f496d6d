import sys, os, torch, numpy as np
sys.path.insert(0, os.path.join(os.getcwd(), "mppi-isaac_amd"))
from mppiisaac.planner.isaacgym_wrapper import IsaacGymWrapper
from mppiisaac.utils.config_store import load_config
cfg = load_config({"defaults": [{"isaacgym": "normal"}]})
sim = IsaacGymWrapper(cfg.isaacgym, actors=["boxer", "block", "goal"], init_positions=[[0, 0, 0.05]], num_envs=1)
u = torch.tensor([0.0, 0.0])
w = []
for i in range(200):
    sim.apply_robot_cmd(u); sim.step()
    if i >= 100: w.append(sim._root_state[0, 0, 10:13].abs().max().item())
print("resting |omega| max over steps 100..200: %.3e" % max(w))

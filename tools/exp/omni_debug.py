"""omni_panda_pick closed loop: where does the hand stop, and why?"""
import importlib.util, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mppi-isaac_amd"))
import logging; logging.disable(logging.WARNING)
spec = importlib.util.spec_from_file_location("examples_run", os.path.join(ROOT, "mppi-isaac_amd", "examples", "run.py"))
run = importlib.util.module_from_spec(spec); spec.loader.exec_module(run)
name = sys.argv[1] if len(sys.argv) > 1 else "omni_panda_pick"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
cfg = run.config(name)
planner = run.make_planner(name, cfg)
print("light pairs", planner.sim._c_model.contact_flags, "K", cfg.mppi.num_samples, "H", cfg.mppi.horizon, "dt", cfg.isaacgym.dt, "u", cfg.mppi.u_min, cfg.mppi.u_max, "lambda", cfg.mppi.lambda_)
def hook(i, sim):
    if i % max(1, steps // 15) == 0 or i == steps - 1:
        names = [a.name for a in sim.scene.env_cfg]
        blk = next(n for n in names if "block" in n)
        robot = names[sim.scene.robot_idx]
        hand = sim.get_actor_link_by_name(robot, "panda_hand")[0, 0:3].cpu().numpy()
        b = sim._root_state[0, sim.scene.actor_index(blk), 0:3].cpu().numpy()
        q = sim._dof_state[0, 0::2].cpu().numpy()
        cost = float(planner.objective.compute_cost(sim)[0])
        print(f"{i:5d} hand {np.round(hand, 3)} block {np.round(b, 3)} |hand-block| {np.linalg.norm(hand - b):.3f} q {np.round(q, 2)} cost {cost:.2f}")
run.run_world(name, cfg, planner, steps, report=False, hook=hook)

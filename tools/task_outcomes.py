#!/usr/bin/env python3
"""Do the reference's example tasks get DONE on this backend?  Every example of examples/run.py in closed loop through the bytes API
(reference examples/<name>/world.py + planner.py: a K = 1 world stepped from Python, MPPIisaacPlanner.compute_action_tensor), for
STEPS control iterations with the example's own conf/mppi parameters; reported: the example's own stage cost on the world state
and the task's distances (block -> goal, robot / end effector -> its target) at the start, every STEPS/6 iterations and at the end.
    python tools/task_outcomes.py [STEPS] [example ...] > profiles/r05x_task_outcomes.txt"""
import importlib.util
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mppi-isaac_amd"))
import logging  # noqa: E402

logging.disable(logging.WARNING)
import numpy as np  # noqa: E402

spec = importlib.util.spec_from_file_location("examples_run", os.path.join(ROOT, "mppi-isaac_amd", "examples", "run.py"))
run = importlib.util.module_from_spec(spec)
spec.loader.exec_module(run)

args = sys.argv[1:]
steps = int(args.pop(0)) if args and args[0].isdigit() else 1200
names = args or sorted(run.EXAMPLES)


def pos(sim, actor):
    return sim._root_state[0, sim.scene.actor_index(actor), 0:3].cpu().numpy()


LINKS = {"boxer": "ee_link", "heijn": "front_link", "panda": "panda_ee", "albert": "panda_ee", "omnipanda": "panda_hand", "jackal_a": "base_link", "anymal": "base"}


def distances(name, sim):
    """the task's own distances [m]: block -> goal (xy), and the link the example's Objective watches -> the block, or -> the goal"""
    names = [a.name for a in sim.scene.env_cfg]
    out = {}
    block = next((n for n in names if "block" in n), None)
    if block and "goal" in names:
        out["block->goal (xy)"] = float(np.linalg.norm(pos(sim, block)[:2] - pos(sim, "goal")[:2]))
    robot = names[sim.scene.robot_idx]
    link = LINKS.get(robot)
    try:
        p = sim.get_actor_link_by_name(robot, link)[0, 0:3].cpu().numpy()
    except Exception:
        p, link = pos(sim, robot), "base"
    target = block or ("goal" if "goal" in names else None)
    if target:
        n3 = 3 if robot in ("panda", "albert", "omnipanda") else 2
        out[f"{robot}:{link}->{target}"] = float(np.linalg.norm(p[:n3] - pos(sim, target)[:n3]))
    if block:
        out["block z"] = float(pos(sim, block)[2])
    return out


for name in names:
    cfg = run.config(name)
    planner = run.make_planner(name, cfg)
    log = []

    def hook(i, sim, log=log, name=name):
        if i % max(1, steps // 6) == 0 or i == steps - 1:
            log.append((i, distances(name, sim)))
    t0 = time.perf_counter()
    first, last, rate = run.run_world(name, cfg, planner, steps, report=False, hook=hook)
    print(f"{name}: K = {cfg.mppi.num_samples}, H = {cfg.mppi.horizon}; {steps} iterations at {rate:.0f} Hz through the bytes API; the example's stage cost on the world state {first:.4f} -> {last:.4f}")
    for i, d in log:
        print(f"    iteration {i:5d}: " + ", ".join(f"{k} {v:.3f} m" for k, v in d.items()))
    planner.sim.stop_sim()

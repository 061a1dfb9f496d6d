"""trace of a pushing example in closed loop (positions every N iterations): python tools/exp/push_trace.py boxer_push 400 20"""
import importlib.util, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, os.path.join(ROOT, "mppi-isaac_amd"))
import logging; logging.disable(logging.WARNING)
import numpy as np
spec = importlib.util.spec_from_file_location("examples_run", os.path.join(ROOT, "mppi-isaac_amd", "examples", "run.py"))
run = importlib.util.module_from_spec(spec); spec.loader.exec_module(run)
name, steps, every = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
cfg = run.config(name)
planner = run.make_planner(name, cfg)
yaw = lambda q: float(np.arctan2(2 * (q[3] * q[2] + q[0] * q[1]), 1 - 2 * (q[1] ** 2 + q[2] ** 2)))
def hook(i, sim):
    if i % every == 0 or i == steps - 1:
        rs = sim._root_state[0].cpu().numpy()
        names = [a.name for a in sim.scene.env_cfg]
        if i == 0:
            for n, r in zip(names, rs): print(f"   {n:14s} at {np.round(r[:3], 3)}  size {getattr(sim.scene.env_cfg[names.index(n)], 'size', None)}")
        dofs = sim._dof_state[0].cpu().numpy()
        out = []
        for n in names:
            if sim.scene.env_cfg[names.index(n)].type == "robot" or "block" in n:
                r = rs[names.index(n)]
                out.append(f"{n} ({r[0]:.3f}, {r[1]:.3f}, z {r[2]:.3f}, yaw {yaw(r[3:7]):.2f})")
        cf = sim.get_actor_contact_forces_by_name("block", "box")[0].cpu().numpy() if "block" in names else None
        print(f"  it {i:4d}: " + "; ".join(out) + f"; dof q {np.round(dofs[0::2], 2)}; force on block {np.round(cf, 1) if cf is not None else ''}")
first, last, rate = run.run_world(name, cfg, planner, steps, report=False, hook=hook)
print(f"{name}: stage cost {first:.3f} -> {last:.3f}, {rate:.0f} Hz")

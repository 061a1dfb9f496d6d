"""K = 1 world of an example stepped on the GPU (closed loop) against the fp64 oracle step from the same state and command, every iteration:
    python tools/exp/world_step_check.py panda_pick 200        -> first iterations where they part, worst difference"""
import importlib.util, logging, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mppi-isaac_amd"), os.path.join(ROOT, "tests")]
logging.disable(logging.WARNING)
spec = importlib.util.spec_from_file_location("examples_run", os.path.join(ROOT, "mppi-isaac_amd", "examples", "run.py"))
run = importlib.util.module_from_spec(spec); spec.loader.exec_module(run)
from oracle.oracle import Oracle
name, steps = sys.argv[1], int(sys.argv[2])
cfg = run.config(name)
planner = run.make_planner(name, cfg)
o = Oracle("f64")
prev = {}
worst = []
def hook(i, sim):
    dof = sim._dof_state[0].cpu().numpy().astype(float).copy(); root = sim._root_state[0].cpu().numpy().astype(float).copy()
    if prev:
        m = sim.scene.to_c()
        u = np.asarray(sim._last_cmd, float)     # (the command of THIS iteration took the previous state here)
        ro, q, qd, cf = o.scene_step(m, prev["root"].copy(), prev["dof"][0::2].copy(), prev["dof"][1::2].copy(), o.cmd_map(m, u))
        e = max(np.abs(ro[:, :3] - root[:, :3]).max(), np.abs(q - dof[0::2]).max())
        worst.append((e, i))
        if e > 1e-3 or not np.isfinite(e):
            print(f"iteration {i}: world vs oracle step differ by {e:.3e}; block gpu {np.round(root[3, :3], 4)} oracle {np.round(ro[3, :3], 4)} before {np.round(prev['root'][3, :3], 4)} q err {np.abs(q - dof[0::2]).max():.2e}")
    prev["dof"], prev["root"] = dof, root
orig = planner.sim.__class__.apply_robot_cmd
def spy(self, u, *a, **k):
    try:
        self._last_cmd = np.asarray(u.detach().cpu().numpy() if hasattr(u, "detach") else u, float).reshape(-1).copy()
    except Exception:
        pass
    return orig(self, u, *a, **k)
planner.sim.__class__.apply_robot_cmd = spy
run.run_world(name, cfg, planner, steps, report=False, hook=hook)
print("worst", sorted(worst, reverse=True)[:5])

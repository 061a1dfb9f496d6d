"""closed loop of one example on the GPU, every iteration's world state dumped: python tools/exp/stick_debug.py <name> <steps> <out.npz> [explicit]"""
import importlib.util, logging, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mppi-isaac_amd"))
logging.disable(logging.WARNING)
spec = importlib.util.spec_from_file_location("examples_run", os.path.join(ROOT, "mppi-isaac_amd", "examples", "run.py"))
run = importlib.util.module_from_spec(spec); spec.loader.exec_module(run)
from mppiisaac.planner.isaacgym_wrapper import Scene
name, steps, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
Scene.EXPLICIT_LIGHT = "explicit" in sys.argv
cfg = run.config(name)
planner = run.make_planner(name, cfg)
_orig = planner.sim.__class__.apply_robot_cmd
def _spy(self, u, *a, **k):
    self._last_cmd = np.asarray(u.detach().cpu().numpy() if hasattr(u, "detach") else u, float).reshape(-1).copy()
    return _orig(self, u, *a, **k)
planner.sim.__class__.apply_robot_cmd = _spy
dofs, roots, cmds = [], [], []
def hook(i, sim):
    dofs.append(sim._dof_state[0].cpu().numpy().copy()); roots.append(sim._root_state[0].cpu().numpy().copy()); cmds.append(np.asarray(getattr(sim, '_last_cmd', np.zeros(1)), float))
first, last, rate = run.run_world(name, cfg, planner, steps, report=False, hook=hook)
np.savez_compressed(out, dof=np.array(dofs), root=np.array(roots), cmd=np.array(cmds))
print(name, "cost", first, "->", last, "rate", rate)
planner.sim.stop_sim()

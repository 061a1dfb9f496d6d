"""How often does panda_pick pick its block up, hold it, lose it?  The example in closed loop (tools/task_outcomes.py) from N start
positions of the block - its init_pos moved by up to 10 mm in x / y before the first iteration - STEPS iterations each:
    python tools/exp/pick_robustness.py [N=8] [STEPS=600] [example=panda_pick]
per run: the highest the block got above where it lay, where it is at the end (held / on the table / off the table), block -> goal."""
import importlib.util, logging, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mppi-isaac_amd"))
logging.disable(logging.WARNING)
spec = importlib.util.spec_from_file_location("examples_run", os.path.join(ROOT, "mppi-isaac_amd", "examples", "run.py"))
run = importlib.util.module_from_spec(spec); spec.loader.exec_module(run)
from mppiisaac.utils.transport import bytes_to_torch, torch_to_bytes
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 600
name = sys.argv[3] if len(sys.argv) > 3 else "panda_pick"
rng = np.random.default_rng(1)
offs = [(0.0, 0.0)] + [tuple(rng.uniform(-0.01, 0.01, 2)) for _ in range(N - 1)]
summary = []
for k, (dx, dy) in enumerate(offs):
    cfg = run.config(name)
    planner = run.make_planner(name, cfg)
    sim = run.make_world(name, cfg)
    names = [a.name for a in sim.scene.env_cfg]
    blk = next(n for n in names if "block" in n)
    bi, gi = sim.scene.actor_index(blk), sim.scene.actor_index("goal")
    p0 = sim._root_state[0, bi, 0:3].cpu().numpy().copy()
    sim.set_actor_position_by_name([float(p0[0] + dx), float(p0[1] + dy), float(p0[2])], blk)
    zs, ds = [], []
    for i in range(STEPS):
        action = bytes_to_torch(planner.compute_action_tensor(torch_to_bytes(sim._dof_state), torch_to_bytes(sim._root_state)))
        sim.apply_robot_cmd(action.to(sim.device).reshape(1, -1))
        sim.step()
        r = sim._root_state[0].cpu().numpy()
        zs.append(float(r[bi, 2])); ds.append(float(np.linalg.norm(r[bi, :2] - r[gi, :2])))
    zs, ds = np.array(zs), np.array(ds)
    rest = zs[40:80].min()
    up = zs[80:].max() - rest
    frac_up = float(np.mean(zs[80:] > rest + 0.05))
    end = "in the air" if zs[-1] > rest + 0.05 else ("on the table" if zs[-1] > rest - 0.02 else "OFF the table")
    print(f"start ({dx * 1e3:+5.1f}, {dy * 1e3:+5.1f}) mm: highest {up * 100:5.1f} cm above the table, {frac_up * 100:3.0f} % of the iterations > 5 cm up, "
          f"at the end {end} (z {zs[-1]:.3f}), block -> goal {ds[0]:.2f} -> {ds[-1]:.2f} m (closest {ds.min():.2f}), finite {bool(np.isfinite(zs).all())}", flush=True)
    summary.append((up, end))
    sim.stop_sim(); planner.sim.stop_sim()
print(f"{sum(u > 0.10 for u, _ in summary)} of {len(summary)} runs lift the block more than 10 cm; at the end: "
      + ", ".join(f"{sum(e == w for _, e in summary)} {w}" for w in ("in the air", "on the table", "OFF the table")))

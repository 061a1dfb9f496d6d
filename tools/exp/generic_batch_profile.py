"""Where a control iteration of the generic Objective mode goes (panda reach K=4096 H=20, Python Objective): experiment."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mppi-isaac_amd"))
import torch
from mppiisaac.objectives import PandaReachObjective
from mppiisaac.planner.mppi_isaac import MPPIisaacPlanner
from mppiisaac.utils.config_store import load_config
class G(PandaReachObjective):
    fused_spec = None
cfg = load_config({"defaults": [{"mppi": "panda"}, {"isaacgym": "normal"}], "actors": ["panda_stick", "goal"], "initial_actor_positions": [[0.0, 0.0, 0.0]], "nx": 14}, overrides={"mppi.num_samples": 4096, "mppi.horizon": 20})
pl = MPPIisaacPlanner(cfg, G(cfg))
pl.sim.set_actor_position_by_name([0.5, -0.4, 0.3], "goal")
q = [0.0, -0.94, 0.0, -2.8, 0.0, 1.8675, 0.0]
for _ in range(3): pl.compute_action(q, [0.0]*7)
m = pl.mppi
def timed(f, n=50):
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return 1e3*(time.perf_counter()-t)/n
print("compute_action total", timed(lambda: pl.compute_action(q, [0.0]*7)))
print("simulated horizon (fused rollout with state dump + one materialise over H*K env-steps)", timed(lambda: m._simulate_horizon()))
b = m._batch_buf
def cost():
    with pl.sim._horizon_view(b, m.T*m.K):
        return m._running_cost(None)
print("cost call over H*K rows", timed(cost))
c = cost()
print("discount+sum", timed(lambda: (c.view(m.T, m.K) * m._batch_disc).sum(0).contiguous()))
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(5): cost()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=14, max_name_column_width=60))

"""Generic Objective mode on the contact scenes (what an unmodified reference example runs): control rate with the
Python compute_cost(sim) per horizon step, next to the fused mode.  Experiment."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(__file__), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mppi-isaac_amd"))
import torch
import bench
import mppiisaac.objectives as objectives
from mppiisaac.planner.mppi_isaac import MPPIisaacPlanner


def rate(workload, generic, n=30):
    wl = bench.WORKLOADS[workload]
    cfg = bench.make_cfg(wl, wl["K"])
    cfg.mppi.device = "cuda:0"
    base = getattr(objectives, wl["objective"])

    class Generic(base):
        fused_spec = None
    planner = MPPIisaacPlanner(cfg, (Generic if generic else base)(cfg))
    planner._bind_objective()
    for _ in range(3): planner.command()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): planner.command()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / n
    print(f"{workload:12s} {'generic' if generic else 'fused  '} {1/dt:8.1f} Hz ({1e3*dt:.2f} ms per command(), K={wl['K']} H={wl['H']})", flush=True)

for w in ("boxer_push", "panda_pick"):
    rate(w, False)
    rate(w, True)

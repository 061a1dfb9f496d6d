"""what does a validation of a traced Objective cost? (MPPIPlanner._trace_check: every TRACE_RECHECK-th command) - stage clock"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mppi-isaac_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from bench import ReferenceStyleReach
from test_gpu_trace import _planner, Q0
from mppiisaac.backend import capi
p = _planner(ReferenceStyleReach(), K=4096, H=20)
mp = p.mppi
lib, ctx = mp._lib, mp._ctx
rows = []
def timed_check(state):
    t = [time.perf_counter()]
    capi.check(lib, lib.mppi_sim_reset(ctx)); t.append(time.perf_counter())
    mp.sim._needs_reset = False
    done = mp._horizon_batched(state); t.append(time.perf_counter())
    if done != "reduced":
        capi.check(lib, lib.mppi_sim_finish(ctx))
    mp.sim._stale = True
    S_py = mp.get_costs(); t.append(time.perf_counter())
    capi.check(lib, lib.mppi_rollout(ctx)); t.append(time.perf_counter())
    S_k = mp.get_costs(); t.append(time.perf_counter())
    rows.append(np.diff(t) * 1e3)
    return bool((S_py - S_k).abs().max() <= 2e-3 * S_py.abs().max())
mp._trace_check = timed_check
mp.TRACE_RECHECK = 8
ts = []
for i in range(200):
    t0 = time.perf_counter(); p.compute_action(Q0, [0.0] * 7); ts.append((time.perf_counter() - t0) * 1e3)
print("per command ms: median %.3f" % np.median(ts))
print("check: sim_reset | horizon_batched | get_costs | rollout | get_costs   [ms]")
for r in rows: print("   " + "  ".join("%7.3f" % v for v in r))

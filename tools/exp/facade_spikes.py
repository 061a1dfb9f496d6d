"""where do the latency spikes of the reference-API loop with a TRACED Objective come from?  stage clock per iteration"""
import gc, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mppi-isaac_amd"))
import torch
import bench
from mppiisaac.planner.isaacgym_wrapper import IsaacGymWrapper
from mppiisaac.planner.mppi_isaac import MPPIisaacPlanner
from mppiisaac.utils.transport import bytes_to_torch, torch_to_bytes
wl = bench.WORKLOADS["panda_reach"]
cfg = bench.make_cfg(wl, wl["K"]); cfg.mppi.device = "cuda:0"
planner = MPPIisaacPlanner(cfg, bench.ReferenceStyleReach())
world = IsaacGymWrapper(cfg.isaacgym, actors=cfg.actors, init_positions=cfg.initial_actor_positions, num_envs=1, device="cuda:0")
for sim in (planner.sim, world): sim.set_actor_position_by_name(wl["goal"], "goal")
dof0 = world._dof_state[0].cpu().numpy().copy(); dof0[0::2] = wl["q0"]
world._push_single_state(dof0, world._root_state[0].cpu().numpy())
if os.environ.get("NO_D2H"):   # experiment: the validation without reading the costs back
    import torch as _t
    planner.mppi.get_costs = lambda: _t.zeros(planner.mppi.K)
if os.environ.get("SYNC_AFTER"):  # experiment: a device-wide synchronise at the end of every validation
    _orig = planner.mppi._trace_check
    def _chk(state):
        ok = _orig(state); torch.cuda.synchronize(); return ok
    planner.mppi._trace_check = _chk
if os.environ.get("NO_TORCH_COST"):   # experiment: the validation's library launches only, the Objective replaced by zeros
    z = torch.zeros(planner.mppi.T * planner.mppi.K, dtype=torch.float32, device="cuda:0")
    planner.mppi._horizon_costs = lambda state, b, single, fold=None: z
if os.environ.get("ONLY_TORCH"):      # experiment: no validation at all, but a burst of 80 torch kernels every 64th iteration
    planner.mppi.TRACE_RECHECK = 10**9
    xx = torch.randn(81920, 9, device="cuda:0")
STAGES = os.environ.get("LIB_STAGES")    # experiment: no validation, selected library calls of one every 64th iteration
if STAGES:
    import ctypes as C
    from mppiisaac.backend import capi
    planner.mppi.TRACE_RECHECK = 10**9
    lib, ctx = planner.mppi._lib, planner.mppi._ctx
    link_rows = torch.zeros((planner.mppi.T * planner.mppi.K, 13), dtype=torch.float32, device="cuda:0")
    zc = torch.zeros(planner.mppi.T * planner.mppi.K, dtype=torch.float32, device="cuda:0")
    def fake_validation():
        if "r" in STAGES: capi.check(lib, lib.mppi_sim_reset(ctx))
        if "t" in STAGES: capi.check(lib, lib.mppi_rollout_trajectory(ctx))
        if "m" in STAGES: capi.check(lib, lib.mppi_materialise_trajectory_link(ctx, planner.sim.scene.rigid_body_index("panda", "panda_ee_tip"), C.c_void_p(link_rows.data_ptr())))
        if "h" in STAGES: capi.check(lib, lib.mppi_reduce_horizon_costs(ctx, C.c_void_p(zc.data_ptr()), None))
        if "f" in STAGES: capi.check(lib, lib.mppi_rollout(ctx))
        if "c" in STAGES: planner.mppi.get_costs()
rows = []
if os.environ.get("SIDE_STREAM"):   # experiment: the whole validation on a stream of its own
    import ctypes as C
    from mppiisaac.backend import capi
    side = torch.cuda.Stream()
    _orig_chk = planner.mppi._trace_check
    def _side_chk(state):
        mp = planner.mppi
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        capi.check(mp._lib, mp._lib.mppi_set_stream(mp._ctx, C.c_void_p(side.cuda_stream)))
        try:
            with torch.cuda.stream(side):
                ok = _orig_chk(state)
            side.synchronize()
        finally:
            capi.check(mp._lib, mp._lib.mppi_set_stream(mp._ctx, C.c_void_p(main.cuda_stream)))
        capi.check(mp._lib, mp._lib.mppi_rollout(mp._ctx))   # (this command's rollout again, on the main stream)
        return ok
    planner.mppi._trace_check = _side_chk
import faulthandler, signal, traceback
caught = []
def _on_alarm(signum, frame):
    caught.append("".join(traceback.format_stack(frame, limit=6)))
if os.environ.get("ALARM"):
    signal.signal(signal.SIGALRM, _on_alarm)
def iterate():
    if os.environ.get("ALARM"): signal.setitimer(signal.ITIMER_REAL, 0.02)
    if os.environ.get("WATCHDOG"):
        faulthandler.cancel_dump_traceback_later(); faulthandler.dump_traceback_later(0.02, repeat=False, file=sys.stderr)
    t0 = time.perf_counter()
    a, b = torch_to_bytes(world._dof_state), torch_to_bytes(world._root_state); t1 = time.perf_counter()
    act = planner.compute_action_tensor(a, b); t2 = time.perf_counter()
    action = bytes_to_torch(act); world.apply_robot_cmd(action); world.step(); t3 = time.perf_counter()
    if os.environ.get("ALARM"): signal.setitimer(signal.ITIMER_REAL, 0)
    rows.append((t1 - t0, t2 - t1, t3 - t2))
    if STAGES and len(rows) % 64 == 0: fake_validation()
    if os.environ.get("ONLY_TORCH") and len(rows) % 64 == 0:
        y = xx
        for _ in range(int(os.environ.get("BURST", "80"))): y = y * 1.0001 + 0.1
for _ in range(30): iterate()
if os.environ.get("NO_RECHECK"): planner.mppi.TRACE_RECHECK = 10**9
if os.environ.get("EMPTY_CACHE"): torch.cuda.empty_cache()
torch.cuda.synchronize(); gc.collect(); gc.disable(); rows.clear()
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 400): iterate()
r = np.array(rows) * 1e3
print("median ms: world bytes %.3f planner %.3f apply+step %.3f" % tuple(np.median(r, 0)))
for i in np.where(r.sum(1) > 1.0)[0]: print("iteration", i, "stages", np.round(r[i], 2), "checks so far", len(getattr(planner.mppi, "trace_check_ms", [])))

for c in caught[-4:]: print("ALARM stack:\n" + c)

"""What the per-step (not per-substep) part of k_rollout_quad is made of: panda reach K=4096 H=20 with the rollout
visualisation off and with a trivial stage cost.  Experiment."""
import ctypes as C, os, sys
ROOT = os.path.join(os.path.dirname(__file__), "..", "..")
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "mppi-isaac_amd"))
import numpy as np
from mppiisaac.backend import capi
from scenes import panda_reach
lib = capi.load_library()

def t(name, viz=True, cheap_cost=False):
    scene, m, cfg, cost, dof, root = panda_reach(K=4096, H=20)
    cfg.want_rollouts = 1 if viz else 0
    if cheap_cost:
        cost.kind = capi.COST_POINT_REACH
    ctx = C.c_void_p()
    capi.check(lib, lib.mppi_create(C.byref(m), C.byref(cfg), 0, C.byref(ctx)))
    capi.check(lib, lib.mppi_set_cost(ctx, C.byref(cost)))
    d, r = np.ascontiguousarray(dof, np.float32), np.ascontiguousarray(root, np.float32)
    capi.check(lib, lib.mppi_set_state(ctx, capi.fptr(d), capi.fptr(r)))
    capi.check(lib, lib.mppi_sample(ctx, C.c_uint32(0)))
    for _ in range(20): capi.check(lib, lib.mppi_rollout(ctx))
    capi.check(lib, lib.mppi_set_profiling(ctx, 1))
    for _ in range(200): capi.check(lib, lib.mppi_rollout(ctx))
    ms = C.c_float()
    capi.check(lib, lib.mppi_kernel_ms(ctx, 0, C.byref(ms)))
    print(f"{name:34s} {1e3 * ms.value:7.1f} us", flush=True)
    lib.mppi_destroy(ctx)

t("reach cost + rollout visualisation")
t("reach cost, no visualisation", viz=False)
t("trivial cost + visualisation", cheap_cost=True)
t("trivial cost, no visualisation", viz=False, cheap_cost=True)

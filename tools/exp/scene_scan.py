"""Contact-scene rollout kernel time against horizon and substeps (open loop from the initial state): per-substep cost,
per-step overhead and what the kernel spends outside its loop.  Experiment."""
import ctypes as C, os, sys
ROOT = os.path.join(os.path.dirname(__file__), "..", "..")
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "mppi-isaac_amd"))
import numpy as np
from mppiisaac.backend import capi
from scenes import boxer_push, panda_pick
lib = capi.load_library()

def t(make, K, H, sub):
    scene, m, cfg, cost, dof, root = make(K=K, H=H)
    m.substeps = sub
    ctx = C.c_void_p()
    capi.check(lib, lib.mppi_create(C.byref(m), C.byref(cfg), 0, C.byref(ctx)))
    capi.check(lib, lib.mppi_set_cost(ctx, C.byref(cost)))
    d, r = np.ascontiguousarray(dof, np.float32), np.ascontiguousarray(root, np.float32)
    capi.check(lib, lib.mppi_set_state(ctx, capi.fptr(d), capi.fptr(r)))
    capi.check(lib, lib.mppi_sample(ctx, C.c_uint32(0)))
    for _ in range(5): capi.check(lib, lib.mppi_rollout(ctx))
    capi.check(lib, lib.mppi_set_profiling(ctx, 1))
    for _ in range(30): capi.check(lib, lib.mppi_rollout(ctx))
    ms = C.c_float()
    capi.check(lib, lib.mppi_kernel_ms(ctx, 0, C.byref(ms)))
    lib.mppi_destroy(ctx)
    return 1e3 * ms.value

for make, K in ((boxer_push, 8192), (panda_pick, 8192)):
    for H, sub in ((8, 2), (16, 2), (16, 1), (16, 4)):
        print(f"{make.__name__:11s} H={H:2d} substeps={sub}  {t(make, K, H, sub):8.1f} us", flush=True)

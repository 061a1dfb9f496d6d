"""A/B timing of the contact-scene rollout kernel with parts of the scene removed (experiment, not a test)."""
import ctypes as C, sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "mppi-isaac_amd"))
import numpy as np, torch
from mppiisaac.backend import capi
from scenes import boxer_push, panda_pick

lib = capi.load_library()

def run(name, make, K, H, edit=None):
    scene, m, cfg, cost, dof, root = make(K=K, H=H)
    if edit: edit(m)
    ctx = C.c_void_p()
    capi.check(lib, lib.mppi_create(C.byref(m), C.byref(cfg), 0, C.byref(ctx)))
    capi.check(lib, lib.mppi_set_cost(ctx, C.byref(cost)))
    d, r = np.ascontiguousarray(dof, np.float32), np.ascontiguousarray(root, np.float32)
    capi.check(lib, lib.mppi_set_state(ctx, capi.fptr(d), capi.fptr(r)))
    capi.check(lib, lib.mppi_sample(ctx, C.c_uint32(0)))
    for _ in range(3): capi.check(lib, lib.mppi_rollout(ctx))
    torch.cuda.synchronize()
    t = time.perf_counter()
    n = 20
    for _ in range(n): capi.check(lib, lib.mppi_rollout(ctx))
    capi.check(lib, lib.mppi_synchronize(ctx))
    print(f"{name:40s} {1e3 * (time.perf_counter() - t) / n:8.3f} ms", flush=True)
    lib.mppi_destroy(ctx)

def no_pairs(m): m.n_pairs = 0
def only(ks):
    def f(m):
        keep = [m.pairs[i] for i in ks]
        for i, p in enumerate(keep): m.pairs[i] = p
        m.n_pairs = len(keep)
    return f
for make, K, H in ((boxer_push, 8192, 25), (panda_pick, 8192, 30)):
    run(make.__name__ + " full", make, K, H)
    run(make.__name__ + " no pairs", make, K, H, no_pairs)
run("panda_pick block pairs only (ground+table)", panda_pick, 8192, 30, only([21, 22]))

"""panda_pick rollout kernel with parts of the scene switched off (open loop from the initial state): what the contact-free
part of a substep is made of.  Experiment."""
import ctypes as C, os, sys
ROOT = os.path.join(os.path.dirname(__file__), "..", "..")
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "mppi-isaac_amd"))
import numpy as np
from mppiisaac.backend import capi
from scenes import panda_pick
lib = capi.load_library()

def t(name, edit):
    scene, m, cfg, cost, dof, root = panda_pick(K=8192, H=30)
    edit(m, cfg)
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
    print(f"{name:44s} {1e3 * ms.value:8.1f} us", flush=True)

def nothing(m, c): pass
def no_pairs(m, c): m.n_pairs = 0
def no_pairs_shapes(m, c): m.n_pairs = 0; m.n_shapes = 0
def no_viz(m, c): m.n_pairs = 0; m.n_shapes = 0; c.want_rollouts = 0
t("full", nothing)
t("no pairs", no_pairs)
t("no pairs, no shapes (no pose cache)", no_pairs_shapes)
t("... and no rollout visualisation", no_viz)

"""Per-horizon-step time of the contact-free rollout kernel against the number of wavefronts in flight (K), quad and octet layout:
does a lone wavefront slow down when more SIMDs of the chip are busy (clock / shared front end)?  Experiment."""
import ctypes as C, os, sys
ROOT = os.path.join(os.path.dirname(__file__), "..", "..")
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "mppi-isaac_amd"))
import numpy as np
from mppiisaac.backend import capi
from scenes import panda_reach
lib = capi.load_library()
def t(K, H):
    scene, m, cfg, cost, dof, root = panda_reach(K=K, H=H)
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
    info = C.create_string_buffer(256); lib.mppi_kernel_info(ctx, info, 256)
    lib.mppi_destroy(ctx)
    return 1e3 * ms.value, dict(kv.split("=") for kv in info.value.decode().split())
for K in (256, 1024, 2048, 4096, 8192, 16384):
    a, info = t(K, 8); b, _ = t(K, 40)
    print(f"{info['rollout']:5s} K={K:6d} waves={info['waves']:>5s}: {(b - a) / 32:6.3f} us per horizon step, {a - 8 * (b - a) / 32:5.1f} us outside the loop", flush=True)

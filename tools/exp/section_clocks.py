"""Where does a wavefront of the contact-scene rollout kernel spend its time?  Needs the instrumented library:
    MPPI_BUILD_VARIANT=sec python __graft_entry__.py
    MPPI_HIP_LIB=$PWD/mppi-isaac_amd/csrc/libmppi_hip_sec.so python tools/exp/section_clocks.py [boxer_push panda_pick]
Shader-clock time per section (MPPI_SEC marks in csrc/mppi_scene*.hpp), per wavefront, at the recorded closed-loop states of
tests/golden/closed_loop_states.npz: mean share over all wavefronts and over the slowest 5 %."""
import ctypes as C
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mppi-isaac_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from mppiisaac.backend import capi
from scenes import boxer_push, panda_pick

NAMES = ["kinematics + frame stores", "acc clears + shape poses", "dealt broad phase", "pair loop (narrow phase, accumulate)", "inertias / bias (prepare)",
         "first articulated solve", "saturation check + second solve", "integration + free bodies", "controls + command map", "stage cost + viz",
         "init + record tail", "pair: record, poses, sizes", "pair: broad-phase arithmetic", "pair: contact law, velocities", "pair: feature points + cross-lane sum",
         "pair: accumulate into LDS rows"]
lib = capi.load_library()
lib.mppi_get_section_clock.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int]
for name in (sys.argv[1:] or ["boxer_push", "panda_pick"]):
    make, K, H = (boxer_push, 8192, 25) if name == "boxer_push" else (panda_pick, 8192, 30)
    scene, m, cfg, cost, dof, root = make(K=K, H=H)
    m.randomize_seed = 0
    z = np.load(os.path.join(ROOT, "tests", "golden", "closed_loop_states.npz"))   # (the states tools/exp/ab_time.py times at)
    st = os.environ.get("STATE", "recorded")   # (STATE=held: the gripper scene's second recorded state)
    dof, root, U = (np.ascontiguousarray(z[f"{name}_{st}_{k}"], np.float32) for k in ("dof", "root", "U"))
    ctx = C.c_void_p()
    capi.check(lib, lib.mppi_create(C.byref(m), C.byref(cfg), 0, C.byref(ctx)))
    capi.check(lib, lib.mppi_set_cost(ctx, C.byref(cost)))
    capi.check(lib, lib.mppi_set_state(ctx, capi.fptr(dof), capi.fptr(root)))
    capi.check(lib, lib.mppi_set_nominal(ctx, capi.fptr(U)))
    capi.check(lib, lib.mppi_sample(ctx, C.c_uint32(0)))
    capi.check(lib, lib.mppi_set_wave_clock(ctx, 1))
    for _ in range(2):
        capi.check(lib, lib.mppi_rollout(ctx))
    n = (K + 7) // 8
    sec = np.zeros((n, 16), np.uint64)
    capi.check(lib, lib.mppi_get_section_clock(ctx, sec.ctypes.data_as(C.POINTER(C.c_uint64)), n))
    sec = sec.astype(np.float64)
    tot = sec.sum(1)
    slow = tot >= np.percentile(tot, 95)
    print(f"{name}: {n} wavefronts; ticks per wavefront mean {tot.mean():.3e}, max {tot.max():.3e} (mean/max {tot.mean() / tot.max():.3f})")
    for j in range(16):
        print(f"   {NAMES[j]:40s} {100 * sec[:, j].sum() / tot.sum():5.1f} %   slowest 5 %: {100 * sec[slow, j].sum() / tot[slow].sum():5.1f} %   "
              f"(extra ticks of the slow ones: {sec[slow, j].mean() - sec[:, j].mean():+.2e})")
    lib.mppi_destroy(ctx)

"""Same-box A/B timing of the rollout kernels of several builds of the library at EQUAL state:
    python tools/exp/ab_time.py [workload,...] lib_a.so lib_b.so ...
workloads: panda_reach (K=4096, H=20), boxer_push / panda_pick (K=8192 at the recorded closed-loop state and nominal plan,
tests/golden/closed_loop_states.npz).  The builds are timed in turn, several rounds, and the minimum / median of the rounds'
means is printed (rollout launch + synchronise, 20 launches per round); the costs of the first build are the reference the
others are compared with (max relative difference), so a variant that changes results shows up here."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "mppi-isaac_amd"), os.path.join(ROOT, "tests")]
import numpy as np

from mppiisaac.backend import capi
from scenes import boxer_push, panda_pick, panda_reach, point_reach

args = sys.argv[1:]
workloads = ["panda_reach", "boxer_push", "panda_pick"]
if args and not args[0].endswith(".so"):
    workloads = args.pop(0).split(",")
libs = [(os.path.basename(p), capi.load_library(p)) for p in args] or [("product", capi.load_library())]
Z = np.load(os.path.join(ROOT, "tests", "golden", "closed_loop_states.npz"))
SPEC = {"panda_reach": (panda_reach, 4096, 20), "point_reach": (point_reach, 1024, 15), "boxer_push": (boxer_push, 8192, 25), "panda_pick": (panda_pick, 8192, 30)}
ROUNDS, REPS = int(os.environ.get("ROUNDS", 5)), int(os.environ.get("REPS", 20))

for w in workloads:
    make, K, H = SPEC[w]
    scene, m, cfg, cost, dof, root = make(K=K, H=H)
    U = np.zeros((H, cfg.nu), np.float32)
    st = os.environ.get("STATE", "recorded")   # (STATE=held: the gripper scene's second recorded state)
    if f"{w}_{st}_dof" in Z.files:
        dof, root, U = Z[f"{w}_{st}_dof"], Z[f"{w}_{st}_root"], Z[f"{w}_{st}_U"]
    ctxs = []
    for name, lib in libs:
        ctx = C.c_void_p()
        capi.check(lib, lib.mppi_create(C.byref(m), C.byref(cfg), 0, C.byref(ctx)))
        capi.check(lib, lib.mppi_set_cost(ctx, C.byref(cost)))
        d, r, u = np.ascontiguousarray(dof, np.float32), np.ascontiguousarray(root, np.float32), np.ascontiguousarray(U, np.float32)
        capi.check(lib, lib.mppi_set_state(ctx, capi.fptr(d), capi.fptr(r)))
        capi.check(lib, lib.mppi_set_nominal(ctx, capi.fptr(u)))
        capi.check(lib, lib.mppi_sample(ctx, C.c_uint32(0)))
        for _ in range(3):
            capi.check(lib, lib.mppi_rollout(ctx))
        capi.check(lib, lib.mppi_synchronize(ctx))
        S = np.zeros(K, np.float32)
        capi.check(lib, lib.mppi_get_costs(ctx, capi.fptr(S)))
        info = C.create_string_buffer(512)
        lib.mppi_kernel_info(ctx, info, 512)
        ctxs.append((name, lib, ctx, S, [], info.value.decode()))
    for _ in range(ROUNDS):
        for name, lib, ctx, S, ts, _ in ctxs:
            t = time.perf_counter()
            for _ in range(REPS):
                capi.check(lib, lib.mppi_rollout(ctx))
            capi.check(lib, lib.mppi_synchronize(ctx))
            ts.append(1e3 * (time.perf_counter() - t) / REPS)
    S0 = ctxs[0][3]
    for name, lib, ctx, S, ts, info in ctxs:
        rel = np.abs(S - S0) / np.abs(S0)
        print(f"{w:12s} {name:28s} min {min(ts):8.4f} ms  median {np.median(ts):8.4f} ms | vs first build: max rel {rel.max():.1e}, within 1e-3 {np.mean(rel <= 1e-3):.4f} | {info[:90]}", flush=True)
        lib.mppi_destroy(ctx)

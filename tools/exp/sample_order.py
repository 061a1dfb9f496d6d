"""Does the ORDER of the samples matter for the contact kernels?  One wavefront per SIMD at K = 8192: the kernel waits for its
slowest wavefront, and a wavefront waits for its busiest sample at every substep.  The same noise set in another order (columns
of eps permuted: the same rollouts, grouped differently - external noise, mppi_set_noise_dev) at the recorded closed-loop states:
natural (Halton) order, random orders, sorted by the samples' cost (heavy samples together), and dealt by cost (every wavefront
gets the same mix).  Prints the rollout time of each order."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "mppi-isaac_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import torch

from mppiisaac.backend import capi
from scenes import boxer_push, panda_pick

lib = capi.load_library()
Z = np.load(os.path.join(ROOT, "tests", "golden", "closed_loop_states.npz"))
for w, make, K, H in (("boxer_push", boxer_push, 8192, 25), ("panda_pick", panda_pick, 8192, 30)):
    scene, m, cfg, cost, dof, root = make(K=K, H=H)
    dof, root, U = Z[f"{w}_recorded_dof"], Z[f"{w}_recorded_root"], Z[f"{w}_recorded_U"]
    ctx = C.c_void_p()
    capi.check(lib, lib.mppi_create(C.byref(m), C.byref(cfg), 0, C.byref(ctx)))
    capi.check(lib, lib.mppi_set_cost(ctx, C.byref(cost)))
    f = lambda a: np.ascontiguousarray(a, np.float32)
    d, r, u = f(dof), f(root), f(U)
    capi.check(lib, lib.mppi_set_state(ctx, capi.fptr(d), capi.fptr(r)))
    capi.check(lib, lib.mppi_set_nominal(ctx, capi.fptr(u)))
    capi.check(lib, lib.mppi_sample(ctx, C.c_uint32(0)))
    nu = cfg.nu
    eps = np.zeros((H, nu, K), np.float32)
    capi.check(lib, lib.mppi_get_noise(ctx, capi.fptr(eps)))

    def timed(perm, label):
        e = torch.from_numpy(np.ascontiguousarray(eps[:, :, perm])).cuda()
        capi.check(lib, lib.mppi_set_noise_dev(ctx, C.c_void_p(e.data_ptr())))
        for _ in range(3):
            capi.check(lib, lib.mppi_rollout(ctx))
        capi.check(lib, lib.mppi_synchronize(ctx))
        ts = []
        for _ in range(5):
            t = time.perf_counter()
            for _ in range(20):
                capi.check(lib, lib.mppi_rollout(ctx))
            capi.check(lib, lib.mppi_synchronize(ctx))
            ts.append(1e3 * (time.perf_counter() - t) / 20)
        S = np.zeros(K, np.float32)
        capi.check(lib, lib.mppi_get_costs(ctx, capi.fptr(S)))
        print(f"{w:11s} {label:46s} {min(ts):.4f} ms", flush=True)
        return S
    nat = np.arange(K)
    S = timed(nat, "natural (Halton) order")
    rng = np.random.default_rng(0)
    for i in range(3):
        timed(rng.permutation(K), f"random order {i}")
    order = np.argsort(S)                                   # (cost as a proxy of how violent a rollout is)
    timed(order, "sorted by cost (like with like)")
    dealt = order.reshape(8, K // 8).T.reshape(-1)          # sample j of every wavefront from the j-th octile of the costs
    timed(dealt, "dealt by cost (every wavefront the same mix)")
    lib.mppi_destroy(ctx)

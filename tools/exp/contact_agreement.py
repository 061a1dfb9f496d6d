"""How tightly do the rollout kernels agree with the oracle (and with each other) on the contact scenes at BASELINE size,
with and without the touch-down ramp of the contact law?  (experiment: evidence for the tolerances in tests/)"""
import ctypes as C
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "mppi-isaac_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

from mppiisaac.backend import capi
from mppiisaac.planner.isaacgym_wrapper import Scene
from mppiisaac.planner.mppi import make_config
from mppiisaac.utils.config_store import load_config
from oracle.oracle import Oracle
from scenes import boxer_push, panda_pick

lib = capi.load_library()
o = Oracle("f64")


def costs(m, cfg, cost, dof, root, mode=None):
    if mode:
        os.environ["MPPI_ROLLOUT"] = mode
    ctx = C.c_void_p()
    capi.check(lib, lib.mppi_create(C.byref(m), C.byref(cfg), 0, C.byref(ctx)))
    os.environ.pop("MPPI_ROLLOUT", None)
    capi.check(lib, lib.mppi_set_cost(ctx, C.byref(cost)))
    d, r = np.ascontiguousarray(dof, np.float32), np.ascontiguousarray(root, np.float32)
    capi.check(lib, lib.mppi_set_state(ctx, capi.fptr(d), capi.fptr(r)))
    capi.check(lib, lib.mppi_sample(ctx, C.c_uint32(0)))
    capi.check(lib, lib.mppi_rollout(ctx))
    S = np.zeros(cfg.num_samples, np.float32)
    capi.check(lib, lib.mppi_get_costs(ctx, capi.fptr(S)))
    eps = np.zeros((cfg.horizon, cfg.nu, cfg.num_samples), np.float32)
    capi.check(lib, lib.mppi_get_noise(ctx, capi.fptr(eps)))
    lib.mppi_destroy(ctx)
    return S, eps


for ramp in (None,):
    Scene.CONTACT_RAMP_DEPTH = ramp
    for make, name, K, H in ((boxer_push, "boxer_push", 8192, 25), (panda_pick, "panda_pick", 8192, 30)):
        for label, state in (("initial", None), ("closed-loop", os.path.join(os.path.dirname(os.path.abspath(__file__)), "states", f"state_{name}.npz"))):
            scene, m, cfg, cost, dof, root = make(K=K, H=H)
            if state:
                z = np.load(state)
                dof, root = z["dof"], z["root"]
            S, eps = costs(m, cfg, cost, dof, root)
            Sq, _ = costs(m, cfg, cost, dof, root, "quad")
            Sl, _ = costs(m, cfg, cost, dof, root, "lane")
            ex = load_config({"defaults": [{"mppi": name}, {"isaacgym": "normal"}]}, overrides={"mppi.num_samples": K, "mppi.horizon": H})
            idx = list(range(3, K, K // 96))
            rel = []
            for k in idx:
                sc = make_config(ex.mppi, k_offset=int(k), k_local=1, viz_link=scene.viz_link_index())
                So, _, _ = o.rollout(m, sc, cost, dof, root, np.zeros((H, cfg.nu)), eps[:, :, k:k + 1])
                rel.append(abs(S[k] - So[0]) / abs(So[0]))
            rel = np.array(rel)
            rl, rq = np.abs(S - Sl) / np.abs(Sl), np.abs(S - Sq) / np.abs(Sq)
            print(f"ramp={'on' if ramp is None else 'off'} {name:11s} {label:11s} vs oracle ({len(idx)} samples): median {np.median(rel):.1e} p90 {np.percentile(rel, 90):.1e} max {rel.max():.1e} "
                  f"within 1e-4: {(rel < 1e-4).mean():.3f} 1e-3: {(rel < 1e-3).mean():.3f} 1e-2: {(rel < 1e-2).mean():.3f} | oct vs lane within 1e-3: {(rl < 1e-3).mean():.4f} "
                  f"1e-2: {(rl < 1e-2).mean():.4f} max {rl.max():.1e} | oct vs quad within 1e-3: {(rq < 1e-3).mean():.4f} max {rq.max():.1e}", flush=True)

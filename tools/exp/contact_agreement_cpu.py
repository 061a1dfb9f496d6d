"""CPU-only proxy of tools/exp/contact_agreement.py: at the recorded closed-loop states, how far apart are the fp64 oracle, the
fp32 oracle and the host-emulated kernel arithmetic (lane / octet dealing)?  The fp32-vs-fp64 ORACLE gap isolates how much of the
disagreement is the contact model amplifying rounding (same code, same formulation, two precisions)."""
import ctypes as C
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "mppi-isaac_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

from oracle.oracle import Oracle
from scenes import boxer_push, panda_pick

o64, o32 = Oracle("f64"), Oracle("f32")
emu = C.CDLL(os.path.join(ROOT, "tests", "hostemu", "libhostemu.so"))
f32 = lambda a: np.ascontiguousarray(a, np.float32)
fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float)) if a is not None else None


def emu_rollout(m, cfg, cost, dof, root, U, eps, split=1):
    emu.emu_set_scene_split(split)
    d0, r0, U32, e32 = f32(dof), f32(root), f32(U), f32(eps)
    S = np.zeros(cfg.num_samples, np.float32)
    du = np.zeros_like(e32)
    rc = emu.emu_rollout(C.byref(m), C.byref(cfg), C.byref(cost), fp(d0), fp(r0), fp(U32), fp(e32), None, fp(S), fp(du), None)
    assert rc == 0
    emu.emu_set_scene_split(1)
    return S


def stats(a, b):
    rel = np.abs(a - b) / np.abs(b)
    return (f"median {np.median(rel):.1e} p90 {np.percentile(rel, 90):.1e} max {rel.max():.1e} within 1e-4: {(rel < 1e-4).mean():.3f} "
            f"1e-3: {(rel < 1e-3).mean():.3f} 1e-2: {(rel < 1e-2).mean():.3f}")


K = int(os.environ.get("K", 256))
which = sys.argv[1:] or ["boxer_push", "panda_pick"]
for make, name, H in ((boxer_push, "boxer_push", 25), (panda_pick, "panda_pick", 30)):
    if name not in which:
        continue
    for label in ("initial", "closed-loop"):
        scene, m, cfg, cost, dof, root = make(K=K, H=H)
        if label == "closed-loop":
            z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "states", f"state_{name}.npz"))
            dof, root = z["dof"], z["root"]
        eps = o64.sample(cfg)
        U = np.zeros((H, cfg.nu))
        S64, _, _ = o64.rollout(m, cfg, cost, dof, root, U, eps)
        S32, _, _ = o32.rollout(m, cfg, cost, dof, root, U, eps)
        Sl = emu_rollout(m, cfg, cost, dof, root, U, eps, 1)
        So = emu_rollout(m, cfg, cost, dof, root, U, eps, 8)
        print(f"{name:11s} {label:11s} K={K}")
        print("   oracle f32 vs f64 :", stats(S32, S64))
        print("   emu lane  vs f64  :", stats(Sl, S64))
        print("   emu octet vs f64  :", stats(So, S64))
        print("   emu octet vs lane :", stats(So, Sl), flush=True)

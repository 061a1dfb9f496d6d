"""How often do joint drives saturate at their effort limit in the gripper scene, and how well does the previous substep's
set predict the next one?  (CPU experiment on the oracle: sizing of the speculative second solve of the octet kernel)"""
import ctypes as C
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mppi-isaac_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle.oracle import Oracle
from scenes import panda_pick

o = Oracle("f32")
K, H = 512, 30
scene, m, cfg, cost, dof, root = panda_pick(K=K, H=H)
for label, state in (("initial", None), ("closed-loop", os.path.join(os.path.dirname(os.path.abspath(__file__)), "states", "state_panda_pick.npz"))):
    U = np.zeros((H, cfg.nu), np.float32)
    if state:
        z = np.load(state)
        dof, root, U = z["dof"], z["root"], z["U"]
    eps = o.sample(cfg)
    nsub = H * m.substeps
    logs = np.zeros((K, nsub), np.uint32)
    o.lib.orc_rollout_satlog.restype = C.c_float
    for k in range(K):
        o.lib.orc_rollout_satlog(C.byref(m), C.byref(cfg), C.byref(cost), o.p(o.arr(dof)), o.p(o.arr(root)), o.p(o.arr(U)), o.p(o.arr(eps)), C.c_int(k),
                                 logs[k].ctypes.data_as(C.POINTER(C.c_uint32)))
    sat = logs != 0
    prev = np.concatenate([np.zeros((K, 1), np.uint32), logs[:, :-1]], 1)                # previous substep's set
    prev2 = np.concatenate([np.zeros((K, 2), np.uint32), logs[:, :-2]], 1)               # same substep of the previous step
    print(f"{label}: sample-substeps with a saturated drive {sat.mean():.3f}; joints hit:", {i: round(float(((logs >> i) & 1).mean()), 3) for i in range(m.n_bodies) if ((logs >> i) & 1).any()})
    for spw in (8, 16):
        w_any = sat.reshape(K // spw, spw, nsub).any(1)
        w_miss1 = (logs != prev).reshape(K // spw, spw, nsub).any(1)
        w_miss2 = (logs != prev2).reshape(K // spw, spw, nsub).any(1)
        print(f"   {spw} samples/wave: wave-substeps that re-solve today {w_any.mean():.3f}; with a speculative solve on the previous substep's set "
              f"{w_miss1.mean():.3f}; on the set of the same substep one step earlier {w_miss2.mean():.3f}")

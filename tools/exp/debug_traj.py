"""Debug: per-step dof trajectory of the quad rollout vs the oracle stepping, first deviation per sample (point robot)."""
import ctypes as C, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "mppi-isaac_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
from mppiisaac.backend import capi
from oracle.oracle import Oracle
from scenes import point_reach, panda_reach
import test_gpu_parity as tg
lib = capi.load_library()
o = Oracle("f64")
K, H = 256, 15
scene, m, cfg, cost, dof, root = point_reach(K=K, H=H)
c = tg.Ctx(m, cfg)
c.call("mppi_sample", C.c_uint32(0)); c.set_state(dof, root)
A, B, n = len(scene.env_cfg), scene.n_rb, scene.n_dof
f32 = dict(dtype=torch.float32, device="cuda")
T = {"dof": torch.zeros((H * K, 2 * n), **f32), "root": torch.zeros((H * K, A, 13), **f32), "rb": torch.zeros((H * K, B, 13), **f32), "cf": torch.zeros((H * K, B, 3), **f32)}
ptr = lambda t: C.c_void_p(t.data_ptr())
c.call("mppi_rollout_trajectory")
c.call("mppi_materialise_trajectory", ptr(T["dof"]), ptr(T["root"]), ptr(T["rb"]), ptr(T["cf"]))
nu = cfg.nu
du = c.get("mppi_get_perturbations", (H, nu, K))
dofs = T["dof"].cpu().numpy().reshape(H, K, 2 * n)
worst = 0
for k in range(0, K, 7):
    q, qd = dof[0::2].astype(np.float64), dof[1::2].astype(np.float64)
    for t in range(H):
        u = du[t, :, k].astype(np.float64)
        q0, qd0 = q, qd
        q, qd = o.step(m, root, q, qd, o.cmd_map(m, u))
        eq, ev = np.abs(dofs[t, k, 0::2] - q).max(), np.abs(dofs[t, k, 1::2] - qd).max()
        if max(eq, ev) > 1e-4 and worst < 6:
            worst += 1
            print(f"k={k} t={t} u={u} \n  q_prev={q0} qd_prev={qd0}\n  oracle q={q} qd={qd}\n  gpu    q={dofs[t, k, 0::2]} qd={dofs[t, k, 1::2]}")
            break
print("done; deviations printed:", worst)

"""ctypes loader of the CPU oracle (oracle/mppi_oracle.c).  TEST INFRASTRUCTURE ONLY: imported by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg - never by the product package.
It reuses the C-ABI struct mirrors of the product binding (mppiisaac.backend.capi) for its
arguments; the dependency points from the checker to the product, never the other way."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "mppi-isaac_amd"))
from mppiisaac.backend import capi  # noqa: E402


def build():
    subprocess.run(["make", "-C", HERE, "-s"], check=True)


class Oracle:
    def __init__(self, precision="f64"):
        # ORACLE_VARIANT=asan: the AddressSanitizer / UBSan build (make -C oracle asan; the process must have the sanitizer
        # runtime preloaded - tests/test_sanitizers.py)
        variant = os.environ.get("ORACLE_VARIANT", "")
        path = os.path.join(HERE, f"liboracle_{precision}{'_' + variant if variant else ''}.so")
        if not os.path.exists(path):
            subprocess.run(["make", "-C", HERE, "-s", variant], check=True) if variant else build()
        self.lib = C.CDLL(path)
        self.dtype = np.float64 if precision == "f64" else np.float32
        self.ctype = C.c_double if precision == "f64" else C.c_float
        assert self.lib.orc_sizeof_real() == np.dtype(self.dtype).itemsize
        assert self.lib.orc_sizeof_model() == C.sizeof(capi.Model), "mppi_model_t layout mismatch"
        assert self.lib.orc_sizeof_config() == C.sizeof(capi.Config), "mppi_config_t layout mismatch"
        assert self.lib.orc_sizeof_cost() == C.sizeof(capi.Cost), "mppi_cost_t layout mismatch"
        self.lib.orc_cost.restype = self.ctype
        self.lib.orc_halton.restype = C.c_double
        self.lib.orc_halton.argtypes = [C.c_uint32, C.c_int]
        self.lib.orc_norminv.restype = C.c_double
        self.lib.orc_norminv.argtypes = [C.c_double]

    def arr(self, a):
        return np.ascontiguousarray(a, dtype=self.dtype)

    def p(self, a):
        return a.ctypes.data_as(C.POINTER(self.ctype)) if a is not None else None

    def forward_dynamics(self, model, root, q, qd, tau):
        root, q, qd, tau = map(self.arr, (root, q, qd, tau))
        qdd = np.zeros_like(q)
        self.lib.orc_forward_dynamics(C.byref(model), self.p(root), self.p(q), self.p(qd), self.p(tau), self.p(qdd))
        return qdd

    def cmd_map(self, model, u):
        u = self.arr(u)
        t = np.zeros(model.n_bodies, self.dtype)
        self.lib.orc_cmd_map(C.byref(model), self.p(u), self.p(t))
        return t

    def step(self, model, root, q, qd, target):
        root, target = self.arr(root), self.arr(target)
        q, qd = self.arr(q).copy(), self.arr(qd).copy()
        self.lib.orc_step(C.byref(model), self.p(root), self.p(q), self.p(qd), self.p(target))
        return q, qd

    def rigid_body_state(self, model, root, q, qd):
        root, q, qd = map(self.arr, (root, q, qd))
        rb = np.zeros((model.n_rb, 13), self.dtype)
        cf = np.zeros((model.n_rb, 3), self.dtype)
        self.lib.orc_rigid_body_state(C.byref(model), self.p(root), self.p(q), self.p(qd), self.p(rb), self.p(cf))
        return rb, cf

    def cost(self, model, cost, root, q, qd, rb, cf=None):
        root, q, qd, rb = map(self.arr, (root, q, qd, rb))
        cf = self.arr(cf) if cf is not None else None
        return float(self.lib.orc_cost(C.byref(model), C.byref(cost), self.p(root), self.p(q), self.p(qd), self.p(rb), self.p(cf)))

    def is_scene(self, model):
        return bool(self.lib.orc_is_scene(C.byref(model)))

    def scene_step(self, model, root, q, qd, target):
        """one dt step of a contact scene -> (root [A,13], q, qd, cf [n_rb,3])"""
        root, q, qd = self.arr(root).copy(), self.arr(q).copy(), self.arr(qd).copy()
        target = self.arr(target)
        cf = np.zeros((model.n_rb, 3), self.dtype)
        self.lib.orc_scene_step(C.byref(model), self.p(root), self.p(q), self.p(qd), self.p(target), self.p(cf))
        return root, q, qd, cf

    def randomise_draws(self, model, g):
        """[n_actors, 5] size deltas xyz, mass scale, friction that sample g simulates"""
        out = np.zeros((model.n_actors, 5), np.float64)
        self.lib.orc_randomise_draws(C.byref(model), C.c_int(g), out.ctypes.data_as(C.c_void_p))
        return out

    def randomise_model(self, model, g):
        """the model sample g simulates (noisy actors take their per-sample size / mass / friction)"""
        out = type(model)()
        self.lib.orc_randomise_model(C.byref(model), C.c_int(g), C.byref(out))
        return out

    def sample(self, cfg, index_base=0):
        eps = np.zeros((cfg.horizon, cfg.nu, cfg.num_samples), self.dtype)
        self.lib.orc_sample(C.byref(cfg), C.c_uint32(index_base), self.p(eps))
        return eps

    def sample_normal(self, cfg, iteration=0):
        eps = np.zeros((cfg.horizon, cfg.nu, cfg.num_samples), self.dtype)
        self.lib.orc_sample_normal(C.byref(cfg), C.c_uint32(iteration), self.p(eps))
        return eps

    def philox(self, ctr, key):
        c, k, o = (C.c_uint32 * 4)(*ctr), (C.c_uint32 * 2)(*key), (C.c_uint32 * 4)()
        self.lib.orc_philox4x32(c, k, o)
        return [int(v) for v in o]

    def normal_knot(self, seed, iteration, g, c, i):
        self.lib.orc_normal_knot.restype = C.c_double
        return float(self.lib.orc_normal_knot(C.c_uint32(seed), C.c_uint32(iteration), C.c_uint32(g), C.c_int(c), C.c_int(i)))

    def rollout(self, model, cfg, cost, dof0, root0, U, eps, prior=None, want_viz=False):
        dof0, root0, U, eps = map(self.arr, (dof0, root0, U, eps))
        K, H, nu = cfg.num_samples, cfg.horizon, cfg.nu
        S = np.zeros(K, self.dtype)
        du = np.zeros((H, nu, K), self.dtype)
        viz = np.zeros((H, K, 3), self.dtype) if want_viz else None
        pr = self.arr(prior) if prior is not None else None
        self.lib.orc_rollout(C.byref(model), C.byref(cfg), C.byref(cost), self.p(dof0), self.p(root0), self.p(U), self.p(eps),
                             self.p(pr), self.p(S), self.p(du), self.p(viz))
        return S, du, viz

    def record(self, cfg, S, du):
        S, du = self.arr(S), self.arr(du)
        rec = np.zeros(2 + cfg.horizon * cfg.nu, self.dtype)
        self.lib.orc_record(C.byref(cfg), self.p(S), self.p(du), self.p(rec))
        return rec

    def update(self, cfg, recs, U):
        recs = self.arr(recs).reshape(-1, 2 + cfg.horizon * cfg.nu)
        U = self.arr(U).copy()
        action = np.zeros(cfg.nu, self.dtype)
        be = np.zeros(2, self.dtype)
        self.lib.orc_update(C.byref(cfg), self.p(recs), C.c_int(recs.shape[0]), self.p(U), self.p(action), self.p(be))
        return U, action, be

    def command(self, model, cfg, cost, dof0, root0, U, eps):
        dof0, root0, eps = map(self.arr, (dof0, root0, eps))
        U = self.arr(U).copy()
        K, H, nu = cfg.num_samples, cfg.horizon, cfg.nu
        S = np.zeros(K, self.dtype)
        du = np.zeros((H, nu, K), self.dtype)
        action = np.zeros(nu, self.dtype)
        self.lib.orc_command(C.byref(model), C.byref(cfg), C.byref(cost), self.p(dof0), self.p(root0), self.p(U), self.p(eps),
                             self.p(S), self.p(du), self.p(action))
        return U, action, S

"""The reference-structured CPU pipeline (SURVEY.md 8d "CPU baseline beside it"): the control iteration with the CALL
STRUCTURE of the reference - `mppi_torch.MPPIPlanner.command` driving `MPPIisaacPlanner.dynamics` / `.running_cost`
(reference mppiisaac/planner/mppi_isaac.py:57-69): a Python `for t in range(H)` of  apply command -> batched env step
(the oracle's C step, OpenMP over the K envs, standing in for Isaac Gym's CPU pipeline) -> the Objective's torch-CPU
`compute_cost(sim)` on reference-layout tensors, then the exp-weight update in torch.

TEST INFRASTRUCTURE / BASELINE ONLY (bench.py's cpu_baseline leg and tests): never imported by the product package."""
import ctypes as C
import time

import numpy as np
import torch

from oracle.oracle import Oracle


class CpuSim:
    """the sim-getter protocol Objectives read (reference isaacgym_wrapper.py:298-356) over CPU tensors"""

    def __init__(self, scene, K):
        self.scene, self.num_envs, self.device = scene, K, "cpu"
        self.env_cfg = scene.env_cfg
        n, A, B = scene.n_dof, len(scene.env_cfg), scene.n_rb
        self._dof_state = torch.zeros((K, 2 * n), dtype=torch.float32)
        self._root_state = torch.zeros((K, A, 13), dtype=torch.float32)
        self._rigid_body_state = torch.zeros((K, B, 13), dtype=torch.float32)
        self._net_contact_force = torch.zeros((K, B, 3), dtype=torch.float32)

    def get_actor_position_by_name(self, name):
        return self._root_state[:, self.scene.actor_index(name), 0:3]

    def get_actor_velocity_by_name(self, name):
        return self._root_state[:, self.scene.actor_index(name), 7:10]

    def get_actor_orientation_by_name(self, name):
        return self._root_state[:, self.scene.actor_index(name), 3:7]

    def get_actor_link_by_name(self, actor_name, link_name):
        return self._rigid_body_state[:, self.scene.rigid_body_index(actor_name, link_name), :]

    def get_actor_contact_forces_by_name(self, actor_name, link_name):
        return self._net_contact_force[:, self.scene.rigid_body_index(actor_name, link_name)]

    def get_dof_state(self):
        return self._dof_state


class CpuPipeline:
    def __init__(self, scene, c_model, c_cfg, objective, threads: int):
        self.o = Oracle("f32")
        self.o.lib.orc_set_threads(C.c_int(threads))
        torch.set_num_threads(threads)
        self.threads = threads
        self.m, self.cfg, self.objective = c_model, c_cfg, objective
        self.K, self.H, self.nu = c_cfg.num_samples, c_cfg.horizon, c_cfg.nu
        self.sim = CpuSim(scene, self.K)
        self.u_min = torch.tensor([c_cfg.u_min[j] for j in range(self.nu)], dtype=torch.float32)
        self.u_max = torch.tensor([c_cfg.u_max[j] for j in range(self.nu)], dtype=torch.float32)
        self.inv_sigma = torch.tensor([1.0 / c_cfg.noise_sigma_diag[j] for j in range(self.nu)], dtype=torch.float32)

    def _ptr(self, t):
        return C.cast(t.data_ptr(), C.POINTER(C.c_float))

    def command(self, dof0, root0, U, eps):
        """one control iteration; U [H,nu] and eps [H,nu,K] torch float32 (CPU).  Returns (U_new, action, S)."""
        cfg, sim, K, H = self.cfg, self.sim, self.K, self.H
        sim._dof_state[:] = torch.as_tensor(dof0, dtype=torch.float32).view(1, -1)          # reset_rollout_sim
        sim._root_state[:] = torch.as_tensor(root0, dtype=torch.float32).view(1, -1, 13)
        sim._net_contact_force.zero_()
        S = torch.zeros(K)
        ctrl = torch.zeros(K)
        du = torch.empty((H, K, self.nu))
        disc = 1.0
        for t in range(H):                                                                      # mppi_torch's horizon loop
            u = torch.clamp(U[t].view(1, -1) + eps[t].T, self.u_min, self.u_max)
            if cfg.sample_null_action and cfg.k_offset + K == cfg.k_total:
                u[-1] = torch.clamp(torch.zeros(self.nu), self.u_min, self.u_max)
            u = u.contiguous()
            du[t] = u - U[t].view(1, -1)
            ctrl += cfg.lambda_ * ((U[t] * self.inv_sigma).view(1, -1) * du[t]).sum(1)
            self.o.lib.orc_envs_step(C.byref(self.m), C.c_int(K), C.c_int(cfg.k_offset), self._ptr(u), self._ptr(sim._dof_state),   # dynamics(): apply + step
                                     self._ptr(sim._root_state), self._ptr(sim._rigid_body_state), self._ptr(sim._net_contact_force))
            S += disc * self.objective.compute_cost(sim)                                        # running_cost()
            disc *= cfg.rollout_var_discount
        S = S + ctrl
        fin = torch.isfinite(S)
        beta = S[fin].min()
        w = torch.where(fin, torch.exp(-(S - beta) / cfg.lambda_), torch.zeros(()))
        U_new = U + torch.einsum("k,tkc->tc", w, du) / w.sum()
        action = U_new[0].clone()
        U_new = torch.cat([U_new[1:], torch.full((1, self.nu), float(cfg.u_init))])
        return U_new, action, S

    def time_iterations(self, dof0, root0, eps, budget_s: float, max_iters: int = 20):
        U = torch.zeros((self.H, self.nu))
        t0 = time.perf_counter()
        U, a, S = self.command(dof0, root0, U, eps)
        first = time.perf_counter() - t0
        n = max(1, min(max_iters, int(budget_s / max(first, 1e-4)) - 1))
        t0 = time.perf_counter()
        for _ in range(n):
            U, a, S = self.command(dof0, root0, U, eps)
        return (time.perf_counter() - t0) / n, n


def _main():
    """one timed row in a clean process (bench.py's cpu_baseline leg): thread counts and the OpenMP wait policy are fixed by
    the environment BEFORE any runtime starts; prints one JSON object"""
    import argparse
    import json
    import os
    import sys
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", required=True)
    ap.add_argument("--k-total", type=int, required=True)
    ap.add_argument("--k-offset", type=int, default=0)
    ap.add_argument("--k-local", type=int, required=True)
    ap.add_argument("--horizon", type=int, required=True)
    ap.add_argument("--threads", type=int, required=True)
    ap.add_argument("--budget", type=float, default=6.0)
    ap.add_argument("--max-iters", type=int, default=10)
    ap.add_argument("--mode", choices=("pipeline", "fused"), default="pipeline")
    ap.add_argument("--eps", default="", help=".npy with the GPU run's noise [H, nu, K_local] (default: the oracle's sampler)")
    ap.add_argument("--state", default="", help=".npz with dof0 / root0 (default: the scene's initial state + the workload's q0 / goal)")
    a = ap.parse_args()
    root_dir = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root_dir)
    import bench
    import mppiisaac.objectives as objectives
    from mppiisaac.planner.isaacgym_wrapper import Scene
    from mppiisaac.planner.mppi import make_config
    from mppiisaac.utils.isaacgym_utils import load_actor_cfgs, load_asset
    wl = bench.WORKLOADS[a.workload]
    cfg = bench.make_cfg(wl, a.k_total, a.horizon)
    env_cfg = load_actor_cfgs(wl["actors"])
    robots = [x for x in env_cfg if x.type == "robot"]
    robots[0].init_pos = list(wl["init"][0])
    scene = Scene(env_cfg, cfg.isaacgym, load_asset(robots[0]))
    scene.randomize_seed = 0 if a.k_total > 1 else -1      # as the planner's K rollout envs (IsaacGymWrapper.__init__)
    c_cfg = make_config(cfg.mppi, k_offset=a.k_offset, k_local=a.k_local, viz_link=scene.viz_link_index())
    if a.state:
        z = np.load(a.state)
        dof, root = z["dof0"], z["root0"]
    else:
        dof, root = scene.initial_state()
        if wl["q0"] is not None:
            dof[0::2] = wl["q0"]
        if wl["goal"] is not None:
            root[scene.actor_index("goal"), 0:3] = wl["goal"]
    o = Oracle("f32")
    eps = np.load(a.eps) if a.eps else o.sample(c_cfg)
    eps = np.ascontiguousarray(eps[:, :, :a.k_local], np.float32)
    objective = getattr(objectives, wl["objective"])(cfg)
    m = scene.to_c()
    if a.mode == "pipeline":
        p = CpuPipeline(scene, m, c_cfg, objective, a.threads)
        dt, n = p.time_iterations(dof, root, torch.from_numpy(eps), a.budget, max_iters=a.max_iters)
    else:
        o.lib.orc_set_threads(C.c_int(a.threads))

        class _S:  # fused_spec only needs name -> index lookups
            pass
        s = _S()
        s.scene = scene
        cost = objective.fused_spec(s)
        U = np.zeros((a.horizon, c_cfg.nu), np.float32)
        t0 = time.perf_counter()
        U, act, S = o.command(m, c_cfg, cost, dof, root, U, eps)
        first = time.perf_counter() - t0
        n = max(1, min(a.max_iters, int(a.budget / max(first, 1e-3)) - 1))
        t0 = time.perf_counter()
        for _ in range(n):
            U, act, S = o.command(m, c_cfg, cost, dof, root, U, eps)
        dt = (time.perf_counter() - t0) / n
    print(json.dumps({"threads": a.threads, "seconds_per_iteration": dt, "iterations": n, "samples_timed": a.k_local,
                      "omp_max_threads": int(o.lib.orc_get_max_threads()), "torch_threads": torch.get_num_threads()}))


if __name__ == "__main__":
    _main()

"""MPPI core: stands in for the reference's external dependency `mppi_torch.mppi`
(MPPIConfig / MPPIPlanner; call sites reference mppiisaac/planner/mppi_isaac.py:3,43-49,84,113
and mppiisaac/utils/config_store.py:2,13).  The arithmetic runs in the HIP library
(sample / rollout / reduce / update kernels); this module is the thin host driver.

Field list of MPPIConfig: reference benchmarks/point_robot/setup/mppi.yaml:5-37 plus
eta_u_bound / eta_l_bound / seed_val (reference conf/mppi/omnipanda_effort.yaml:29-31).
Semantics: SURVEY.md section A (build-normative; mppi_torch itself is not in the reference tree).
"""
from __future__ import annotations

import ctypes as C
import os
import warnings
from dataclasses import dataclass
from typing import Callable, List, Optional

import numpy as np
import torch

from mppiisaac.backend import capi


@dataclass
class MPPIConfig:
    num_samples: int = 100
    horizon: int = 30
    mppi_mode: str = "halton-spline"      # "halton-spline" | "simple"
    sampling_method: str = "halton"       # "halton" | "random"
    noise_sigma: Optional[List[List[float]]] = None
    noise_mu: Optional[List[float]] = None
    device: str = "cuda:0"
    lambda_: float = 1.0
    update_lambda: bool = False
    update_cov: bool = False
    u_min: Optional[List[float]] = None
    u_max: Optional[List[float]] = None
    u_init: float = 0.0
    U_init: Optional[List[List[float]]] = None
    u_scale: float = 1.0
    u_per_command: int = 1
    rollout_var_discount: float = 0.95
    sample_null_action: bool = False
    noise_abs_cost: bool = False
    filter_u: bool = False
    use_priors: bool = False
    seed_val: int = 0
    eta_u_bound: float = 10.0
    eta_l_bound: float = 5.0


def bspline_basis(horizon: int, n_knots: int, degree: int = 2) -> np.ndarray:
    """[H, n_knots] clamped uniform B-spline basis evaluated at t/(H-1) (Cox-de Boor)."""
    H, n, p = horizon, n_knots, degree
    if n <= p:  # too few control points for the degree: piecewise-constant hold
        B = np.zeros((H, n))
        for t in range(H):
            B[t, min(n - 1, t * n // H)] = 1.0
        return B
    knots = np.concatenate([np.zeros(p), np.linspace(0.0, 1.0, n - p + 1), np.ones(p)])
    xs = np.linspace(0.0, 1.0, H) if H > 1 else np.zeros(1)
    B = np.zeros((H, n))
    for ti, x in enumerate(xs):
        N = np.zeros(len(knots) - 1)
        for j in range(len(knots) - 1):  # degree 0
            if (knots[j] <= x < knots[j + 1]) or (x == 1.0 and knots[j] < knots[j + 1] == 1.0):
                N[j] = 1.0
        for d in range(1, p + 1):
            Nn = np.zeros(len(knots) - 1 - d)
            for j in range(len(Nn)):
                a = 0.0 if knots[j + d] == knots[j] else (x - knots[j]) / (knots[j + d] - knots[j]) * N[j]
                b = 0.0 if knots[j + d + 1] == knots[j + 1] else (knots[j + d + 1] - x) / (knots[j + d + 1] - knots[j + 1]) * N[j + 1]
                Nn[j] = a + b
            N = Nn
        B[ti] = N[:n]
    return B


def savgol_matrix(horizon: int, window: int = 9, order: int = 3) -> np.ndarray:
    """[H, H] Savitzky-Golay smoothing operator with polynomial edge handling (the behaviour of
    scipy.signal.savgol_filter(..., mode="interp")): row t fits a degree-`order` polynomial to the `window`
    samples nearest to t (clamped inside the horizon) and evaluates it at t.  `filter_u: True` applies it to
    the nominal control sequence after every update.  mppi_torch's own window/order are not visible from
    the reference tree: window 9 / order 3 are this build's choice (PARITY UNPINNED)."""
    H = horizon
    w = min(window, H if H % 2 == 1 else H - 1)
    if w < 3:
        return np.eye(H)
    p = min(order, w - 1)
    F = np.zeros((H, H))
    for t in range(H):
        a = int(np.clip(t - w // 2, 0, H - w))
        x = np.arange(a, a + w, dtype=float) - t
        V = np.vander(x, p + 1, increasing=True)       # [w, p+1]
        F[t, a:a + w] = np.linalg.pinv(V)[0]            # value of the fitted polynomial at x = 0
    return F


def knots_for_horizon(horizon: int) -> int:
    """halton-spline: n_knots = H // 4; below 3 knots sample every step directly."""
    nk = horizon // 4
    return nk if nk >= 3 else horizon


def make_config(cfg: MPPIConfig, *, k_offset: int = 0, k_local: Optional[int] = None, viz_link: int = -1) -> capi.Config:
    """MPPIConfig -> C-ABI mppi_config_t for the shard [k_offset, k_offset + k_local)."""
    if cfg.update_cov:
        raise NotImplementedError("update_cov=True is not supported (False in every shipped conf/mppi file)")
    if cfg.update_lambda and not (cfg.eta_u_bound > cfg.eta_l_bound > 0):
        raise ValueError("update_lambda needs eta_u_bound > eta_l_bound > 0")
    if cfg.mppi_mode not in ("halton-spline", "simple") or cfg.sampling_method not in ("halton", "random"):
        raise ValueError(f"unknown mppi_mode / sampling_method: {cfg.mppi_mode!r} / {cfg.sampling_method!r}")
    if not 1 <= int(cfg.u_per_command) <= int(cfg.horizon):
        raise ValueError("u_per_command must lie in [1, horizon]")
    if cfg.noise_sigma is None:
        raise ValueError("noise_sigma is required: its size defines the control dimension")
    sigma = np.atleast_2d(np.asarray(cfg.noise_sigma, dtype=np.float64))
    nu = sigma.shape[0]
    if sigma.shape != (nu, nu) or np.abs(sigma - np.diag(np.diag(sigma))).max() > 0:
        raise NotImplementedError("noise_sigma must be a diagonal [nu x nu] matrix")
    if nu > capi.MAX_NU or cfg.horizon > capi.MAX_H:
        raise ValueError(f"nu={nu} / horizon={cfg.horizon} exceed MPPI_MAX_NU / MPPI_MAX_H")
    c = capi.Config()
    c.abi_version = capi.ABI_VERSION
    c.num_samples = int(cfg.num_samples if k_local is None else k_local)
    c.horizon = int(cfg.horizon)
    c.nu = nu
    c.k_offset = int(k_offset)
    c.k_total = int(cfg.num_samples)
    c.sample_null_action = int(bool(cfg.sample_null_action))
    c.use_priors = int(bool(cfg.use_priors))
    # halton-spline + halton: the fixed low-discrepancy set, sampled once.  Everything else draws fresh Gaussian
    # noise on the device at every command (counter-based, MPPI_SAMPLE_NORMAL): "simple" mode one draw per horizon
    # step - whatever sampling_method says, e.g. reference conf/mppi/omnipanda_effort.yaml:4-5 -, halton-spline +
    # random through the same B-spline as the Halton knots.
    halton = cfg.mppi_mode == "halton-spline" and cfg.sampling_method == "halton"
    spline = cfg.mppi_mode == "halton-spline"
    c.sampling = capi.SAMPLE_HALTON_SPLINE if halton else capi.SAMPLE_NORMAL
    nk = knots_for_horizon(cfg.horizon) if spline else cfg.horizon
    if nk > capi.MAX_KNOTS and nk != cfg.horizon:
        raise ValueError(f"n_knots={nk} exceeds MPPI_MAX_KNOTS")
    if halton and nk > capi.MAX_KNOTS:
        raise ValueError(f"halton sampling without a spline needs horizon <= {capi.MAX_KNOTS} (n_knots = horizon = {nk})")
    c.n_knots = nk
    c.noise_abs_cost = int(bool(cfg.noise_abs_cost))
    c.want_rollouts = int(viz_link >= 0)
    c.viz_link = max(int(viz_link), 0)
    c.seed = int(cfg.seed_val)
    c.lambda_ = float(cfg.lambda_)
    c.rollout_var_discount = float(cfg.rollout_var_discount)
    c.u_init = float(cfg.u_init)

    def bcast(v, default):
        if v is None:
            return [default] * nu
        v = list(v)
        return v * nu if len(v) == 1 else v
    umin, umax = bcast(cfg.u_min, -1e30), bcast(cfg.u_max, 1e30)
    for j in range(nu):
        c.u_min[j], c.u_max[j] = float(umin[j]) * cfg.u_scale, float(umax[j]) * cfg.u_scale
        c.noise_sigma_diag[j] = float(sigma[j, j])
        c.noise_mu[j] = float(cfg.noise_mu[j]) if cfg.noise_mu is not None else 0.0
    if nk != cfg.horizon or halton:
        B = bspline_basis(cfg.horizon, nk) if nk != cfg.horizon else np.eye(cfg.horizon)
        flat = B.reshape(-1)
        for j, v in enumerate(flat):
            c.spline_basis[j] = float(v)
    return c


class MPPIPlanner:
    """Host driver with the call surface the reference uses from mppi_torch.MPPIPlanner
    (constructed at reference mppi_isaac.py:43-49, driven by `.command(state)` :84,113).

    Two execution modes, chosen per call:
      fused    the Objective declares a cost spec (`fused_spec(sim)`): ONE persistent rollout kernel runs
               all K samples over the whole horizon with the stage cost evaluated in-kernel.
      generic  any Objective with `compute_cost(sim) -> [K]`: H step launches; after each the
               reference-layout state tensors are materialised and the Python callback runs, exactly
               the loop of reference mppi_isaac.py:57-69 (apply -> step -> cost).
    Sharded over `torch.distributed` ranks when a process group is initialised and `shard=True`:
    each rank owns K/world samples; the shard records (beta, eta, sum w*du) are all-gathered
    (RCCL on GPUs) and combined identically on every rank.
    """

    def __init__(self, cfg: MPPIConfig, nx: int, dynamics: Callable = None, running_cost: Callable = None,
                 prior: Optional[Callable] = None, *, sim=None, process_group=None, shard: bool = False):
        if sim is None:
            raise ValueError("MPPIPlanner needs the HIP simulator context (sim=IsaacGymWrapper)")
        self.cfg, self.nx = cfg, nx
        self._dynamics, self._running_cost, self._prior = dynamics, running_cost, prior
        self.sim = sim
        self._lib, self._ctx = sim._lib, sim._ctx
        self.K, self.T = sim.num_envs, cfg.horizon
        self.nu = len(cfg.noise_sigma)
        self.lambda_ = cfg.lambda_
        self._pg, self._shard = process_group, shard
        self._world = 1
        if shard:
            import torch.distributed as dist
            self._world = dist.get_world_size(process_group)
        self._rec_floats = self._lib.mppi_record_floats(self._ctx)
        self._records = torch.zeros((self._world, self._rec_floats), dtype=torch.float32, device=sim.device)
        # fused mode on the quad kernels: the rollout's own tail folds its wave records to `_fold_n` records per shard and
        # writes them straight into this rank's rows of the tensor that is all-gathered in place (no reduce launch)
        self._fold_n = self._lib.mppi_shard_record_count(self._ctx) if shard else 0
        self._records_fold = None
        if self._fold_n > 0:
            import torch.distributed as dist
            r = dist.get_rank(process_group)
            self._records_fold = torch.zeros((self._world * self._fold_n, self._rec_floats), dtype=torch.float32, device=sim.device)
            capi.check(self._lib, self._lib.mppi_set_record_out(self._ctx, C_void(self._records_fold[r * self._fold_n:(r + 1) * self._fold_n])))
        # how the shard records travel: "rccl" = torch.distributed all-gather (the portable path), "mailbox" = the
        # library's direct peer-to-peer exchange (one node; the inboxes are connected through hipIpc handles at the first command)
        self._exchange = os.environ.get("MPPI_EXCHANGE", "rccl") if shard else None
        if self._exchange not in (None, "rccl", "mailbox"):
            raise ValueError(f"MPPI_EXCHANGE={self._exchange!r}: 'rccl' or 'mailbox'")
        self._mailbox_state = None
        self._fused_cost = None
        self._sample_index = 0
        self._action = np.zeros(self.nu, np.float32)
        if cfg.U_init is not None:
            U = np.ascontiguousarray(np.asarray(cfg.U_init, np.float32).reshape(self.T, self.nu))
            capi.check(self._lib, self._lib.mppi_set_nominal(self._ctx, capi.fptr(U)))
        if cfg.filter_u:
            F = np.ascontiguousarray(savgol_matrix(self.T), np.float32)
            capi.check(self._lib, self._lib.mppi_set_filter(self._ctx, capi.fptr(F)))
        self._graph, self._graph_sig, self._graph_viz = None, None, []
        # MPPI_GENERIC_GRAPH: "0" never capture, "1" capture any Objective, unset: capture Objectives that declare
        # `graph_safe = True` (see _replay_horizon)
        self._graph_state = {"0": "off", "1": "on"}.get(os.environ.get("MPPI_GENERIC_GRAPH", ""), "auto")
        # MPPI_GENERIC_BATCH: "0" per-step cost calls only, "1" one cost call over the whole horizon without the check,
        # unset: checked on the first command of every Objective (see _horizon_batched)
        self._batch_state = {"0": "off", "1": "on"}.get(os.environ.get("MPPI_GENERIC_BATCH", ""), "auto")
        self._batch_sig, self._batch_buf, self._batch_graph = None, None, None
        self._external_noise = None
        self._iteration = 0   # control iterations so far: the counter of the device-side Gaussian sampler
        self._sampling = sim._mppi_config.sampling
        if self._sampling == capi.SAMPLE_HALTON_SPLINE:
            capi.check(self._lib, self._lib.mppi_sample(self._ctx, np.uint32(cfg.seed_val)))
        elif self._sampling == capi.SAMPLE_NORMAL:
            capi.check(self._lib, self._lib.mppi_sample_normal(self._ctx, np.uint32(0)))

    # -- sampling -----------------------------------------------------------------------
    def set_external_noise(self, eps: Optional[torch.Tensor]):
        """caller-owned noise [H][nu][K] (float32, on the sim's device) for the following commands; None returns to
        the configured sampler.  The tensor is kept alive here: the library reads it at every rollout."""
        if eps is None:
            # (a captured horizon has the released tensor's pointer baked into its step kernels: recapture)
            self._graph = self._batch_graph = None
            capi.check(self._lib, self._lib.mppi_set_noise_dev(self._ctx, None))
            self._external_noise = None
            return
        if tuple(eps.shape) != (self.T, self.nu, self.K) or eps.dtype != torch.float32 or not eps.is_cuda:
            raise ValueError(f"external noise must be a float32 device tensor of shape [{self.T}, {self.nu}, {self.K}]")
        self._external_noise = eps.contiguous()
        self._graph = self._batch_graph = None  # a captured horizon has the previous noise pointer baked in
        capi.check(self._lib, self._lib.mppi_set_noise_dev(self._ctx, C_void(self._external_noise)))

    # -- properties mirroring mppi_torch attributes used by callers -----------------------
    @property
    def U(self) -> torch.Tensor:
        out = np.zeros((self.T, self.nu), np.float32)
        capi.check(self._lib, self._lib.mppi_get_nominal(self._ctx, capi.fptr(out)))
        return torch.from_numpy(out)

    def set_fused_cost(self, cost: Optional[capi.Cost]):
        """called before every command (the Objective's weights are mutable): the struct is uploaded only when it changed"""
        self._fused_cost = cost
        if cost is not None:
            blob = bytes(cost)
            if blob != getattr(self, "_fused_cost_blob", None):
                capi.check(self._lib, self._lib.mppi_set_cost(self._ctx, C.byref(cost)))
                self._fused_cost_blob = blob
                # which kernel runs depends on the cost (a cost program on a contact-free scene: the one-lane kernel, which
                # folds no records): the folded all-gather is used only while the library says it folds
                self._fold_active = self._records_fold is not None and self._lib.mppi_shard_record_count(self._ctx) == self._fold_n
        else:
            self._fused_cost_blob = None

    # -- traced Objectives ---------------------------------------------------------------------
    TRACE_RECHECK = 64   # commands between two validations of a traced cost program against the eager Objective

    def set_trace_guard(self, on_fail: Optional[Callable]):
        """`on_fail(detail)`: the fused cost is a program TRACED from a Python Objective (MPPIisaacPlanner._bind_objective) - on the
        first command with it and every TRACE_RECHECK-th the same rollouts are also costed by the Objective itself (generic mode: the
        horizon's states, one compute_cost call) and the per-sample trajectory costs compared; on_fail is told when they differ"""
        if on_fail is None:
            self._trace_guard = None
        elif getattr(self, "_trace_guard", None) is None or self._trace_guard[0] is not on_fail.__self__:
            self._trace_guard = [on_fail.__self__, on_fail, 0]

    def _trace_check(self, state) -> bool:
        """-> True when the traced program stands (the context then holds the FUSED rollout of this command, as on any other)"""
        lib, ctx = self._lib, self._ctx
        import time
        clk = [time.perf_counter()]
        capi.check(lib, lib.mppi_sim_reset(ctx))
        self.sim._needs_reset = False
        done = self._horizon_batched(state)
        if done != "reduced":
            if not done and not self._replay_horizon(state):
                self._horizon_eager(state)
            capi.check(lib, lib.mppi_sim_finish(ctx))
        self.sim._stale = True
        clk.append(time.perf_counter())
        S_py = self._costs_np()
        clk.append(time.perf_counter())
        capi.check(lib, lib.mppi_rollout(ctx))
        S_k = self._costs_np()
        clk.append(time.perf_counter())
        # (what the validations cost, for whoever reports rates: [Objective on the horizon, wait for its costs, fused rollout] in ms)
        self.trace_check_ms = getattr(self, "trace_check_ms", []) + [[round(1e3 * (b - a), 3) for a, b in zip(clk, clk[1:])]]
        # numpy, not torch, for the host-side comparison: a torch op on CPU tensors wakes the OpenMP pool - one spinning thread per
        # hardware thread of the host (256 on the MI355X boxes) - and inside a container with a CPU quota (16 cores there) the whole
        # process is then throttled for the rest of the scheduler period: measured as 50-70 ms stalls of kernel launches a few dozen
        # commands AFTER a validation, one per ~180 commands, gone with OMP_NUM_THREADS=1 (tools/exp/facade_spikes.py)
        fin = np.isfinite(S_py) & np.isfinite(S_k)
        scale = max(float(np.abs(S_py[fin]).max()), 1e-6) if fin.any() else 1.0
        err = float(np.abs(S_py[fin] - S_k[fin]).max()) if fin.any() else 0.0
        # (contact-free rollouts agree to fp32 rounding; the states of a contact scene's DUMP instantiation and of its fused one are the
        # same arithmetic, the costs differ by the Objective's own fp32 evaluation order)
        if err <= 2e-3 * scale and bool((np.isfinite(S_py) == np.isfinite(S_k)).all()):
            return True
        self._trace_guard[1](f"trajectory costs differ by {err:.3g} of {scale:.3g}")
        return False

    # -- one control iteration ---------------------------------------------------------------
    def command(self, state=None) -> torch.Tensor:
        lib, ctx = self._lib, self._ctx
        if self._sampling == capi.SAMPLE_NORMAL and self._external_noise is None:
            # fresh noise every command, drawn on the device into the context's own (fixed-address) buffer: a captured
            # generic horizon keeps reading the right memory
            capi.check(lib, lib.mppi_sample_normal(ctx, np.uint32(self._iteration)))
        self._iteration += 1
        with_prior = self._prior is not None and self.cfg.use_priors
        if with_prior and self._fused_cost is not None:
            # fused mode: the whole horizon runs inside one kernel, so the prior can only be evaluated beforehand, on
            # the state the rollout starts from (open loop); generic mode evaluates it at every rollout step
            pr = np.ascontiguousarray(np.stack([_prior_row(self._prior(state, t), self.nu) for t in range(self.T)]))
            capi.check(lib, lib.mppi_set_prior(ctx, capi.fptr(pr)))
        guard = getattr(self, "_trace_guard", None) if self._fused_cost is not None else None
        if guard is not None and not with_prior:
            guard[2] += 1
            if (guard[2] == 1 or guard[2] % self.TRACE_RECHECK == 0) and not self._trace_check(state):
                self._fused_cost = None       # (this command: generic mode, below; the facade binds no program from now on)
                self._fused_cost_blob = None
                guard = None
            elif guard[2] == 1 or guard[2] % self.TRACE_RECHECK == 0:
                guard = "checked"             # (the fused rollout of this command has run inside the check)
        if self._fused_cost is not None:
            if guard != "checked":
                capi.check(lib, lib.mppi_rollout(ctx))
        else:
            # generic mode.  _horizon_batched returns "reduced" when the whole horizon went through the library in four launches
            # (rollout with the states kept -> materialise what the Objective reads -> [Objective] -> costs folded and reduced):
            # the control-cost finish and the separate reduce of the step-by-step bookkeeping are part of the last one
            capi.check(lib, lib.mppi_sim_reset(ctx))
            self.sim._needs_reset = False
            done = self._horizon_batched(state)
            if done != "reduced":
                if not done and not self._replay_horizon(state):
                    self._horizon_eager(state)
                capi.check(lib, lib.mppi_sim_finish(ctx))
            self.sim._stale = True
        if self._shard and self._exchange == "mailbox":
            # the library's own exchange (include/mppi_hip.h mppi_mailbox_*): every rank stores its records into every rank's
            # inbox over xGMI and polls its own - no collective library on the per-iteration path
            gathered, n = self._mailbox()
            capi.check(lib, lib.mppi_exchange(ctx))
            capi.check(lib, lib.mppi_update(ctx, gathered, n))
        elif self._shard and self._fused_cost is not None and self._records_fold is not None and getattr(self, "_fold_active", True):
            allgather_records(self._records_fold, _dist_rank(self._pg), self._pg, per=self._fold_n)
            capi.check(lib, lib.mppi_update(ctx, C_void(self._records_fold), self._world * self._fold_n))
        elif self._shard:
            rank = _dist_rank(self._pg)
            capi.check(lib, lib.mppi_reduce(ctx, C_void(self._records[rank])))
            allgather_records(self._records, rank, self._pg)
            capi.check(lib, lib.mppi_update(ctx, C_void(self._records), self._world))
        else:
            capi.check(lib, lib.mppi_reduce(ctx, None))
            capi.check(lib, lib.mppi_update(ctx, None, 1))
        # the action as soon as the update kernel has published it (sequence number in mapped host memory: no copy operation, no
        # wait for whatever else is still queued on the stream)
        capi.check(lib, lib.mppi_wait_action(ctx, capi.fptr(self._action)))
        if self._shard and self._exchange == "mailbox":
            # a peer that did not publish in time (dead, or > 2 s late): the library made its records neutral for THIS update - the
            # action returned below comes from the ranks that did publish - and set the status word, which is read-and-clear: the
            # caller hears about the iterations that were affected, a transient stall does not poison the ones after it
            late = C.c_int(0)
            capi.check(lib, lib.mppi_exchange_status(ctx, C.byref(late)))
            if late.value:
                self.late_exchanges = getattr(self, "late_exchanges", 0) + 1
                warnings.warn("mailbox exchange: a rank did not publish its shard records in time; this update ran without them "
                              f"({self.late_exchanges} late iteration(s) so far)", RuntimeWarning)
        if self.cfg.update_lambda:
            self._adapt_lambda()
        n = int(self.cfg.u_per_command)
        if n > 1:
            # mppi_torch: `action = U[:u_per_command]` of the updated (and, with filter_u, smoothed) nominal, which is then shifted by
            # ONE step as always [RECALLED: pytorch_mppi's command(); SURVEY.md A].  Row 0 is the action the update kernel publishes,
            # rows 1 .. n-1 are the first rows of the shifted nominal: [n, nu]
            return torch.cat((torch.from_numpy(self._action.copy()).unsqueeze(0), self.U[: n - 1]), dim=0)
        return torch.from_numpy(self._action.copy())

    # mppi_torch's `update_lambda` (False in every shipped conf; its code is not in the reference tree).  Restated from the MPPI
    # it derives from (STORM's mppi: beta_lm = 0.9, beta_um = 1.2) [RECALLED - unpinned, SURVEY.md A]: the normaliser eta = sum of
    # the weights measures how many samples carry weight; above eta_u_bound the temperature is too soft -> lambda *= 0.9, below
    # eta_l_bound too greedy -> lambda *= 1.2.  Every rank sees the same eta after the combine, so shards stay in step.
    LAMBDA_DOWN, LAMBDA_UP = 0.9, 1.2

    def _adapt_lambda(self):
        stats = np.zeros(2, np.float32)
        capi.check(self._lib, self._lib.mppi_get_weights_stats(self._ctx, capi.fptr(stats)))
        eta = float(stats[1])
        lam = self.lambda_
        if eta > self.cfg.eta_u_bound:
            lam *= self.LAMBDA_DOWN
        elif eta < self.cfg.eta_l_bound:
            lam *= self.LAMBDA_UP
        if lam != self.lambda_:
            self.lambda_ = lam
            capi.check(self._lib, self._lib.mppi_set_lambda(self._ctx, C.c_double(lam)))

    def _mailbox(self):
        """(gathered records pointer, count) of this rank's mailbox; created and connected on first use - after the cost is set,
        because the number of records a shard publishes depends on the kernel the cost selects.  The 64-byte IPC handles of the
        inboxes travel once through the process group (as CPU bytes: any backend)."""
        if self._mailbox_state is None:
            import torch.distributed as dist
            lib, ctx = self._lib, self._ctx
            rank, world = dist.get_rank(self._pg), self._world
            capi.check(lib, lib.mppi_mailbox_create(ctx, rank, world))
            h = (C.c_ubyte * 64)()
            capi.check(lib, lib.mppi_mailbox_ipc_handle(ctx, h))
            handles = [None] * world
            dist.all_gather_object(handles, bytes(h), group=self._pg)
            for r, hb in enumerate(handles):
                if r != rank:
                    capi.check(lib, lib.mppi_mailbox_open(ctx, r, (C.c_ubyte * 64).from_buffer_copy(hb)))
            p, n = C.c_void_p(), C.c_int()
            capi.check(lib, lib.mppi_mailbox_gathered(ctx, C.byref(p), C.byref(n)))
            dist.barrier(group=self._pg)   # every inbox is open before anybody publishes
            self._mailbox_state = (p, n.value)
        return self._mailbox_state

    # -- generic Objective mode: the horizon loop ---------------------------------------------------
    BATCH_RECHECK = 64   # commands between two validations of the one-call-per-horizon evaluation of an Objective

    def _horizon_batched(self, state) -> bool:
        """The horizon without a simulator launch per step.  The rollout dynamics never depend on the running cost, so the H
        steps are simulated first (`_simulate_horizon`: the fused rollout kernel with every step's env state kept, or 2H step /
        materialise launches replayed as a graph) into [H*K, ...] state tensors - row t*K + k = env k after horizon step t.
        The Objective then sees them either
          * in ONE call, through a view of the simulator whose `num_envs` is H*K: a reference-style Objective is a row-wise
            function of the state tensors (reference examples/*/planner.py), so this is the same cost with ~H times fewer
            kernel launches (generic mode is launch-bound); or
          * in H calls on the [K]-row blocks, in horizon order - the reference's call pattern (mppi_isaac.py:57-69) on
            precomputed states, valid for any Objective that only READS the simulator.
        Which one: on the first command of every Objective object (and after its `.weights` change) both are evaluated on the
        same states; the single call is adopted if the costs agree, otherwise (a call counter, a dependence on the horizon
        step, tensors sized for K envs) the H calls stay, with a warning.  S_k = sum_t gamma^t c_t[k] goes into the same
        accumulator.  MPPI_GENERIC_BATCH=0 keeps the reference loop shape (step, cost, step, ...), =1 skips the check;
        priors are host callbacks that see the envs' state at every rollout step: they keep the step-by-step loop too."""
        if self._batch_state == "off" or (self._prior is not None and self.cfg.use_priors):
            return False
        sig = self._objective_signature()
        b = self._simulate_horizon()
        single = self._batch_state == "on" or self._batch_sig == ("ok", sig)
        # an Objective whose Python-side state drifts AFTER its first command (a call counter that starts to matter, a schedule)
        # would go stale silently: the adopted single call is re-validated every BATCH_RECHECK-th command
        self._batch_age = getattr(self, "_batch_age", 0) + 1
        if single and self._batch_state != "on" and self._batch_age >= self.BATCH_RECHECK:
            # (measured: the full check - H calls next to the single one - costs 7.5 ms, 19 % of a loop that re-validates every
            # 64th command; a drifting call counter or schedule shows on any row block, so the re-validation compares TWO of them -
            # the first and one that moves through the horizon - against the single call's rows: ~1 ms)
            self._batch_age = 0
            if not self._revalidate_single(state, b):
                single, self._batch_sig = False, None
        if not single and self._batch_sig != ("no", sig):       # first command of this Objective (or a re-validation): check
            self._batch_age = 0
            self._batch_sig = ("no", sig)
            viz = list(self.sim.visualize_link_buffer)
            S_one = None
            try:
                S_one = self._horizon_costs(state, b, single=True)
            except Exception as e:  # noqa: BLE001 - e.g. an Objective with tensors sized for K envs
                warnings.warn(f"the Objective cannot be evaluated over a whole horizon in one call ({type(e).__name__}: {e}); "
                              "keeping one compute_cost call per horizon step")
            self.sim.visualize_link_buffer = viz
            S_add = self._horizon_costs(state, b, single=False)   # (this command: the reference's call pattern)
            if S_one is not None:
                scale = float(S_add.abs().max().clamp_min(1e-6))
                if bool(torch.isfinite(S_one).all()) and float((S_one - S_add).abs().max()) <= 1e-5 * scale:
                    self._batch_sig = ("ok", sig)
                else:
                    warnings.warn("the Objective's cost of a whole horizon evaluated in one call differs from its per-step costs "
                                  "(it depends on more than the sim tensors): keeping one compute_cost call per horizon step")
        elif single and self._cost_graph_wanted():
            S_add = self._horizon_costs_graph(state, b, sig)
        else:
            S_add = self._horizon_costs(state, b, single=single, fold=False if single else None)
        if S_add.shape[0] == self.T * self.K:   # the stage costs of all H*K env-steps: folded over the horizon and reduced in ONE launch
            rc = self._lib.mppi_reduce_horizon_costs(self._ctx, C_void(S_add), None)
            if rc == capi.MPPI_OK:
                return "reduced"
            if rc != capi.MPPI_EUNSUPPORTED:
                capi.check(self._lib, rc)
            S_add = (S_add.view(self.T, self.K) * self._batch_disc).sum(0).contiguous()
        capi.check(self._lib, self._lib.mppi_sim_accumulate_cost(self._ctx, 0, C_void(S_add)))
        return True

    def _revalidate_single(self, state, b) -> bool:
        """does the Objective still give the same costs on a [K]-row block alone as inside the one call over the whole horizon?"""
        H, K, sim = self.T, self.K, self.sim
        viz = list(sim.visualize_link_buffer)
        try:
            c_all = self._horizon_costs(state, b, single=True, fold=False).view(H, K)
            self._reval_t = (getattr(self, "_reval_t", 0) + 7) % H
            for t in {0, self._reval_t}:
                rows_of = lambda idx, t=t: (lambda r: None if r is None else r[t * K:(t + 1) * K])(self._lazy_link(idx))
                with sim._horizon_view({k: v[t * K:(t + 1) * K] for k, v in b.items()}, K, lazy=self._lazy_materialise, link=rows_of), torch.no_grad():
                    c_t = self._running_cost(state).to(dtype=torch.float32, device=sim.device)
                scale = float(c_all[t].abs().max().clamp_min(1e-6))
                if c_t.shape != (K,) or not bool(torch.isfinite(c_t).all()) or float((c_t - c_all[t]).abs().max()) > 1e-5 * scale:
                    return False
            return True
        except Exception:  # noqa: BLE001 - whatever it is, the full check of the next command reports it
            return False
        finally:
            sim.visualize_link_buffer = viz

    def _cost_graph_wanted(self) -> bool:
        """the adopted one-call evaluation as a captured HIP graph: for Objectives that declare `graph_safe = True` (compute_cost is
        a pure tensor program of sim tensors and `.weights` - the same opt-in as _replay_horizon), or any Objective with
        MPPI_GENERIC_GRAPH=1.  A reference-style compute_cost is ~60 small torch kernels; launching them costs the host ~5 us
        each, a replay costs one launch."""
        if self._graph_state == "off" or getattr(self, "_cost_graph", None) is False:
            return False
        if self._graph_state == "on":
            return True
        obj = getattr(self._running_cost, "__self__", None)
        obj = getattr(obj, "objective", obj)
        return bool(getattr(obj, "graph_safe", False))

    def _horizon_costs_graph(self, state, b, sig) -> torch.Tensor:
        g = getattr(self, "_cost_graph", None)
        if g is not None and g[2] != sig:
            g = None                                   # another Objective / other weights: capture again
        if g is None:
            try:
                S = self._horizon_costs(state, b, single=True, fold=False)     # (this command's costs, and the warm-up of the capture)
                viz0 = list(self.sim.visualize_link_buffer)
                torch.cuda.synchronize()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    S_static = self._horizon_costs(state, b, single=True, fold=False)
                viz = self.sim.visualize_link_buffer[len(viz0):]
                self.sim.visualize_link_buffer = viz0
                self._cost_graph = (graph, S_static, sig, viz)
                return S
            except Exception as e:  # noqa: BLE001 - host synchronisation inside compute_cost etc.: the eager call stays
                warnings.warn(f"the Objective's horizon evaluation is not graph-capturable ({type(e).__name__}: {e}); evaluating it eagerly")
                self._cost_graph = False
                torch.cuda.synchronize()
                return self._horizon_costs(state, b, single=True, fold=False)
        graph, S_static, _, viz = g
        graph.replay()
        if self.sim._visualize_link_present:
            self.sim.visualize_link_buffer.extend(viz)
        return S_static

    def _simulate_horizon(self) -> dict:
        """the envs' state after every horizon step as reference-layout tensors [H*K, ...] (row t*K + k)"""
        lib, ctx, sim = self._lib, self._ctx, self.sim
        H, K = self.T, self.K
        if self._batch_buf is None:
            f32 = dict(dtype=torch.float32, device=sim.device)
            t = sim._state_t
            self._batch_buf = {k: torch.zeros((H * K,) + tuple(v.shape[1:]), **f32) for k, v in t.items()}
            self._batch_gamma = [float(self.cfg.rollout_var_discount) ** t for t in range(H)]
            self._batch_disc = torch.tensor(self._batch_gamma, **f32)[:, None].contiguous()
            self._batch_graph = None
            self._batch_fused = None   # None: not tried yet; True / False: the library's whole-horizon rollout is / is not available
            self._batch_want, self._batch_done = set(), set()
            self._batch_links, self._batch_links_want, self._batch_links_done = {}, set(), set()   # dense rows of single rigid bodies
        b = self._batch_buf
        # the whole horizon in two launches: the fused rollout kernel (no cost, per-step states kept), then one materialise
        # over all H*K env-steps; contexts without that kernel simulate step by step below
        if self._batch_fused is not False:
            rc = lib.mppi_rollout_trajectory(ctx)
            if self._batch_fused is None and rc == capi.MPPI_EUNSUPPORTED:
                self._batch_fused = False   # this context has no whole-horizon kernel: step by step below
            else:
                capi.check(lib, rc)         # (any other failure - a hipMalloc of the trajectory buffer, a launch error - is an error)
                self._batch_fused = True
        if self._batch_fused:
            # only the tensors the Objective read at the last command are produced up front (one launch); any other one is
            # materialised when it is asked for (_lazy_materialise through IsaacGymWrapper._fresh)
            self._batch_done = set(self._batch_want)
            if self._batch_done:
                capi.check(lib, lib.mppi_materialise_trajectory(ctx, *[C_void(b[k]) if k in self._batch_done else None for k in ("dof", "root", "rb", "cf")]))
            self._batch_links_done = set()
            if "rb" not in self._batch_done:
                for idx in self._batch_links_want:
                    capi.check(lib, lib.mppi_materialise_trajectory_link(ctx, idx, C_void(self._batch_links[idx])))
                    self._batch_links_done.add(idx)
            return b
        self._batch_done = {"dof", "root", "rb", "cf"}
        self._batch_links_done = set()

        def simulate():
            for t in range(H):
                capi.check(lib, lib.mppi_sim_step_horizon(ctx, t))
                capi.check(lib, lib.mppi_sim_materialise(ctx, C_void(b["dof"][t * K]), C_void(b["root"][t * K]), C_void(b["rb"][t * K]),
                                                         C_void(b["cf"][t * K])))
        # the 2H launches are library kernels on fixed buffers (no Python inside): captured once, replayed as one launch
        if self._batch_graph is None and self._graph_state != "off":
            simulate()                                  # warm-up outside the capture
            capi.check(lib, lib.mppi_sim_reset(ctx))
            torch.cuda.synchronize()
            g, main = torch.cuda.CUDAGraph(), torch.cuda.current_stream()
            try:
                with torch.cuda.graph(g):
                    capi.check(lib, lib.mppi_set_stream(ctx, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
                    simulate()
                self._batch_graph = g
            except Exception as e:  # noqa: BLE001
                warnings.warn(f"the simulated horizon could not be captured into a HIP graph ({type(e).__name__}: {e}); launching it step by step")
                self._batch_graph = False
            finally:
                capi.check(lib, lib.mppi_set_stream(ctx, C.c_void_p(main.cuda_stream)))
            capi.check(lib, lib.mppi_sim_reset(ctx))
        if self._batch_graph:
            self._batch_graph.replay()
        else:
            simulate()
        return b

    def _lazy_materialise(self, key):
        """a state tensor of the horizon view is about to be read: produce it now if this command has not yet (and up front from the
        next command on)"""
        if key in self._batch_done:
            return
        b = self._batch_buf
        capi.check(self._lib, self._lib.mppi_materialise_trajectory(self._ctx, *[C_void(b[k]) if k == key else None for k in ("dof", "root", "rb", "cf")]))
        self._batch_done.add(key)
        self._batch_want.add(key)

    def _lazy_link(self, idx: int):
        """dense [H*K, 13] rows of rigid body `idx` over the horizon (None: take the slice of the full rigid-body tensor - it exists
        already, or this body / scene has no single-body kernel)"""
        if "rb" in self._batch_done or not self._batch_fused or self._batch_links.get(idx, 0) is None:
            return None
        if idx not in self._batch_links_done:
            buf = self._batch_links.get(idx)
            if buf is None:
                buf = torch.zeros((self.T * self.K, 13), dtype=torch.float32, device=self.sim.device)
            rc = self._lib.mppi_materialise_trajectory_link(self._ctx, idx, C_void(buf))
            if rc == capi.MPPI_EUNSUPPORTED:
                self._batch_links[idx] = None      # (remembered: no second attempt)
                return None
            capi.check(self._lib, rc)
            self._batch_links[idx] = buf
            self._batch_links_done.add(idx)
            self._batch_links_want.add(idx)
        return self._batch_links[idx]

    def _horizon_costs(self, state, b, single: bool, fold=None) -> torch.Tensor:
        """S_add [K] = sum_t gamma^t c_t from ONE compute_cost over the [H*K]-env view, or from H calls on its [K]-row blocks;
        fold=False (single call only): the un-discounted stage costs [H*K] themselves, for mppi_reduce_horizon_costs"""
        sim, H, K = self.sim, self.T, self.K
        if single:
            with sim._horizon_view(b, H * K, lazy=self._lazy_materialise, link=self._lazy_link), torch.no_grad():
                c = self._running_cost(state)
                if sim._visualize_link_present:
                    # the whole horizon as ONE [H, K, 3] block (H per-step views cost the host 18 us per command;
                    # MPPIisaacPlanner.get_rollouts concatenates blocks and per-step entries alike)
                    sim.visualize_link_buffer.append(sim.visualize_link_pos.reshape(H, K, 3))
            c = c.to(dtype=torch.float32, device=sim.device)
            if c.shape != (H * K,):
                raise ValueError(f"compute_cost must return one cost per env ([{H * K}] over the horizon view), got {tuple(c.shape)}")
            return c.contiguous() if fold is False else (c.view(H, K) * self._batch_disc).sum(0).contiguous()
        S = torch.zeros(K, dtype=torch.float32, device=sim.device)
        for t in range(H):
            rows_of = lambda idx, t=t: (lambda r: None if r is None else r[t * K:(t + 1) * K])(self._lazy_link(idx))
            with sim._horizon_view({k: v[t * K:(t + 1) * K] for k, v in b.items()}, K, lazy=self._lazy_materialise, link=rows_of):
                if sim._visualize_link_present:
                    sim.visualize_link_buffer.append(sim.visualize_link_pos.clone())
                with torch.no_grad():
                    c = self._running_cost(state).to(dtype=torch.float32, device=sim.device)
            if c.shape != (K,):
                raise ValueError(f"compute_cost must return a [{K}] tensor, got {tuple(c.shape)}")
            S += self._batch_gamma[t] * c
        return S

    def _horizon_eager(self, state):
        """reference loop shape (mppi_isaac.py:57-69): per horizon step dynamics() = apply + step, then running_cost()"""
        lib, ctx = self._lib, self._ctx
        with_prior = self._prior is not None and self.cfg.use_priors
        for t in range(self.T):
            if with_prior:  # prior(state, t) on the envs' state AT step t, as the reference's callback sees it (mppi_isaac.py:39)
                capi.check(lib, lib.mppi_set_prior_row(ctx, t, capi.fptr(_prior_row(self._prior(state, t), self.nu))))
            capi.check(lib, lib.mppi_sim_step_horizon(ctx, t))
            self.sim._materialise()
            if self.sim._visualize_link_present:
                self.sim.visualize_link_buffer.append(self.sim.visualize_link_pos.clone())
            c = self._running_cost(state)
            c = c.to(dtype=torch.float32, device=self.sim.device).contiguous()
            if c.shape != (self.K,):
                raise ValueError(f"compute_cost must return a [{self.K}] tensor, got {tuple(c.shape)}")
            capi.check(lib, lib.mppi_sim_accumulate_cost(ctx, t, C_void(c)))

    def _objective_signature(self):
        """what a captured horizon bakes in from the Python side: the objective object and its weights"""
        obj = getattr(self._running_cost, "__self__", None)
        obj = getattr(obj, "objective", obj)
        w = getattr(obj, "weights", None)
        return (id(obj), repr(sorted(w.items())) if isinstance(w, dict) else repr(w))

    def _replay_horizon(self, state) -> bool:
        """The generic horizon is launch-bound: H x (step kernel, materialise, ~20 small torch kernels of the Python
        Objective, accumulate).  It is therefore captured ONCE into a HIP graph (torch.cuda.CUDAGraph) and replayed:
        the Objective's Python code runs at capture time only, its tensor program runs every iteration.  Anything
        the Objective reads through `sim` tensors (goal, obstacles, states) stays live; Python-side numbers are
        baked in, so the capture is redone when the objective object or its `.weights` change.  Because ANY other
        Python-side state of an Objective (float goals, counters, tensors re-created in reset(), data-dependent
        branches) would silently go stale, the capture is opt-in: an Objective declares `graph_safe = True` when its
        compute_cost is a pure tensor program of `sim` tensors and `.weights` (the in-tree Objectives are);
        MPPI_GENERIC_GRAPH=1 forces the capture for any Objective, =0 switches it off.  Objectives that cannot be
        captured (host synchronisation such as .item()/.cpu() inside compute_cost) fall back to the eager loop for good."""
        if self._graph_state == "off" or (self._prior is not None and self.cfg.use_priors):
            return False  # (a prior is arbitrary host code evaluated per step: eager loop)
        if self._graph_state == "auto":
            obj = getattr(self._running_cost, "__self__", None)
            obj = getattr(obj, "objective", obj)
            if not getattr(obj, "graph_safe", False):
                return False
        sig = self._objective_signature()
        if self._graph is not None and sig != self._graph_sig:
            self._graph = None                         # objective or weights changed: capture again
        lib, ctx = self._lib, self._ctx
        if self._graph is None:
            try:
                self._horizon_eager(state)             # warm-up: lazy initialisation must not happen under capture
                self.sim.visualize_link_buffer = []
                capi.check(lib, lib.mppi_sim_reset(ctx))
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                main = torch.cuda.current_stream()
                try:
                    with torch.cuda.graph(g):
                        capi.check(lib, lib.mppi_set_stream(ctx, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
                        self._horizon_eager(state)
                finally:
                    capi.check(lib, lib.mppi_set_stream(ctx, C.c_void_p(main.cuda_stream)))
                self._graph, self._graph_sig = g, sig
                self._graph_viz = list(self.sim.visualize_link_buffer)
                self.sim.visualize_link_buffer = []
                capi.check(lib, lib.mppi_sim_reset(ctx))
            except Exception as e:  # not capturable: keep the reference loop shape
                warnings.warn(f"generic Objective horizon is not graph-capturable ({type(e).__name__}: {e}); running it eagerly")
                self._graph, self._graph_state = None, "off"
                torch.cuda.synchronize()
                self.sim.visualize_link_buffer = []
                capi.check(lib, lib.mppi_sim_reset(ctx))
                return False
        self._graph.replay()
        if self.sim._visualize_link_present:
            self.sim.visualize_link_buffer = list(self._graph_viz)
        return True

    def _costs_np(self) -> np.ndarray:
        S = np.zeros(self.K, np.float32)
        capi.check(self._lib, self._lib.mppi_get_costs(self._ctx, capi.fptr(S)))
        return S

    def get_costs(self) -> torch.Tensor:
        return torch.from_numpy(self._costs_np())


def _prior_row(p, nu: int) -> np.ndarray:
    """what a prior callback returns (list, numpy, CPU or device tensor) -> contiguous float32 [nu]"""
    row = torch.as_tensor(p).detach().to(dtype=torch.float32, device="cpu").reshape(-1).numpy()
    if row.shape[0] != nu:
        raise ValueError(f"prior returned {row.shape[0]} values, the control dimension is {nu}")
    return np.ascontiguousarray(row)


def C_void(t: torch.Tensor):
    return C.c_void_p(t.data_ptr())


def _dist_rank(group=None) -> int:
    import torch.distributed as dist
    return dist.get_rank(group)


def allgather_records(records: torch.Tensor, rank: int, group=None, per: int = 1) -> None:
    """All-gather the shard records [world * per, 2+H*nu] in place (rows rank*per .. rank*per+per-1 hold this shard's
    records on entry).  One small collective per control iteration: RCCL over xGMI for device tensors (backend "nccl":
    truly in place, the send buffer is this rank's slice of the receive buffer), gloo for the CPU tests.
    1-9 KB per rank: latency-bound."""
    import torch.distributed as dist
    mine = records[rank * per:(rank + 1) * per].reshape(-1)
    dist.all_gather_into_tensor(records.view(-1), mine if records.is_cuda else mine.clone(), group=group)


def _get_rollouts(self) -> torch.Tensor:
    """[H, K, 3] positions of the robot's visualize_link over the last fused rollout."""
    out = np.zeros((self.T, self.K, 3), np.float32)
    capi.check(self._lib, self._lib.mppi_get_rollouts(self._ctx, capi.fptr(out)))
    return torch.from_numpy(out)


MPPIPlanner.get_rollouts = _get_rollouts

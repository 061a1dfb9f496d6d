#!/usr/bin/env python3
"""Closed-loop MPPI benchmark on the BASELINE.json workload (Panda 7-DoF reach, K=4096, H=20).

One "step" = one control iteration: rollout kernel (K samples x H horizon steps of articulated-body dynamics +
fused cost + per-wave / per-XCD softmax records) -> (in-place all-gather of the shard records when --gpus > 1) ->
combine + nominal update + the K=1 world stepped with the action, its new state fed back (closed loop, everything
device-resident) -> the action reaches the host.  Weak scaling: every GPU owns K samples, the softmax weights are
combined over all ranks; `value` counts K-sample control iterations per second summed over ranks (at --gpus 1 it is
exactly the control-loop Hz at K=4096, H=20).

Protocol: W untimed warm-up iterations, then EXACTLY K timed ones bracketed by barrier +
torch.cuda.synchronize(); `value` = K / elapsed (max over ranks); the per-iteration wall times of the same run give
median / p5 / p95.  Prints ONE JSON line (rank 0).  DESIGN.md "Measurement" defines roofline / cpu_baseline.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "mppi-isaac_amd"))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
FP32_LANE_OPS_PEAK = 157.3e12 / 2  # fp32 vector peak 157.3 TFLOP/s = 78.6e12 lane-instructions/s (an FMA counts 2 flops)
N_SIMD = 1024               # 256 CUs x 4 SIMDs
# BASELINE.json configs.  The default - and the only one the driver's bench line uses - is panda_reach.
WORKLOADS = {
    "panda_reach": dict(desc="panda_stick reach (BASELINE configs[2]): ABA from URDF, no contact, fused reach cost",
                        actors=["panda_stick", "goal"], mppi="panda", nx=14, K=4096, H=20, init=[[0.0, 0.0, 0.0]],
                        q0=[0.0, -0.94, 0.0, -2.8, 0.0, 1.8675, 0.0],   # conf/actors/panda_stick.yaml init_joint_pose
                        goal=[0.5, -0.4, 0.3], objective="PandaReachObjective"),  # reference benchmarks/panda_arm/setup/exp.yaml:21-24
    "point_reach": dict(desc="point_robot reach (BASELINE configs[1]): 3-DoF velocity-driven base",
                        actors=["point_robot", "goal"], mppi="pointbot", nx=6, K=1024, H=15, init=[[0.0, 0.0, 0.05]],
                        q0=[0.1, 0.0, 0.0], goal=[4.5, 0.2, 0.0], objective="PointReachObjective"),
    "boxer_push": dict(desc="boxer_push (BASELINE configs[3]): floating diff-drive base + block + obstacles, penalty contact",
                       actors=["boxer", "block", "paper_obst1", "paper_obst2", "goal"], mppi="boxer_push", nx=4, K=8192, H=25,
                       init=[[0.0, 2.5, 0.05]], q0=None, goal=None, objective="BoxerPushObjective"),
    "panda_pick": dict(desc="panda_pick (BASELINE configs[4], 8192 samples per GPU): gripper arm + block + table contact",
                       actors=["panda_gripper", "xaxis", "yaxis", "panda_pick_block", "table", "goal"], mppi="panda_pick", nx=18,
                       K=8192, H=30, init=[[0.0, 0.0, 0.0]], q0=None, goal=None, objective="PandaPickObjective"),
}


def make_cfg(w, k_total, H=None, shipped=False):
    """conf/mppi/<workload>.yaml with num_samples / horizon overridden.  The measured configuration also switches
    `filter_u` off (mppi_torch's Savitzky-Golay window / order cannot be verified) and `use_priors` off;
    shipped=True keeps both as the reference's file ships them (`value_shipped_conf` of the result line)."""
    from mppiisaac.utils.config_store import load_config
    over = {"mppi.num_samples": k_total, "mppi.horizon": H or w["H"]}
    if not shipped:
        over.update({"mppi.filter_u": False, "mppi.use_priors": False})
    return load_config({"defaults": [{"mppi": w["mppi"]}, {"isaacgym": "normal"}], "actors": w["actors"],
                        "initial_actor_positions": w["init"], "nx": w["nx"]}, overrides=over)


class Loop:
    """planner (K rollout envs + MPPI core) and K=1 world of one workload on this rank's GPU, and its iteration"""

    def __init__(self, name, k_per_gpu, env, sync=True, horizon=None, shipped=False):
        import torch
        import torch.distributed as dist
        from mppiisaac.backend import capi
        import mppiisaac.objectives as objectives
        from mppiisaac.planner.isaacgym_wrapper import IsaacGymWrapper
        from mppiisaac.planner.mppi_isaac import MPPIisaacPlanner
        self.torch, self.dist, self.capi, self.env = torch, dist, capi, env
        wl = WORKLOADS[name]
        self.name, self.wl, self.K, self.H, self.sync = name, wl, k_per_gpu, horizon or wl["H"], sync
        world_size, rank, sharded = env["world_size"], env["rank"], env["sharded"]
        cfg = make_cfg(wl, k_per_gpu * world_size, self.H, shipped=shipped)
        if shipped and cfg.mppi.use_priors:
            cfg.mppi.use_priors = False   # (a prior is a host callback of the example scripts; none is part of the conf file)
        cfg.mppi.device = f"cuda:{env['local_rank']}"
        self.cfg = cfg
        self.objective = getattr(objectives, wl["objective"])(cfg)
        if os.environ.get("MPPI_BENCH_PROGRAM"):  # A/B: the same cost through the in-kernel term interpreter (MPPI_COST_PROGRAM)
            self.objective.fused_spec = self.objective.program_spec
        self.planner = MPPIisaacPlanner(cfg, self.objective, shard=sharded)
        self.world = IsaacGymWrapper(cfg.isaacgym, actors=cfg.actors, init_positions=cfg.initial_actor_positions, num_envs=1,
                                     device=cfg.mppi.device)
        self.lib, self.P, self.W = self.planner.sim._lib, self.planner.sim._ctx, self.world._ctx
        if wl["goal"] is not None:
            for sim in (self.planner.sim, self.world):
                sim.set_actor_position_by_name(wl["goal"], "goal")
        self.dof0 = self.world._dof_state[0].cpu().numpy().copy()
        if wl["q0"] is not None:
            self.dof0[0::2] = wl["q0"]
        self.root0 = self.world._root_state[0].cpu().numpy()
        for sim in (self.planner.sim, self.world):
            sim._push_single_state(self.dof0, self.root0)
        self.planner._bind_objective()
        self.nu = self.planner.sim.scene.nu
        self.action = np.zeros(self.nu, np.float32)
        self._ap = capi.fptr(self.action)
        self.graph = None
        self.n_records = 0
        lib, P = self.lib, self.P
        if sharded:
            # this rank's folded records are written by the rollout kernel straight into its rows of the tensor that is
            # all-gathered IN PLACE: rollout -> all-gather -> combine/update, no separate reduce launch, no staging copy
            RF = lib.mppi_record_floats(P)
            self.cnt = lib.mppi_shard_record_count(P)
            self.inplace = self.cnt > 0
            per = self.cnt if self.inplace else 1
            self.records = torch.zeros((world_size * per, RF), dtype=torch.float32, device=cfg.mppi.device)
            self.mine = self.records[rank * per:(rank + 1) * per]
            if self.inplace:
                capi.check(lib, lib.mppi_set_record_out(P, ctypes.c_void_p(self.mine.data_ptr())))
            self.n_records = world_size * per
            self.exchange, self.exchange_why, self.probe = "rccl", "requested (MPPI_BENCH_EXCHANGE / default)", None
            if env.get("exchange") == "mailbox":
                ok, why = self._connect_mailbox()
                self.exchange, self.exchange_why = ("mailbox", "probe passed: " + why) if ok else ("rccl", "mailbox refused: " + why)

    def _connect_mailbox(self):
        """The library's own exchange of the shard records (mppi_mailbox_*: every rank stores its records into every rank's inbox
        and polls its own) instead of the RCCL all-gather.  Guarded: the inboxes are connected through hipIpc
        handles, then three probe iterations run BOTH exchanges on the same records and compare them bit for bit on every rank;
        any refusal, timed-out wait or mismatch on any rank keeps the RCCL path for the whole job."""
        torch, dist, lib, P, capi, env = self.torch, self.dist, self.lib, self.P, self.capi, self.env
        rank, world = env["rank"], env["world_size"]
        ok, why = 1, ""
        self.probe = {"fine_grained": None, "records_per_rank": None, "iterations": 0, "late": None, "equal_per_rank": None}
        try:
            if os.environ.get("MPPI_BENCH_TEST_REFUSE_RANK") == str(rank):   # (test hook: one rank cannot take part)
                raise RuntimeError("refused (MPPI_BENCH_TEST_REFUSE_RANK)")
            capi.check(lib, lib.mppi_mailbox_create(P, rank, world))
            fine, nrec, nr = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
            capi.check(lib, lib.mppi_mailbox_info(P, ctypes.byref(fine), ctypes.byref(nrec), ctypes.byref(nr)))
            self.probe.update(fine_grained=bool(fine.value), records_per_rank=nrec.value)
            h = (ctypes.c_ubyte * 64)()
            capi.check(lib, lib.mppi_mailbox_ipc_handle(P, h))     # (refuses a coarse-grained inbox)
        except Exception as e:  # noqa: BLE001
            ok, why, h = 0, f"create on rank {rank}: {e}", (ctypes.c_ubyte * 64)()
        handles = [None] * world
        dist.all_gather_object(handles, (ok, why, bytes(h)))
        if not all(o for o, _, _ in handles):
            why = "; ".join(w for o, w, _ in handles if not o)
            print(f"[bench] mailbox exchange not available ({why}); RCCL all-gather", file=sys.stderr)
            return False, why
        try:
            for r, (_, _, hb) in enumerate(handles):
                if r != rank:   # (opens the peer's inbox and checks the mapping with a copy-engine write + read-back before any kernel stores to it)
                    capi.check(lib, lib.mppi_mailbox_open(P, r, (ctypes.c_ubyte * 64).from_buffer_copy(hb)))
            gp, gn = ctypes.c_void_p(), ctypes.c_int()
            capi.check(lib, lib.mppi_mailbox_gathered(P, ctypes.byref(gp), ctypes.byref(gn)))
            self.gathered, self.n_gathered = gp, gn.value
        except Exception as e:  # noqa: BLE001
            ok, why = 0, f"open on rank {rank}: {e}"
        flags = [None] * world
        dist.all_gather_object(flags, (ok, why))
        if not all(o for o, _ in flags):
            why = "; ".join(w for o, w in flags if not o)
            print(f"[bench] mailbox exchange not available ({why}); RCCL all-gather", file=sys.stderr)
            return False, why
        # probe: same records through both exchanges
        RF = lib.mppi_record_floats(P)
        hip = ctypes.CDLL("libamdhip64.so")
        hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        mine = torch.zeros((self.n_records, RF), dtype=torch.float32, device=self.records.device)
        for it in range(3):
            capi.check(lib, lib.mppi_rollout(P))
            if not self.inplace:
                capi.check(lib, lib.mppi_reduce(P, ctypes.c_void_p(self.mine.data_ptr())))
            if env["backend"] == "nccl":
                dist.all_gather_into_tensor(self.records.view(-1), self.mine.view(-1))
            else:
                host = torch.empty(self.records.shape, dtype=self.records.dtype)
                torch.cuda.current_stream().synchronize()
                dist.all_gather_into_tensor(host.view(-1), self.mine.cpu().view(-1))
                self.records.copy_(host)
            capi.check(lib, lib.mppi_exchange(P))
            late = ctypes.c_int(0)
            capi.check(lib, lib.mppi_exchange_status(P, ctypes.byref(late)))
            torch.cuda.synchronize()
            same = 0
            if self.n_gathered == self.n_records and not late.value:
                hip.hipMemcpy(ctypes.c_void_p(mine.data_ptr()), self.gathered, self.n_records * RF * 4, 3)   # device to device
                torch.cuda.synchronize()
                same = int(torch.equal(mine, self.records))
            res = [None] * world
            dist.all_gather_object(res, (same, late.value))
            self.probe.update(iterations=it + 1, late=[l for _, l in res], equal_per_rank=[e for e, _ in res])
            if not all(e for e, _ in res):
                why = f"probe iteration {it}: late per rank {[l for _, l in res]}, gathered == all-gather per rank {[e for e, _ in res]}"
                print(f"[bench] mailbox probe failed ({why}); RCCL all-gather", file=sys.stderr)
                return False, why
            capi.check(lib, lib.mppi_update(P, self.gathered, self.n_gathered))
        return True, "3 iterations, gathered records bit-equal to the all-gather on every rank, no late rank"

    def exchange_report(self):
        """what this rank ended up doing with its shard records, for the result line (SCALE records explain themselves)"""
        if not self.env["sharded"]:
            return None
        late = None
        if self.exchange == "mailbox":
            v = ctypes.c_int(0)
            self.capi.check(self.lib, self.lib.mppi_exchange_status(self.P, ctypes.byref(v)))
            late = v.value
        # (what the collective library itself says about the job: world size and backend as torch.distributed sees them, the device
        # this rank computes on; which other devices of the node it can reach as a peer - what the mailbox's inboxes need)
        peers = None
        try:
            hip = ctypes.CDLL("libamdhip64.so")
            n_dev, me = self.torch.cuda.device_count(), self.env["local_rank"]
            peers = {}
            for d in range(n_dev):
                if d != me:
                    can = ctypes.c_int(-1)
                    rc = hip.hipDeviceCanAccessPeer(ctypes.byref(can), ctypes.c_int(me), ctypes.c_int(d))
                    peers[str(d)] = can.value if rc == 0 else f"hip error {rc}"
        except Exception as e:  # noqa: BLE001
            peers = f"not queried: {e}"
        return {"rank": self.env["rank"], "device": self.env["local_rank"], "device_name": self.torch.cuda.get_device_name(self.env["local_rank"]),
                "dist_world_size": self.dist.get_world_size(), "dist_rank": self.dist.get_rank(), "dist_backend": str(self.dist.get_backend()),
                "selected": self.exchange, "why": self.exchange_why, "probe": self.probe, "peer_access": peers,
                "mppi_exchange_status": late, "graph": self.graph is not None}

    def time_exchange(self, n=50):
        """per-iteration cost of the exchange alone (same records again and again): enqueue + wait, mean over n"""
        torch, lib, P, capi = self.torch, self.lib, self.P, self.capi
        if not self.env["sharded"]:
            return None
        self.barrier()
        t0 = time.perf_counter()
        for _ in range(n):
            if self.exchange == "mailbox":
                capi.check(lib, lib.mppi_exchange(P))
            elif self.env["backend"] == "nccl":
                self.dist.all_gather_into_tensor(self.records.view(-1), self.mine.view(-1))
            else:
                return None
            torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / n

    # ---- one control iteration, enqueued (no host wait)
    def enqueue(self):
        lib, P, W, capi, env = self.lib, self.P, self.W, self.capi, self.env
        capi.check(lib, lib.mppi_rollout(P))
        if env["sharded"] and self.exchange == "mailbox":
            # publish into every inbox, poll the own one, combine, step the world: ONE launch for the contact-free scenes
            # (the exchange is the head of the combine + world kernel), exchange kernel + tail kernels otherwise
            capi.check(lib, lib.mppi_exchange_update_step_world(P, W))
        elif env["sharded"]:
            if not self.inplace:
                capi.check(lib, lib.mppi_reduce(P, ctypes.c_void_p(self.mine.data_ptr())))
            if env["backend"] == "nccl":
                self.dist.all_gather_into_tensor(self.records.view(-1), self.mine.view(-1))
            else:  # gloo smoke test on a single-GPU box: records staged through the host
                host = self.torch.empty(self.records.shape, dtype=self.records.dtype)
                self.torch.cuda.current_stream().synchronize()
                self.dist.all_gather_into_tensor(host.view(-1), self.mine.cpu().view(-1))
                self.records.copy_(host)
            capi.check(lib, lib.mppi_update_step_world(P, ctypes.c_void_p(self.records.data_ptr()), self.n_records, W))
        else:
            capi.check(lib, lib.mppi_update_step_world(P, None, 1, W))  # combine + update + world step + state feedback

    def capture(self):
        """the sharded iteration as ONE HIP graph (library launches + the RCCL all-gather captured by torch): per-iteration
        host work shrinks to one graph launch.  Returns False (eager loop stays) if the capture is refused."""
        torch, lib, capi = self.torch, self.lib, self.capi
        main = torch.cuda.current_stream()
        try:
            g = torch.cuda.CUDAGraph()
            torch.cuda.synchronize()
            with torch.cuda.graph(g):
                cs = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
                for c in (self.P, self.W):
                    capi.check(lib, lib.mppi_set_stream(c, cs))
                self.enqueue()
            capi.check(lib, lib.mppi_note_graph_update(self.P, -1))  # (the update launched under capture was recorded, not run)
            self.graph = g
        except Exception as e:  # noqa: BLE001 - any refusal (RCCL capture unsupported, sync inside) keeps the eager loop
            print(f"[bench] graph capture refused: {type(e).__name__}: {e}", file=sys.stderr)
            self.graph = None
            torch.cuda.synchronize()
        finally:
            for c in (self.P, self.W):
                capi.check(lib, lib.mppi_set_stream(c, ctypes.c_void_p(main.cuda_stream)))
        return self.graph is not None

    def iterate(self):
        lib, P, capi = self.lib, self.P, self.capi
        if self.graph is not None:
            self.graph.replay()
            capi.check(lib, lib.mppi_note_graph_update(P, 1))
        else:
            self.enqueue()
        if self.sync:
            # the controller output reaches the host as soon as the update kernel has published it (polled sequence
            # number in mapped host memory); MPPI_BENCH_ACTION=sync waits for the whole stream instead
            capi.check(lib, (lib.mppi_get_action if self.env["action_sync"] else lib.mppi_wait_action)(P, self._ap))

    def barrier(self):
        if self.env["sharded"]:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def run(self, steps, warmup, profile=True):
        """W warm-up + exactly `steps` timed iterations; returns elapsed (max over ranks), per-iteration times, kernel ms"""
        torch, lib, P, capi = self.torch, self.lib, self.P, self.capi
        for _ in range(warmup):
            self.iterate()
        prof = profile and self.graph is None
        if prof:
            capi.check(lib, lib.mppi_set_profiling(P, 4))  # hipEvent brackets around every 4th launch of each kernel
        stamps = np.zeros(steps + 1)
        self.barrier()
        stamps[0] = t0 = time.perf_counter()
        for i in range(steps):
            self.iterate()
            stamps[i + 1] = time.perf_counter()
        self.barrier()
        elapsed = time.perf_counter() - t0
        kms = [0.0, 0.0, 0.0]
        if prof:
            for which in range(3):
                ms = ctypes.c_float()
                rc = lib.mppi_kernel_ms(P, which, ctypes.byref(ms))  # rc != 0: that kernel was not launched (fused tail)
                kms[which] = ms.value if rc == 0 else 0.0
            capi.check(lib, lib.mppi_set_profiling(P, 0))
        if self.env["sharded"]:
            t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if self.env["backend"] == "nccl" else "cpu")
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed, np.diff(stamps), kms

    def profile_kernels(self, steps=40):
        """kernel durations by hipEvents in an eager pass (a captured graph cannot carry the event brackets)"""
        g, self.graph = self.graph, None
        _, _, kms = self.run(steps, 0, profile=True)
        self.graph = g
        return kms


def hbm_copy_ceiling(torch, min_ms=0.0):
    """measured HBM ceiling next to the 8 TB/s spec: read + write of a 256 MiB device-to-device copy, best of the
    batches of 10 copies that fit into `min_ms` (at least one: the first batch on a device that has just been opened reads
    5.1-5.3 TB/s, later ones 5.5-5.6).  Taken between the construction of the loop and its timed run.  (Measured, round 4: neither
    this load, nor 1000 launches of the fp64 sampler, nor up to 2000 extra iterations in front of the warm-up change what the driver's short command - 5 warm-up + 20
    timed iterations, 3 ms in all - reports: 7840-7950 Hz against 7925-7983 Hz over 200-2000 iterations; the spread between
    boxes is larger than that)"""
    a = torch.empty(64 << 20, dtype=torch.float32, device="cuda")
    b = torch.empty_like(a)
    b.copy_(a)
    torch.cuda.synchronize()
    best, t_end = 0.0, time.perf_counter() + 1e-3 * min_ms
    while True:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            b.copy_(a)
        e1.record()
        torch.cuda.synchronize()
        best = max(best, 10 * 2 * a.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9)
        if time.perf_counter() >= t_end:
            return best


def time_sampler(loop, n=20):
    """the halton-spline set is fixed (sampled once at construction, never inside the loop): its one-off cost"""
    import torch
    lib, P, capi = loop.lib, loop.P, loop.capi
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        capi.check(lib, lib.mppi_sample(P, ctypes.c_uint32(0)))
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def cpu_baseline(loop, budget_s=24.0):
    """CPU rows beside the GPU number, on this box's host cores, bounded to ~budget_s of CPU work.  Every row runs in its own
    process (oracle/cpu_pipeline.py) so that thread counts and the OpenMP wait policy are fixed before any runtime starts:
      rows[0], rows[1]  the REFERENCE-STRUCTURED pipeline (Python horizon loop -> batched C env step -> torch-CPU
                        Objective.compute_cost, reference mppi_isaac.py:57-69) with 1 thread and with all host threads
                        (OMP_NUM_THREADS = torch threads = os.cpu_count()), on the bench workload and ITS noise;
      rows[2]           the same pipeline on BASELINE configs[0] (point_robot K=64 H=10, the reference's CPU-runnable case);
      rows[3]           the oracle's fused C loop (oracle/mppi_oracle.c, OpenMP over samples): faster than the reference
                        structure, kept as the conservative comparison.
    A row may time the first `samples_timed` samples of the set and scale to K (per-sample work is independent, time is
    linear in K): contact scenes at K=8192 would take minutes on one thread.  `value` = rows[1]."""
    import subprocess
    import tempfile
    sim, capi = loop.planner.sim, loop.capi
    cores = os.cpu_count() or 1
    K, H, nu = loop.K, loop.H, loop.nu
    eps = np.zeros((H, nu, K), np.float32)
    capi.check(sim._lib, sim._lib.mppi_get_noise(sim._ctx, capi.fptr(eps)))
    tmp = tempfile.mkdtemp(prefix="mppi_cpu_")
    np.save(os.path.join(tmp, "eps.npy"), eps)
    np.savez(os.path.join(tmp, "state.npz"), dof0=loop.dof0, root0=loop.root0)
    unit = f"Hz (K={K},H={H} control iterations/s)"
    k_total = K * loop.env["world_size"]

    def row(label, threads, mode, k_local, budget, workload=loop.name, kt=k_total, horizon=H, own_inputs=True, max_iters=10):
        env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads), OMP_WAIT_POLICY="PASSIVE", GOMP_SPINCOUNT="0",
                   OMP_PROC_BIND="false")
        cmd = [sys.executable, "-m", "oracle.cpu_pipeline", "--workload", workload, "--k-total", str(kt), "--k-local", str(k_local),
               "--horizon", str(horizon), "--threads", str(threads), "--budget", str(budget), "--mode", mode, "--max-iters", str(max_iters)]
        if own_inputs:
            cmd += ["--eps", os.path.join(tmp, "eps.npy"), "--state", os.path.join(tmp, "state.npz")]
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
        if out.returncode != 0:
            raise RuntimeError(f"cpu baseline row failed: {out.stderr[-1500:]}")
        r = json.loads(out.stdout.strip().splitlines()[-1])
        dt = r["seconds_per_iteration"] * (kt // loop.env["world_size"] if workload == loop.name else kt) / k_local
        return {"pipeline": label, "threads": threads, "value": 1.0 / dt, "unit": unit if workload == loop.name else f"Hz (K={kt},H={horizon} control iterations/s)",
                "ms_per_iteration": dt * 1e3, "iterations": r["iterations"], "samples_timed": k_local, "torch_threads": r["torch_threads"]}

    per_sample_cost = H * (30 if loop.name in ("boxer_push", "panda_pick") else 1)
    k1, kn = min(K, max(256, 65536 // per_sample_cost)), min(K, max(2048, 2 ** 21 // per_sample_cost))
    ref = "reference-structured (python horizon loop -> batched C step -> torch-CPU Objective)"
    rows = [row(ref, 1, "pipeline", k1, budget_s * 0.35), row(ref, cores, "pipeline", kn, budget_s * 0.25),
            row(ref + ", BASELINE configs[0] point_robot K=64 H=10", 1, "pipeline", 64, budget_s * 0.1, workload="point_reach", kt=64, horizon=10,
                own_inputs=False, max_iters=200),
            row("oracle fused C loop (oracle/mppi_oracle.c, OpenMP over samples)", cores, "fused", kn, budget_s * 0.25, max_iters=20)]
    return {"value": rows[1]["value"], "unit": unit, "cores": cores, "kind": "port",
            "sample": f"{rows[1]['iterations']} open-loop control iterations of the same K={K} x H={H} workload (same noise set) through the "
                      f"reference-structured CPU pipeline (oracle/cpu_pipeline.py, fp32) on {cores} host threads "
                      f"({rows[1]['ms_per_iteration']:.1f} ms/iteration); rows: 1 thread, all threads, configs[0], fused C loop",
            "rows": rows}


class ReferenceStyleReach(object):
    """A Python Objective written the way the reference's examples write theirs (examples/panda/planner.py:10-40: weights dict,
    reset(), compute_cost(sim) over the gym getters with torch; pytorch3d's two rotation helpers come from mppiisaac.utils.conversions
    because pytorch3d is not in the image).  It declares nothing else - no cost program, no fused_spec, no graph_safe - so the
    planner runs it in GENERIC mode: horizon simulated by the HIP kernels, cost evaluated by this Python code."""

    def __init__(self, cfg=None):
        self.weights = {"robot_to_goal": 1.0, "robot_ori": 0.5}
        self.reset()

    def reset(self):
        pass

    def compute_cost(self, sim):
        import torch
        from mppiisaac.utils.conversions import matrix_to_euler_angles, quaternion_to_matrix
        ee = sim.get_actor_link_by_name("panda", "panda_ee_tip")
        goal = sim.get_actor_position_by_name("goal")
        to_goal = torch.linalg.norm(ee[:, 0:3] - goal[:, 0:3], axis=1)
        tilt = torch.linalg.norm(matrix_to_euler_angles(quaternion_to_matrix(ee[:, 3:7]), "ZYX")[:, 0:2], axis=1)
        return self.weights["robot_to_goal"] * to_goal + self.weights["robot_ori"] * tilt


class ReferenceStyleReachGraphSafe(ReferenceStyleReach):
    """the same Objective with the one-line opt-in `graph_safe = True` (a pure tensor program of sim tensors and .weights): the
    planner may capture its evaluation over the horizon into a HIP graph (planner/mppi.py)"""
    graph_safe = True


def facade_loop(wl_name, objective, device, steps, warmup, profile_to=None):
    """The loop a user of the REFERENCE runs (examples/panda/world.py:32-50 + planner.py:43-48, both in one process): a K = 1
    IsaacGymWrapper world, an MPPIisaacPlanner, state and action exchanged as torch.save blobs through
    `compute_action_tensor(dof_bytes, root_bytes)`, the world stepped from Python with `apply_robot_cmd(action); step()`.
    Every iteration waits for its action on the host by construction (the action is the payload of the returned blob).
    -> (Hz, ms per iteration, final end-effector distance to the goal)"""
    import torch
    from mppiisaac.planner.isaacgym_wrapper import IsaacGymWrapper
    from mppiisaac.planner.mppi_isaac import MPPIisaacPlanner
    from mppiisaac.utils.transport import bytes_to_torch, torch_to_bytes
    wl = WORKLOADS[wl_name]
    cfg = make_cfg(wl, wl["K"])
    cfg.mppi.device = device
    planner = MPPIisaacPlanner(cfg, objective)
    world = IsaacGymWrapper(cfg.isaacgym, actors=cfg.actors, init_positions=cfg.initial_actor_positions, num_envs=1, device=device)
    if wl["goal"] is not None:
        for sim in (planner.sim, world):
            sim.set_actor_position_by_name(wl["goal"], "goal")
    if wl["q0"] is not None:
        dof0 = world._dof_state[0].cpu().numpy().copy()
        dof0[0::2] = wl["q0"]
        world._push_single_state(dof0, world._root_state[0].cpu().numpy())

    def iterate():
        action = bytes_to_torch(planner.compute_action_tensor(torch_to_bytes(world._dof_state), torch_to_bytes(world._root_state)))
        world.apply_robot_cmd(action)
        world.step()
    import gc
    for _ in range(warmup):
        iterate()
    torch.cuda.synchronize()
    # (as timeit does: no cyclic garbage collection inside the timed region - a generation-2 pass over what the earlier phases of a
    # bench run left behind costs milliseconds, the loop itself allocates a few dozen short-lived objects per iteration)
    gc.collect()
    gc.disable()
    stamps = np.zeros(steps + 1)
    try:
        stamps[0] = t0 = time.perf_counter()
        for i in range(steps):
            iterate()
            stamps[i + 1] = time.perf_counter()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
    finally:
        gc.enable()
    facade_loop.last_latency_ms = {"median": float(np.median(np.diff(stamps)) * 1e3), "p5": float(np.percentile(np.diff(stamps), 5) * 1e3),
                                   "p95": float(np.percentile(np.diff(stamps), 95) * 1e3), "max": float(np.diff(stamps).max() * 1e3),
                                   "over_1ms": [(int(i), round(float(v) * 1e3, 2)) for i, v in enumerate(np.diff(stamps)) if v > 1e-3][:12]}
    dist = None
    if wl_name == "panda_reach":
        ee = world.get_actor_link_by_name("panda", "panda_ee_tip")[0, 0:3].cpu().numpy()
        dist = float(np.linalg.norm(ee - np.asarray(wl["goal"])))
    if profile_to is not None:
        import cProfile
        import io
        import pstats
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(steps):
            iterate()
        torch.cuda.synchronize()
        pr.disable()
        buf = io.StringIO()
        pstats.Stats(pr, stream=buf).sort_stats("cumulative").print_stats(40)
        with open(profile_to, "a") as f:
            f.write(f"==== {type(objective).__name__}: {1 / dt:.0f} Hz, {dt * 1e3:.4f} ms / iteration (un-profiled), cProfile of {steps} iterations\n")
            f.write(buf.getvalue())
    # which path ran: the in-kernel cost kind the planner bound (a traced Objective says so), or generic mode
    fc = planner.mppi._fused_cost
    facade_loop.last_checks = getattr(planner.mppi, "trace_check_ms", None)
    facade_loop.last_mode = ("generic (torch on the simulated horizon)" if fc is None else
                             ("traced -> " if getattr(planner.mppi, "_trace_guard", None) is not None else "declared -> ") + f"in-kernel cost kind {fc.kind}")
    # (the K = 1 wrapper registers host mirrors that hold it alive - stop it explicitly, `del` alone leaks the context)
    world.stop_sim()
    planner.sim.stop_sim()
    del planner, world
    return 1.0 / dt, dt * 1e3, dist


def facade_rows(wl_name, device):
    """`value_facade` / `value_generic_objective` of the result line: the reference-API loop on the metric's workload"""
    import mppiisaac.objectives as objectives
    n, w = int(os.environ.get("MPPI_BENCH_FACADE_STEPS", "400")), 30
    prof = os.environ.get("MPPI_BENCH_FACADE_PROFILE")
    rows = {}
    # `generic`: the reference-style Objective as the planner runs it by default since round 6 - traced once into a cost program
    # (mppiisaac/trace.py), validated against the Python code on the first command and every 64th; `generic_untraced` /
    # `generic_graph_safe`: the same Objective with MPPI_TRACE_OBJECTIVE=0 - torch evaluates it on the kernel-simulated horizon
    for key, obj, traced in (("fused", getattr(objectives, WORKLOADS[wl_name]["objective"])(None), True), ("generic", ReferenceStyleReach(), True),
                             ("generic_untraced", ReferenceStyleReach(), False), ("generic_graph_safe", ReferenceStyleReachGraphSafe(), False)):
        if key != "fused" and wl_name != "panda_reach":
            continue
        if not traced:
            os.environ["MPPI_TRACE_OBJECTIVE"] = "0"
        try:
            hz, ms, dist = facade_loop(wl_name, obj, device, n, w, profile_to=prof)
        finally:
            os.environ.pop("MPPI_TRACE_OBJECTIVE", None)
        rows[key] = {"value": hz, "ms_per_step": ms, "steps": n, "final_ee_to_goal_m": dist, "latency_ms": facade_loop.last_latency_ms,
                     "mode": facade_loop.last_mode, "trace_validations_ms": facade_loop.last_checks}
    return rows


def roofline(loop, kms, hbm_measured, n_waves):
    """HBM view (north_star's yardstick) and instruction-issue view (the bound that matters here) of the rollout kernel"""
    K, H, nu = loop.K, loop.H, loop.nu
    bytes_alg = 4 * (3 * K * H * nu + 2 * K + H * nu)  # SURVEY.md 8d, per GPU per control iteration
    achieved = bytes_alg / (kms * 1e-3) / 1e9 if kms > 0 else 0.0
    key = loop.wl["desc"].split(" ")[0]
    traffic, traffic_src, issue = None, None, None
    pmc_path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    if os.path.exists(pmc_path):
        pmc = json.load(open(pmc_path)).get("by_workload", {}).get(key)
        if pmc and pmc.get("K", K) == K:
            traffic = pmc.get("k_rollout", {}).get("hbm_traffic_bytes_per_launch")
            traffic_src = f"profiles/{pmc.get('tag')}_pmc_summary.json (2*FETCH_SIZE + WRITE_SIZE, separate --pmc passes of this command)"
    sq_path = os.path.join(ROOT, "profiles", "sq_latest.json")
    if os.path.exists(sq_path):
        sq = json.load(open(sq_path)).get("by_workload", {}).get(key)
        if sq and sq.get("K", K) == K and kms > 0:
            k = sq["k_rollout"]
            valu, waves = k["SQ_INSTS_VALU_per_wave"], k["SQ_WAVES"]
            issue = {"valu_per_wave": valu, "salu_per_wave": k.get("SQ_INSTS_SALU_per_wave"), "lds_per_wave": k.get("SQ_INSTS_LDS_per_wave"),
                     "waves": waves, "kernel_ms": kms,
                     "lane_ops_per_s": valu * waves * 64 / (kms * 1e-3),
                     "frac_of_fp32_issue_peak": valu * waves * 64 / (kms * 1e-3) / FP32_LANE_OPS_PEAK,
                     "simd_occupancy": min(1.0, waves / N_SIMD), "waves_per_simd": waves / N_SIMD,
                     "issue_cycles_frac": k.get("issue_cycles_frac"), "wait_cycles_frac": k.get("wait_cycles_frac"),
                     "lds_bank_conflict_per_wave": k.get("SQ_LDS_BANK_CONFLICT_per_wave"),
                     "source": f"profiles/{sq.get('tag')}_sq_summary.json (rocprofv3 --pmc SQ_* passes of this command; kernel_ms live)"}
    info = ctypes.create_string_buffer(256)
    loop.capi.check(loop.lib, loop.lib.mppi_kernel_info(loop.P, info, 256))
    kind = dict(kv.split("=") for kv in info.value.decode().split())["rollout"]   # lane | quad | oct | scene | scene-quad | scene-oct | scene-oct-pair
    lane = kind in ("lane", "scene")
    scene = kind.startswith("scene")
    lps = 1 if lane else (8 if "oct" in kind else 4)
    # `bound`: what limits this kernel - instruction ISSUE of lone wavefronts, not HBM; achieved / peak / unit / frac keep
    # the HBM view the contract defines (algorithmic bytes per launch / kernel time against 8 TB/s), `issue_frac` is the same kernel
    # against the fp32 vector issue peak
    return {"bound": "issue", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "hbm_frac": achieved / HBM_PEAK_GBS, "issue_frac": issue["frac_of_fp32_issue_peak"] if issue else None,
            "traffic": traffic, "traffic_source": traffic_src, "kernel": ("k_rollout_scene" if scene else "k_rollout") + ("" if lane else "_quad") + ("<.., 8> (octet layout)" if (lps == 8 and not scene) else ""),
            "peak_measured": hbm_measured, "kernel_ms": kms, "bytes_alg_per_launch": bytes_alg, "wavefronts": n_waves, "lanes_per_sample": lps,
            "issue": issue,
            "note": "instruction-issue-bound path (SURVEY 8d; DESIGN.md 5-6): one sample per 8-lane octet (MPPI_ROLLOUT=quad: 4-lane quad) = K/8 (K/16) wavefronts on 1024 SIMDs; "
                    "`issue` restates the kernel against the fp32 vector issue rate from the committed SQ counters; "
                    "peak_measured = device-to-device copy of 256 MiB (read + write bytes / time) on this GPU"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)     # 20 warm-up + 200 timed iterations
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-facade", action="store_true", help="skip the reference-API loops (value_facade, value_generic_objective)")
    ap.add_argument("--async-loop", action="store_true", help="do not wait for the action on the host every iteration")
    ap.add_argument("--workload", default="panda_reach", choices=sorted(WORKLOADS), help="BASELINE config (default: the metric's)")
    ap.add_argument("--k-total", type=int, default=0,
                    help="total number of samples over all GPUs (default: the workload's K per GPU x --gpus); "
                         "`--workload panda_pick --k-total 65536` is BASELINE configs[4] at its stated size on however many GPUs")
    args = ap.parse_args()
    wl = WORKLOADS[args.workload]

    import torch
    import torch.distributed as dist

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: this process becomes the launcher - one rank per GPU through torch.distributed.run on
        # 127.0.0.1, rank 0's result line on this stdout, the launcher's exit code handed on
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        print(f"[bench] --gpus {args.gpus} without WORLD_SIZE: launching {' '.join(cmd[1:9])} ...", file=sys.stderr, flush=True)
        raise SystemExit(subprocess.run(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))).returncode)
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world_size:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world_size}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if args.k_total and args.k_total % world_size:
        raise SystemExit("--k-total must be a multiple of --gpus")
    K_PER_GPU = args.k_total // world_size if args.k_total else wl["K"]
    # one process per GPU over RCCL.  MPPI_BENCH_BACKEND=gloo (ranks may then share a GPU, records staged
    # through the host) exists only to smoke-test the sharded loop on a single-GPU box.
    backend = os.environ.get("MPPI_BENCH_BACKEND", "nccl")
    local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    # MPPI_BENCH_FORCE_DIST=1: run the sharded code path (process group, record all-gather) even with one rank, so
    # that the RCCL path can be exercised and its per-iteration overhead measured on a single-GPU box
    sharded = world_size > 1 or bool(os.environ.get("MPPI_BENCH_FORCE_DIST"))
    # stdout carries the ONE result line and nothing else: RCCL / gloo print version and connection banners to fd 1 from C,
    # so everything written to fd 1 before the result goes to stderr instead
    sys.stdout.flush()
    stdout_fd = os.dup(1)
    os.dup2(2, 1)
    if sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", str(rank))
        os.environ.setdefault("WORLD_SIZE", str(world_size))
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world_size, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world_size)
    # MPPI_BENCH_EXCHANGE: "rccl" = all-gather of the shard records (default: the path that has run on multi-GPU hardware before -
    # RCCL itself), "mailbox" = the library's own exchange (guarded by a probe against the all-gather, falls back to it).  For
    # real multi-rank jobs the mailbox is measured as well, in an A/B pass AFTER the result line is out (below): it has never
    # moved a byte between two GPUs, and the line the driver records must not depend on it.
    exchange = os.environ.get("MPPI_BENCH_EXCHANGE", "rccl")
    env = dict(world_size=world_size, rank=rank, local_rank=local_rank, sharded=sharded, backend=backend, exchange=exchange,
               action_sync=os.environ.get("MPPI_BENCH_ACTION") == "sync")
    sync = not args.async_loop
    # The sharded iteration as one captured HIP graph (library launches + the RCCL all-gather): measured with one rank
    # (MPPI_BENCH_FORCE_DIST=1) 0.1897 -> 0.1813 ms per iteration.  Default: on for the one-rank measurement, OFF for real
    # multi-rank runs unless MPPI_BENCH_GRAPH=1 - a capture of an 8-rank RCCL collective could not be exercised on the
    # one-GPU development box, and a hang inside a replay cannot be caught.
    # With the mailbox exchange the captured iteration contains library kernels only, and their waits are bounded: the graph is on.
    use_graph_rccl = sharded and backend == "nccl" and os.environ.get("MPPI_BENCH_GRAPH", "1" if world_size == 1 else "0") == "1"

    hbm = {}

    def measure(name, k_per_gpu, steps, warmup, shipped=False, env=env, ceiling=False):
        loop = Loop(name, k_per_gpu, env, sync=sync, shipped=shipped)
        if ceiling:  # (between the construction of the loop - host work, the device idles - and its timed run)
            hbm["measured"] = hbm_copy_ceiling(torch, float(os.environ.get("MPPI_BENCH_CEILING_MS", "20")))
        graphed = False
        use_graph = (use_graph_rccl and getattr(loop, "exchange", "") != "mailbox") or \
                    (sharded and getattr(loop, "exchange", "") == "mailbox" and os.environ.get("MPPI_BENCH_GRAPH", "1") == "1")
        if use_graph:
            for _ in range(3):
                loop.iterate()     # lazy initialisation (RCCL channels, LDS limits) must not happen under capture
            graphed = loop.capture()
        elapsed, per_iter, kms = loop.run(steps, warmup)
        if graphed:
            kms = loop.profile_kernels()
        return loop, elapsed, per_iter, kms, graphed

    def gather_reports(loop):
        """every rank's exchange report on rank 0 (None when not sharded)"""
        rep = loop.exchange_report()
        if not sharded:
            return None
        out = [None] * world_size
        dist.all_gather_object(out, rep)
        return out

    def strong_row(env=env):
        """BASELINE configs[4] at its stated size, K_total = 65536 x H = 30 split over the ranks: the STRONG-scaling row (the
        driver's N = 1, 2, 4, 8 lines hold the same total work), with the exchange timed on its own"""
        kt = 65536
        if kt % world_size:
            return None
        l2, e2, p2, k2, g2 = measure("panda_pick", kt // world_size, max(10, min(args.steps, 40)), min(args.warmup, 5), env=env)
        row = {"workload": WORKLOADS["panda_pick"]["desc"].replace("8192 samples per GPU", "K_total = 65536 over all GPUs"), "scaling": "strong",
               "K_per_gpu": l2.K, "K_total": kt, "H": l2.H, "steps": len(p2), "ms_per_step": 1e3 * e2 / len(p2), "loop_hz": len(p2) / e2,
               "env_steps_per_s": len(p2) / e2 * kt * l2.H, "rollout_kernel_ms": k2[0], "graph": g2,
               "exchange": getattr(l2, "exchange", None), "exchange_ms": l2.time_exchange() if sharded else None}
        del l2
        return row

    loop, elapsed, per_iter, kms, graphed = measure(args.workload, K_PER_GPU, args.steps, args.warmup, ceiling=True)
    exchange_ms = loop.time_exchange() if sharded else None   # (every rank takes part)
    reports = gather_reports(loop)
    # SURVEY 8e's scaling argument is made on BASELINE configs[4] (compute >> exchange): its strong-scaling row is timed in
    # the same job at every N, the weak row (4096 panda samples per GPU) being the line's own `value`
    second = None
    if args.workload == "panda_reach" and not args.k_total and os.environ.get("MPPI_BENCH_SECOND", "1") != "0":
        second = strong_row()

    # the same workload with the conf file AS THE REFERENCE SHIPS IT (filter_u: True in conf/mppi/panda.yaml:22): the smoothing
    # operator U <- F U runs inside the same combine kernel, the iteration costs the same
    shipped = None
    if os.environ.get("MPPI_BENCH_SHIPPED", "1") != "0":
        ls, es, ps, ks, gs = measure(args.workload, K_PER_GPU, max(20, args.steps // 2), min(args.warmup, 10), shipped=True)
        shipped = {"value": world_size * len(ps) / es, "ms_per_step": 1e3 * es / len(ps), "filter_u": bool(ls.cfg.mppi.filter_u),
                   "steps": len(ps), "graph": gs}
        del ls

    # final state sanity (default workload): the closed loop must have moved the end effector to the goal
    world = loop.world
    world._materialise()
    dist_to_goal = None
    final_root = [round(float(v), 4) for v in world._root_state[0, :, 0:3].reshape(-1).cpu().numpy()]  # actor positions: run-to-run sanity
    if args.workload == "panda_reach":
        ee = world.get_actor_link_by_name("panda", "panda_ee_tip")[0, 0:3].cpu().numpy()
        dist_to_goal = float(np.linalg.norm(ee - np.asarray(wl["goal"])))
    # task outcome of the contact workloads (what their Objectives drive down: examples/boxer_push/planner.py:26-67,
    # examples/panda_pick/planner.py:24-53) after the warm-up + timed iterations of this run
    outcome = None
    if args.workload in ("boxer_push", "panda_pick"):
        o = loop.objective
        pos = lambda name: world.get_actor_position_by_name(name)[0, 0:3].cpu().numpy()
        ee = world.get_actor_link_by_name(o.robot, o.link)[0, 0:3].cpu().numpy()
        nd = 2 if args.workload == "boxer_push" else 3
        outcome = {"final_block_to_goal_m": float(np.linalg.norm((pos(o.block) - pos(o.goal))[:nd])),
                   "final_ee_to_block_m": float(np.linalg.norm((ee - pos(o.block))[:nd])),
                   "iterations": args.steps + args.warmup, "dims": nd}
        if args.workload == "panda_pick":
            # (round 6: the one-gram block is held implicitly by the links that touch it, DESIGN.md 3 "light bodies"; whether THIS run's
            # few hundred iterations got as far as lifting it is in the height - tests/test_gpu_task_outcomes.py runs the task to its end)
            table = next((a for a in world.scene.env_cfg if a.name == o.table), None)
            blk = next((a for a in world.scene.env_cfg if a.name == o.block), None)
            if table is not None and blk is not None:
                outcome["final_block_height_above_table_m"] = float(pos(o.block)[2] - (pos(o.table)[2] + 0.5 * table.size[2] + 0.5 * blk.size[2]))
        if args.workload == "boxer_push":
            # the goal of the reference's pushing examples sits INSIDE the footprint of paper_obst1 (both at (1, 1): reference
            # examples/boxer_push/config_boxer_push.yaml, conf/actors/paper_obst1.yaml): the block is done when it rests against that
            # obstacle - the distance it can reach is the obstacle's and its own half extent, not zero (tools/task_outcomes.py)
            obst = next((a for a in world.scene.env_cfg if a.name == o.obstacles[0]), None)
            blk = next((a for a in world.scene.env_cfg if a.name == o.block), None)
            d = pos(o.goal)[:2] - pos(o.obstacles[0])[:2]
            if obst is not None and blk is not None and abs(d[0]) < 0.5 * obst.size[0] and abs(d[1]) < 0.5 * obst.size[1]:
                outcome["goal_is_inside_the_footprint_of"] = o.obstacles[0]
                outcome["closest_the_block_can_get_m"] = float(0.5 * min(obst.size[0], obst.size[1]) + 0.5 * min(blk.size[0], blk.size[1]))
    # the loop a user of the reference runs (MPPIisaacPlanner.compute_action_tensor with torch.save blobs, a K = 1 world stepped from
    # Python): with the example's Objective fused, and with a reference-style Python compute_cost (generic mode)
    facade = None
    if world_size == 1 and not args.no_facade and os.environ.get("MPPI_BENCH_FACADE", "1") != "0":
        facade = facade_rows(args.workload, loop.cfg.mppi.device)

    if rank == 0:
        loop_hz = args.steps / elapsed
        K, H, nu = loop.K, loop.H, loop.nu
        lat = per_iter * 1e3
        info = ctypes.create_string_buffer(256)
        loop.capi.check(loop.lib, loop.lib.mppi_kernel_info(loop.P, info, 256))
        info = dict(kv.split("=") for kv in info.value.decode().split())
        n_waves = int(info["waves"])
        sampler_ms = time_sampler(loop) if loop.planner.sim._mppi_config.sampling == 0 else None
        out = {
            "metric": "MPPI control-loop Hz (K samples x H horizon), Panda 7-DoF K=4096 H=20" if (args.workload == "panda_reach" and K == 4096)
                      else f"MPPI control-loop Hz, {args.workload} K={K} H={H} (not the BASELINE metric)",
            "value": loop_hz * world_size,
            "value_shipped_conf": shipped["value"] if shipped else None,
            "value_facade": facade["fused"]["value"] if facade else None,
            "value_generic_objective": facade["generic"]["value"] if facade and "generic" in facade else None,
            "value_generic_objective_untraced": facade["generic_untraced"]["value"] if facade and "generic_untraced" in facade else None,
            "unit": f"Hz ({K}-sample x {H}-step control iterations per second, summed over GPUs)",
            "n_gpus": world_size, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "latency_ms": {"median": float(np.median(lat)), "p5": float(np.percentile(lat, 5)), "p95": float(np.percentile(lat, 95)),
                           "hz_at_median": 1e3 / float(np.median(lat)),
                           "how": "host wall time between consecutive published actions of the SAME timed run" if sync
                                  else "enqueue time only (--async-loop)"},
            "config": {"workload": wl["desc"],
                       "K_per_gpu": K, "K_total": K * world_size, "H": H, "nu": nu, "dt": loop.cfg.isaacgym.dt,
                       "substeps": loop.cfg.isaacgym.substeps, "substeps_integrated": loop.planner.sim.substeps_integrated, "closed_loop": True, "action_to_host_every_step": sync,
                       "parallelism": (f"sample-shard x{world_size}: rollout -> "
                                       + (f"mailbox exchange (library kernels: store into every rank's inbox, poll the own one) of {loop.n_gathered} records"
                                          if loop.exchange == "mailbox" else f"in-place {backend} all-gather of {loop.n_records} folded records") + " -> combine"
                                       + (" (one captured HIP graph per iteration)" if graphed else "")) if sharded else "single GPU",
                       "noise": ("fixed halton-spline set, sampled once at construction (k_sample %.1f us, outside the loop)" % (1e3 * sampler_ms))
                                if sampler_ms is not None else "gaussian, redrawn on the device every iteration",
                       "loop_hz": loop_hz, "env_steps_per_s": loop_hz * K * H * world_size,
                       "final_ee_to_goal_m": dist_to_goal, "final_actor_positions": final_root, "task_outcome": outcome,
                       "facade": {"what": "the reference's own loop (examples/<x>/world.py:32-50): torch.save blobs through "
                                          "MPPIisaacPlanner.compute_action_tensor, K = 1 IsaacGymWrapper world stepped from Python with apply_robot_cmd + step; "
                                          "`fused`: the example's Objective as an in-kernel cost; `generic`: a reference-style Python compute_cost(sim) "
                                          "(bench.py ReferenceStyleReach, nothing declared) as the planner runs it by default - traced once into a cost program, "
                                          "validated against the Python code on the first command and every 64th (the validations are inside the timed loop); "
                                          "`generic_untraced`: the same Objective with MPPI_TRACE_OBJECTIVE=0, evaluated by torch on the kernel-simulated horizon "
                                          "(what `generic` was until round 5); `generic_graph_safe`: that with the `graph_safe = True` opt-in",
                                  **facade} if facade else None,
                       "exchange_ms": exchange_ms,
                       "exchange": {"selected": loop.exchange, "why": loop.exchange_why, "exchange_ms": exchange_ms, "per_rank": reports} if sharded else None,
                       "shipped_conf": shipped,
                       "cfg5_strong": second},
            "roofline": roofline(loop, kms[0], hbm["measured"], n_waves),
            "kernels_ms": {"k_rollout(+record tail)": kms[0], "k_reduce(generic mode only)": kms[1], "k_combine_update(+world step)": kms[2]},
        }
        if world_size == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(loop)
        ctypes.CDLL(None).fflush(None)  # (C stdio of the libraries: out through the redirected fd before it is restored)
        sys.stdout.flush()
        os.dup2(stdout_fd, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    # ---- A/B of the library's own exchange on real peers, AFTER the result line: same workloads through the mailbox (probe first;
    # any refusal keeps RCCL and says why).  Its numbers go to stderr and gpurun_out/ - they are evidence, not the metric.
    # (MPPI_BENCH_MAILBOX_AB=force: also with gloo staging - ranks sharing one device, tests/test_bench_contract.py)
    ab_mode = os.environ.get("MPPI_BENCH_MAILBOX_AB", "1")
    if world_size > 1 and (backend == "nccl" or ab_mode == "force") and exchange == "rccl" and ab_mode != "0":
        env_mb = dict(env, exchange="mailbox")
        try:
            lm, em, pm, km, gm = measure(args.workload, K_PER_GPU, max(20, args.steps // 2), min(args.warmup, 10), env=env_mb)
            ab = {"n_gpus": world_size, "workload": args.workload, "K_per_gpu": K_PER_GPU, "selected": lm.exchange, "why": lm.exchange_why,
                  "ms_per_step": 1e3 * em / len(pm), "value": world_size * len(pm) / em, "graph": gm, "exchange_ms": lm.time_exchange(),
                  "rccl_ms_per_step": 1e3 * elapsed / args.steps, "rccl_exchange_ms": exchange_ms}
            ab["per_rank"] = gather_reports(lm)
            if second is not None:
                ab["cfg5_strong"] = strong_row(env=env_mb)
            if rank == 0:
                print("[bench] mailbox_ab " + json.dumps(ab), file=sys.stderr, flush=True)
                try:
                    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                    with open(os.path.join(ROOT, "gpurun_out", f"mailbox_ab_n{world_size}.json"), "w") as f:
                        json.dump(ab, f)
                except OSError:
                    pass
        except Exception as e:  # noqa: BLE001 - the result line is out already
            # the peers may be inside a collective of the A/B pass: leaving quietly would strand them until a watchdog fires;
            # a non-zero exit makes the launcher end the job at once
            print(f"[bench] mailbox A/B failed on rank {rank}: {type(e).__name__}: {e}", file=sys.stderr, flush=True)
            os._exit(3)
    if sharded:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

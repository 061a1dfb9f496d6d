#!/usr/bin/env python3
"""Closed-loop MPPI benchmark on the BASELINE.json workload (Panda 7-DoF reach, K=4096, H=20).

One "step" = one control iteration: rollout kernel (K samples x H horizon steps of articulated-body
dynamics + fused cost) -> reduce -> (all-gather of shard records when --gpus > 1) -> nominal update
-> the K=1 world is stepped with the action and its new state is fed back (closed loop, everything
device-resident) -> the action is copied to the host.  Weak scaling: every GPU owns 4096 samples, the
softmax weights are combined over all ranks; `value` counts 4096-sample control iterations per second
summed over ranks (at --gpus 1 it is exactly the control-loop Hz at K=4096, H=20).

Prints ONE JSON line (rank 0).  See DESIGN.md "Measurement" for how roofline/cpu_baseline are defined.
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

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s measured achievable
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


def make_cfg(w, k_total):
    from mppiisaac.utils.config_store import load_config
    return load_config({"defaults": [{"mppi": w["mppi"]}, {"isaacgym": "normal"}], "actors": w["actors"],
                        "initial_actor_positions": w["init"], "nx": w["nx"]},
                       overrides={"mppi.num_samples": k_total, "mppi.horizon": w["H"], "mppi.filter_u": False,
                                  "mppi.use_priors": False})


def cpu_baseline(planner, dof, K_PER_GPU, HORIZON, seconds_budget=20.0):
    """The oracle (C restatement, fp32, OpenMP over samples) timed on this box's host cores on the same
    K=4096 x H=20 control iteration.  Bounded sample: as many iterations as fit ~seconds_budget."""
    from mppiisaac.backend import capi
    from oracle.oracle import Oracle
    sim = planner.sim
    cores = os.cpu_count() or 1
    os.environ["OMP_NUM_THREADS"] = str(cores)
    o = Oracle("f32")
    nu = sim.scene.nu
    eps = np.zeros((HORIZON, nu, K_PER_GPU), np.float32)
    capi.check(sim._lib, sim._lib.mppi_get_noise(sim._ctx, capi.fptr(eps)))
    root = sim._root_state[0].cpu().numpy()
    U = np.zeros((HORIZON, nu), np.float32)
    cost = planner.objective.fused_spec(sim)
    t0 = time.perf_counter()
    U, a, S = o.command(sim._c_model, sim._mppi_config, cost, dof, root, U, eps)
    first = time.perf_counter() - t0
    n = max(1, min(20, int(seconds_budget / max(first, 1e-3)) - 1))
    t0 = time.perf_counter()
    for _ in range(n):
        U, a, S = o.command(sim._c_model, sim._mppi_config, cost, dof, root, U, eps)
    dt = (time.perf_counter() - t0) / n
    return {"value": 1.0 / dt, "unit": f"Hz (K={K_PER_GPU},H={HORIZON} control iterations/s)", "cores": cores, "kind": "port",
            "sample": f"{n} open-loop control iterations of the same K={K_PER_GPU}xH={HORIZON} workload, oracle/mppi_oracle.c fp32, "
                      f"OpenMP over samples on {cores} host threads ({dt * 1e3:.1f} ms/iteration)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--async-loop", action="store_true", help="do not copy the action to the host every iteration")
    ap.add_argument("--workload", default="panda_reach", choices=sorted(WORKLOADS), help="BASELINE config (default: the metric's)")
    args = ap.parse_args()
    wl = WORKLOADS[args.workload]
    K_PER_GPU, HORIZON = wl["K"], wl["H"]

    import torch
    import torch.distributed as dist
    from mppiisaac.backend import capi
    import mppiisaac.objectives as objectives
    from mppiisaac.planner.isaacgym_wrapper import IsaacGymWrapper
    from mppiisaac.planner.mppi_isaac import MPPIisaacPlanner

    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world_size:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world_size}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
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

    cfg = make_cfg(wl, K_PER_GPU * world_size)
    cfg.mppi.device = f"cuda:{local_rank}"
    objective = getattr(objectives, wl["objective"])(cfg)
    planner = MPPIisaacPlanner(cfg, objective, shard=sharded)
    world = IsaacGymWrapper(cfg.isaacgym, actors=cfg.actors, init_positions=cfg.initial_actor_positions, num_envs=1,
                            device=cfg.mppi.device)
    lib, P, W = planner.sim._lib, planner.sim._ctx, world._ctx
    GOAL = wl["goal"]
    if GOAL is not None:
        for sim in (planner.sim, world):
            sim.set_actor_position_by_name(GOAL, "goal")
    dof0 = world._dof_state[0].cpu().numpy().copy()
    if wl["q0"] is not None:
        dof0[0::2] = wl["q0"]
    root0 = world._root_state[0].cpu().numpy()
    for sim in (planner.sim, world):
        sim._push_single_state(dof0, root0)
    planner._bind_objective()
    records = planner.mppi._records
    send = torch.zeros_like(records[0])  # this rank's shard record (written by mppi_reduce, gathered into `records`)
    nu = planner.sim.scene.nu
    action = np.zeros(nu, np.float32)
    ap_ = capi.fptr(action)

    action_sync = os.environ.get("MPPI_BENCH_ACTION") == "sync"

    def iterate(sync):
        capi.check(lib, lib.mppi_rollout(P))
        if sharded:
            capi.check(lib, lib.mppi_reduce(P, ctypes.c_void_p(send.data_ptr())))
            if backend == "nccl":
                dist.all_gather_into_tensor(records.view(-1), send)
            else:
                host = torch.empty(records.shape, dtype=records.dtype)
                dist.all_gather_into_tensor(host.view(-1), send.cpu())
                records.copy_(host)
            capi.check(lib, lib.mppi_update_step_world(P, ctypes.c_void_p(records.data_ptr()), world_size, W))
        else:
            capi.check(lib, lib.mppi_reduce(P, None))
            capi.check(lib, lib.mppi_update_step_world(P, None, 1, W))  # update + world step + state feedback
        if sync:
            # the controller output reaches the host as soon as the update kernel has published it (polled sequence
            # number in mapped host memory); MPPI_BENCH_ACTION=sync waits for the whole stream instead
            capi.check(lib, (lib.mppi_get_action if action_sync else lib.mppi_wait_action)(P, ap_))

    def barrier():
        if sharded:
            dist.barrier()
        torch.cuda.synchronize()

    sync = not args.async_loop
    for _ in range(args.warmup):
        iterate(sync)
    capi.check(lib, lib.mppi_set_profiling(P, 4))  # hipEvent brackets around every 4th launch of each kernel
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        iterate(sync)
    barrier()
    elapsed = time.perf_counter() - t0
    kms = []
    for which in range(3):
        ms = ctypes.c_float()
        rc = lib.mppi_kernel_ms(P, which, ctypes.byref(ms))  # rc != 0: that kernel was not launched (fused tail)
        kms.append(ms.value if rc == 0 else 0.0)
    capi.check(lib, lib.mppi_set_profiling(P, 0))
    if sharded:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # final state sanity (default workload): the closed loop must have moved the end effector to the goal
    world._materialise()
    dist_to_goal = None
    final_root = [round(float(v), 4) for v in world._root_state[0, :, 0:3].reshape(-1).cpu().numpy()]  # actor positions: run-to-run sanity
    if args.workload == "panda_reach":
        ee = world.get_actor_link_by_name("panda", "panda_ee_tip")[0, 0:3].cpu().numpy()
        dist_to_goal = float(np.linalg.norm(ee - np.asarray(GOAL)))

    if rank == 0:
        loop_hz = args.steps / elapsed
        K, H = K_PER_GPU, HORIZON
        bytes_alg = 4 * (3 * K * H * nu + 2 * K + H * nu)  # SURVEY.md 8d, per GPU per control iteration
        achieved = bytes_alg / (kms[0] * 1e-3) / 1e9
        traffic, traffic_src = None, None  # HBM bytes/launch from the last committed rocprofv3 PMC passes
        pmc_path = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc_path):
            pmc = json.load(open(pmc_path)).get("by_workload", {}).get(wl["desc"].split(" ")[0])
            if pmc:
                traffic = pmc.get("k_rollout", {}).get("hbm_traffic_bytes_per_launch")
                traffic_src = f"profiles/{pmc.get('tag')}_pmc_summary.json (2*FETCH_SIZE + WRITE_SIZE, separate --pmc passes of this command)"
        lane = os.environ.get("MPPI_ROLLOUT") == "lane"
        scene = args.workload in ("boxer_push", "panda_pick")
        kernel_name = ("k_rollout_scene" if scene else "k_rollout") + ("" if lane else "_quad")
        # measured HBM ceiling next to the 8 TB/s spec: read + write of a 256 MiB device-to-device copy
        a = torch.empty(64 << 20, dtype=torch.float32, device="cuda")
        b = torch.empty_like(a)
        b.copy_(a)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            b.copy_(a)
        e1.record()
        torch.cuda.synchronize()
        hbm_measured = 10 * 2 * a.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9
        del a, b
        out = {
            "metric": "MPPI control-loop Hz (K samples x H horizon), Panda 7-DoF K=4096 H=20" if args.workload == "panda_reach"
                      else f"MPPI control-loop Hz, {args.workload} K={K} H={H} (not the BASELINE metric)",
            "value": loop_hz * world_size,
            "unit": f"Hz ({K}-sample x {H}-step control iterations per second, summed over GPUs)",
            "n_gpus": world_size, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": wl["desc"],
                       "K_per_gpu": K, "K_total": K * world_size, "H": H, "nu": nu, "dt": cfg.isaacgym.dt,
                       "substeps": cfg.isaacgym.substeps, "closed_loop": True, "action_to_host_every_step": sync,
                       "parallelism": f"sample-shard x{world_size} ({backend} all-gather of the shard records)" if sharded else "single GPU",
                       "loop_hz": loop_hz, "env_steps_per_s": loop_hz * K * H * world_size,
                       "final_ee_to_goal_m": dist_to_goal, "final_actor_positions": final_root},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src, "kernel": kernel_name, "peak_measured": hbm_measured,
                         "kernel_ms": kms[0], "bytes_alg_per_launch": bytes_alg,
                         "note": "instruction-issue-bound path (SURVEY 8d; DESIGN.md 6): one sample per 4-lane quad = K/16 wavefronts, one per CU at K=4096; "
                                 "peak_measured = device-to-device copy of 256 MiB (read + write bytes / time) on this GPU"},
            "kernels_ms": {"k_rollout(+record tail)": kms[0], "k_reduce(generic mode only)": kms[1], "k_combine_update": kms[2]},
        }
        if world_size == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(planner, dof0, K_PER_GPU, HORIZON)
        ctypes.CDLL(None).fflush(None)  # (C stdio of the libraries: out through the redirected fd before it is restored)
        sys.stdout.flush()
        os.dup2(stdout_fd, 1)
        print(json.dumps(out), flush=True)
    if sharded:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Where the microseconds of the REFERENCE-API loop go (bench.py `value_facade` / `value_generic_objective`): the loop of
reference examples/panda/world.py:32-50 - torch.save blobs through MPPIisaacPlanner.compute_action_tensor, a K = 1 world stepped
from Python - on the metric's workload (panda reach, K = 4096, H = 20).

  1. the rates (un-instrumented), fused / generic / generic + graph_safe;
  2. a stage clock of the fused and the generic loop: host wall time per stage, averaged;
  3. cProfile of the same loops (cumulative).
Usage (GPU box): python tools/facade_profile.py > profiles/r05x_facade_profile.txt"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mppi-isaac_amd"))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import mppiisaac.objectives as objectives  # noqa: E402
from mppiisaac.planner.isaacgym_wrapper import IsaacGymWrapper  # noqa: E402
from mppiisaac.planner.mppi_isaac import MPPIisaacPlanner  # noqa: E402
from mppiisaac.utils.transport import bytes_to_torch, torch_to_bytes  # noqa: E402

N = int(os.environ.get("FACADE_STEPS", "500"))


def build(objective):
    wl = bench.WORKLOADS["panda_reach"]
    cfg = bench.make_cfg(wl, wl["K"])
    planner = MPPIisaacPlanner(cfg, objective)
    world = IsaacGymWrapper(cfg.isaacgym, actors=cfg.actors, init_positions=cfg.initial_actor_positions, num_envs=1)
    for sim in (planner.sim, world):
        sim.set_actor_position_by_name(wl["goal"], "goal")
    dof0 = world._dof_state[0].cpu().numpy().copy()
    dof0[0::2] = wl["q0"]
    world._push_single_state(dof0, world._root_state[0].cpu().numpy())
    return planner, world


def stage_clock(objective, label):
    planner, world = build(objective)
    T = {}

    def tick(name, t0):
        t1 = time.perf_counter()
        T[name] = T.get(name, 0.0) + (t1 - t0)
        return t1
    mppi = planner.mppi
    orig_command = mppi.command

    def it(clock):
        t = time.perf_counter()
        dof_t = world._dof_state
        root_t = world._root_state
        if clock: t = tick("world: state tensors (materialise + mirror launch)", t)
        b_dof = torch_to_bytes(dof_t)
        if clock: t = tick("world: torch_to_bytes(dof) incl. wait for the mirrored state", t)
        b_root = torch_to_bytes(root_t)
        if clock: t = tick("world: torch_to_bytes(root)", t)
        planner.objective.reset()
        planner.reset_rollout_sim(b_dof, b_root)
        if clock: t = tick("planner: reset_rollout_sim (2 x bytes_to_array + mppi_set_state launch)", t)
        planner._bind_objective()
        if clock: t = tick("planner: _bind_objective (fused_spec + compare)", t)
        a = orig_command(planner.state_place_holder)
        if clock: t = tick("planner: mppi.command (launches + wait for the action)", t)
        b_a = torch_to_bytes(a)
        if clock: t = tick("planner: torch_to_bytes(action)", t)
        action = bytes_to_torch(b_a)
        if clock: t = tick("world: bytes_to_torch(action)", t)
        world.apply_robot_cmd(action)
        world.step()
        if clock: t = tick("world: apply_robot_cmd + step (launch)", t)
    for _ in range(30):
        it(False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        it(True)
    torch.cuda.synchronize()
    total = (time.perf_counter() - t0) / N
    print(f"\n== stage clock, {label}: {1 / total:.0f} Hz with the clock in the loop, {total * 1e6:.1f} us / iteration")
    for k, v in T.items():
        print(f"   {v / N * 1e6:8.1f} us  {k}")
    print(f"   {(total - sum(T.values()) / N) * 1e6:8.1f} us  (clock overhead / rest)")
    # device side of the same iteration: kernel times by hipEvents
    import ctypes
    from mppiisaac.backend import capi
    lib, P = planner.sim._lib, planner.sim._ctx
    capi.check(lib, lib.mppi_set_profiling(P, 1))
    for _ in range(100):
        it(False)
    for which, name in ((0, "rollout kernel"), (1, "reduce kernel (generic mode)"), (2, "combine + update kernel")):
        ms = ctypes.c_float()
        if lib.mppi_kernel_ms(P, which, ctypes.byref(ms)) == 0:
            print(f"   device: {name}: {ms.value * 1e3:.1f} us")
    capi.check(lib, lib.mppi_set_profiling(P, 0))
    del planner, world


if __name__ == "__main__":
    print("reference-API loop, panda reach K=4096 H=20 (bench.py facade_loop):")
    for label, obj in (("fused (PandaReachObjective as in-kernel cost)", objectives.PandaReachObjective(None)),
                       ("generic (reference-style Python compute_cost)", bench.ReferenceStyleReach()),
                       ("generic + graph_safe", bench.ReferenceStyleReachGraphSafe())):
        hz, ms, dist = bench.facade_loop("panda_reach", obj, "cuda:0", N, 30)
        print(f"  {label:52s} {hz:8.1f} Hz  {ms * 1e3:7.1f} us / iteration   final ee-goal distance {dist:.3f} m")
    stage_clock(objectives.PandaReachObjective(None), "fused")
    stage_clock(bench.ReferenceStyleReach(), "generic")
    prof = os.path.join(ROOT, "gpurun_out", "facade_cprofile.txt")
    os.makedirs(os.path.dirname(prof), exist_ok=True)
    open(prof, "w").close()
    bench.facade_loop("panda_reach", objectives.PandaReachObjective(None), "cuda:0", N, 30, profile_to=prof)
    bench.facade_loop("panda_reach", bench.ReferenceStyleReach(), "cuda:0", N, 30, profile_to=prof)
    print("\n" + open(prof).read())

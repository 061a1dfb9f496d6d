#!/usr/bin/env python3
"""Record the closed-loop states that tests/test_gpu_parity.py, tests/test_gpu_state_parity.py and tools/exp/ab_time.py work at
(tests/golden/closed_loop_states.npz): a bench workload (K x H of BASELINE.json) is run closed loop for N iterations, then
    <workload>_recorded_{dof,root,U}   the world's state and the planner's nominal plan at that moment (<workload>:N1:N2 - a second
                                       state `held` after N2 iterations), and
    <workload>_violent<t>_{dof,root}   the env state of a sample after t steps of a rollout from the last one (violent states: the
                                       samples with the highest costs whose state is still finite and within 10 m)
go into the file; the entries of workloads that are not named stay as they are.  Needs a GPU.
    python tools/record_closed_loop_states.py panda_pick:70:lift [boxer_push:300]   (lift: until the block is 8 cm above where it lay) [--out tests/golden/closed_loop_states.npz]"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "mppi-isaac_amd"))
import bench  # noqa: E402


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    out = next((a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--out=")), os.path.join(ROOT, "tests", "golden", "closed_loop_states.npz"))
    keep = dict(np.load(out)) if os.path.exists(out) else {}
    env = {"world_size": 1, "rank": 0, "local_rank": 0, "sharded": False, "backend": None, "action_sync": False, "exchange": None}
    for spec in args:
        name, *ns = spec.split(":")
        ns = [v if v == "lift" else int(v) for v in ns]
        loop = bench.Loop(name, bench.WORKLOADS[name]["K"], env)
        world, planner, capi = loop.world, loop.planner, loop.capi
        for k in [k for k in keep if k.startswith(name + "_")]:
            del keep[k]
        done = 0
        # <name>:N -> `recorded` after N iterations; <name>:N1:N2 -> `recorded` after N1 (the task under way), `held` after N2 (its
        # contact-rich phase: the gripper holding the block), the violent states derived from the last one
        blk = next((i for i, a in enumerate(world.scene.env_cfg) if "block" in a.name), None)
        for tag, n in zip(("recorded", "held"), ns):
            if n == "lift":   # until the block is 8 cm above where it lay at the state before (the gripper holds it in the air)
                loop.torch.cuda.synchronize(); world._stale = True
                z0 = float(world._root_state[0, blk, 2])
                n = done
                while n < 1500:
                    loop.iterate(); n += 1
                    if n % 5 == 0:
                        loop.torch.cuda.synchronize(); world._stale = True
                        if float(world._root_state[0, blk, 2]) > z0 + 0.08:
                            break
            else:
                for _ in range(n - done):
                    loop.iterate()
            done = n
            loop.torch.cuda.synchronize()
            world._stale = True
            dof, root = world._dof_state[0].cpu().numpy().copy(), world._root_state[0].cpu().numpy().copy()
            U = planner.mppi.U.numpy().copy()
            keep[f"{name}_{tag}_dof"], keep[f"{name}_{tag}_root"], keep[f"{name}_{tag}_U"] = dof, root, U
            print(f"{name}: `{tag}` = state after {n} closed-loop iterations: dof {np.round(dof, 4)}")
            for a, row in zip(world.scene.env_cfg, root):
                print(f"    {a.name:20s} pos {np.round(row[0:3], 4)} vel {np.round(row[7:10], 3)} w {np.round(row[10:13], 3)}")
        # violent states: a rollout from the last state with every step's env state kept (generic-mode trajectory kernels)
        lib, P = loop.lib, loop.P
        K, H = loop.K, loop.H
        capi.check(lib, lib.mppi_rollout(P))
        S = planner.mppi.get_costs().numpy()
        capi.check(lib, lib.mppi_sim_reset(P))
        b = planner.mppi._simulate_horizon()
        for key in ("dof", "root"):
            planner.mppi._lazy_materialise(key)
        loop.torch.cuda.synchronize()
        planner.sim._stale = True
        dofs, roots = b["dof"].view(H, K, -1).cpu().numpy(), b["root"].view(H, K, root.shape[0], 13).cpu().numpy()
        order = np.argsort(-np.where(np.isfinite(S), S, -np.inf))
        picked = 0
        for k in order:
            t = (9, 20)[picked % 2] if H > 20 else (5, H - 2)[picked % 2]
            if np.isfinite(dofs[t, k]).all() and np.isfinite(roots[t, k]).all() and np.abs(roots[t, k, :, 0:3]).max() < 10.0:
                keep[f"{name}_violent{t}_dof"], keep[f"{name}_violent{t}_root"] = dofs[t, k].copy(), roots[t, k].copy()
                print(f"    violent state: sample {k} (cost {S[k]:.1f}, median {np.median(S):.1f}) after {t} steps -> {name}_violent{t}")
                picked += 1
            if picked == 2:
                break
        world.stop_sim()
        planner.sim.stop_sim()
    np.savez(out, **keep)
    print("wrote", out, sorted(keep))


if __name__ == "__main__":
    main()

"""A/B timing of the contact-scene rollout kernel with parts of the scene removed (experiment, not a test)."""
import ctypes as C, sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "mppi-isaac_amd"))
import numpy as np, torch
from mppiisaac.backend import capi
from scenes import boxer_push, panda_pick

lib = capi.load_library(sys.argv[1] if len(sys.argv) > 1 else None)  # optional: a variant .so (tools/exp/ab_build.sh)

def run(name, make, K, H, edit=None):
    scene, m, cfg, cost, dof, root = make(K=K, H=H)
    if edit: edit(m)
    ctx = C.c_void_p()
    capi.check(lib, lib.mppi_create(C.byref(m), C.byref(cfg), 0, C.byref(ctx)))
    capi.check(lib, lib.mppi_set_cost(ctx, C.byref(cost)))
    d, r = np.ascontiguousarray(dof, np.float32), np.ascontiguousarray(root, np.float32)
    capi.check(lib, lib.mppi_set_state(ctx, capi.fptr(d), capi.fptr(r)))
    capi.check(lib, lib.mppi_sample(ctx, C.c_uint32(0)))
    for _ in range(3): capi.check(lib, lib.mppi_rollout(ctx))
    torch.cuda.synchronize()
    t = time.perf_counter()
    n = 20
    for _ in range(n): capi.check(lib, lib.mppi_rollout(ctx))
    capi.check(lib, lib.mppi_synchronize(ctx))
    print(f"{name:40s} {1e3 * (time.perf_counter() - t) / n:8.3f} ms", flush=True)
    lib.mppi_destroy(ctx)

def no_pairs(m): m.n_pairs = 0
def only(ks):
    def f(m):
        keep = [m.pairs[i] for i in ks]
        for i, p in enumerate(keep): m.pairs[i] = p
        m.n_pairs = len(keep)
    return f
if not os.environ.get("CLOSED_LOOP_ONLY"):
    for make, K, H in ((boxer_push, 8192, 25), (panda_pick, 8192, 30)):
        run(make.__name__ + " full", make, K, H)
        run(make.__name__ + " no pairs", make, K, H, no_pairs)
    run("panda_pick block pairs only (ground+table)", panda_pick, 8192, 30, only([21, 22]))


def closed_loop_variants(workload, steps=150):
    """reach the steady closed-loop state with the bench loop, then time the rollout kernel on variants of the scene"""
    import copy
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
    import bench
    import mppiisaac.objectives as objectives
    from mppiisaac.planner.isaacgym_wrapper import IsaacGymWrapper
    from mppiisaac.planner.mppi_isaac import MPPIisaacPlanner
    wl = bench.WORKLOADS[workload]
    cfg = bench.make_cfg(wl, wl["K"])
    cfg.mppi.device = "cuda:0"
    planner = MPPIisaacPlanner(cfg, getattr(objectives, wl["objective"])(cfg))
    world = IsaacGymWrapper(cfg.isaacgym, actors=cfg.actors, init_positions=cfg.initial_actor_positions, num_envs=1, device="cuda:0")
    P, W = planner.sim._ctx, world._ctx
    cost = planner.objective.fused_spec(planner.sim)
    capi.check(lib, lib.mppi_set_cost(P, C.byref(cost)))
    capi.check(lib, lib.mppi_set_state_from_world(P, W))
    for _ in range(steps):
        capi.check(lib, lib.mppi_rollout(P))
        capi.check(lib, lib.mppi_update_step_world(P, None, 1, W))
    n, A = planner.sim.scene.n_dof, len(planner.sim.env_cfg)
    dof, root = np.zeros(2 * n, np.float32), np.zeros((A, 13), np.float32)
    capi.check(lib, lib.mppi_get_state(P, capi.fptr(dof), capi.fptr(root)))
    U = np.zeros((wl["H"], planner.sim.scene.nu), np.float32)
    capi.check(lib, lib.mppi_get_nominal(P, capi.fptr(U)))
    # equal-state A/B of two builds: MPPI_STATE_SAVE=<prefix> records this state, MPPI_STATE_LOAD=<prefix> times at a recorded one
    if os.environ.get("MPPI_STATE_SAVE"):
        np.savez(os.environ["MPPI_STATE_SAVE"] + "_" + workload + ".npz", dof=dof, root=root, U=U)
    default_state = os.path.join(os.path.dirname(os.path.abspath(__file__)), "states", "state")   # recorded with the r02d build
    load = os.environ.get("MPPI_STATE_LOAD", default_state if not os.environ.get("MPPI_STATE_SAVE") else "")
    if load and os.path.exists(load + "_" + workload + ".npz"):
        z = np.load(load + "_" + workload + ".npz")
        dof, root, U = np.ascontiguousarray(z["dof"]), np.ascontiguousarray(z["root"]), np.ascontiguousarray(z["U"])
    print(workload, "closed-loop state after", steps, "steps: q =", np.round(dof[0::2], 2), flush=True)
    model0 = planner.sim._c_model
    sc = planner.sim.scene
    names = [f"{sc.env_cfg[sh['actor']].name}:{sh['link']}" for sh in sc.shapes]
    print("  pairs:", ", ".join(f"{i}={names[a]}/{names[b] if b >= 0 else 'ground'}" for i, (a, b) in enumerate(sc.pairs)), flush=True)

    def timed(name, edit):
        m = type(model0).from_buffer_copy(model0)
        if edit: edit(m)
        ctx = C.c_void_p()
        capi.check(lib, lib.mppi_create(C.byref(m), C.byref(planner.sim._mppi_config), 0, C.byref(ctx)))
        capi.check(lib, lib.mppi_set_cost(ctx, C.byref(cost)))
        capi.check(lib, lib.mppi_set_state(ctx, capi.fptr(dof), capi.fptr(root)))
        capi.check(lib, lib.mppi_set_nominal(ctx, capi.fptr(U)))
        capi.check(lib, lib.mppi_sample(ctx, C.c_uint32(0)))
        for _ in range(3): capi.check(lib, lib.mppi_rollout(ctx))
        capi.check(lib, lib.mppi_synchronize(ctx))
        t = time.perf_counter()
        for _ in range(20): capi.check(lib, lib.mppi_rollout(ctx))
        capi.check(lib, lib.mppi_synchronize(ctx))
        print(f"  {name:44s} {1e3 * (time.perf_counter() - t) / 20:8.3f} ms", flush=True)
        lib.mppi_destroy(ctx)
    def no_rnd(m): m.randomize_seed = -1
    def no_effort(m):
        for i in range(m.n_bodies): m.bodies[i].effort = 0.0
    timed("full (seeded noise)", None)
    timed("full, noise off", no_rnd)
    timed("full, no joint effort limits (never a 2nd solve)", no_effort)
    timed("no pairs", no_pairs)
    if workload == "boxer_push":
        ground = [i for i, (a, b) in enumerate(sc.pairs) if b < 0]
        timed("ground pairs only", only(ground))
        timed("box-box pairs only", only([i for i in range(len(sc.pairs)) if i not in ground]))
        for i in range(len(sc.pairs)):
            timed(f"without pair {i}", only([j for j in range(len(sc.pairs)) if j != i]))
    if workload == "panda_pick":
        timed("block pairs only", only([21, 22]))
        timed("block + finger/hand-block pairs", only([13, 15, 17, 19, 21, 22]))
        timed("no link-table pairs", only([0, 1, 3, 5, 7, 9, 11, 13, 15, 17, 19, 21, 22]))


closed_loop_variants("panda_pick")
closed_loop_variants("boxer_push")

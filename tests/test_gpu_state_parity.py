"""Full-K, PER-STEP state parity of the contact scenes (VERDICT r4 "what's weak" 2): where, along the horizon, does a sample of the
HIP rollout leave the fp64 oracle?  The trajectory-dumping rollout kernels (mppi_rollout_trajectory + mppi_materialise_trajectory:
the env state after EVERY horizon step, all K samples) against the oracle's batched env step (orc_envs_step, fp64, OpenMP over the
envs, the same per-sample actor randomisation) driven with the same commands u_t = clamp(U_t + eps_t).

What is asserted (and printed as a table per state): after the FIRST steps - before contact dynamics had time to amplify the last
bits - every sample agrees to 1e-4 (joint positions [rad | m] and actor positions [m]); further on the fraction inside each
tolerance band falls off as the chaotic samples (tumbling block, chassis on its side) part ways, and the test reports
the horizon step at which each band is left by more than 0.1 % of the samples."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from mppiisaac.backend import capi
from scenes import boxer_push, panda_pick
from test_gpu_parity import CLOSED_LOOP_STATES, Ctx

pytestmark = pytest.mark.gpu
BANDS = (1e-5, 1e-4, 1e-3, 1e-2)


def oracle_states(o, m, cfg, dof0, root0, U, eps, n_steps):
    """[n_steps][K] joint positions / actor positions of the oracle's envs after every step"""
    K, nu, n, A, B = cfg.num_samples, cfg.nu, m.n_bodies, m.n_actors, m.n_rb
    f = o.dtype
    dof = np.tile(np.asarray(dof0, f).reshape(1, -1), (K, 1))
    root = np.tile(np.asarray(root0, f).reshape(1, A, 13), (K, 1, 1))
    rb, cf = np.zeros((K, B, 13), f), np.zeros((K, B, 3), f)
    umin, umax = np.array([cfg.u_min[j] for j in range(nu)]), np.array([cfg.u_max[j] for j in range(nu)])
    qs, ps = [], []
    for t in range(n_steps):
        u = np.clip(np.asarray(U[t], np.float64)[None, :] + eps[t].T.astype(np.float64), umin, umax)
        if cfg.sample_null_action and cfg.k_offset + K == cfg.k_total:
            u[-1] = np.clip(np.zeros(nu), umin, umax)
        u = np.ascontiguousarray(u, f)
        o.lib.orc_envs_step(C.byref(m), C.c_int(K), C.c_int(cfg.k_offset), o.p(u), o.p(dof), o.p(root), o.p(rb), o.p(cf))
        qs.append(dof[:, 0::2].copy())
        ps.append(root[:, :, 0:3].copy())
    return np.stack(qs), np.stack(ps)


@pytest.mark.parametrize("make,name,K,H,nu,first_steps", [(boxer_push, "boxer_push", 8192, 25, 2, 2), (panda_pick, "panda_pick", 8192, 30, 9, 2)])
def test_per_step_states_of_all_samples_against_the_oracle(make, name, K, H, nu, first_steps, oracle64):
    Z = np.load(CLOSED_LOOP_STATES)
    scene, m, cfg, cost, dof0, root0 = make(K=K, H=H)
    n_cmp = 12
    for st in ("initial", "recorded") + (("held",) if f"{name}_held_dof" in Z.files else ()):
        dof, root = (dof0, root0) if st == "initial" else (Z[f"{name}_{st}_dof"], Z[f"{name}_{st}_root"])
        U = Z[f"{name}_{st}_U"] if st != "initial" and f"{name}_{st}_U" in Z.files else np.zeros((H, nu), np.float32)
        c = Ctx(m, cfg, cost)
        c.call("mppi_sample", C.c_uint32(0)); c.set_state(dof, root); c.set_U(U)
        eps = c.get("mppi_get_noise", (H, nu, K))
        c.call("mppi_rollout_trajectory")
        f32 = dict(dtype=torch.float32, device="cuda")
        t_dof, t_root = torch.zeros((H * K, 2 * m.n_bodies), **f32), torch.zeros((H * K, m.n_actors, 13), **f32)
        c.call("mppi_materialise_trajectory", C.c_void_p(t_dof.data_ptr()), C.c_void_p(t_root.data_ptr()), None, None)
        torch.cuda.synchronize()
        q_hip = t_dof.cpu().numpy().reshape(H, K, -1)[:n_cmp, :, 0::2]
        p_hip = t_root.cpu().numpy().reshape(H, K, m.n_actors, 13)[:n_cmp, :, :, 0:3]
        c.close()
        q_orc, p_orc = oracle_states(oracle64, m, cfg, dof, root, U, eps, n_cmp)
        err = np.maximum(np.abs(q_hip - q_orc).max(-1), np.abs(p_hip - p_orc).reshape(n_cmp, K, -1).max(-1))    # [n_cmp][K]
        assert np.isfinite(err).all()
        frac = np.array([[np.mean(err[t] <= b) for b in BANDS] for t in range(n_cmp)])
        left = {b: next((t for t in range(n_cmp) if frac[t, i] < 0.999), None) for i, b in enumerate(BANDS)}
        print(f"\n{name}, {st} state: per-step state error of all {K} samples vs the fp64 oracle (max over joint and actor positions)")
        print("   step | within 1e-5   1e-4     1e-3     1e-2   | median    max")
        for t in range(n_cmp):
            print(f"   {t + 1:4d} |  {frac[t, 0]:.4f}   {frac[t, 1]:.4f}   {frac[t, 2]:.4f}   {frac[t, 3]:.4f} | {np.median(err[t]):.1e}  {err[t].max():.1e}")
        print("   more than 0.1 % of the samples outside a band from step: " + ", ".join(f"{b:g}: {'never (12 steps)' if s is None else s + 1}" for b, s in left.items()))
        if st == "held":
            # the gripper holding the one-gram block (round 6): rollouts that let go of it are chaotic from the first step on (see
            # test_contact_rich_states_match_oracle) - asserted: 99 % within 1e-3 after the first step, the median sample within 1e-3
            # after twelve, 80 % within 1e-2 throughout
            assert frac[0, 2] >= 0.99 and np.median(err[-1]) <= 1e-3 and frac[:, 3].min() >= 0.8
            continue
        # before the chaotic regime: EVERY sample within 1e-4 over the first steps, the typical sample within 1e-5 for all twelve
        assert err[:first_steps].max() <= 1e-4, err[:first_steps].max()
        assert np.median(err[-1]) <= 1e-4
        # ... and no cliff: 99 % of the samples stay within 1e-2 for the twelve steps
        assert frac[:, 3].min() >= 0.99


@pytest.mark.parametrize("make,name,K,H,nu", [(boxer_push, "boxer_push", 8192, 25, 2), (panda_pick, "panda_pick", 2048, 30, 9)])
def test_group_cull_of_candidate_pairs_changes_no_cost(make, name, K, H, nu, monkeypatch):
    """pair groups (DevModel::Group, csrc/mppi_scene.hpp contact_forces): the robot's pairs against one shape of another actor are
    skipped together when that shape is out of the robot's reach in every sample of a wavefront (mobile bases).  Conservative like
    the broad phase: with the groups switched off (MPPI_GROUP_CULL=0) every pair is visited - and every cost is the same to the
    last bit, at the initial and at the recorded closed-loop state."""
    Z = np.load(CLOSED_LOOP_STATES)
    scene, m, cfg, cost, dof0, root0 = make(K=K, H=H)
    for st in ("initial", "recorded"):
        dof, root = (dof0, root0) if st == "initial" else (Z[f"{name}_recorded_dof"], Z[f"{name}_recorded_root"])
        U = Z[f"{name}_recorded_U"] if st == "recorded" and f"{name}_recorded_U" in Z.files else np.zeros((H, nu), np.float32)
        S = {}
        for sw in ("1", "0"):
            monkeypatch.setenv("MPPI_GROUP_CULL", sw)
            c = Ctx(m, cfg, cost)
            c.call("mppi_sample", C.c_uint32(0)); c.set_state(dof, root); c.set_U(U); c.call("mppi_rollout")
            S[sw] = c.get("mppi_get_costs", (K,))
            c.close()
        if make is boxer_push and st == "recorded":       # ... and with every sample simulating its own block size (the groups' reach covers the draws)
            scene.randomize_seed = 3
            mr = scene.to_c()
            scene.randomize_seed = -1
            for sw in ("1", "0"):
                monkeypatch.setenv("MPPI_GROUP_CULL", sw)
                c = Ctx(mr, cfg, cost)
                c.call("mppi_sample", C.c_uint32(0)); c.set_state(dof, root); c.set_U(U); c.call("mppi_rollout")
                S["r" + sw] = c.get("mppi_get_costs", (K,))
                c.close()
            assert np.array_equal(S["r1"], S["r0"]) and not np.array_equal(S["r1"], S["1"])
        rel = np.abs(S["1"] - S["0"]) / np.abs(S["0"])
        print(f"\n{name} {st}: group cull on vs off: bit-equal costs {np.mean(S['1'] == S['0']):.4f}, max rel diff {rel.max():.1e}")
        assert np.mean(S["1"] == S["0"]) == 1.0, "a skipped pair is one the broad phase would have culled: bit-identical costs"


def test_an_arm_among_ten_obstacle_spheres(oracle64):
    """reference `IsaacGymConfig.num_obstacles: int = 10` (isaacgym_wrapper.py:16) and the obstacle lists of compute_action
    (mppi_isaac.py:75-81, isaacgym_wrapper.py:695-742): the panda with its gripper among TEN fixed spheres = 12 actors, 100 candidate
    pairs (MPPI_MAX_ACTORS 12 / MPPI_MAX_PAIRS 128 since ABI 8; 8 / 48 before, when this scene was refused).  Three of the
    spheres stand where the arm's links reach them: rollouts of the contact-scene kernel against the fp64 oracle on every sample,
    and the contact forces on the obstacles are felt."""
    from mppiisaac.planner.isaacgym_wrapper import ActorWrapper, Scene
    from mppiisaac.planner.mppi import make_config
    from mppiisaac.utils.config_store import load_config
    from mppiisaac.utils.isaacgym_utils import load_actor_cfgs, load_asset
    K, H = 1024, 15
    env = load_actor_cfgs(["panda_gripper", "goal"])
    rng = np.random.default_rng(5)
    spots = [[0.45, 0.0, 0.55], [0.3, 0.25, 0.75], [0.35, -0.2, 0.35]] + [[float(v) for v in rng.uniform([0.9, -1.0, 0.1], [1.6, 1.0, 1.2])] for _ in range(7)]
    for i, p in enumerate(spots):
        env.append(ActorWrapper(type="sphere", name=f"sphere{i}", size=[0.08], fixed=True, init_pos=p))
    ex = load_config({"defaults": [{"mppi": "panda_pick"}, {"isaacgym": "normal"}]}, overrides={"mppi.num_samples": K, "mppi.horizon": H})
    scene = Scene(env, ex.isaacgym, load_asset(env[0]))
    m = scene.to_c()
    assert m.n_actors == 12 and m.n_pairs == 100 and m.n_shapes == 21, (m.n_actors, m.n_pairs, m.n_shapes)
    cfg = make_config(ex.mppi, viz_link=scene.viz_link_index())
    nu = cfg.nu
    cost = capi.Cost()
    cost.kind, cost.n_terms = capi.COST_PROGRAM, 4
    t = cost.terms[0]
    t.op, t.n, t.w = capi.OP_DIST, 3, 10.0
    t.src[0], t.idx[0] = capi.SRC_RB, scene.rigid_body_index("panda", "panda_ee")
    t.src[1], t.idx[1] = capi.SRC_ACTOR, scene.actor_index("goal")
    for j in range(3):
        t = cost.terms[1 + j]
        t.op, t.n, t.w = capi.OP_FORCE_L1, 3, 0.05
        t.src[0], t.idx[0] = capi.SRC_RB, scene.rigid_body_index(f"sphere{j}", "sphere")
    dof, root = scene.initial_state()
    root[scene.actor_index("goal"), 0:3] = [0.5, 0.1, 0.5]
    c = Ctx(m, cfg, cost)
    c.call("mppi_sample", C.c_uint32(0)); c.set_state(dof, root)
    U = np.zeros((H, nu), np.float32)
    U[:, 1], U[:, 3] = 0.6, 0.5                      # lean the arm forward, into the first spheres
    c.set_U(U); c.call("mppi_rollout")
    S, eps = c.get("mppi_get_costs", (K,)), c.get("mppi_get_noise", (H, nu, K))
    c.close()
    So, _, _ = oracle64.rollout(m, cfg, cost, dof, root, U, eps)
    rel = np.abs(S - So) / np.abs(So)
    m0 = scene.to_c()
    m0.n_pairs = 0
    S_free, _, _ = oracle64.rollout(m0, cfg, cost, dof, root, U, eps)
    touched = np.mean(np.abs(So - S_free) > 1e-3 * np.abs(S_free))
    print(f"\narm among ten spheres, {m.n_pairs} pairs: all {K} samples vs fp64 oracle within 1e-4 {np.mean(rel <= 1e-4):.4f} 1e-3 {np.mean(rel <= 1e-3):.4f} max {rel.max():.1e}; "
          f"{touched:.2f} of the rollouts touch an obstacle")
    assert np.isfinite(S).all() and touched > 0.2
    assert np.mean(rel <= 1e-3) >= 0.99 and np.mean(rel <= 1e-2) >= 0.998


def test_a_mobile_base_among_nine_boxes_takes_the_one_wavefront_kernel(oracle64):
    """The pushing scene's kernel with a helper wavefront keeps two words of dead-pair masks (its register budget is full:
    mppi_scene.hpp, pair groups) - a scene with more than 64 candidate pairs (boxer: chassis, two wheels, two casters and the
    block against NINE obstacle boxes = 12 actors, 65 pairs; reference isaacgym_wrapper.py:16 `num_obstacles = 10`) is given the one-wavefront octet kernel by
    mppi_create; costs against the fp64 oracle either way."""
    from mppiisaac.planner.isaacgym_wrapper import ActorWrapper, Scene
    from mppiisaac.planner.mppi import make_config
    from mppiisaac.utils.config_store import load_config
    from mppiisaac.utils.isaacgym_utils import load_actor_cfgs, load_asset
    K, H = 1024, 20
    for n_obst, kernel in ((9, "scene-oct"), (4, "scene-oct-pair")):
        env = load_actor_cfgs(["boxer", "block", "goal"])
        rng = np.random.default_rng(11)
        spots = [[0.85, 0.05, 0.15], [0.3, -0.75, 0.15], [0.1, 0.8, 0.15]] + [[float(v) for v in rng.uniform([-2.0, -2.0, 0.15], [2.5, 2.0, 0.15])] for _ in range(n_obst - 3)]
        for i, p in enumerate(spots):
            env.append(ActorWrapper(type="box", name=f"obst{i}", size=[0.3, 0.3, 0.3], fixed=True, init_pos=p))
        ex = load_config({"defaults": [{"mppi": "boxer_push"}, {"isaacgym": "normal"}]}, overrides={"mppi.num_samples": K, "mppi.horizon": H})
        scene = Scene(env, ex.isaacgym, load_asset(env[0]))
        m = scene.to_c()
        assert (m.n_pairs > 64) == (n_obst == 9), m.n_pairs
        cfg = make_config(ex.mppi, viz_link=scene.viz_link_index())
        cost = capi.Cost()
        cost.kind, cost.n_terms = capi.COST_PROGRAM, 3
        t = cost.terms[0]
        t.op, t.n, t.w = capi.OP_DIST, 2, 1.0
        t.src[0], t.idx[0] = capi.SRC_ACTOR, scene.actor_index("block")
        t.src[1], t.idx[1] = capi.SRC_ACTOR, scene.actor_index("goal")
        for j in range(2):
            t = cost.terms[1 + j]
            t.op, t.n, t.w = capi.OP_FORCE_L1, 3, 0.001
            t.src[0], t.idx[0] = capi.SRC_RB, scene.rigid_body_index(f"obst{j}", "box")
        dof, root = scene.initial_state()
        c = Ctx(m, cfg, cost)
        buf = C.create_string_buffer(256)
        c.call("mppi_kernel_info", buf, C.c_int(256))
        assert f"rollout={kernel} " in buf.value.decode(), buf.value
        c.call("mppi_sample", C.c_uint32(0)); c.set_state(dof, root)
        U = np.zeros((H, cfg.nu), np.float32)
        U[:, 0] = 1.2                                # straight at the first boxes
        c.set_U(U); c.call("mppi_rollout")
        S, eps = c.get("mppi_get_costs", (K,)), c.get("mppi_get_noise", (H, cfg.nu, K))
        c.close()
        So, _, _ = oracle64.rollout(m, cfg, cost, dof, root, U, eps)
        rel = np.abs(S - So) / np.abs(So)
        m0 = scene.to_c()
        m0.n_pairs = 0
        S_free, _, _ = oracle64.rollout(m0, cfg, cost, dof, root, U, eps)
        touched = np.mean(np.abs(So - S_free) > 1e-3 * np.abs(S_free))
        print(f"\nboxer among {n_obst} boxes, {m.n_pairs} pairs, {kernel}: vs fp64 oracle within 1e-4 {np.mean(rel <= 1e-4):.4f} 1e-3 {np.mean(rel <= 1e-3):.4f} 1e-2 {np.mean(rel <= 1e-2):.4f} max {rel.max():.1e}; "
              f"{touched:.2f} of the rollouts differ from the scene without contacts")
        assert np.isfinite(S).all() and touched > 0.2
        assert np.mean(rel <= 1e-3) >= 0.97 and np.mean(rel <= 1e-2) >= 0.99

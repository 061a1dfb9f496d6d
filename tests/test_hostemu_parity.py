"""The per-sample device functions (csrc/mppi_device.hpp: world-frame fp32 ABA, z-framed model, fused
costs) compiled for the host by tests/hostemu and compared with the oracle (body-frame fp64).  This is
the pre-GPU check of the kernel arithmetic; the -m gpu tests repeat it through the C-ABI on the device."""
import ctypes as C

import numpy as np
import pytest

from scenes import build_scene, panda_reach, point_reach

f32 = lambda a: np.ascontiguousarray(a, np.float32)
fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float)) if a is not None else None


def emu_rollout(lib, m, cfg, cost, dof, root, U, eps, want_viz=False):
    d0, r0, U32, e32 = f32(dof), f32(root), f32(U), f32(eps)
    S = np.zeros(cfg.num_samples, np.float32)
    du = np.zeros_like(e32)
    viz = np.zeros((cfg.horizon, cfg.num_samples, 3), np.float32) if want_viz else None
    rc = lib.emu_rollout(C.byref(m), C.byref(cfg), C.byref(cost), fp(d0), fp(r0), fp(U32), fp(e32), None, fp(S), fp(du), fp(viz))
    assert rc == 0
    return S, du, viz


@pytest.mark.parametrize("make", [panda_reach, point_reach])
def test_rollout_costs_match_oracle(make, hostemu, oracle64):
    scene, m, cfg, cost, dof, root = make(K=128)
    eps = oracle64.sample(cfg)
    rng = np.random.default_rng(0)
    U = 0.05 * rng.normal(size=(cfg.horizon, cfg.nu))
    S, du, viz = oracle64.rollout(m, cfg, cost, dof, root, U, eps, want_viz=True)
    Se, due, vize = emu_rollout(hostemu, m, cfg, cost, dof, root, U, eps, want_viz=cfg.want_rollouts)
    np.testing.assert_allclose(Se, S, rtol=2e-5)          # fp32 kernel arithmetic vs fp64 oracle
    np.testing.assert_allclose(due, du, atol=1e-6)
    if cfg.want_rollouts:
        np.testing.assert_allclose(vize, viz, atol=2e-5)  # visualize_link positions [H,K,3]


def test_step_and_rigid_body_state_match_oracle(hostemu, oracle64):
    for actors in (["panda_stick", "goal"], ["panda_gripper", "goal"], ["point_robot", "goal"]):
        scene = build_scene(actors)
        scene.robot.gravity = True  # exercise the gravity path as well
        m = scene.to_c()
        dof, root = scene.initial_state()
        rng = np.random.default_rng(5)
        n = scene.n_dof
        q, qd = dof[0::2].astype(np.float64) + 0.1 * rng.normal(size=n), 0.3 * rng.normal(size=n)
        u = rng.uniform(-0.2, 0.2, scene.nu)
        qe, qde = f32(q).copy(), f32(qd).copy()
        for _ in range(10):
            q, qd = oracle64.step(m, root, q, qd, oracle64.cmd_map(m, u))
            assert hostemu.emu_step(C.byref(m), fp(f32(root)), fp(qe), fp(qde), fp(f32(u))) == 0
        np.testing.assert_allclose(qe, q, atol=1e-4)      # joint position after 10 steps (SURVEY 8c tolerance)
        np.testing.assert_allclose(qde, qd, atol=2e-4)
        rb, _ = oracle64.rigid_body_state(m, root, q, qd)
        rbe = np.zeros((m.n_rb, 13), np.float32)
        cfe = np.ones((m.n_rb, 3), np.float32)
        assert hostemu.emu_rigid_body_state(C.byref(m), fp(f32(root)), fp(f32(q)), fp(f32(qd)), fp(rbe), fp(cfe)) == 0
        np.testing.assert_allclose(rbe, rb, atol=2e-5)
        assert not cfe.any()


def test_fused_panda_cost_matches_oracle(hostemu, oracle64):
    scene, m, cfg, cost, dof, root = panda_reach()
    rng = np.random.default_rng(6)
    for _ in range(16):
        q = rng.uniform(-2.5, 2.5, 7)
        rb, _ = oracle64.rigid_body_state(m, root, q, np.zeros(7))
        want = oracle64.cost(m, cost, root, q, np.zeros(7), rb)
        got = hostemu.emu_cost(C.byref(m), C.byref(cost), fp(f32(root)), fp(f32(q)), fp(f32(np.zeros(7))))
        assert got == pytest.approx(want, rel=2e-5, abs=2e-5)


def test_cross_sample_determinism(hostemu, oracle64):
    """the one property the reference's own test asserts (test_isaacgym_wrapper.py:35): identical inputs in
    every env give identical outputs."""
    scene, m, cfg, cost, dof, root = panda_reach(K=64, sample_null_action=False)
    eps = np.repeat(oracle64.sample(cfg)[:, :, :1], 64, axis=2)
    S, du, _ = emu_rollout(hostemu, m, cfg, cost, dof, root, np.zeros((cfg.horizon, 7)), eps)
    assert np.all(S == S[0]) and np.all(du == du[:, :, :1])


@pytest.mark.parametrize("make", [panda_reach, point_reach])
def test_quad_parallel_rollout_matches_oracle(make, hostemu, oracle64):
    """csrc/mppi_quad.hpp (one sample per 4-lane quad, DPP permutations emulated by array shuffles): same
    trajectory costs as the oracle, and every replicated scalar identical across the four lanes."""
    scene, m, cfg, cost, dof, root = make(K=96)
    eps = oracle64.sample(cfg)
    rng = np.random.default_rng(1)
    U = 0.05 * rng.normal(size=(cfg.horizon, cfg.nu))
    S, du, viz = oracle64.rollout(m, cfg, cost, dof, root, U, eps, want_viz=True)
    Se = np.zeros(cfg.num_samples, np.float32)
    due = np.zeros_like(f32(eps))
    vize = np.zeros((cfg.horizon, cfg.num_samples, 3), np.float32) if cfg.want_rollouts else None
    rc = hostemu.emu_rollout_quad(C.byref(m), C.byref(cfg), C.byref(cost), fp(f32(dof)), fp(f32(root)), fp(f32(U)), fp(f32(eps)), None,
                                  fp(Se), fp(due), fp(vize))
    assert rc == 0
    assert not np.isnan(Se).any()                       # NaN marks a quad whose lanes disagreed
    np.testing.assert_allclose(Se, S, rtol=2e-5)
    np.testing.assert_allclose(due, du, atol=1e-6)
    if vize is not None:
        np.testing.assert_allclose(vize, viz, atol=2e-5)


def test_position_mode_matches_oracle(hostemu, oracle64):
    """dof_mode "position" (teleport to the command, then the stiffness drive; reference isaacgym_wrapper.py:501-504,571-572):
    one-lane step, one-lane and quad rollouts of the arm, and the gripper scene's contact step, against the oracle"""
    scene = build_scene(["panda_stick", "goal"], robot_overrides={"dof_mode": "position"})
    scene.robot.gravity = True
    m = scene.to_c()
    assert m.drive_mode == 2 and m.drive_kp == 80.0 and m.drive_kd == 0.0
    dof, root = scene.initial_state()
    rng = np.random.default_rng(11)
    q, qd = dof[0::2].astype(np.float64), 0.3 * rng.normal(size=7)
    qe, qde = f32(q).copy(), f32(qd).copy()
    for _ in range(6):
        u = dof[0::2] + rng.uniform(-0.3, 0.3, 7)                                  # commanded joint positions
        q, qd = oracle64.step(m, root, q, qd, oracle64.cmd_map(m, u))
        assert hostemu.emu_step(C.byref(m), fp(f32(root)), fp(qe), fp(qde), fp(f32(u))) == 0
        assert np.abs(q - u).max() < 0.1 and np.abs(qd).max() > 1e-3               # teleported, then sagging under gravity
        np.testing.assert_allclose(qe, q, atol=2e-5)
        np.testing.assert_allclose(qde, qd, atol=2e-3)
    # whole rollouts: nominal U = the start pose, noise around it (panda_reach's cost and sampler)
    _, _, cfg, cost, _, root = panda_reach(K=64, H=8)
    eps = oracle64.sample(cfg)
    U = np.tile(dof[0::2], (cfg.horizon, 1))
    for j in range(7):
        cfg.u_min[j], cfg.u_max[j] = -3.0, 3.0
    S, du, _ = oracle64.rollout(m, cfg, cost, dof, root, U, eps)
    Se, due, _ = emu_rollout(hostemu, m, cfg, cost, dof, root, U, eps)
    np.testing.assert_allclose(Se, S, rtol=5e-5)
    Sq, duq = np.zeros(cfg.num_samples, np.float32), np.zeros_like(f32(eps))
    assert hostemu.emu_rollout_quad(C.byref(m), C.byref(cfg), C.byref(cost), fp(f32(dof)), fp(f32(root)), fp(f32(U)), fp(f32(eps)), None,
                                    fp(Sq), fp(duq), None) == 0
    assert not np.isnan(Sq).any()
    np.testing.assert_allclose(Sq, S, rtol=5e-5)
    np.testing.assert_allclose(duq, du, atol=1e-6)

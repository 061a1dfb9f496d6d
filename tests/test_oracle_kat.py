"""Known-answer tests that pin the CPU oracle (oracle/mppi_oracle.c).  The reference has no golden
values for dynamics / MPPI arithmetic, so the oracle is checked against physics
and against independent formulations: a numpy RNEA, closed-form pendulum/drive results, scipy."""
import ctypes as C
import json
import math
import os

import numpy as np
import pytest

from mppiisaac.backend import capi
from scenes import build_scene, panda_reach, point_reach


# ------------------------------------------------------------------ independent numpy RNEA (body frame)
def rot_axis(a, q):
    a = np.asarray(a, float)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + math.sin(q) * K + (1 - math.cos(q)) * K @ K


def rnea(model_json, q, qd, qdd, gravity):
    """tau = RNEA(q, qd, qdd): classic Newton-Euler with 3-vectors (Craig / Luh-Walker-Paul),
    independent of the oracle's spatial-vector ABA."""
    bodies = model_json["bodies"]
    n = len(bodies)
    R, p, w, wd, a, F, N = [None] * n, [None] * n, [None] * n, [None] * n, [None] * n, [None] * n, [None] * n
    for i, b in enumerate(bodies):
        ax = np.asarray(b["axis"])
        Rt, pt = np.asarray(b["R_tree"]), np.asarray(b["p_tree"])
        par = b["parent"]
        if b["jtype"] == "revolute":
            R[i], p[i] = Rt @ rot_axis(ax, q[i]), pt
        else:
            R[i], p[i] = Rt, pt + Rt @ ax * q[i]
        wp, wdp, ap = (np.zeros(3), np.zeros(3), -np.asarray(gravity, float)) if par < 0 else (w[par], wd[par], a[par])
        E = R[i].T
        if b["jtype"] == "revolute":
            w[i] = E @ wp + ax * qd[i]
            wd[i] = E @ wdp + ax * qdd[i] + np.cross(E @ wp, ax * qd[i])
            a[i] = E @ (ap + np.cross(wdp, p[i]) + np.cross(wp, np.cross(wp, p[i])))
        else:
            w[i] = E @ wp
            wd[i] = E @ wdp
            a[i] = E @ (ap + np.cross(wdp, p[i]) + np.cross(wp, np.cross(wp, p[i]))) + 2 * np.cross(w[i], ax * qd[i]) + ax * qdd[i]
        I = b["inertia"]
        m = I["mass"]
        c = np.asarray(I["h"]) / m if m > 0 else np.zeros(3)
        Io = np.array([[I["Io"][0], I["Io"][1], I["Io"][2]], [I["Io"][1], I["Io"][3], I["Io"][4]], [I["Io"][2], I["Io"][4], I["Io"][5]]])
        Ic = Io - m * (c @ c * np.eye(3) - np.outer(c, c))
        ac = a[i] + np.cross(wd[i], c) + np.cross(w[i], np.cross(w[i], c))
        F[i] = m * ac
        N[i] = Ic @ wd[i] + np.cross(w[i], Ic @ w[i]) + np.cross(c, F[i])  # moment about the body origin
    f, nn = [F[i].copy() for i in range(n)], [N[i].copy() for i in range(n)]
    tau = np.zeros(n)
    for i in reversed(range(n)):
        b = bodies[i]
        ax = np.asarray(b["axis"])
        tau[i] = (nn[i] if b["jtype"] == "revolute" else f[i]) @ ax
        par = b["parent"]
        if par >= 0:
            fp = R[i] @ f[i]
            f[par] += fp
            nn[par] += R[i] @ nn[i] + np.cross(p[i], fp)
    return tau


@pytest.mark.parametrize("actors,gravity_on", [(["panda_stick", "goal"], False), (["panda_stick", "goal"], True),
                                               (["panda_gripper", "goal"], True), (["point_robot", "goal"], True)])
def test_aba_inverts_rnea(actors, gravity_on, oracle64):
    scene = build_scene(actors)
    scene.robot.gravity = gravity_on
    m = scene.to_c()
    _, root = scene.initial_state()
    rng = np.random.default_rng(0)
    n = scene.n_dof
    for _ in range(5):
        q, qd, tau = rng.uniform(-1.5, 1.5, n), rng.uniform(-1, 1, n), rng.uniform(-5, 5, n)
        qdd = oracle64.forward_dynamics(m, root, q, qd, tau)
        g = (0, 0, -9.8) if gravity_on else (0, 0, 0)
        np.testing.assert_allclose(rnea(scene.robot_model, q, qd, qdd, g), tau, rtol=1e-8, atol=1e-8)


def pendulum_model(n_links=1, L=1.0, mass=1.0, dt=0.001, substeps=1, kd=0.0, mode=capi.DRIVE_EFFORT):
    """chain of point masses on massless rods, revolute about y, hanging along -z, gravity on."""
    m = capi.Model()
    m.abi_version = capi.ABI_VERSION
    m.n_actors, m.robot_actor = 1, 0
    a = m.actors[0]
    a.type, a.fixed, a.collision, a.gravity, a.first_rb, a.n_rb = capi.ACTOR_ROBOT, 1, 0, 1, 0, n_links
    m.n_bodies = m.n_links = m.n_rb = n_links
    eye = [1, 0, 0, 0, 1, 0, 0, 0, 1]
    for i in range(n_links):
        b = m.bodies[i]
        b.parent, b.jtype = i - 1, capi.JOINT_REVOLUTE
        for j, v in enumerate((0, 1, 0)):
            b.axis[j] = v
        for j in range(9):
            b.R_tree[j] = eye[j]
        b.p_tree[2] = 0.0 if i == 0 else -L
        b.mass = mass
        b.h[2] = -mass * L                      # point mass at (0,0,-L)
        b.Io[0] = b.Io[3] = mass * L * L        # Ixx = Iyy = m L^2 about the joint
        m.cmd_col[i][0], m.cmd_coef[i][0] = i, 1.0
        l = m.links[i]
        l.body = i
        for j in range(9):
            l.R[j] = eye[j]
    m.drive_mode, m.substeps, m.drive_kd, m.dt, m.nu = mode, substeps, kd, dt, n_links
    m.gravity[2] = -9.8
    return m


ROOT1 = np.array([[0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0]], float)


def test_pendulum_acceleration_and_period(oracle64):
    L = 0.7
    m = pendulum_model(L=L)
    for th in (0.1, 0.8, -2.0):
        qdd = oracle64.forward_dynamics(m, ROOT1, [th], [0.0], [0.0])
        assert qdd[0] == pytest.approx(-9.8 / L * math.sin(th), rel=1e-12)
    # small-angle period 2 pi sqrt(L/g): count time to return to the start through one full swing
    q, qd = np.array([0.05]), np.array([0.0])
    t, crossings, prev = 0.0, [], q[0]
    for step in range(4000):
        q, qd = oracle64.step(m, ROOT1, q, qd, [0.0])
        t += m.dt
        if prev < 0 <= q[0]:
            crossings.append(t)
        prev = q[0]
    period = crossings[1] - crossings[0]
    assert period == pytest.approx(2 * math.pi * math.sqrt(L / 9.8), rel=2e-3)


def test_position_mode_teleports_then_holds_with_the_stiffness_drive(oracle64):
    """dof_mode "position" (reference isaacgym_wrapper.py:501-504 stiffness 80 / damping 0, :571-572 apply_robot_cmd overwrites the
    DOF state with the command): q <- target, qd <- 0 at the start of the step whatever the state was, then the step runs with the
    drive tau = kp (target - q) treated implicitly.  Closed form for the one-link pendulum, one substep of h:
    qdd = -m g L sin(u) / (m L^2 + h^2 kp),  qd+ = h qdd,  q+ = u + h qd+."""
    L, mass, h, kp = 0.7, 1.3, 0.01, 80.0
    m = pendulum_model(L=L, mass=mass, dt=h, substeps=1, kd=0.0, mode=capi.DRIVE_POSITION)
    m.drive_kp = kp
    for u in (0.0, 0.4, -1.1):
        for q0, qd0 in (([2.0], [5.0]), ([-0.3], [0.0])):                         # the previous state does not matter
            q, qd = oracle64.step(m, ROOT1, q0, qd0, [u])
            qdd = -mass * 9.8 * L * math.sin(u) / (mass * L * L + h * h * kp)
            assert qd[0] == pytest.approx(h * qdd, rel=1e-12, abs=1e-15)
            assert q[0] == pytest.approx(u + h * h * qdd, rel=1e-12, abs=1e-15)
    # without gravity the commanded pose is held exactly, and two substeps equal two half steps of the closed form
    m.gravity[2] = 0.0
    q, qd = oracle64.step(m, ROOT1, [1.0], [3.0], [0.25])
    assert q[0] == 0.25 and qd[0] == 0.0
    m.gravity[2], m.substeps = -9.8, 2
    q, qd = oracle64.step(m, ROOT1, [0.0], [0.0], [0.4])
    hh, I = h / 2, mass * L * L
    kde = hh * kp
    x, v = 0.4, 0.0
    for _ in range(2):
        a = (-mass * 9.8 * L * math.sin(x) + kp * (0.4 - x) - kde * v) / (I + kde * hh)
        v += hh * a
        x += hh * v
    assert q[0] == pytest.approx(x, rel=1e-12) and qd[0] == pytest.approx(v, rel=1e-12)
    # the state is overwritten EVERY step: 200 steps with the same command end where one step ends; over one long step
    # (50 substeps of 40 ms) the spring works against gravity, and a stiffer drive sags less
    m.drive_kp = 80.0
    (q1,), (qd1,) = oracle64.step(m, ROOT1, [0.0], [0.0], [0.9])
    q, qd = 0.3, -2.0
    for _ in range(200):
        (q,), (qd,) = oracle64.step(m, ROOT1, [q], [qd], [0.9])
    assert (q, qd) == (q1, qd1) and 0.0 < 0.9 - q1 < 1e-2
    m.substeps, m.dt = 50, 2.0
    sag = []
    for k in (80.0, 8000.0):
        m.drive_kp = k
        (q,), _ = oracle64.step(m, ROOT1, [0.0], [0.0], [0.9])
        sag.append(0.9 - q)
    assert 0.0 < sag[1] < 0.1 * sag[0] < 0.09


def test_double_pendulum_energy_drift(oracle64):
    L, mass = 0.5, 1.3
    m = pendulum_model(n_links=2, L=L, mass=mass, dt=0.0005)

    def energy(q, qd):
        z1 = -L * math.cos(q[0]); x1 = -L * math.sin(q[0])
        z2 = z1 - L * math.cos(q[0] + q[1]); x2 = x1 - L * math.sin(q[0] + q[1])
        vx1 = -L * math.cos(q[0]) * qd[0]; vz1 = L * math.sin(q[0]) * qd[0]
        vx2 = vx1 - L * math.cos(q[0] + q[1]) * (qd[0] + qd[1]); vz2 = vz1 + L * math.sin(q[0] + q[1]) * (qd[0] + qd[1])
        return 0.5 * mass * (vx1 ** 2 + vz1 ** 2 + vx2 ** 2 + vz2 ** 2) + mass * 9.8 * (z1 + z2)
    q, qd = np.array([1.0, -0.5]), np.array([0.0, 0.0])
    e0 = energy(q, qd)
    for _ in range(2000):  # 1 s
        q, qd = oracle64.step(m, ROOT1, q, qd, [0.0, 0.0])
    assert abs(energy(q, qd) - e0) < 2e-2 * abs(e0)  # symplectic Euler: bounded O(h) drift


def test_velocity_drive_is_first_order_lag_and_saturates(oracle64):
    """SURVEY.md D (point_robot): qd+ = (M qd + h kd v*)/(M + h kd) while |drive force| <= 87 N."""
    scene = build_scene(["point_robot", "goal"])
    m = scene.to_c()
    dof, root = scene.initial_state()
    M, kd, h = 23.0, 600.0, 0.025
    # small target: unsaturated.  force = kd M (v*-qd)/(M+h kd) = 363 * 0.1 = 36 N < 87
    vstar = 0.1
    q, qd = oracle64.step(m, root, np.zeros(3), np.zeros(3), [vstar, 0, 0])
    v1 = h * kd * vstar / (M + h * kd)
    v2 = (M * v1 + h * kd * vstar) / (M + h * kd)
    # (the off-axis feature_link couples x with the drive-held yaw joint at the 1e-6 level)
    assert qd[0] == pytest.approx(v2, rel=1e-5)
    assert q[0] == pytest.approx(h * v1 + h * v2, rel=1e-5)
    # large target: saturated at the URDF effort limit 87 N -> a = 87/23 in both substeps
    q, qd = oracle64.step(m, root, np.zeros(3), np.zeros(3), [1.5, 0, 0])
    assert qd[0] == pytest.approx(2 * h * 87.0 / M, rel=1e-4)


def test_joint_limits_are_inelastic_clamps(oracle64):
    scene = build_scene(["panda_stick", "goal"])
    m = scene.to_c()
    dof, root = scene.initial_state()
    q = dof[0::2].astype(float); qd = np.zeros(7)
    q[3] = -0.0698 - 1e-4   # joint4 upper limit is -0.0698
    q, qd = oracle64.step(m, root, q, qd, [0, 0, 0, 0.2, 0, 0, 0])
    assert q[3] == pytest.approx(-0.0698) and qd[3] <= 0.0


def test_rigid_body_state_against_numpy_fk(oracle64):
    scene = build_scene(["panda_stick", "goal"])
    m = scene.to_c()
    dof, root = scene.initial_state()
    rng = np.random.default_rng(1)
    q, qd = rng.uniform(-1, 1, 7), rng.uniform(-1, 1, 7)
    rb, cf = oracle64.rigid_body_state(m, root, q, qd)
    assert rb.shape == (11, 13) and not cf.any()
    # numpy FK of the tip from the compiled model
    R, p = np.eye(3), np.zeros(3)
    for i, b in enumerate(scene.robot_model["bodies"]):
        p = p + R @ np.asarray(b["p_tree"])
        R = R @ np.asarray(b["R_tree"]) @ rot_axis(b["axis"], q[i])
    tip = scene.robot_model["links"][scene.link_names.index("panda_ee_tip")]
    ptip = p + R @ np.asarray(tip["p"])
    row = rb[scene.rigid_body_index("panda", "panda_ee_tip")]
    np.testing.assert_allclose(row[0:3], ptip, atol=1e-12)
    assert np.linalg.norm(row[3:7]) == pytest.approx(1.0) and row[6] >= 0  # unit xyzw, canonical w >= 0
    # linear velocity of the tip = d/dt FK (finite difference)
    e = 1e-7
    rb2, _ = oracle64.rigid_body_state(m, root, q + e * qd, qd)
    np.testing.assert_allclose(row[7:10], (rb2[scene.rigid_body_index("panda", "panda_ee_tip"), 0:3] - row[0:3]) / e, atol=1e-5)
    # goal actor row = its root state
    np.testing.assert_allclose(rb[scene.rigid_body_index("goal", "sphere")], root[1])


def test_zyx_euler_restatement_against_scipy():
    """pytorch3d's quaternion_to_matrix / matrix_to_euler_angles('ZYX') as restated in
    mppiisaac.utils.conversions: proper wxyz input must reproduce scipy's intrinsic ZYX angles."""
    import torch
    from scipy.spatial.transform import Rotation
    from mppiisaac.utils.conversions import matrix_to_euler_angles, quaternion_to_matrix
    rot = Rotation.random(32, random_state=3)
    xyzw = rot.as_quat()
    wxyz = torch.tensor(np.concatenate([xyzw[:, 3:4], xyzw[:, 0:3]], 1))
    Mx = quaternion_to_matrix(wxyz)
    np.testing.assert_allclose(Mx.numpy(), rot.as_matrix(), atol=1e-12)
    np.testing.assert_allclose(matrix_to_euler_angles(Mx, "ZYX").numpy(), rot.as_euler("ZYX"), atol=1e-9)


def test_panda_reach_cost_matches_reference_expression(oracle64):
    """oracle cost == the reference Objective's torch expression (examples/panda/planner.py:22-40) fed
    with the same rigid-body rows (incl. the xyzw-into-(r,i,j,k) quirk)."""
    import torch
    from mppiisaac.utils.conversions import matrix_to_euler_angles, quaternion_to_matrix
    scene, m, cfg, cost, dof, root = panda_reach()
    rng = np.random.default_rng(2)
    for _ in range(8):
        q = rng.uniform(-2, 2, 7)
        rb, _ = oracle64.rigid_body_state(m, root, q, np.zeros(7))
        ee = torch.tensor(rb[cost.link[0]])[None]
        goal = torch.tensor(root[cost.actor[0], 0:3])[None]
        d = torch.linalg.norm(ee[:, 0:3] - goal, axis=1)
        rpy = matrix_to_euler_angles(quaternion_to_matrix(ee[:, 3:7]), "ZYX")[:, 0:2]
        want = 1.0 * d + 0.5 * torch.linalg.norm(rpy, axis=1)
        assert oracle64.cost(m, cost, root, q, np.zeros(7), rb) == pytest.approx(float(want), rel=1e-10)


def test_halton_and_norminv(oracle64):
    from scipy.stats import norm
    lib = oracle64.lib
    # dimension 0: base 2, multiplier 1 -> plain van der Corput
    assert [lib.orc_halton(n, 0) for n in (1, 2, 3, 4, 5)] == [0.5, 0.25, 0.75, 0.125, 0.625]
    for dim in (1, 7, 34, 140):
        u = np.array([lib.orc_halton(n, dim) for n in range(1, 4097)])
        assert u.min() > 0 and u.max() < 1
        hist, _ = np.histogram(u, bins=16, range=(0, 1))
        assert np.abs(hist - 256).max() <= 24, dim      # low discrepancy: near-uniform bins
    for p in (1e-6, 0.01, 0.3, 0.5, 0.9, 1 - 1e-6):  # halton points stay >= 1/p^digits away from 0 and 1
        assert lib.orc_norminv(p) == pytest.approx(norm.ppf(p), rel=1e-9, abs=1e-12)
    # neighbouring high dimensions must not be correlated (why the digits are scrambled)
    a = np.array([lib.orc_halton(n, 33) for n in range(1, 513)])
    b = np.array([lib.orc_halton(n, 34) for n in range(1, 513)])
    assert abs(np.corrcoef(a, b)[0, 1]) < 0.15


def test_sampler_statistics(oracle64):
    scene, m, cfg, cost, dof, root = panda_reach(K=4096, H=20)
    eps = oracle64.sample(cfg)
    assert eps.shape == (20, 7, 4096)
    assert abs(eps.mean()) < 5e-3
    # first and last horizon step interpolate a single knot -> full sigma = sqrt(0.1)
    assert eps[0].std() == pytest.approx(math.sqrt(0.1), rel=0.02)
    assert eps[-1].std() == pytest.approx(math.sqrt(0.1), rel=0.02)
    # shards index the same global sequence
    from mppiisaac.planner.mppi import make_config
    from mppiisaac.utils.config_store import load_config
    ex = load_config({"defaults": [{"mppi": "panda"}]})
    ex.mppi.num_samples, ex.mppi.horizon = 4096, 20
    shard = make_config(ex.mppi, k_offset=1024, k_local=1024)
    np.testing.assert_array_equal(oracle64.sample(shard), eps[:, :, 1024:2048])


def test_update_math_and_nan_rejection(oracle64):
    scene, m, cfg, cost, dof, root = point_reach(K=32, H=10)
    rng = np.random.default_rng(4)
    S = rng.uniform(1, 3, 32)
    S[5] = np.nan; S[9] = np.inf
    du = rng.normal(size=(10, 3, 32))
    U0 = rng.normal(size=(10, 3))
    rec = oracle64.record(cfg, S, du)
    ok = np.isfinite(S)
    beta = S[ok].min()
    w = np.where(ok, np.exp(-(np.where(ok, S, 0) - beta) / cfg.lambda_), 0.0)
    assert rec[0] == pytest.approx(beta) and rec[1] == pytest.approx(w.sum())
    U, action, be = oracle64.update(cfg, rec, U0)
    Unew = U0 + np.einsum("k,tck->tc", w / w.sum(), du)
    np.testing.assert_allclose(action, Unew[0], rtol=1e-12)
    np.testing.assert_allclose(U[:-1], Unew[1:], rtol=1e-12)       # shifted left
    np.testing.assert_allclose(U[-1], cfg.u_init)                  # u_init appended


def test_shard_records_combine_to_the_unsharded_update(oracle64):
    scene, m, cfg, cost, dof, root = panda_reach(K=256, H=12)
    eps = oracle64.sample(cfg)
    U0 = np.zeros((12, 7))
    S, du, _ = oracle64.rollout(m, cfg, cost, dof, root, U0, eps)
    U_ref, a_ref, be_ref = oracle64.update(cfg, oracle64.record(cfg, S, du), U0)
    from mppiisaac.planner.mppi import make_config
    from mppiisaac.utils.config_store import load_config
    ex = load_config({"defaults": [{"mppi": "panda"}]})
    ex.mppi.num_samples, ex.mppi.horizon = 256, 12
    recs = []
    for r in range(4):
        sh = make_config(ex.mppi, k_offset=64 * r, k_local=64, viz_link=scene.viz_link_index())
        Ss, dus, _ = oracle64.rollout(m, sh, cost, dof, root, U0, oracle64.sample(sh))
        np.testing.assert_array_equal(Ss, S[64 * r:64 * r + 64])   # null-action sample lands on the last shard
        recs.append(oracle64.record(sh, Ss, dus))
    U_sh, a_sh, be_sh = oracle64.update(cfg, np.stack(recs), U0)
    np.testing.assert_allclose(a_sh, a_ref, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(U_sh, U_ref, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(be_sh, be_ref, rtol=1e-12)


def test_null_action_sample_and_clamp(oracle64):
    scene, m, cfg, cost, dof, root = panda_reach(K=64, H=12)
    eps = oracle64.sample(cfg) * 3.0           # force clamping
    U0 = np.full((12, 7), 0.05)
    S, du, _ = oracle64.rollout(m, cfg, cost, dof, root, U0, eps)
    u = U0[:, :, None] + du
    assert u.max() <= 0.2 + 1e-12 and u.min() >= -0.2 - 1e-12
    np.testing.assert_allclose(u[:, :, -1], 0.0, atol=1e-15)   # sample K-1: u == 0 over the whole horizon


# ------------------------------------------------------------------ external pins of the sampler pieces
def test_radical_inverse_matches_scipy_halton(oracle64):
    """the oracle's radical inverse with the digit scramble switched off (multiplier 1) IS the Halton sequence:
    bit-for-bit against scipy.stats.qmc.Halton(scramble=False) in the first 16 bases, indices 1..2048"""
    from scipy.stats import qmc
    lib = oracle64.lib
    lib.orc_radical_inverse.restype = C.c_double
    lib.orc_radical_inverse.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32]
    d, n = 16, 2048
    want = qmc.Halton(d=d, scramble=False).random(n + 1)[1:]      # scipy starts at index 0 (the origin)
    got = np.array([[lib.orc_radical_inverse(i, lib.orc_prime(j), 1) for j in range(d)] for i in range(1, n + 1)])
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-16)
    # the sampler's sequence is the same digits through a fixed per-base permutation: a bijection of each digit
    for dim in (3, 40, 139):
        p = lib.orc_prime(dim)
        mult = int(0.6180339887498949 * p + 0.5)
        assert sorted((dg * mult) % p for dg in range(p)) == list(range(p))
        assert lib.orc_halton(p + 2, dim) == pytest.approx(((2 * mult) % p) / p + (mult % p) / p ** 2, abs=1e-15)


def test_norminv_matches_scipy_ndtri(oracle64):
    from scipy.special import ndtri
    p = np.concatenate([np.logspace(-12, -1, 45), np.linspace(0.1, 0.9, 33), 1 - np.logspace(-12, -1, 45)])
    got = np.array([oracle64.lib.orc_norminv(float(x)) for x in p])
    np.testing.assert_allclose(got, ndtri(p), rtol=2e-10, atol=1e-12)


def test_bspline_basis_matches_scipy_design_matrix():
    """planner/mppi.py:bspline_basis (the matrix both the kernel and the oracle multiply the knots with) against
    scipy.interpolate.BSpline.design_matrix on the same clamped uniform knot vector"""
    from scipy.interpolate import BSpline
    from mppiisaac.planner.mppi import bspline_basis, knots_for_horizon
    for H in (12, 16, 20, 25, 30, 48, 64):
        nk, deg = knots_for_horizon(H), 2
        t = np.concatenate([np.zeros(deg), np.linspace(0.0, 1.0, nk - deg + 1), np.ones(deg)])
        x = np.linspace(0.0, 1.0, H)
        want = BSpline.design_matrix(x, t, deg).toarray()
        np.testing.assert_allclose(bspline_basis(H, nk), want, atol=1e-13)


PHILOX_KAT = [  # Random123 known-answer vectors for philox4x32-10 (counter, key) -> output
    ([0, 0, 0, 0], [0, 0], [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]),
    ([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2, [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]),
    ([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0], [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]),
]


def test_philox_known_answers_and_normal_sampler(oracle64):
    for ctr, key, want in PHILOX_KAT:
        assert oracle64.philox(ctr, key) == want
    from mppiisaac.planner.mppi import make_config
    from mppiisaac.utils.config_store import load_config
    ex = load_config({"defaults": [{"mppi": "panda"}]})
    ex.mppi.num_samples, ex.mppi.horizon, ex.mppi.mppi_mode, ex.mppi.seed_val = 4096, 20, "simple", 3
    ex.mppi.noise_mu = [0.01 * j for j in range(7)]
    cfg = make_config(ex.mppi)
    assert cfg.sampling == capi.SAMPLE_NORMAL and cfg.n_knots == 20
    eps = oracle64.sample_normal(cfg, 5)
    # knot (g, c, i) is word pair (i % 4) // 2 of philox(counter = (g, c, i // 4, iteration), key = (seed, "MPPI")) through Box-Muller
    g, c, i = 77, 4, 13
    x = oracle64.philox([g, c, i // 4, 5], [3, 0x4D505049])
    ua, ub = (x[0] + 0.5) / 2 ** 32, (x[1] + 0.5) / 2 ** 32
    z = math.sqrt(-2 * math.log(ua)) * math.sin(2 * math.pi * ub)   # i % 4 == 1: second value of the first pair
    assert eps[i, c, g] == pytest.approx(0.01 * c + math.sqrt(0.1) * z, abs=1e-12)
    # N(mu, sigma) per control dimension, independent over the horizon, and a different set every iteration
    z = (eps - np.array(ex.mppi.noise_mu)[None, :, None]) / math.sqrt(0.1)
    assert abs(z.mean()) < 5e-3 and z.std() == pytest.approx(1.0, rel=5e-3)
    assert abs(np.corrcoef(z[3, 2], z[4, 2])[0, 1]) < 0.05
    from scipy.stats import kstest
    assert kstest(z[:, 0, :].ravel(), "norm").pvalue > 1e-3
    assert np.abs(oracle64.sample_normal(cfg, 6) - eps).max() > 1.0
    np.testing.assert_array_equal(oracle64.sample_normal(cfg, 5), eps)
    # shards draw by GLOBAL sample id
    shard = make_config(ex.mppi, k_offset=1024, k_local=512)
    np.testing.assert_array_equal(oracle64.sample_normal(shard, 5), eps[:, :, 1024:1536])
    # halton-spline + random: Gaussian knots through the same B-spline as the Halton knots
    ex.mppi.mppi_mode, ex.mppi.sampling_method = "halton-spline", "random"
    cs = make_config(ex.mppi)
    assert cs.sampling == capi.SAMPLE_NORMAL and cs.n_knots == 5
    es = oracle64.sample_normal(cs, 0)
    assert ((es[0] - np.array(ex.mppi.noise_mu)[:, None]) / math.sqrt(0.1)).std() == pytest.approx(1.0, rel=0.03)  # clamped ends = one knot

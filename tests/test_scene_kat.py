"""Contact scenes (floating base, free bodies, penalty contact): physics known-answer tests of the oracle,
and the device arithmetic (host build, tests/hostemu) against the oracle.  There is nothing in the
reference to pin these against (PhysX is absent); the numbers below are physics, not fixtures."""
import ctypes as C

import numpy as np
import pytest

from scenes import boxer_push, build_scene, panda_pick

f32 = lambda a: np.ascontiguousarray(a, np.float32)
fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))


def yaw_of(q):
    return np.arctan2(2 * (q[3] * q[2] + q[0] * q[1]), q[3] ** 2 + q[0] ** 2 - q[1] ** 2 - q[2] ** 2)


def settle(o, m, root, q, qd, steps=20, u=(0.0, 0.0)):
    cf = None
    for _ in range(steps):
        root, q, qd, cf = o.scene_step(m, root, q, qd, o.cmd_map(m, u))
    return root, q, qd, cf


@pytest.fixture(scope="module")
def open_floor():
    scene = build_scene(["boxer", "block", "goal"], [[0.0, 2.5, 0.05]])
    dof, root = scene.initial_state()
    return scene, scene.to_c(), dof[0::2].astype(float), dof[1::2].astype(float), root.astype(float)


def test_free_fall_without_contacts(oracle64, open_floor):
    scene, m, q, qd, root = open_floor
    m2 = scene.to_c()
    m2.n_pairs = 0
    r, _, _, _ = oracle64.scene_step(m2, root, q, qd, [0, 0])
    # two substeps of h = 0.025 under g = 9.8: v = -0.49, z drops by h^2 g (1 + 2)
    assert r[0, 9] == pytest.approx(-0.49, abs=1e-9) and r[1, 9] == pytest.approx(-0.49, abs=1e-9)
    assert r[0, 2] == pytest.approx(0.05 - 9.8 * 0.025 ** 2 * 3, abs=1e-9)
    assert np.abs(r[0, 10:13]).max() < 1e-9  # no spurious rotation


def test_resting_contact_carries_the_weight(oracle64, open_floor):
    scene, m, q, qd, root = open_floor
    root, q, qd, cf = settle(oracle64, m, root, q, qd, 40)
    assert np.abs(root[:, 7:13]).max() < 1e-3                      # everything at rest
    total = 254.0 + 20.0                                           # base cluster (mass override) + two wheels
    robot_rows = [scene.rigid_body_index("boxer", n) for n in scene.link_names]
    assert cf[robot_rows, 2].sum() == pytest.approx(total * 9.8, rel=1e-3)
    assert cf[scene.rigid_body_index("block", "box"), 2] == pytest.approx(1.0 * 9.8, rel=1e-3)
    # wheels (r = 0.08 at z = 0.058 in the chassis) carry the robot a few mm into the penalty layer
    assert 0.015 < root[0, 2] < 0.022
    assert 0.09 < root[1, 2] < 0.1                                 # block half height 0.1 minus the penalty sink


def test_differential_drive_kinematics(oracle64, open_floor):
    """v = r * mean(wheel speed), yaw rate = r (w_R - w_L) / L - the inverse of the reference's _ik
    (isaacgym_wrapper.py:510-522) must come out of wheel-ground traction."""
    scene, m, q, qd, root = open_floor
    root, q, qd, _ = settle(oracle64, m, root, q, qd, 20)
    r1, q1, qd1, _ = settle(oracle64, m, root, q, qd, 40, u=(0.5, 0.0))
    heading = np.array([np.sin(yaw_of(r1[0, 3:7])), -np.cos(yaw_of(r1[0, 3:7]))])   # forward is -y at yaw 0
    assert r1[0, 7:9] @ heading == pytest.approx(0.5, rel=2e-2)
    np.testing.assert_allclose(qd1, [6.25, 6.25], rtol=1e-2)
    r2, q2, qd2, _ = settle(oracle64, m, root, q, qd, 40, u=(0.0, 1.0))
    assert r2[0, 12] == pytest.approx(1.0, rel=2e-2)
    assert yaw_of(r2[0, 3:7]) == pytest.approx(2.0, rel=3e-2)
    np.testing.assert_allclose(np.abs(qd2), [3.0875, 3.0875], rtol=2e-2)


def test_pushing_moves_the_block_and_reports_contact(oracle64):
    scene, m, cfg, cost, dof, root = boxer_push()
    root = root.astype(float)
    root[scene.actor_index("block"), 0:3] = [0.0, 1.9, 0.1]          # straight ahead of the robot (heading -y)
    q, qd = dof[0::2].astype(float), dof[1::2].astype(float)
    root, q, qd, _ = settle(oracle64, m, root, q, qd, 10)
    y0 = root[1, 1]
    root, q, qd, cf = settle(oracle64, m, root, q, qd, 20, u=(0.5, 0.0))
    assert root[1, 1] < y0 - 0.15                                     # the block was pushed along -y
    assert abs(root[1, 0]) < 0.05
    # Newton's third law in the reported forces: chassis and block see opposite pushes
    fb, fc = cf[scene.rigid_body_index("block", "box")], cf[scene.rigid_body_index("boxer", "chassis_link")]
    assert fb[1] < 0 < fc[1]
    # driving on pushes the block into paper_obst2 (y in [0.6, 1.4]): the obstacle must then report a contact force
    root, q, qd, cf = settle(oracle64, m, root, q, qd, 30, u=(0.5, 0.0))
    assert np.abs(cf[scene.rigid_body_index("paper_obst2", "box"), 0:2]).sum() > 1.0


def test_device_arithmetic_matches_oracle_stepwise(hostemu, oracle64):
    """fp32 world-frame structured code (LDS-indexed frames, Cholesky) vs fp64 dense oracle, re-synchronised
    every step (contact switching makes long open-loop comparisons chaotic)."""
    scene, m, cfg, cost, dof, root = boxer_push()
    root = root.astype(float)
    root[scene.actor_index("block"), 0:3] = [0.05, 1.9, 0.1]
    q, qd = dof[0::2].astype(float), dof[1::2].astype(float)
    rb = np.zeros((m.n_rb, 13), np.float32)
    cf = np.zeros((m.n_rb, 3), np.float32)
    seq = [((0.0, 0.0), 8), ((0.6, 0.0), 20), ((0.4, 0.9), 12), ((-0.3, -1.5), 10)]
    worst = 0.0
    for u, n in seq:
        for _ in range(n):
            de = np.zeros(2 * scene.n_dof, np.float32)
            de[0::2], de[1::2] = q, qd
            re = f32(root).copy()
            assert hostemu.emu_scene_step(C.byref(m), fp(de), fp(re.reshape(-1)), fp(f32(u)), fp(rb), fp(cf)) == 0
            root, q, qd, cfo = oracle64.scene_step(m, root, q, qd, oracle64.cmd_map(m, u))
            np.testing.assert_allclose(re[:, 0:7], root[:, 0:7], atol=2e-5)       # poses
            np.testing.assert_allclose(re[:, 7:13], root[:, 7:13], atol=2e-3)     # velocities
            np.testing.assert_allclose(de[1::2], qd, atol=2e-3)
            scale = max(1.0, np.abs(cfo).max())
            worst = max(worst, np.abs(cf - cfo).max() / scale)
            rbo, _ = oracle64.rigid_body_state(m, root, q, qd)
            np.testing.assert_allclose(rb[:, 0:7], rbo[:, 0:7], atol=1e-4)        # link poses incl. ee_link on the base
    assert worst < 5e-3  # net contact forces, relative to the largest force in the scene


def test_boxer_push_cost_matches_reference_expression(oracle64):
    """oracle BOXER_PUSH cost == the reference Objective's torch expression (examples/boxer_push/planner.py:26-67)."""
    import torch
    from mppiisaac.objectives import BoxerPushObjective
    scene, m, cfg, cost, dof, root = boxer_push()
    root = root.astype(float)
    root[scene.actor_index("block"), 0:3] = [0.05, 1.9, 0.1]
    q, qd = dof[0::2].astype(float), dof[1::2].astype(float)
    root, q, qd, cf = settle(oracle64, m, root, q, qd, 25, u=(0.6, 0.3))
    rb, _ = oracle64.rigid_body_state(m, root, q, qd)

    class FakeSim:  # the getters an Objective uses, on one env
        device = "cpu"
        def get_actor_link_by_name(self, actor_name, link_name): return torch.tensor(rb[scene.rigid_body_index(actor_name, link_name)])[None]
        def get_actor_position_by_name(self, n): return torch.tensor(root[scene.actor_index(n), 0:3])[None]
        def get_actor_velocity_by_name(self, n): return torch.tensor(root[scene.actor_index(n), 7:10])[None]
        def get_actor_orientation_by_name(self, n): return torch.tensor(root[scene.actor_index(n), 3:7])[None]
        def get_actor_contact_forces_by_name(self, actor_name, link_name): return torch.tensor(cf[scene.rigid_body_index(actor_name, link_name)])[None]
    want = float(BoxerPushObjective().compute_cost(FakeSim()))
    assert oracle64.cost(m, cost, root, q, qd, rb, cf) == pytest.approx(want, rel=1e-10)


def test_rollout_costs_match_oracle_on_open_floor(hostemu, oracle64):
    """whole-horizon rollouts without block/obstacle interaction are smooth: strict cost parity."""
    scene, m, cfg, cost, dof, root = boxer_push(K=32, H=12)
    root[scene.actor_index("block"), 0:3] = [2.5, 1.8, 0.0923]
    root[0, 2] = 0.019                                              # start settled on the wheels
    eps = oracle64.sample(cfg) * 0.3
    U = np.zeros((12, 2))
    S, du, viz = oracle64.rollout(m, cfg, cost, dof, root, U, eps, want_viz=True)
    Se = np.zeros(32, np.float32)
    due = np.zeros((12, 2, 32), np.float32)
    vize = np.zeros((12, 32, 3), np.float32)
    assert hostemu.emu_rollout(C.byref(m), C.byref(cfg), C.byref(cost), fp(f32(dof)), fp(f32(root)), fp(f32(U)), fp(f32(eps)), None,
                               fp(Se), fp(due), fp(vize)) == 0
    np.testing.assert_allclose(Se, S, rtol=5e-4)
    np.testing.assert_allclose(vize, viz, atol=1e-3)


def test_panda_pick_scene(hostemu, oracle64):
    """BASELINE config 5 scene: fixed-base gripper arm, 1-gram block on a static table (implicit contact),
    goal marker; rollouts through the device arithmetic match the oracle."""
    scene, m, cfg, cost, dof, root = panda_pick(K=32, H=12)
    assert scene.nu == 9 and m.n_pairs == len(scene.pairs) and oracle64.is_scene(m)
    q, qd, ro = dof[0::2].astype(float), dof[1::2].astype(float), root.astype(float)
    for _ in range(40):
        ro, q, qd, cf = oracle64.scene_step(m, ro, q, qd, np.zeros(9))
    blk, tab = scene.actor_index("panda_pick_block"), scene.rigid_body_index("table", "box")
    # resting on the table top (z = 0.14 + 0.02), at rest: the box normal's blend has compact support (a face further away than
    # five times the depth has no share in it; until round 4 (depth / distance)^3 of the side faces leaked in: a creep of ~1 um/s)
    assert 0.155 < ro[blk, 2] < 0.16 and np.abs(ro[blk, 7:13]).max() < 1e-7
    assert cf[tab, 2] == pytest.approx(-0.001 * 9.8, rel=1e-3)                     # the table carries the block's weight
    np.testing.assert_allclose(q, dof[0::2], atol=1e-6)                            # gravity is off for the arm: it holds its pose
    eps = oracle64.sample(cfg)
    U = np.zeros((12, 9))
    S, du, _ = oracle64.rollout(m, cfg, cost, dof, root, U, eps)
    Se = np.zeros(32, np.float32)
    due = np.zeros((12, 9, 32), np.float32)
    assert hostemu.emu_rollout(C.byref(m), C.byref(cfg), C.byref(cost), fp(f32(dof)), fp(f32(root)), fp(f32(U)), fp(f32(eps)), None,
                               fp(Se), fp(due), None) == 0
    np.testing.assert_allclose(Se, S, rtol=1e-4)


# ---------------------------------------------------------------- per-sample actor randomisation
def test_randomisation_draws_are_seeded_per_sample_and_bounded(oracle64):
    """reference isaacgym_wrapper.py:430-475 / isaacgym_utils.py:30-52: size ~ N(0, sigma), mass and friction
    uniform within +-percentage - here a counter-based hash of (seed, sample, actor) instead of np.random"""
    scene, m, cfg, cost, dof, root = boxer_push()
    a = scene.actor_index("block")
    scene.randomize_seed = -1
    off = oracle64.randomise_draws(scene.to_c(), 3)
    np.testing.assert_array_equal(off[:, 0:4], np.tile([0.0, 0.0, 0.0, 1.0], (m.n_actors, 1)))
    scene.randomize_seed = 11
    m = scene.to_c()
    d = np.stack([oracle64.randomise_draws(m, g) for g in range(4096)])
    np.testing.assert_array_equal(d, np.stack([oracle64.randomise_draws(m, g) for g in range(4096)]))
    assert np.all(d[:, scene.actor_index("boxer")] == [0, 0, 0, 1, m.actors[scene.actor_index("boxer")].friction])
    blk = d[:, a]
    assert abs(blk[:, 0].mean()) < 4e-4 and abs(blk[:, 0].std() - 0.005) < 3e-4   # block.yaml noise_sigma_size
    assert np.all(blk[:, 2] == 0.0)
    assert 0.7 <= blk[:, 3].min() < 0.71 and 1.29 < blk[:, 3].max() <= 1.3
    f0 = m.actors[a].friction
    assert 0.7 * f0 <= blk[:, 4].min() and blk[:, 4].max() <= 1.3 * f0 and blk[:, 4].std() > 0.1 * f0
    assert abs(np.corrcoef(blk[:, 0], blk[:, 1])[0, 1]) < 0.05
    scene.randomize_seed = 12
    assert not np.allclose(oracle64.randomise_draws(scene.to_c(), 0)[a], d[0, a])


def test_randomised_samples_device_arithmetic_matches_oracle(hostemu, oracle64):
    """every sample simulates its own block (size / mass / friction): device code with per-sample draws in LDS vs the
    oracle stepping an explicitly perturbed model, re-synchronised every step"""
    hostemu.emu_scene_step_g.restype = C.c_int
    scene, m, cfg, cost, dof, root0 = boxer_push()
    scene.randomize_seed = 5
    m = scene.to_c()
    differs = steps = switched = 0
    for g in (0, 1, 77, 4095):
        mg = oracle64.randomise_model(m, g)
        root = root0.astype(float)
        root[scene.actor_index("block"), 0:3] = [0.05, 1.9, 0.1]       # right in front of the robot: pushed within a few steps
        q, qd = dof[0::2].astype(float), dof[1::2].astype(float)
        rb = np.zeros((m.n_rb, 13), np.float32)
        cf = np.zeros((m.n_rb, 3), np.float32)
        worst = 0.0
        for u, n in [((0.0, 0.0), 8), ((0.6, 0.0), 20), ((0.4, 0.9), 12), ((-0.3, -1.5), 10)]:
            for _ in range(n):
                de = np.zeros(2 * scene.n_dof, np.float32)
                de[0::2], de[1::2] = q, qd
                re = f32(root).copy()
                assert hostemu.emu_scene_step_g(C.byref(m), fp(de), fp(re.reshape(-1)), fp(f32(u)), fp(rb), fp(cf), C.c_int(g)) == 0
                nominal = oracle64.scene_step(m, root, q, qd, oracle64.cmd_map(m, u))[0]
                root, q, qd, cfo = oracle64.scene_step(mg, root, q, qd, oracle64.cmd_map(m, u))
                differs += int(np.abs(nominal[:, 7:13] - root[:, 7:13]).max() > 1e-4)
                steps += 1
                # a support point touching down inside a substep switches its damper on (a velocity jump of beta * v_n):
                # fp32 and fp64 can take that event one substep apart.  Such steps are counted, not hidden (none occur
                # since the damper also acts on separation; an approach-only damper gave up to 2 % of them).
                if np.abs(re[:, 0:7] - root[:, 0:7]).max() > 2e-5 or np.abs(re[:, 7:13] - root[:, 7:13]).max() > 2e-3:
                    switched += 1
                    np.testing.assert_allclose(re[:, 0:7], root[:, 0:7], atol=5e-3)
                    continue
                worst = max(worst, np.abs(cf - cfo).max() / max(1.0, np.abs(cfo).max()))
        assert worst < 1e-3
    assert switched <= 1, (switched, steps)
    assert differs > 20      # the perturbed worlds really do evolve differently from the nominal one


@pytest.mark.parametrize("lanes", [4, 8])
def test_quad_split_of_contact_points_is_the_same_contact_model(lanes, hostemu, oracle64):
    """k_rollout_scene_quad deals the feature points of every pair over the 4 (quad) or 8 (octet) lanes of a sample and sums
    the partial wrenches / dampings: emulated on the host (kSplitEmulate), same steps as the one-lane arithmetic up to fp32
    summation order, and the same parity with the oracle."""
    scene, m, cfg, cost, dof, root0 = boxer_push()
    q, qd = dof[0::2].astype(float), dof[1::2].astype(float)
    root = root0.astype(float)
    root[scene.actor_index("block"), 0:3] = [0.05, 1.9, 0.1]
    rb = np.zeros((m.n_rb, 13), np.float32)
    cf1, cf4 = np.zeros((m.n_rb, 3), np.float32), np.zeros((m.n_rb, 3), np.float32)
    try:
        for u, n in [((0.0, 0.0), 8), ((0.6, 0.0), 20), ((0.4, 0.9), 12)]:
            for _ in range(n):
                out = []
                for split, cf in ((1, cf1), (lanes, cf4)):
                    hostemu.emu_set_scene_split(split)
                    de = np.zeros(2 * scene.n_dof, np.float32)
                    de[0::2], de[1::2] = q, qd
                    re = f32(root).copy()
                    assert hostemu.emu_scene_step(C.byref(m), fp(de), fp(re.reshape(-1)), fp(f32(u)), fp(rb), fp(cf)) == 0
                    out.append((de, re))
                np.testing.assert_allclose(out[1][1][:, 0:7], out[0][1][:, 0:7], atol=1e-5)     # poses
                np.testing.assert_allclose(out[1][1][:, 7:13], out[0][1][:, 7:13], atol=1e-3)   # velocities (summation order)
                np.testing.assert_allclose(out[1][0], out[0][0], atol=1e-3)
                np.testing.assert_allclose(cf4, cf1, atol=2e-3 * max(1.0, np.abs(cf1).max()))
                root, q, qd, cfo = oracle64.scene_step(m, root, q, qd, oracle64.cmd_map(m, u))
                np.testing.assert_allclose(out[1][1][:, 0:7], root[:, 0:7], atol=2e-5)
        hostemu.emu_set_scene_split(lanes)
        scene, m, cfg, cost, dof, root = panda_pick(K=16, H=10)
        eps = oracle64.sample(cfg)
        U = np.zeros((10, cfg.nu))
        S, du, _ = oracle64.rollout(m, cfg, cost, dof, root, U, eps)
        Se = np.zeros(16, np.float32)
        due = np.zeros((10, cfg.nu, 16), np.float32)
        assert hostemu.emu_rollout(C.byref(m), C.byref(cfg), C.byref(cost), fp(f32(dof)), fp(f32(root)), fp(f32(U)), fp(f32(eps)), None,
                                   fp(Se), fp(due), None) == 0
        np.testing.assert_allclose(Se, S, rtol=2e-3)
    finally:
        hostemu.emu_set_scene_split(1)


def _sym(Io):
    return np.array([[Io[0], Io[1], Io[2]], [Io[1], Io[3], Io[4]], [Io[2], Io[4], Io[5]]], float)


def _quat_R(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _momentum(scene, model_c, rb):
    """total linear momentum and angular momentum about the world origin of the robot, from the reference-layout rigid-body
    rows (pose and velocity of each body frame) and the compiled inertias (mass, first moment h = m c, Io about the frame
    origin, all in the body frame) - numpy, independent of the oracle's spatial algebra"""
    rm = scene.robot_model
    frames = [("base_link", model_c.base_mass, np.array(model_c.base_h[:3]), _sym(list(model_c.base_Io[:6])))]
    for b in rm["bodies"]:
        I = b["inertia"]
        frames.append((b["name"], I["mass"], np.asarray(I["h"], float), _sym(I["Io"])))
    P, L = np.zeros(3), np.zeros(3)
    for name, m, h, Io in frames:
        row = rb[scene.rigid_body_index(scene.robot.name, name)]
        r, R, v, w = row[0:3], _quat_R(row[3:7]), row[7:10], row[10:13]
        hw, Iw = R @ h, R @ Io @ R.T
        p_lin = m * v + np.cross(w, hw)                      # m v_c
        H_O = Iw @ w + np.cross(hw, v)                       # angular momentum about the frame origin
        P += p_lin
        L += H_O + np.cross(r, p_lin)
    return P, L


def test_floating_tree_conserves_momentum_without_external_forces(oracle64):
    """albert (diff-drive base as a free-floating body, 7-joint arm, two wheel joints) in empty space: no gravity, no contact
    pairs, joint drives working against each other.  Drives are internal forces, so the total linear and angular momentum
    (computed here from the rigid-body rows and the compiled inertias, not by the oracle) stay what they were - up to the
    first-order error of the semi-implicit Euler step, which must halve when the step is halved."""
    scene = build_scene(["albert", "goal"], [[0.0, 0.0, 1.0]])
    dof, root = scene.initial_state()
    q0, qd0 = dof[0::2].astype(float), dof[1::2].astype(float)
    rng = np.random.default_rng(3)
    qd0 = rng.uniform(-0.5, 0.5, qd0.shape)                     # the arm and the wheels already move
    root = root.astype(float)
    root[0, 7:10] = [0.2, -0.1, 0.05]                           # ... and so does the base
    root[0, 10:13] = [0.1, 0.2, -0.3]
    target = rng.uniform(-1.0, 1.0, scene.n_dof)                # joint velocity targets: the drives push the links around
    drift = []
    for substeps in (2, 4, 8):
        m = scene.to_c()
        m.n_pairs = 0
        for j in range(3):
            m.gravity[j] = 0.0
        m.substeps = substeps
        r, q, qd = root.copy(), q0.copy(), qd0.copy()
        rb, _ = oracle64.rigid_body_state(m, r, q, qd)
        P0, L0 = _momentum(scene, m, rb)
        assert np.linalg.norm(P0) > 10.0 and np.linalg.norm(L0) > 1.0       # (the state carries real momentum)
        for _ in range(10):
            r, q, qd, _ = oracle64.scene_step(m, r, q, qd, target)
        rb, _ = oracle64.rigid_body_state(m, r, q, qd)
        P1, L1 = _momentum(scene, m, rb)
        drift.append((np.linalg.norm(P1 - P0) / np.linalg.norm(P0), np.linalg.norm(L1 - L0) / np.linalg.norm(L0)))
        assert np.abs(qd - qd0).max() > 0.1                                 # the drives did change the joint velocities
    (p2, l2), (p4, l4), (p8, l8) = drift
    assert p2 < 2e-3 and l2 < 3e-3, drift                                   # measured: 6.6e-4 / 1.2e-3 at h = 25 ms
    assert 0.45 * p2 < p4 < 0.55 * p2 and 0.45 * p4 < p8 < 0.55 * p4, drift   # first order in h: halves with the step
    assert 0.45 * l2 < l4 < 0.55 * l2 and 0.45 * l4 < l8 < 0.55 * l4, drift


# ---------------------------------------------------------------- sphere against sphere
def _ball_scene(offset_x=0.0):
    """a fixed-base point robot (irrelevant here), a FIXED ball of radius 0.1 resting on the ground and a FREE 1-kg ball of the
    same radius dropped onto it - the sphere obstacles of the plannerbenchmark adapters against sphere-shaped bodies
    (reference benchmarks/panda_arm/mppi_planner/mppi_planner_wrapper.py:58-79)"""
    import copy
    from mppiisaac.planner.isaacgym_wrapper import Scene
    from mppiisaac.utils.config_store import load_config
    from mppiisaac.utils.isaacgym_utils import load_actor_cfgs, load_asset
    env = load_actor_cfgs(["point_robot", "goal", "goal"])
    fixed, free = env[1], copy.deepcopy(env[2])
    fixed.name, fixed.collision, fixed.fixed, fixed.size, fixed.init_pos = "ball_fixed", True, True, [0.1], [1.0, 1.0, 0.1]
    free.name, free.collision, free.fixed, free.gravity, free.mass, free.size = "ball_free", True, False, True, 1.0, [0.1]
    free.init_pos = [1.0 + offset_x, 1.0, 0.35]
    env = [env[0], fixed, free]
    scene = Scene(env, load_config({"defaults": [{"isaacgym": "normal"}]}).isaacgym, [load_asset(env[0])])
    return scene, scene.to_c()


def test_sphere_rests_on_sphere(oracle64):
    scene, m = _ball_scene()
    pairs = {(m.pairs[i].a, m.pairs[i].b) for i in range(m.n_pairs)}
    balls = [i for i in range(m.n_shapes) if m.shapes[i].actor in (1, 2)]
    assert (balls[0], balls[1]) in pairs                                     # the sphere-sphere pair exists (it used to be refused)
    dof, root = scene.initial_state()
    q, qd, r = dof[0::2].astype(float), dof[1::2].astype(float), root.astype(float)
    for _ in range(30):
        r, q, qd, cf = oracle64.scene_step(m, r, q, qd, np.zeros(scene.n_dof))
    d0 = 9.8 * 0.025 ** 2 / 0.8                                              # static sag of a one-point contact
    assert r[2, 2] == pytest.approx(0.3 - d0, abs=1e-6) and np.abs(r[2, 7:10]).max() < 1e-6
    assert cf[scene.rigid_body_index("ball_free", "sphere"), 2] == pytest.approx(9.8, rel=1e-6)
    assert cf[scene.rigid_body_index("ball_fixed", "sphere"), 2] == pytest.approx(-9.8, rel=1e-6)


def test_sphere_rolls_off_sphere_device_arithmetic_matches_oracle(hostemu, oracle64):
    """dropped 3 cm off centre the ball slides / rolls down the other one and lands on the ground; the host build of the device
    code follows the oracle step by step"""
    scene, m = _ball_scene(offset_x=0.03)
    dof, root = scene.initial_state()
    q, qd, r = dof[0::2].astype(float), dof[1::2].astype(float), root.astype(float)
    rb = np.zeros((m.n_rb, 13), np.float32)
    cfe = np.zeros((m.n_rb, 3), np.float32)
    for i in range(40):                     # (re-synchronised every step, as in test_device_arithmetic_matches_oracle_stepwise)
        de = np.zeros(2 * scene.n_dof, np.float32)
        de[0::2], de[1::2] = q, qd
        re = f32(r).copy()
        assert hostemu.emu_scene_step(C.byref(m), fp(de), fp(re.reshape(-1)), fp(f32(np.zeros(scene.nu))), fp(rb), fp(cfe)) == 0
        r, q, qd, cf = oracle64.scene_step(m, r, q, qd, oracle64.cmd_map(m, np.zeros(scene.nu)))
        np.testing.assert_allclose(re[:, 0:7], r[:, 0:7], atol=2e-5)
        np.testing.assert_allclose(re[:, 7:13], r[:, 7:13], atol=2e-3)
        np.testing.assert_allclose(cfe, cf, atol=5e-3 * max(1.0, np.abs(cf).max()))
    assert r[2, 0] > 1.15 and r[2, 2] < 0.1                                   # it has left the top and sits on the ground


def _without_pairs(m, drop):
    """copy of the model without the candidate pairs for which drop(shape a, shape b) holds"""
    import copy
    m2 = copy.deepcopy(m) if False else type(m).from_buffer_copy(bytes(m))
    keep = [(m.pairs[i].a, m.pairs[i].b) for i in range(m.n_pairs) if not drop(m.pairs[i].a, m.pairs[i].b)]
    for i, (a, b) in enumerate(keep):
        m2.pairs[i].a, m2.pairs[i].b = a, b
    m2.n_pairs = len(keep)
    return m2


def test_a_caster_driven_into_a_block_pushes_it(oracle64, open_floor):
    """round 5: wheels and casters meet the boxes of other actors (reference: one collision group per env,
    isaacgym_wrapper.py:436-442), not the ground only.  The chassis-block pair is taken OUT of the model here, so the only
    shapes of the robot that can touch the block are its casters (front: y = -0.274 in the chassis, radius 0.062, heading -y)
    and wheels: driving forward, the casters' rims reach the block's face and push it along - the face stays at the rims, a
    millimetre or two inside the penalty layer; with the disc-box pairs removed as well (the model of rounds 1-4) the robot
    drives straight through the block."""
    from mppiisaac.backend import capi
    scene, m, q0, qd0, root0 = open_floor
    shapes = scene.shapes
    chassis = next(i for i, s in enumerate(shapes) if s["link"] == "chassis_link")
    block = next(i for i, s in enumerate(shapes) if scene.env_cfg[s["actor"]].name == "block")
    discs = {i for i, s in enumerate(shapes) if s["type"] == capi.SHAPE_DISC}
    assert sum(1 for i in range(m.n_pairs) if m.pairs[i].a in discs and m.pairs[i].b == block) == 4      # two wheels, two casters
    assert not scene.dropped_pairs                                                                       # nothing is left out
    only_discs = _without_pairs(m, lambda a, b: {a, b} == {chassis, block})
    nothing = _without_pairs(m, lambda a, b: b == block and (a == chassis or a in discs))
    bi, rb_block = scene.actor_index("block"), scene.rigid_body_index("block", "box")
    res = {}
    for name, mm in (("discs", only_discs), ("none", nothing)):
        root = root0.copy()
        root[bi, 0:3] = [0.0, 2.5 - 0.336 - 0.15 - 0.04, 0.1]        # block face 4 cm in front of the casters' rims
        root, q, qd, _ = settle(oracle64, mm, root, q0.copy(), qd0.copy(), 10)
        y0, peak = root[bi, 1], 0.0
        for _ in range(30):
            root, q, qd, cf = oracle64.scene_step(mm, root, q, qd, oracle64.cmd_map(mm, (0.5, 0.0)))
            peak = max(peak, abs(cf[rb_block, 1]))
        res[name] = (y0 - root[bi, 1], peak, (root[scene.robot_idx, 1] - 0.336) - (root[bi, 1] + 0.15), root[bi, 8])
    pushed, force, gap, vy = res["discs"]
    assert pushed > 0.6 and force > 1.0 and vy == pytest.approx(-0.5, abs=0.02), res      # 1.5 s at 0.5 m/s, minus the 4 cm of approach
    assert -0.005 < gap < 0.0005, res                                     # the block's face rides on the casters' rims
    assert abs(res["none"][0]) < 1e-6 and res["none"][1] < 1e-9, res      # nothing touches the block without the pairs


def test_disc_box_contact_device_arithmetic_matches_oracle(hostemu, oracle64, open_floor):
    """the disc-box narrow phase of csrc/mppi_scene.hpp (disc_in_box) against the oracle's, with the contact ACTIVE: chassis-block
    pair out of the model, the robot drives its casters into the block and then turns with a wheel against it; host build of the
    device functions vs the fp64 oracle, re-synchronised every step"""
    scene, m, q0, qd0, root0 = open_floor
    shapes = scene.shapes
    chassis = next(i for i, s in enumerate(shapes) if s["link"] == "chassis_link")
    block = next(i for i, s in enumerate(shapes) if scene.env_cfg[s["actor"]].name == "block")
    mm = _without_pairs(m, lambda a, b: {a, b} == {chassis, block})
    bi, rbb = scene.actor_index("block"), scene.rigid_body_index("block", "box")
    root = root0.copy()
    root[bi, 0:3] = [0.02, 2.5 - 0.336 - 0.15 - 0.01, 0.1]
    q, qd = q0.copy(), qd0.copy()
    rb, cf = np.zeros((m.n_rb, 13), np.float32), np.zeros((m.n_rb, 3), np.float32)
    active, worst = 0, 0.0
    for u, n in (((0.0, 0.0), 8), ((0.5, 0.0), 14), ((0.3, 1.2), 14), ((0.4, -1.0), 10)):
        for _ in range(n):
            de = np.zeros(2 * scene.n_dof, np.float32)
            de[0::2], de[1::2] = q, qd
            re = f32(root).copy()
            assert hostemu.emu_scene_step(C.byref(mm), fp(de), fp(re.reshape(-1)), fp(f32(u)), fp(rb), fp(cf)) == 0
            root, q, qd, cfo = oracle64.scene_step(mm, root, q, qd, oracle64.cmd_map(mm, u))
            np.testing.assert_allclose(re[:, 0:7], root[:, 0:7], atol=2e-5)
            np.testing.assert_allclose(re[:, 7:13], root[:, 7:13], atol=2e-3)
            active += int(np.abs(cfo[rbb, 0:2]).max() > 0.1)
            worst = max(worst, np.abs(cf - cfo).max() / max(1.0, np.abs(cfo).max()))
    assert active >= 15 and worst < 5e-3, (active, worst)


def test_a_wheel_meets_a_sphere_obstacle(oracle64, hostemu):
    """round 5: disc-sphere pairs (csrc/mppi_scene.hpp disc_sphere) - the wheels of a mobile base against the sphere obstacles the
    benchmark adapters place (reference benchmarks/point_robot/mppi_planner/mppi_planner_wrapper.py:58-79).  With the chassis-sphere
    pair out of the model a fixed sphere in the path of the boxer's front casters stops the robot at their rims; oracle, and the
    host build of the device function stepwise against it with the contact active."""
    from mppiisaac.backend import capi
    from mppiisaac.planner.isaacgym_wrapper import ActorWrapper, Scene
    from mppiisaac.utils.config_store import load_config
    from mppiisaac.utils.isaacgym_utils import load_actor_cfgs, load_asset
    env = load_actor_cfgs(["boxer", "goal"])
    env[0].init_pos = [0.0, 2.5, 0.05]
    env.append(ActorWrapper(type="sphere", name="sphere0", size=[0.15], fixed=True, init_pos=[0.177, 2.5 - 0.336 - 0.15 - 0.05, 0.06]))
    scene = Scene(env, load_config({"defaults": [{"isaacgym": "normal"}]}).isaacgym, load_asset(env[0]))
    assert not scene.dropped_pairs                       # wheels against spheres are tested; only wheel-wheel pairs would be listed
    m = scene.to_c()
    shapes = scene.shapes
    chassis = next(i for i, s in enumerate(shapes) if s["link"] == "chassis_link")
    sph = next(i for i, s in enumerate(shapes) if s["type"] == capi.SHAPE_SPHERE)
    discs = [i for i, s in enumerate(shapes) if s["type"] == capi.SHAPE_DISC]
    assert sum(1 for i in range(m.n_pairs) if m.pairs[i].a in discs and m.pairs[i].b == sph) == 4
    mm = _without_pairs(m, lambda a, b: {a, b} == {chassis, sph})
    dof, root = scene.initial_state()
    q, qd, root = dof[0::2].astype(float), dof[1::2].astype(float), root.astype(float)
    root, q, qd, _ = settle(oracle64, mm, root, q, qd, 10)
    rb, cf = np.zeros((m.n_rb, 13), np.float32), np.zeros((m.n_rb, 3), np.float32)
    rbs = scene.rigid_body_index("sphere0", "sphere")
    active, worst = 0, 0.0
    for t in range(40):
        u = (0.5, 0.0)
        de = np.zeros(2 * scene.n_dof, np.float32)
        de[0::2], de[1::2] = q, qd
        re = f32(root).copy()
        assert hostemu.emu_scene_step(C.byref(mm), fp(de), fp(re.reshape(-1)), fp(f32(u)), fp(rb), fp(cf)) == 0
        root, q, qd, cfo = oracle64.scene_step(mm, root, q, qd, oracle64.cmd_map(mm, u))
        np.testing.assert_allclose(re[:, 0:7], root[:, 0:7], atol=3e-5)
        np.testing.assert_allclose(re[:, 7:13], root[:, 7:13], atol=3e-3)
        active += int(np.abs(cfo[rbs]).max() > 0.1)
        worst = max(worst, np.abs(cf - cfo).max() / max(1.0, np.abs(cfo).max()))
    # the right caster's rim (0.336 ahead of the chassis centre, at x = 0.177) runs into the sphere: for half a dozen steps the sphere
    # pushes back (kilonewtons on a 274-kg robot), the robot is turned aside and passes the obstacle
    assert active >= 4 and worst < 5e-3, (active, worst)
    assert yaw_of(root[0, 3:7]) > 0.15 and root[0, 0] > 0.1, (yaw_of(root[0, 3:7]), root[0, 0:3])
    # without the pair it would have driven straight through
    nothing = _without_pairs(mm, lambda a, b: b == sph and a in discs)
    r2, q2, qd2 = f32(scene.initial_state()[1]).astype(float), dof[0::2].astype(float), dof[1::2].astype(float)
    r2, q2, qd2, _ = settle(oracle64, nothing, r2, q2, qd2, 10)
    r2, q2, qd2, _ = settle(oracle64, nothing, r2, q2, qd2, 40, u=(0.5, 0.0))
    assert abs(yaw_of(r2[0, 3:7])) < 1e-3 and abs(r2[0, 0]) < 1e-3


def _two_jackals(tmp_path, offset, pair_normal=True):
    import yaml
    from mppiisaac.planner.isaacgym_wrapper import Scene
    from mppiisaac.utils.config_store import load_config
    from mppiisaac.utils.isaacgym_utils import load_actor_cfgs, load_asset
    from test_host_logic import JACKAL
    paths = []
    for k in (1, 2):
        p = tmp_path / f"jackal{k}.yaml"
        p.write_text(yaml.safe_dump({**JACKAL, "name": f"jackal{k}"}))
        paths.append(str(p))
    ig = load_config({"defaults": [{"isaacgym": "normal"}]}).isaacgym
    env = load_actor_cfgs(paths + ["goal"])
    env[0].init_pos, env[1].init_pos = [0.0, 0.0, 0.1], [1.2, offset, 0.1]
    env[1].init_ori = [0.0, 0.0, 1.0, 0.0]                               # facing the first one
    scene = Scene(env, ig, [load_asset(env[0]), load_asset(env[1])])
    m = scene.to_c()
    if not pair_normal:
        from mppiisaac.backend import capi
        m.contact_flags = capi.CONTACT_POINT_NORMALS
    return scene, m


def test_two_jackals_head_on_stop_instead_of_interpenetrating(oracle64, hostemu, tmp_path):
    """round 5: the moving-base robots of one env meet each other - chassis box against chassis box
    (reference: one collision group per env, isaacgym_wrapper.py:436-442) - and two DYNAMIC boxes get ONE normal per pair from the
    15-axis separating-axis test (oracle box_pair_sat).  Two jackals (46.5 kg, chassis 0.42 m long) driven at each
    other at 0.5 m/s commanded each, squarely and 5 / 15 cm off the line: they meet at a centre distance of 0.42 m, stay within
    5 cm of it and on the ground.  With the per-point normals of rounds 1-4 (`MPPI_CONTACT_POINT_NORMALS`) the pair that is
    5 cm off the line ends up INSIDE each other, one chassis lifted onto the other: the corners of the front's top edge are nearer
    to the other chassis' top face than to its front as soon as the robots pitch, and are pushed up.  The host build of the device
    function follows the oracle through the collision."""
    scene, m = _two_jackals(tmp_path, 0.0)
    shapes = scene.shapes
    chassis = [i for i, s in enumerate(shapes) if s["link"] == "chassis_link"]
    assert len(chassis) == 2 and shapes[chassis[0]]["owner"] != shapes[chassis[1]]["owner"]
    cross = [(m.pairs[i].a, m.pairs[i].b) for i in range(m.n_pairs) if m.pairs[i].b >= 0]
    assert cross == [tuple(chassis)], cross                              # one pair between the robots; wheels meet the ground only
    assert len(scene.dropped_pair_shapes) == 24 and all(2 in (shapes[a]["type"], shapes[b]["type"]) for a, b in scene.dropped_pair_shapes)   # 4 x 4 wheels, 2 x 4 wheel-chassis: listed

    def drive(m, steps=100, v=0.5, emu=False):
        dof, root = scene.initial_state()
        ro, q, qd = root.astype(float), dof[0::2].astype(float), dof[1::2].astype(float)
        dist, z, worst = [], [], 0.0
        rb, cf = np.zeros((m.n_rb, 13), np.float32), np.zeros((m.n_rb, 3), np.float32)
        for _ in range(steps):
            u = np.array([v, 0.0, v, 0.0])
            if emu:
                de = np.zeros(16, np.float32)
                de[0::2], de[1::2] = q, qd
                re = f32(ro).copy()
                assert hostemu.emu_scene_step(C.byref(m), fp(de), fp(re.reshape(-1)), fp(f32(u)), fp(rb), fp(cf)) == 0
            ro, q, qd, _ = oracle64.scene_step(m, ro, q, qd, oracle64.cmd_map(m, u))
            if emu:
                worst = max(worst, float(np.abs(re[:, 0:7] - ro[:, 0:7]).max()))
            dist.append(float(np.hypot(*(ro[1, 0:2] - ro[0, 0:2]))))
            z.append(float(max(ro[0, 2], ro[1, 2])))
        return min(dist), dist[-1], max(z), worst

    for off in (0.0, 0.05, 0.15):
        scene, m = _two_jackals(tmp_path, off)
        closest, final, top, _ = drive(m)
        assert closest > 0.37 and final > 0.38 and top < 0.1, (off, closest, final, top)     # 0.42 = touching; at rest z = 0.06
    scene, m_old = _two_jackals(tmp_path, 0.05, pair_normal=False)
    closest, final, top, _ = drive(m_old)
    assert closest < 0.05 and top > 0.2, (closest, final, top)            # rounds 1-4: through each other, one on top
    hostemu.emu_set_scene_split(1)
    scene, m = _two_jackals(tmp_path, 0.05)
    closest, _, _, worst = drive(m, steps=60, emu=True)
    assert closest < 0.42 and worst < 5e-5, (closest, worst)              # device arithmetic through the collision


def test_pair_normal_law_of_two_dynamic_boxes(oracle64):
    """box_pair_sat of the oracle on configurations with known answers (two free boxes cannot be posed without a robot in a scene:
    the boxer's chassis - half extents 0.275 x 0.37 x 0.0695, 274 kg - against a free 4-kg box), one substep at rest, no gravity:
    (i) squarely face to face with EQUAL cross sections - every corner and edge midpoint on a face plane of the other box, the
    degenerate case of the feature-point model: the two face centres carry k d / 2 along the approach axis, nothing sideways or
    vertical (k = alpha m_eff / h^2); (ii) a BAR sunk 1 cm into the chassis' top face, lying ACROSS it - no feature point of
    either box inside the other, the boxes intersect: the separating-axis contact carries it with HALF the nominal stiffness,
    straight up; with the per-point law of rounds 1-4 it feels nothing and falls through; (iii) a crate turned 45 degrees whose
    vertical edge stands 1 cm off the chassis' vertical edge - all six face axes overlap, the edge-edge axis separates: no force."""
    from mppiisaac.backend import capi
    from mppiisaac.planner.isaacgym_wrapper import ActorWrapper, Scene
    from mppiisaac.utils.config_store import load_config
    from mppiisaac.utils.isaacgym_utils import load_actor_cfgs, load_asset
    ig = load_config({"defaults": [{"isaacgym": "normal"}]}).isaacgym

    def scene_with(size):
        env = load_actor_cfgs(["boxer", "goal"])
        env.append(ActorWrapper(type="box", name="crate", size=size, fixed=False, mass=4.0, init_pos=[0.0, 0.0, 1.0]))
        scene = Scene(env, ig, load_asset(env[0]))
        dof, root0 = scene.initial_state()
        return scene, dof[0::2].astype(float), dof[1::2].astype(float), root0.astype(float), scene.actor_index("crate"), scene.rigid_body_index("crate", "box")

    def force_on_crate(scene, q, qd, root0, ci, rbc, pos, quat=(0.0, 0.0, 0.0, 1.0), law="pair"):
        m = scene.to_c()
        m.gravity[2] = 0.0                                                # (nothing but the pair acts on the crate)
        m.substeps = 1                                                    # (the reported force is the step's LAST substep's: the first one, at rest)
        if law == "points":
            m.contact_flags = capi.CONTACT_POINT_NORMALS
        M = float(sum(m.bodies[i].mass for i in range(m.n_bodies)) + m.base_mass)
        k = m.contact_alpha * (M * 4.0 / (M + 4.0)) / m.dt ** 2
        root = root0.copy()
        root[scene.robot_idx, 0:3] = [0.0, 0.0, 2.0]                      # both far above the ground, at rest
        root[ci, 0:3], root[ci, 3:7] = pos, quat
        _, _, _, cf = oracle64.scene_step(m, root, q, qd, oracle64.cmd_map(m, (0.0, 0.0)))
        return cf[rbc].copy(), k

    d = 0.012                                                             # 12 mm deep (beyond the ramp depth: full gains)
    # (i) crate of the chassis' cross-section squarely in front of it (chassis centre 0.0695 above the base, heading -y)
    sc = scene_with([0.55, 0.3, 0.139])
    for law in ("pair", "points"):
        f, k = force_on_crate(*sc, pos=[0.0, -(0.37 + 0.15) + d, 2.0 + 0.0695], law=law)
        # (the per-point law measures the face centres' depth as the smooth minimum over three face distances: 1.5 % less)
        assert f[1] < 0 and abs(f[1]) == pytest.approx(0.5 * k * d, rel=1e-5 if law == "pair" else 0.03) and np.abs(f[[0, 2]]).max() < 1e-6 * k * d, (law, f, k * d)
    # (ii) a bar 1.2 m long, 4 cm thick, along x: centre 0.3 m to the side (its middle and both ends outside the chassis'
    # footprint), 0.1 m off the chassis' centre line (the chassis' own top-face points outside the bar), 1 cm into the top face
    sb = scene_with([1.2, 0.04, 0.04])
    top = 2.0 + 2 * 0.0695
    f, k = force_on_crate(*sb, pos=[0.3, 0.1, top + 0.02 - 0.01])
    assert f[2] == pytest.approx(0.5 * k * 0.01, rel=1e-5) and np.abs(f[0:2]).max() < 1e-6 * k * 0.01, (f, k * 0.01)
    f_old, _ = force_on_crate(*sb, pos=[0.3, 0.1, top + 0.02 - 0.01], law="points")
    assert np.abs(f_old).max() == 0.0, f_old
    # (iii) the crate of (i) turned 45 degrees about the vertical, the vertical edge that points at the chassis 1 cm off the
    # chassis' front-right vertical edge (0.275, -0.37), along the diagonal
    c, s_ = np.cos(np.pi / 8), np.sin(np.pi / 8)
    R = np.array([[np.cos(np.pi / 4), -np.sin(np.pi / 4)], [np.sin(np.pi / 4), np.cos(np.pi / 4)]])
    corners = [R @ np.array([sx * 0.275, sy * 0.15]) for sx in (-1, 1) for sy in (-1, 1)]
    inward = np.array([-1.0, 1.0]) / np.sqrt(2)                           # from outside the chassis' corner towards it
    tip = max(corners, key=lambda v: float(v @ inward))                   # the crate's edge that points at the chassis
    centre = np.array([0.275, -0.37]) - 0.01 * inward - tip
    f, _ = force_on_crate(*sc, pos=[centre[0], centre[1], 2.0 + 0.0695], quat=(0.0, 0.0, s_, c))
    assert np.abs(f).max() == 0.0, f
    f_in, k = force_on_crate(*sc, pos=[centre[0] + 0.02 * inward[0], centre[1] + 0.02 * inward[1], 2.0 + 0.0695], quat=(0.0, 0.0, s_, c))
    assert np.linalg.norm(f_in) > 0.2 * k * 0.01 and f_in @ np.array([inward[0], inward[1], 0.0]) < 0, f_in   # 1 cm inside: pushed back out


def test_contact_scenes_never_integrate_with_a_step_the_contact_cannot_carry(oracle64):
    """conf/isaacgym/push.yaml (reference: dt 0.1, substeps 1 - heijn_push, anymal) asks for 100-ms steps.  The penalty contact's
    stiffness is tied to the step, k = alpha m / h^2: a body at rest sags |g| h^2 / alpha into what it rests on - 12 cm at 100 ms,
    the block of heijn_push sank into the floor until the robot's bumper passed over it (block 1.58 m from its goal after 1500
    iterations; 0.51 m - against the obstacle that covers the goal - with the cap, `profiles/r05v_task_outcomes.txt`).  Contact
    scenes multiply the configured substeps up until a step is at most 25 ms (Scene.substeps); dt stays."""
    from mppiisaac.planner.isaacgym_wrapper import Scene
    from mppiisaac.utils.config_store import load_config
    from mppiisaac.utils.isaacgym_utils import load_actor_cfgs, load_asset
    ig = load_config({"defaults": [{"isaacgym": "push"}]}).isaacgym
    assert (ig.dt, ig.substeps) == (0.1, 1)
    env = load_actor_cfgs(["heijn", "block", "paper_obst1", "paper_obst2", "goal"])
    env[0].init_pos = [0.0, 1.5, 0.05]
    scene = Scene(env, ig, load_asset(env[0]))
    m = scene.to_c()
    assert m.substeps == 4 and m.dt == pytest.approx(0.1) and m.contact_ramp_depth == pytest.approx(9.8 * 0.025 ** 2 / 0.8)
    bi = scene.actor_index("block")
    half = 0.5 * scene.env_cfg[bi].size[2]
    res = {}
    for cap in (Scene.MAX_CONTACT_SUBSTEP, 0.0):
        old = Scene.MAX_CONTACT_SUBSTEP
        try:
            Scene.MAX_CONTACT_SUBSTEP = cap
            mm = scene.to_c()
        finally:
            Scene.MAX_CONTACT_SUBSTEP = old
        dof, root = scene.initial_state()
        root, q, qd, _ = settle(oracle64, mm, root.astype(float), dof[0::2].astype(float), dof[1::2].astype(float), 40, u=(0.0,) * scene.nu)
        res[cap] = (mm.substeps, half - root[bi, 2])
    assert res[Scene.MAX_CONTACT_SUBSTEP][0] == 4 and 0.004 < res[Scene.MAX_CONTACT_SUBSTEP][1] < 0.012, res      # sag |g| h^2 / alpha = 7.7 mm
    assert res[0.0][0] == 1 and res[0.0][1] > 0.1, res                                                             # 12 cm at the configured step
    # a contact-free scene keeps its configured steps (the golden fixtures of the contact-free paths were made with them)
    arm = load_actor_cfgs(["panda", "goal"])
    assert Scene(arm, ig, load_asset(arm[0])).to_c().substeps == 1


def test_two_fixed_base_robots_of_an_env_meet_each_other(oracle64, tmp_path):
    """round 5: the moving links of different FIXED-base robots form candidate pairs too (reference conf/mppi/multi-pointbot.yaml, one
    collision group per env).  Two point robots (base cylinder as a box 0.4 m across, joints x / y / yaw of one forest) commanded at
    each other at 0.5 m/s stop 0.4 m apart - 4 mm into the penalty layer - instead of passing through each other."""
    import yaml
    from mppiisaac.planner.isaacgym_wrapper import Scene
    from mppiisaac.utils.config_store import load_config
    from mppiisaac.utils.isaacgym_utils import load_actor_cfgs, load_asset
    second = tmp_path / "point_robot2.yaml"
    second.write_text(yaml.safe_dump({"type": "robot", "name": "point_robot2", "fixed": True, "urdf_file": "point_robot.urdf"}))
    ig = load_config({"defaults": [{"isaacgym": "normal"}]}).isaacgym
    for off in (0.0, 0.1):
        env = load_actor_cfgs(["point_robot", str(second), "goal"])
        env[0].init_pos, env[1].init_pos = [0.0, 0.0, 0.05], [1.0, off, 0.05]
        scene = Scene(env, ig, [load_asset(env[0]), load_asset(env[1])])
        m = scene.to_c()
        assert m.n_pairs == 4 and all(scene.shapes[m.pairs[i].a]["owner"] != scene.shapes[m.pairs[i].b]["owner"] for i in range(4))
        dof, root = scene.initial_state()
        ro, q, qd = root.astype(float), dof[0::2].astype(float), dof[1::2].astype(float)
        gap = []
        for _ in range(80):
            ro, q, qd, cf = oracle64.scene_step(m, ro, q, qd, oracle64.cmd_map(m, np.array([0.5, 0.0, 0.0, -0.5, 0.0, 0.0])))
            gap.append((1.0 + q[3]) - q[0])
        assert 0.38 < min(gap) and 0.39 < gap[-1] < 0.4005, (off, min(gap), gap[-1])
        f1, f2 = cf[scene.rigid_body_index("point_robot", "base_link")], cf[scene.rigid_body_index("point_robot2", "base_link")]
        assert f1[0] < -1.0 and np.allclose(f1, -f2, atol=1e-9), (f1, f2)      # pushed apart, equal and opposite


# ---------------------------------------------------------------- a light body held by robot links (round 6)
def _gripper_over_block(oracle, explicit=False, gap=0.0205):
    """panda_pick with the 40-mm, 1-gram block between the opened fingers (0.37 mm clear of each pad), in free air."""
    from scipy.spatial.transform import Rotation as Rot
    from mppiisaac.backend import capi
    scene, m, cfg, cost, dof, root = panda_pick(K=8, H=4)
    if explicit:
        m.contact_flags |= capi.CONTACT_EXPLICIT_LIGHT
    q, qd, ro = dof[0::2].astype(float).copy(), dof[1::2].astype(float).copy(), root.astype(float).copy()
    q[7] = q[8] = gap
    rb, _ = oracle.rigid_body_state(m, ro, q, qd)
    lf, rf = scene.rigid_body_index("panda", "panda_leftfinger"), scene.rigid_body_index("panda", "panda_rightfinger")
    blk = scene.actor_index("panda_pick_block")
    Rl = Rot.from_quat(rb[lf, 3:7]).as_matrix()
    ro[blk, :3] = 0.5 * (rb[lf, :3] + rb[rf, :3]) + Rl @ np.array([0.0, 0.0, 0.035])   # pads: 0 .. 54 mm along the finger's z
    ro[blk, 3:7] = rb[lf, 3:7]
    ro[blk, 7:13] = 0.0

    def pad_centre(q_, qd_):
        r, _ = oracle.rigid_body_state(m, ro, q_, qd_)
        return 0.5 * (r[lf, :3] + r[rf, :3]) + Rot.from_quat(r[lf, 3:7]).as_matrix() @ np.array([0.0, 0.0, 0.035]), Rot.from_quat(r[lf, 3:7]).as_matrix()
    return scene, m, q, qd, ro, blk, pad_centre


def test_a_gripper_holds_and_lifts_a_one_gram_block(oracle64):
    """reference examples/panda_pick (planner.py:24-53, conf/actors/panda_pick_block.yaml: mass 0.001): PhysX's implicit solver lets
    the fingers close ON the block and the arm lift it (isaacgym_wrapper.py:29-36).  The explicit law of two dynamic bodies is as
    stiff as the LIGHTER one can carry - 1.3 N/m: the finger drives close the fingers through the block (asserted below: that is
    what MPPI_CONTACT_EXPLICIT_LIGHT restores).  With the pair implicit on both bodies ("light bodies"):
    the fingers stop at the block's faces (< 2 mm in), the arm lifts 10 cm and the block follows within 5 mm, and at rest in
    the closed gripper it creeps by less than 1 mm/s."""
    scene, m, q, qd, ro, blk, pad_centre = _gripper_over_block(oracle64)
    close = np.zeros(9); close[7] = close[8] = -0.1            # finger velocity drives (kd = 600, 20 N limit)
    for _ in range(8):
        ro, q, qd, cf = oracle64.scene_step(m, ro, q, qd, close)
    pen = 0.5 * (0.040 - (q[7] + q[8] - 2 * 0.000133))          # pads' inner faces sit 0.133 mm inside the finger frames
    assert 0.0 < pen < 2e-3, pen
    assert abs(q[7] - q[8]) < 4e-3                              # the block is between the fingers, not squeezed out
    pc0, R0 = pad_centre(q, qd)
    off0 = R0.T @ (ro[blk, :3] - pc0)
    assert np.abs(off0[:2]).max() < 1e-3 and abs(off0[2]) < 8e-3   # (it fell 6 mm before the pads had it)
    squeeze = np.linalg.norm(cf[scene.rigid_body_index("panda", "panda_leftfinger")])
    assert 15.0 < squeeze < 25.0                                # the drives' 20 N, not the 240 N of the first touch
    lift = close.copy(); lift[1] = -0.3                         # shoulder up: the hand rises at ~8 cm/s
    z0 = ro[blk, 2]
    worst = 0.0
    for _ in range(26):
        ro, q, qd, cf = oracle64.scene_step(m, ro, q, qd, lift)
        pc, R = pad_centre(q, qd)
        worst = max(worst, np.abs(R.T @ (ro[blk, :3] - pc) - off0).max())
    assert ro[blk, 2] - z0 > 0.10                               # lifted by more than 10 cm ...
    assert worst < 5e-3, worst                                  # ... and never more than 5 mm from where the pads took it
    for _ in range(20):
        ro, q, qd, cf = oracle64.scene_step(m, ro, q, qd, close)
    pc, R = pad_centre(q, qd)
    rb, _ = oracle64.rigid_body_state(m, ro, q, qd)
    v_pad = rb[scene.rigid_body_index("panda", "panda_leftfinger"), 7:10]
    rel = R.T @ (ro[blk, 7:10] - v_pad)
    assert abs(rel[2]) < 1e-3 and abs(rel[0]) < 1e-3, rel       # no creep along the pads (z: gravity)
    assert np.abs(R.T @ (ro[blk, :3] - pc) - off0).max() < 5e-3
    # the law of rounds 1-5: the fingers meet in the middle of the block
    scene, m, q, qd, ro, blk, pad_centre = _gripper_over_block(oracle64, explicit=True)
    for _ in range(8):
        ro, q, qd, cf = oracle64.scene_step(m, ro, q, qd, close)
    assert q[7] + q[8] < 0.005


@pytest.mark.parametrize("lanes", [1, 8])
def test_light_body_pairs_device_arithmetic_matches_oracle(lanes, hostemu, oracle64):
    """the same grasp through the host build of the device functions (fp32, pair evaluated about the block's centre, records in
    the sample's rows; lanes = 8: feature points dealt over an emulated octet), re-synchronised with the fp64 oracle every step"""
    scene, m, q, qd, ro, blk, pad_centre = _gripper_over_block(oracle64)
    close = np.zeros(9); close[7] = close[8] = -0.1
    lift = close.copy(); lift[1] = -0.3
    rb = np.zeros((m.n_rb, 13), np.float32)
    cf = np.zeros((m.n_rb, 3), np.float32)
    hostemu.emu_set_scene_split(lanes)
    try:
        worst_p = worst_v = worst_f = 0.0
        for u, n in ((close, 8), (lift, 12), (close, 6)):
            for _ in range(n):
                de = np.zeros(18, np.float32)
                de[0::2], de[1::2] = q, qd
                re = f32(ro).copy()
                assert hostemu.emu_scene_step(C.byref(m), fp(de), fp(re.reshape(-1)), fp(f32(u)), fp(rb), fp(cf)) == 0
                ro, q, qd, cfo = oracle64.scene_step(m, ro, q, qd, u)
                worst_p = max(worst_p, np.abs(re[:, 0:3] - ro[:, 0:3]).max(), np.abs(de[0::2] - q).max())
                worst_v = max(worst_v, np.abs(re[blk, 7:10] - ro[blk, 7:10]).max(), np.abs(de[1::2] - qd).max())
                worst_f = max(worst_f, np.abs(cf - cfo).max() / max(1.0, np.abs(cfo).max()))
        assert worst_p < 2e-5 and worst_v < 2e-3 and worst_f < 5e-3, (worst_p, worst_v, worst_f)
    finally:
        hostemu.emu_set_scene_split(1)


def test_light_body_law_survives_random_gripper_motion(oracle64):
    """the held block under 80 x 40 steps of random arm and finger commands (fingers opening, closing, hitting their stops, the
    hand whipping around under saturated wrist drives): every state stays finite and the gram never leaves at more than a few m/s.
    (Two earlier formulations of the coupling - an effective contact point per pair, one common reference link - passed the grasp
    test above and failed here: 1000 rad/s and NaN, a pinch ringing at the substep rate.)"""
    scene, m, q0, qd0, ro0, blk, pad_centre = _gripper_over_block(oracle64)
    close = np.zeros(9); close[7] = close[8] = -0.1
    q, qd, ro = q0.copy(), qd0.copy(), ro0.copy()
    for _ in range(6):
        ro, q, qd, cf = oracle64.scene_step(m, ro, q, qd, close)
    rng = np.random.default_rng(0)
    worst = 0.0
    for trial in range(80):
        q1, qd1, ro1 = q.copy(), qd.copy(), ro.copy()
        for t in range(40):
            if t % 4 == 0:
                u = rng.uniform(-0.2, 0.2, 9) * (rng.random(9) < 0.8)
            ro1, q1, qd1, cf = oracle64.scene_step(m, ro1, q1, qd1, u)
            assert np.isfinite(ro1).all() and np.isfinite(q1).all() and np.isfinite(cf).all(), (trial, t)
            worst = max(worst, np.abs(ro1[blk, 7:10]).max())
    assert worst < 8.0, worst    # (free fall from the gripper to the ground: 2.6 m/s; a finger flicking it: a few m/s)


def test_the_mobile_manipulator_holds_and_lifts_its_block(oracle64):
    """reference examples/omni_panda_pick: the omnipanda (three base joints, arm, gripper; EFFORT drives, conf/actors/omnipanda_effort.yaml)
    and the 0.1-kg block of conf/actors/block2.yaml - 427 times lighter than the robot, at most MPPI_LIGHT_BODY_MASS: a light body.  Closing
    forces of 6 N per finger (conf/mppi/omnipanda_effort.yaml u_max) hold it against its weight of 1 N, a shoulder torque lifts it."""
    from scipy.spatial.transform import Rotation as Rot
    scene = build_scene(["omnipanda_effort", "xaxis", "yaxis", "block2", "table2", "goal"], [[1.0, 2.0, 0.0]], isaacgym="pick")
    m = scene.to_c()
    dof, root = scene.initial_state()
    q, qd, ro = dof[0::2].astype(float).copy(), dof[1::2].astype(float).copy(), root.astype(float).copy()
    q[10] = q[11] = 0.0205
    rb, _ = oracle64.rigid_body_state(m, ro, q, qd)
    lf, rf = scene.rigid_body_index("omnipanda", "panda_leftfinger"), scene.rigid_body_index("omnipanda", "panda_rightfinger")
    blk = scene.actor_index("panda_pick_block")
    Rl = Rot.from_quat(rb[lf, 3:7]).as_matrix()
    ro[blk, :3] = 0.5 * (rb[lf, :3] + rb[rf, :3]) + Rl @ np.array([0.0, 0.0, 0.035])
    ro[blk, 3:7] = rb[lf, 3:7]
    ro[blk, 7:13] = 0.0
    squeeze = np.zeros(12); squeeze[10] = squeeze[11] = -6.0

    def pads_now():
        rb, _ = oracle64.rigid_body_state(m, ro, q, qd)
        hand = scene.rigid_body_index("omnipanda", "panda_hand")
        pads = 0.5 * (rb[lf, :3] + rb[rf, :3]) + Rot.from_quat(rb[lf, 3:7]).as_matrix() @ np.array([0.0, 0.0, 0.035])
        return pads, rb[hand, 7:10] + np.cross(rb[hand, 10:13], pads - rb[hand, :3])      # (the point of the hand between the pads)

    for _ in range(10):
        ro, q, qd, cf = oracle64.scene_step(m, ro, q, qd, squeeze)
    z_held = ro[blk, 2]
    pads, v_pads = pads_now()
    # the pads stop at its faces (not closed to zero: the grip neither chatters nor lets it through) ...
    assert 0.0 < 0.5 * (0.040 - (q[10] + q[11] - 2 * 0.000133)) < 2e-3
    # ... and it hangs between them, 1 N of weight on 2 x 6 N of grip: it slipped by millimetres while the 0.5-mm gaps closed, and now moves
    # with the hand (whose arm, effort-driven and without a torque, sags under the block's weight)
    assert np.abs(ro[blk, :3] - pads).max() < 6e-3 and np.abs(ro[blk, 7:10] - v_pads).max() < 5e-3, (ro[blk, :3] - pads, ro[blk, 7:10] - v_pads)
    lift = squeeze.copy(); lift[4] = -20.0                                         # shoulder torque (gravity is off for the arm)
    for _ in range(20):
        ro, q, qd, cf = oracle64.scene_step(m, ro, q, qd, lift)
    pads, v_pads = pads_now()
    assert ro[blk, 2] > z_held + 0.05 and np.abs(ro[blk, :3] - pads).max() < 0.012, (ro[blk, :3], pads)


def test_a_free_actor_spins_no_faster_than_the_engine_lets_it(hostemu, oracle64):
    """Isaac Gym's AssetOptions.max_angular_velocity (64 rad/s by default, which the reference keeps: isaacgym_utils.py:15) limits every rigid
    body's angular velocity; here for free actors (include/mppi_hip.h MPPI_MAX_ANGULAR_VELOCITY): the block of the gripper scene thrown up
    spinning at 300 rad/s leaves its first substep at 64 rad/s about the same axis, its linear velocity is what gravity makes of it -
    oracle and the host build of the device functions alike."""
    scene, m, cfg, cost, dof, root = panda_pick(K=32, H=12)
    blk = scene.actor_index("panda_pick_block")
    ro = root.astype(float).copy()
    ro[blk, 0:3] = [0.5, 0.0, 1.0]
    axis = np.array([2.0, -1.0, 2.0]) / 3.0
    ro[blk, 7:10] = [0.3, 0.0, 1.0]
    ro[blk, 10:13] = 300.0 * axis
    q, qd = dof[0::2].astype(float), dof[1::2].astype(float)
    r1, _, _, _ = oracle64.scene_step(m, ro.copy(), q.copy(), qd.copy(), np.zeros(9))
    w = r1[blk, 10:13]
    assert np.linalg.norm(w) == pytest.approx(64.0, rel=1e-9) and np.allclose(w / 64.0, axis, atol=1e-6)
    assert r1[blk, 7:10] == pytest.approx([0.3, 0.0, 1.0 + m.gravity[2] * m.dt], abs=1e-9)
    de = np.ascontiguousarray(np.stack([q, qd], 1).reshape(-1), np.float32)
    re = np.ascontiguousarray(ro, np.float32)
    rb, cf = np.zeros((m.n_rb, 13), np.float32), np.zeros((m.n_rb, 3), np.float32)
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    assert hostemu.emu_scene_step(C.byref(m), fp(de), fp(re.reshape(-1)), fp(np.zeros(9, np.float32)), fp(rb), fp(cf)) == 0
    # (fp32: the world-frame inertia of the spinning cube is isotropic to rounding only - the axis moves by 6e-4 in that one substep)
    assert np.linalg.norm(re[blk, 10:13]) == pytest.approx(64.0, rel=1e-5) and np.allclose(re[blk, 10:13] / 64.0, axis, atol=2e-3)
    np.testing.assert_allclose(re[blk, 0:3], r1[blk, 0:3], atol=1e-5)

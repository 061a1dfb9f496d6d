"""the remaining robots and drive modes shipped in conf/actors (effort-mode arms,
the holonomic-base arm, heijn, the 4-wheel jackal, the albert mobile manipulator).  Device arithmetic
(host build) against the oracle, re-synchronised every step."""
import ctypes as C

import numpy as np
import pytest

from scenes import build_scene

f32 = lambda a: np.ascontiguousarray(a, np.float32)
fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))

CASES = {
    "panda_effort": (["panda_effort", "goal"], [[0.0, 0.0, 0.0]], 7, 20.0),
    "omnipanda": (["omnipanda", "goal"], [[0.0, 0.0, 0.0]], 12, 0.5),
    "omnipanda_effort": (["omnipanda_effort", "goal"], [[0.0, 0.0, 0.0]], 12, 20.0),
    "heijn": (["heijn", "goal"], [[0.0, 0.0, 0.0]], 3, 1.0),
    "jackal": (["jackal", "goal"], [[0.0, 0.0, 0.1]], 2, 1.0),
    "albert": (["albert", "goal"], [[0.0, 0.0, 0.2]], 9, 0.5),
    # 12-DoF quadruped: free-floating trunk, four 3-joint legs, 37 collision primitives against the ground
    "anymal": (["anymal", "goal"], [[0.0, 0.0, 0.62]], 12, 1.0),
}


# the shipped jackal.yaml names no wheel joints (the reference raises TypeError on it); supply them
JACKAL_WHEELS = {"left_wheel_joints": ["front_left_wheel", "rear_left_wheel"], "right_wheel_joints": ["front_right_wheel", "rear_right_wheel"]}


def test_jackal_yaml_as_shipped_is_rejected():
    with pytest.raises(ValueError, match="left_wheel_joints"):
        build_scene(["jackal", "goal"], [[0.0, 0.0, 0.1]])


@pytest.mark.parametrize("split", [1, 4, 8])
@pytest.mark.parametrize("name", sorted(CASES))
def test_step_parity(name, split, hostemu, oracle64):
    """split = 4: the arithmetic of k_rollout_scene_quad on the host (contact points dealt over an emulated quad, robot
    kinematics and articulated-body solve in the quad layout incl. the gathered floating-base system for albert)"""
    actors, init, nu, umax = CASES[name]
    hostemu.emu_set_scene_split(split)
    scene = build_scene(actors, init, robot_overrides=JACKAL_WHEELS if name == "jackal" else None)
    m = scene.to_c()
    assert scene.nu == nu
    dof, root = scene.initial_state()
    q, qd, ro = dof[0::2].astype(np.float64), dof[1::2].astype(np.float64), root.astype(np.float64)
    rng = np.random.default_rng(7)
    is_scene = oracle64.is_scene(m)
    rb = np.zeros((m.n_rb, 13), np.float32)
    cf = np.zeros((m.n_rb, 3), np.float32)
    for step in range(30):
        if step % 10 == 0:
            u = rng.uniform(-umax, umax, nu)
        de = np.zeros(2 * scene.n_dof, np.float32)
        de[0::2], de[1::2] = q, qd
        re = f32(ro).copy()
        tgt = oracle64.cmd_map(m, u)
        if is_scene:
            assert hostemu.emu_scene_step(C.byref(m), fp(de), fp(re.reshape(-1)), fp(f32(u)), fp(rb), fp(cf)) == 0
            ro, q, qd, _ = oracle64.scene_step(m, ro, q, qd, tgt)
            np.testing.assert_allclose(re[:, 0:7], ro[:, 0:7], atol=5e-5)
            np.testing.assert_allclose(re[:, 7:13], ro[:, 7:13], atol=5e-3)
        else:
            qe, qde = f32(q).copy(), f32(qd).copy()
            assert hostemu.emu_step(C.byref(m), fp(f32(ro)), fp(qe), fp(qde), fp(f32(u))) == 0
            q, qd = oracle64.step(m, ro, q, qd, tgt)
            de[0::2], de[1::2] = qe, qde
        np.testing.assert_allclose(de[0::2], q, atol=5e-5)
        # a joint sitting on its limit is a knife edge: whether the inelastic clamp fires depends on the last bit of q
        lo = np.array([m.bodies[i].lower for i in range(scene.n_dof)]); hi = np.array([m.bodies[i].upper for i in range(scene.n_dof)])
        lim = np.array([bool(m.bodies[i].limited) for i in range(scene.n_dof)])
        free = ~(lim & ((np.abs(q - lo) < 1e-5) | (np.abs(q - hi) < 1e-5)))
        np.testing.assert_allclose(de[1::2][free], qd[free], atol=5e-3)
    hostemu.emu_set_scene_split(1)
    assert np.isfinite(q).all() and np.isfinite(ro).all()


def test_jackal_drives_on_four_wheels(oracle64):
    """4-wheel skid steer (conf/actors/jackal.yaml: wheel_radius 0.14, wheel_base 0.4): forward speed follows the command."""
    scene = build_scene(["jackal", "goal"], [[0.0, 0.0, 0.1]], robot_overrides=JACKAL_WHEELS)
    m = scene.to_c()
    assert scene.dof_names == ["front_left_wheel", "front_right_wheel", "rear_left_wheel", "rear_right_wheel"] and scene.nu == 2
    dof, root = scene.initial_state()
    q, qd, ro = dof[0::2].astype(np.float64), dof[1::2].astype(np.float64), root.astype(np.float64)
    for _ in range(20):
        ro, q, qd, _ = oracle64.scene_step(m, ro, q, qd, oracle64.cmd_map(m, [0.0, 0.0]))
    assert abs(ro[0, 9]) < 1e-3  # settled on its wheels
    for _ in range(40):
        ro, q, qd, _ = oracle64.scene_step(m, ro, q, qd, oracle64.cmd_map(m, [0.5, 0.0]))
    # the conf says wheel_radius 0.14 but the URDF wheels are cylinders of radius 0.098: the wheel speed follows
    # the IK (v / 0.14), the base follows the real rolling radius
    np.testing.assert_allclose(np.abs(qd), 0.5 / 0.14, rtol=0.05)
    assert np.linalg.norm(ro[0, 7:9]) == pytest.approx(0.5 * 0.098 / 0.14, rel=0.05)


def test_effort_mode_gravity_compensation_holds_the_arm(oracle64):
    """dof_mode 'effort' (reference isaacgym_wrapper.py:492-496: damping 10): with zero command the arm sags slowly
    under its damper when gravity is on, and stays put when gravity is off (conf/actors/panda_effort.yaml)."""
    scene = build_scene(["panda_effort", "goal"], [[0.0, 0.0, 0.0]])
    m = scene.to_c()
    assert m.drive_mode == 1 and m.drive_kd == 10.0
    dof, root = scene.initial_state()
    q0 = np.array([0.0, -0.94, 0.0, -2.8, 0.0, 1.8675, 0.0])
    q, qd = q0.copy(), np.zeros(7)
    for _ in range(20):
        q, qd = oracle64.step(m, root, q, qd, np.zeros(7))
    moved = np.abs(q - q0).max()
    if scene.robot.gravity:
        assert 1e-3 < moved < 1.5
    else:
        assert moved < 1e-9

"""Host-side logic that needs no GPU: config composition, C-ABI struct packing, the spline basis,
the built library's symbol table, loud failure of the product path without a GPU."""
import ctypes as C
import os

import numpy as np
import pytest

from mppiisaac.backend import capi
from mppiisaac.planner.mppi import MPPIConfig, bspline_basis, knots_for_horizon, make_config, savgol_matrix
from mppiisaac.utils.config_store import ExampleConfig, load_config
from scenes import build_scene

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_example_config_composition():
    cfg = load_config({"defaults": [{"mppi": "panda"}, {"isaacgym": "normal"}], "n_steps": 10000,
                       "actors": ["panda_stick", "goal"], "initial_actor_positions": [[0.0, 0.0, 0.0]], "nx": 14,
                       "hydra": {"searchpath": ["pkg://conf"]}})
    assert isinstance(cfg, ExampleConfig) and cfg.nx == 14
    # values of the reference's conf/mppi/panda.yaml:6-22 and conf/isaacgym/normal.yaml:4-5
    assert (cfg.mppi.num_samples, cfg.mppi.horizon, cfg.mppi.lambda_) == (200, 12, 0.05)
    assert cfg.mppi.u_min == [-0.2] and cfg.mppi.u_max == [0.2] and cfg.mppi.sample_null_action is True
    assert np.allclose(np.diag(np.asarray(cfg.mppi.noise_sigma)), 0.1) and len(cfg.mppi.noise_sigma) == 7
    assert (cfg.isaacgym.dt, cfg.isaacgym.substeps) == (0.05, 2)
    with pytest.raises(KeyError):
        load_config({"defaults": [{"mppi": "panda"}]}, overrides={"mppi.not_a_field": 1})


def test_make_config_fields_and_broadcast():
    cfg = load_config({"defaults": [{"mppi": "boxer_push"}]}).mppi
    c = make_config(cfg, k_offset=100, k_local=50)
    assert (c.num_samples, c.k_offset, c.k_total, c.nu, c.horizon) == (50, 100, 400, 2, 12)
    assert list(c.u_min)[:2] == [-1.2, -3.5] and list(c.noise_sigma_diag)[:2] == [2.0, 8.0]
    p = make_config(load_config({"defaults": [{"mppi": "panda"}]}).mppi)
    assert list(p.u_max)[:7] == [0.2] * 7 and p.n_knots == 3 and p.sampling == capi.SAMPLE_HALTON_SPLINE
    with pytest.raises(NotImplementedError):
        make_config(MPPIConfig(noise_sigma=[[1.0, 0.1], [0.1, 1.0]]))
    with pytest.raises(NotImplementedError):
        make_config(MPPIConfig(noise_sigma=[[1.0]], update_cov=True))


def test_bspline_basis_properties():
    for H in (12, 15, 20, 30):
        nk = knots_for_horizon(H)
        B = bspline_basis(H, nk)
        assert B.shape == (H, nk) and (B >= -1e-12).all()
        np.testing.assert_allclose(B.sum(1), 1.0, atol=1e-12)     # partition of unity
        assert B[0, 0] == pytest.approx(1.0) and B[-1, -1] == pytest.approx(1.0)  # clamped ends
    assert knots_for_horizon(8) == 8                               # < 3 knots: sample every step


def test_scene_layout_matches_survey_tables():
    s = build_scene(["panda_stick", "goal"])
    assert s.n_dof == 7 and s.nu == 7 and s.n_rb == 11             # B = 10 + 1
    assert s.link_names[-2:] == ["panda_ee_finger", "panda_ee_tip"]
    m = s.to_c()
    assert [m.bodies[i].effort for i in range(7)] == [87, 87, 87, 87, 12, 12, 12]
    assert m.drive_kd == 600.0 and m.substeps == 2 and m.dt == 0.05 and m.gravity[2] == -9.8
    masses = [m.bodies[i].mass for i in range(7)]
    np.testing.assert_allclose(masses, [2.975, 3.004, 2.328, 2.374, 3.419, 1.435, 0.537], atol=2e-3)
    b = build_scene(["boxer", "block", "paper_obst1", "paper_obst2", "goal"])
    assert b.nu == 2 and b.dof_names == ["wheel_right_joint", "wheel_left_joint"] and b.n_rb == 8 + 4
    g = build_scene(["panda_gripper", "xaxis", "yaxis", "panda_pick_block", "table", "goal"])
    assert g.nu == 9 and g.n_rb == 12 + 5
    p = build_scene(["point_robot", "goal"])
    assert [x["inertia"]["mass"] for x in p.robot_model["bodies"]] == [1.0, 1.0, 21.0]  # SURVEY D


def test_library_exports_every_declared_symbol():
    lib_path = capi.LIB_PATH
    assert os.path.exists(lib_path), "run __graft_entry__.build() first"
    lib = C.CDLL(lib_path)                       # loads without a GPU
    for sym in capi.EXPORTED_SYMBOLS:
        assert hasattr(lib, sym), sym
    header = open(os.path.join(ROOT, "include", "mppi_hip.h")).read()
    import re
    declared = set(re.findall(r"\b(mppi_[a-z_]+)\s*\(", header))
    assert declared == set(capi.EXPORTED_SYMBOLS), declared ^ set(capi.EXPORTED_SYMBOLS)
    lib.mppi_abi_version.restype = C.c_int
    assert lib.mppi_abi_version() == capi.ABI_VERSION


def test_stale_or_foreign_handles_are_reported_not_dereferenced():
    """the C-ABI keeps a registry of live contexts: a handle that was destroyed (or never created) yields
    MPPI_EINVAL instead of a use-after-free (no compute, runs without a GPU)."""
    lib = capi.load_library()
    junk = (C.c_char * 4096)()
    bogus = C.cast(junk, C.c_void_p)
    out = np.zeros(8, np.float32)
    assert lib.mppi_get_nominal(bogus, capi.fptr(out)) != 0
    assert b"stale or foreign" in lib.mppi_last_error()
    assert lib.mppi_rollout(bogus) != 0
    assert lib.mppi_destroy(bogus) != 0
    assert lib.mppi_record_floats(bogus) == 0
    assert lib.mppi_destroy(None) == 0


def test_direct_dof_target_setters_exist_with_the_reference_names():
    """reference isaacgym_wrapper.py:402-406 (examples/*/tuning.py, examples/anymal/world.py call them)"""
    from mppiisaac.planner.isaacgym_wrapper import IsaacGymWrapper
    for name in ("get_actor_position_by_actor_index", "get_actor_position_by_robot_index", "get_actor_velocity_by_actor_index",
                 "get_actor_velocity_by_robot_index", "get_actor_orientation_by_actor_index", "get_actor_orientation_by_robot_index",
                 "get_rigid_body_by_rigid_body_index", "_get_actor_index_by_robot_index", "set_actor_position_by_actor_index",
                 "set_actor_position_by_robot_index", "set_actor_velocity_by_actor_index", "set_actor_velocity_by_name",
                 "set_actor_velocity_by_robot_index", "set_root_state_tensor_by_actor_idx", "set_actor_dof_state", "draw_lines",
                 "save_root_state", "reset_root_state", "get_saved_root_state", "add_to_envs",
                 "update_root_state_tensor_by_obstacles", "update_root_state_tensor_by_obstacles_tensor",
                 "set_dof_velocity_target_tensor", "set_dof_actuation_force_tensor", "apply_robot_cmd", "step", "reset_to_initial_poses",
                 "get_actor_link_by_name", "get_actor_position_by_name", "get_actor_velocity_by_name", "get_actor_orientation_by_name",
                 "get_actor_contact_forces_by_name", "get_dof_state", "stop_sim", "start_sim"):
        assert callable(getattr(IsaacGymWrapper, name)), name


def test_product_path_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from mppiisaac.objectives import PandaReachObjective
    from mppiisaac.planner.mppi_isaac import MPPIisaacPlanner
    cfg = load_config({"defaults": [{"mppi": "panda"}, {"isaacgym": "normal"}], "actors": ["panda_stick", "goal"],
                       "initial_actor_positions": [[0.0, 0.0, 0.0]], "nx": 14})
    with pytest.raises(Exception):               # no CPU pipeline, no silent fallback
        MPPIisaacPlanner(cfg, PandaReachObjective(cfg))
    cfg.mppi.device = "cpu"
    with pytest.raises(capi.MppiHipError):
        MPPIisaacPlanner(cfg, PandaReachObjective(cfg))


def test_savgol_matrix_matches_scipy():
    """filter_u operator == scipy.signal.savgol_filter(mode='interp') applied along the horizon."""
    from scipy.signal import savgol_filter
    rng = np.random.default_rng(0)
    for H in (12, 20, 30):
        F = savgol_matrix(H, 9, 3)
        U = rng.normal(size=(H, 4))
        np.testing.assert_allclose(F @ U, savgol_filter(U, 9, 3, axis=0, mode="interp"), atol=1e-10)
        np.testing.assert_allclose(F.sum(1), 1.0, atol=1e-10)      # constants pass through
        t = np.arange(H, dtype=float)
        np.testing.assert_allclose(F @ (t ** 3), t ** 3, atol=1e-6)   # cubics are reproduced exactly
    assert np.array_equal(savgol_matrix(2), np.eye(2))


def test_every_reference_mppi_file_is_restated_and_builds():
    """all 18 conf/mppi files: same values as the reference's (tests/golden/mppi_cfgs.json, captured by
    tools/make_golden.py), and every one of them turns into a valid C config (or is refused for a stated reason)"""
    import glob
    import json
    import os
    from mppiisaac.backend import capi
    from mppiisaac.planner.mppi import MPPIConfig, make_config
    from mppiisaac.utils.config_store import _load_group
    from mppiisaac.utils.isaacgym_utils import CONF_DIR
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "mppi_cfgs.json")))
    names = sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(CONF_DIR, "mppi", "*.yaml")))
    assert names == sorted(gold) and len(names) == 18
    for name in names:
        cfg = _load_group("mppi", name, MPPIConfig)
        for k, v in gold[name].items():
            assert getattr(cfg, k) == v, (name, k)
        if cfg.noise_sigma is None:                # reference conf/mppi/panda_push.yaml ships without a covariance
            with pytest.raises(ValueError, match="noise_sigma"):
                make_config(cfg)
            continue
        c = make_config(cfg)
        assert c.nu == len(cfg.noise_sigma) and c.horizon == cfg.horizon
        if cfg.mppi_mode == "simple":              # e.g. omnipanda_effort: fresh Gaussian noise per command, one draw per step
            assert c.sampling == capi.SAMPLE_NORMAL and c.n_knots == cfg.horizon
        else:
            assert c.sampling == capi.SAMPLE_HALTON_SPLINE
            assert c.n_knots == (cfg.horizon // 4 if cfg.horizon >= 12 else cfg.horizon)


def test_unsupported_mppi_switches_are_refused_not_ignored():
    from mppiisaac.planner.mppi import MPPIConfig, make_config
    with pytest.raises(NotImplementedError):
        make_config(MPPIConfig(noise_sigma=[[1.0]], update_cov=True))
    with pytest.raises(ValueError, match="eta_u_bound"):   # update_lambda: a band of the weight normaliser, upper above lower
        make_config(MPPIConfig(noise_sigma=[[1.0]], update_lambda=True, eta_u_bound=3.0, eta_l_bound=5.0))
    for n in (0, 99):                                   # u_per_command: the first n rows of the updated nominal, 1 <= n <= horizon
        with pytest.raises(ValueError, match="u_per_command"):
            make_config(MPPIConfig(noise_sigma=[[1.0]], horizon=12, u_per_command=n))
    with pytest.raises(ValueError):
        make_config(MPPIConfig(noise_sigma=[[1.0]], mppi_mode="spline"))


def test_transport_fast_path_is_byte_compatible_with_torch_save_and_load():
    """torch_to_bytes / bytes_to_torch patch / view the payload of a cached torch.save archive: what they produce must load
    with plain torch.load, what plain torch.save produces must load through them, for every dtype / shape the planner
    exchanges; anything unusual takes the plain torch path"""
    import io
    import torch
    from mppiisaac.utils import transport as T
    T._SAVE_TEMPLATES.clear(); T._LOAD_LAYOUTS.clear()
    g = torch.Generator().manual_seed(0)
    for shape, dtype in (((1, 14), torch.float32), ((1, 2, 13), torch.float32), ((7,), torch.float32), ((12, 256, 3), torch.float32),
                         ((3,), torch.float64), ((2, 2), torch.int64)):
        for rep in range(3):   # 0: template built (slow path), 1..: patched template / cached layout
            t = (torch.randn(shape, generator=g) * 10).to(dtype)
            blob = T.torch_to_bytes(t)
            ref = io.BytesIO(); torch.save(t, ref)
            assert len(blob) == len(ref.getvalue())
            assert torch.equal(torch.load(io.BytesIO(blob)), t)              # a stock peer reads our blob
            assert torch.equal(T.bytes_to_torch(ref.getvalue()), t)          # we read a stock peer's blob
            back = T.bytes_to_torch(blob, map_location="cpu")
            assert torch.equal(back, t) and back.dtype == dtype and tuple(back.shape) == shape
            back += 1                                                        # a private, writable copy
    assert (torch.float32, (1, 14), "cpu") in T._SAVE_TEMPLATES and len(T._LOAD_LAYOUTS) >= 6
    a = T.bytes_to_array(T.torch_to_bytes(torch.arange(14, dtype=torch.float32).reshape(1, 14)))   # the numbers only, viewed in place
    assert a.dtype == np.float32 and a.shape == (1, 14) and (a[0] == np.arange(14)).all()
    # a payload that does not match its CRC is not served from the fast path: torch.load decides what happens with it
    t = torch.arange(14, dtype=torch.float32).reshape(1, 14)
    blob = bytearray(T.torch_to_bytes(t))
    off = T._SAVE_TEMPLATES[(torch.float32, (1, 14), "cpu")][1]
    blob[off + 1] ^= 0x40
    calls = []
    orig = torch.load
    torch.load = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    try:
        got = T.bytes_to_torch(bytes(blob))
    finally:
        torch.load = orig
    assert calls and torch.equal(got, orig(io.BytesIO(bytes(blob))))
    # non-contiguous / grad tensors: plain torch path, same result
    nc = torch.arange(12, dtype=torch.float32).reshape(3, 4).t()
    assert torch.equal(torch.load(io.BytesIO(T.torch_to_bytes(nc))), nc)


def test_two_fixed_base_robots_form_one_forest(oracle64, tmp_path):
    """several robots per env (reference isaacgym_wrapper.py:101-106,534-559,574-612; conf/mppi/multi-pointbot.yaml, nu = 6):
    fixed-base robots are merged into one articulated forest - commands, joint states and rigid-body rows of the robots follow
    one another in env order - and simulate exactly like the robots on their own; moving bases and scattered robot actors are
    refused with a reason"""
    import yaml
    from mppiisaac.planner.isaacgym_wrapper import Scene
    from mppiisaac.utils.config_store import load_config
    from mppiisaac.utils.isaacgym_utils import load_actor_cfgs, load_asset
    second = tmp_path / "point_robot2.yaml"
    second.write_text(yaml.safe_dump({"type": "robot", "name": "point_robot2", "fixed": True, "urdf_file": "point_robot.urdf"}))
    ig = load_config({"defaults": [{"isaacgym": "normal"}]}).isaacgym
    env = load_actor_cfgs(["point_robot", str(second), "goal"])
    env[0].init_pos, env[1].init_pos = [0.0, 0.0, 0.05], [1.0, -0.5, 0.05]
    env[1].init_ori = [0.0, 0.0, float(np.sin(0.4)), float(np.cos(0.4))]          # the second base is also turned by 0.8 rad
    scene = Scene(env, ig, [load_asset(env[0]), load_asset(env[1])])
    assert scene.n_dof == 6 and scene.nu == 6 and [b["parent"] for b in scene.robot_model["bodies"]] == [-1, 0, 1, -1, 3, 4]
    n1 = len(load_asset(env[0])["links"])
    assert scene.first_rb == [0, n1, 2 * n1] and scene.n_rb == 2 * n1 + 1
    assert scene.rigid_body_index("point_robot2", "base_link") == n1 + scene.rigid_body_index("point_robot", "base_link")
    m = scene.to_c()
    assert m.actors[1].first_rb == n1 and m.actors[1].n_rb == n1 and m.robot_actor == 0
    dof, root = scene.initial_state()
    q, qd = dof[0::2].astype(float), dof[1::2].astype(float)
    singles = []
    for k in (0, 1):
        e = load_actor_cfgs(["point_robot" if k == 0 else str(second), "goal"])
        e[0].init_pos, e[0].init_ori = list(env[k].init_pos), list(env[k].init_ori)
        s = Scene(e, ig, load_asset(e[0]))
        d, r = s.initial_state()
        singles.append([s, s.to_c(), r, d[0::2].astype(float), d[1::2].astype(float)])
    u = np.array([0.5, -0.2, 0.3, -0.4, 0.6, -0.1])
    for _ in range(10):
        q, qd = oracle64.step(m, root, q, qd, oracle64.cmd_map(m, u))
        for k, s in enumerate(singles):
            s[3], s[4] = oracle64.step(s[1], s[2], s[3], s[4], oracle64.cmd_map(s[1], u[3 * k:3 * k + 3]))
    np.testing.assert_allclose(q, np.concatenate([singles[0][3], singles[1][3]]), atol=1e-12)
    np.testing.assert_allclose(qd, np.concatenate([singles[0][4], singles[1][4]]), atol=1e-12)
    rb, _ = oracle64.rigid_body_state(m, root, q, qd)
    for k, s in enumerate(singles):
        rbk, _ = oracle64.rigid_body_state(s[1], s[2], s[3], s[4])
        np.testing.assert_allclose(rb[k * n1:(k + 1) * n1], rbk[:n1], atol=1e-12)
    np.testing.assert_allclose(rb[2 * n1], root[2])                            # the goal's row follows the two robots
    # refused: a moving base next to a fixed one; robot actors that are not listed next to each other
    boxer = load_actor_cfgs(["boxer", "point_robot", "goal"])
    with pytest.raises(NotImplementedError, match="either all fixed or all moving"):
        Scene(boxer, ig, [load_asset(boxer[0]), load_asset(boxer[1])])
    apart = load_actor_cfgs(["point_robot", "goal", str(second)])
    with pytest.raises(NotImplementedError, match="next to each other"):
        Scene(apart, ig, [load_asset(apart[0]), load_asset(apart[2])])


JACKAL = {"type": "robot", "differential_drive": True, "friction": 0.8, "mass": 40.0, "urdf_file": "jackal/jackal.urdf", "wheel_base": 0.4,
          "wheel_count": 4, "wheel_radius": 0.14, "left_wheel_joints": ["front_left_wheel", "rear_left_wheel"],
          "right_wheel_joints": ["front_right_wheel", "rear_right_wheel"]}


def test_two_moving_base_robots_form_a_forest_with_a_base_per_tree(oracle64, hostemu, tmp_path):
    """several MOVING-base robots per env (reference conf/mppi/multi-jackal.yaml; mppi_hip.h ABI 7): the forest carries one
    floating base per tree - a root body's parent and a base link's body index -1 - r name base r -, every robot takes its own
    (v, yaw rate) pair of commands, and on the oracle both robots move bit for bit like each robot on its own; the device
    arithmetic of the one-lane scene kernels (host build) follows the oracle step by step"""
    import ctypes as C
    import yaml
    from mppiisaac.planner.isaacgym_wrapper import Scene
    from mppiisaac.utils.config_store import load_config
    from mppiisaac.utils.isaacgym_utils import load_actor_cfgs, load_asset
    paths = []
    for k in (1, 2):
        p = tmp_path / f"jackal{k}.yaml"
        p.write_text(yaml.safe_dump({**JACKAL, "name": f"jackal{k}"}))
        paths.append(str(p))
    ig = load_config({"defaults": [{"isaacgym": "normal"}]}).isaacgym
    env = load_actor_cfgs(paths + ["goal"])
    env[0].init_pos, env[1].init_pos = [0.0, 0.0, 0.1], [3.0, 1.0, 0.1]
    scene = Scene(env, ig, [load_asset(env[0]), load_asset(env[1])])
    m = scene.to_c()
    assert scene.n_dof == 8 and scene.nu == 4 and [b["parent"] for b in scene.robot_model["bodies"]] == [-1] * 4 + [-2] * 4
    assert m.n_extra_bases == 1 and m.extra_base_actor[0] == 1 and m.extra_base_mass[0] == pytest.approx(m.base_mass)
    assert [(t[0][0], t[1][0]) for t in scene.cmd_terms] == [(0, 1)] * 4 + [(2, 3)] * 4        # (v, yaw rate) per robot
    # 2 x 14 URDF links exceed MPPI_MAX_LINKS: the links nobody can observe are not reported (as for the ANYmal)
    assert scene.n_rb == 13 and scene.rigid_body_index("jackal2", "base_link") == 6
    assert {l["body"] for l, o in zip(scene.robot_model["links"], scene.link_owner) if o == 1} == {-2, 4, 5, 6, 7}
    dof, root = scene.initial_state()
    q, qd, ro = dof[0::2].astype(float), dof[1::2].astype(float), root.astype(float)
    singles = []
    for k in (0, 1):
        e = load_actor_cfgs([paths[k], "goal"])
        e[0].init_pos = list(env[k].init_pos)
        s = Scene(e, ig, load_asset(e[0]))
        d, r = s.initial_state()
        singles.append([s.to_c(), r.astype(float), d[0::2].astype(float), d[1::2].astype(float)])
    f32 = lambda a: np.ascontiguousarray(a, np.float32)
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    rb, cf = np.zeros((m.n_rb, 13), np.float32), np.zeros((m.n_rb, 3), np.float32)
    hostemu.emu_set_scene_split(1)
    for t in range(50):
        u = np.array([0.5, 0.3, -0.3, -0.5]) if t >= 20 else np.zeros(4)
        de = np.zeros(16, np.float32)
        de[0::2], de[1::2] = q, qd
        re = f32(ro).copy()
        assert hostemu.emu_scene_step(C.byref(m), fp(de), fp(re.reshape(-1)), fp(f32(u)), fp(rb), fp(cf)) == 0
        ro, q, qd, _ = oracle64.scene_step(m, ro, q, qd, oracle64.cmd_map(m, u))
        for k, s in enumerate(singles):
            s[1], s[2], s[3], _ = oracle64.scene_step(s[0], s[1], s[2], s[3], oracle64.cmd_map(s[0], u[2 * k:2 * k + 2]))
        np.testing.assert_allclose(re[:, 0:7], ro[:, 0:7], atol=5e-5)                         # device arithmetic vs oracle
        np.testing.assert_allclose(re[:, 7:13], ro[:, 7:13], atol=5e-3)
        np.testing.assert_allclose(de[0::2], q, atol=5e-5)
        np.testing.assert_allclose(rb[[0, 6], 0:3], ro[0:2, 0:3], atol=5e-5)                  # base links follow their own base
    for k, s in enumerate(singles):                                                          # each robot exactly as on its own
        np.testing.assert_array_equal(ro[k], s[1][0])
        np.testing.assert_array_equal(q[4 * k:4 * k + 4], s[2])
    assert np.linalg.norm(ro[0, 0:2]) > 0.3 and np.linalg.norm(ro[1, 0:2] - [3.0, 1.0]) > 0.15    # and they drive
    # the library side refuses what the forest cannot be: a base that names a fixed or a non-robot actor
    m.extra_base_actor[0] = 2
    assert hostemu.emu_scene_step(C.byref(m), fp(de), fp(re.reshape(-1)), fp(f32(u)), fp(rb), fp(cf)) != 0
    # refused on the host: more moving bases than the ABI carries
    many = load_actor_cfgs([paths[0]] * 5 + ["goal"])
    with pytest.raises(ValueError, match="moving bases"):
        Scene(many, ig, [load_asset(a) for a in many[:5]])


def test_pruned_link_list_keeps_what_can_be_observed():
    """urdf_compile.prune_links (ANYmal: 78 URDF links, MPPI_MAX_LINKS = 24): the root, every link with collision geometry and
    every link the example objective names survive, parent indices point at surviving ancestors, bodies and inertias are
    untouched"""
    import json
    from mppiisaac.backend.urdf_compile import prune_links
    from mppiisaac.utils.isaacgym_utils import COMPILED_DIR
    m = json.load(open(os.path.join(COMPILED_DIR, "anymal.json")))
    names = [l["name"] for l in m["links"]]
    assert len(names) == 21 and m["pruned_links"] == 57 and names[0] == "base" and len(m["bodies"]) == 12
    for n in ("base", "face_front", "face_rear", "LF_KFE", "LH_KFE", "RH_KFE", "RF_KFE", "LF_FOOT", "RH_FOOT"):
        assert n in names
    for i, l in enumerate(m["links"]):
        assert -1 <= l["parent_link"] < i and -1 <= l["body"] < 12
        assert i == 0 or l["collision"] or l["name"] in ("base", "face_front", "face_rear", "LF_KFE", "LH_KFE", "RH_KFE", "RF_KFE")
    # pruning again changes nothing; pruning a model to its root only keeps the chain of ancestors consistent
    again = prune_links(m, ("face_front",))
    assert [l["name"] for l in again["links"]] == names
    full = json.load(open(os.path.join(COMPILED_DIR, "franka_panda_stick.json")))
    small = prune_links(full, ("panda_ee_tip",))
    kept = [l["name"] for l in small["links"]]
    assert kept[0] == full["links"][0]["name"] and "panda_ee_tip" in kept and small["bodies"] == full["bodies"]
    tip = small["links"][kept.index("panda_ee_tip")]
    assert 0 <= tip["parent_link"] < kept.index("panda_ee_tip")
    total = sum(b["inertia"]["mass"] for b in m["bodies"]) + m["base"]["inertia"]["mass"]
    assert total == pytest.approx(81.51, abs=0.01)        # every URDF inertial is still in the bodies, whatever the link list says


def test_unseen_kinematic_tree_is_built_on_demand_without_a_gpu(tmp_path, monkeypatch):
    """reference gym.load_asset takes any URDF (isaacgym_utils.py:14-29).  Here: a URDF nobody compiled before is compiled at run
    time by load_asset, and mppi_create BUILDS the kernels of its tree (hipcc cross-compiles without a GPU) into a cached plugin
    and loads it - on this box the call then stops at the first device call, after the build.  (GPU side:
    tests/test_gpu_runtime_tree.py plans on it.)"""
    import ctypes as C
    import test_gpu_runtime_tree as T
    from mppiisaac.planner.isaacgym_wrapper import Scene
    from mppiisaac.utils.isaacgym_utils import load_actor_cfgs, load_asset
    urdf, actor = str(tmp_path / "b5.urdf"), str(tmp_path / "arm5.yaml")
    T.write_branched_urdf(urdf)
    T.actor_yaml(actor, urdf)
    monkeypatch.setenv("MPPI_JIT_CACHE", str(tmp_path / "jit"))
    env = load_actor_cfgs([actor, "goal"])
    sc = Scene(env, load_config({"defaults": [{"isaacgym": "normal"}]}).isaacgym, [load_asset(a) for a in env if a.type == "robot"])
    assert [b["parent"] for b in sc.robot_model["bodies"]] == [-1, 0, 1, 0, 3] and sc.link_names[-1] == "tool"
    m, cfg = sc.to_c(), make_config(MPPIConfig(num_samples=64, horizon=8, noise_sigma=np.eye(5).tolist()), viz_link=-1)
    lib = capi.load_library()
    ctx = C.c_void_p()
    monkeypatch.setenv("MPPI_JIT", "0")
    assert lib.mppi_create(C.byref(m), C.byref(cfg), 0, C.byref(ctx)) == capi.MPPI_EUNSUPPORTED and b"MPPI_JIT=0" in lib.mppi_last_error()
    monkeypatch.delenv("MPPI_JIT")
    rc = lib.mppi_create(C.byref(m), C.byref(cfg), 0, C.byref(ctx))
    info = C.create_string_buffer(512)
    assert lib.mppi_jit_info(info, 512) == 0
    assert info.value.decode().startswith("built ") and "topo_m1_0_1_0_3_free" in info.value.decode(), (rc, lib.mppi_last_error(), info.value)
    built = os.listdir(str(tmp_path / "jit"))
    assert len(built) == 1 and built[0].endswith(".so")           # sources, objects and logs of the build are gone
    if rc != 0:                                                    # (no GPU here: the build went through, the device call did not)
        assert rc == capi.MPPI_EHIP and b"not instantiated" not in lib.mppi_last_error()
    else:
        lib.mppi_destroy(ctx)
    # a damaged file in the cache (a torn copy, another library's plugin) is thrown away and built afresh - seen by a tree this process
    # has not loaded yet: the same URDF without its last link, [-1, 0, 1, 0]
    urdf4, actor4 = str(tmp_path / "b4.urdf"), str(tmp_path / "arm4.yaml")
    T.write_branched_urdf(urdf4, tail=False)
    T.actor_yaml(actor4, urdf4, init_joint_pose=[0.0] * 8)
    env4 = load_actor_cfgs([actor4, "goal"])
    sc4 = Scene(env4, load_config({"defaults": [{"isaacgym": "normal"}]}).isaacgym, [load_asset(a) for a in env4 if a.type == "robot"])
    m4, cfg4 = sc4.to_c(), make_config(MPPIConfig(num_samples=64, horizon=8, noise_sigma=np.eye(4).tolist()), viz_link=-1)
    name4 = built[0].replace("topo_m1_0_1_0_3_free", "topo_m1_0_1_0_free")
    with open(os.path.join(str(tmp_path / "jit"), name4), "wb") as f:
        f.write(b"not a shared object")
    rc = lib.mppi_create(C.byref(m4), C.byref(cfg4), 0, C.byref(ctx))
    assert lib.mppi_jit_info(info, 512) == 0 and info.value.decode().startswith("built ") and name4 in info.value.decode(), (rc, lib.mppi_last_error(), info.value)
    assert os.path.getsize(os.path.join(str(tmp_path / "jit"), name4)) > 100000
    if rc == 0:
        lib.mppi_destroy(ctx)


def test_plugin_cache_is_private_and_foreign_plugins_are_refused(tmp_path, monkeypatch):
    """the plugin of an unseen tree is code that mppi_create loads: the cache directory is created 0700, and a directory or a cached
    plugin that somebody else could have written (group / other write bits; another owner) is refused BEFORE dlopen - the file name of a
    plugin is computable from the tree and the public sources (round-5 advice).  mppi_jit_info says what a running build is doing."""
    import ctypes as C
    import subprocess
    import sys
    import test_gpu_runtime_tree as T
    jit = str(tmp_path / "jit")
    script = f'''
import ctypes as C, os, sys, threading, time
sys.path[:0] = [{os.path.join(ROOT, "tests")!r}, {os.path.join(ROOT, "mppi-isaac_amd")!r}, {ROOT!r}]
import numpy as np
import test_gpu_runtime_tree as T
from mppiisaac.backend import capi
from mppiisaac.planner.isaacgym_wrapper import Scene
from mppiisaac.planner.mppi import make_config, MPPIConfig
from mppiisaac.utils.config_store import load_config
from mppiisaac.utils.isaacgym_utils import load_actor_cfgs, load_asset
env = load_actor_cfgs([{str(tmp_path / "arm4.yaml")!r}, "goal"])
sc = Scene(env, load_config({{"defaults": [{{"isaacgym": "normal"}}]}}).isaacgym, [load_asset(a) for a in env if a.type == "robot"])
m, cfg = sc.to_c(), make_config(MPPIConfig(num_samples=64, horizon=8, noise_sigma=np.eye(4).tolist()), viz_link=-1)
lib = capi.load_library()
seen = []
def watch():
    buf = C.create_string_buffer(512)
    while not seen or seen[-1] != "done":
        lib.mppi_jit_info(buf, 512)
        if buf.value and (not seen or seen[-1] != buf.value.decode()):
            seen.append(buf.value.decode())
        time.sleep(0.2)
t = threading.Thread(target=watch, daemon=True); t.start()
ctx = C.c_void_p()
rc = lib.mppi_create(C.byref(m), C.byref(cfg), 0, C.byref(ctx))
print("RC", rc, lib.mppi_last_error().decode()[:300].replace("\\n", " "))
print("SEEN", [s[:40] for s in seen])
'''
    urdf4, actor4 = str(tmp_path / "b4.urdf"), str(tmp_path / "arm4.yaml")
    T.write_branched_urdf(urdf4, tail=False)
    T.actor_yaml(actor4, urdf4, init_joint_pose=[0.0] * 8)
    env = dict(os.environ, MPPI_JIT_CACHE=jit)

    def run():
        r = subprocess.run([sys.executable, "-c", script], env=env, capture_output=True, text=True, timeout=600)
        assert "RC " in r.stdout, r.stderr[-2000:]
        return r.stdout
    out = run()                                                     # (1) builds; the directory is private; the build was visible as it ran
    assert (os.stat(jit).st_mode & 0o777) == 0o700
    plugin = [f for f in os.listdir(jit) if f.endswith(".so")]
    assert len(plugin) == 1 and "refusing" not in out
    seen = out.split("SEEN", 1)[1]
    assert "building " in seen, out                                 # mppi_jit_info from another thread while mppi_create blocked
    os.chmod(os.path.join(jit, plugin[0]), 0o664)                   # (2) the cached plugin is group-writable: refused, not loaded
    out = run()
    assert "refusing the cached plugin" in out and "writable by group or others" in out, out
    os.chmod(os.path.join(jit, plugin[0]), 0o644)
    os.chmod(jit, 0o770)                                            # (3) ... and so is a cache directory others can write into
    out = run()
    assert "refusing the plugin cache directory" in out, out
    os.chmod(jit, 0o700)
    out = run()                                                     # (4) private again: the cached plugin is loaded
    assert "refusing" not in out, out

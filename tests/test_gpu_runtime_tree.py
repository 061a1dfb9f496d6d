"""Robots the library has never seen (reference gym.load_asset: ANY URDF the actor YAML names, isaacgym_utils.py:14-29,
isaacgym_wrapper.py:429-447): a synthetic branched URDF written into tmp_path is compiled at run time (urdf_compile), its
kinematic tree [-1, 0, 1, 0, 3] is in no shipped instantiation, `mppi_create` builds the kernels of that tree on demand (hipcc ->
plugin library in a cache directory), and the planner plans on it - rollouts against the fp64 oracle at the tolerance of row a6.
Nothing under assets/compiled/ is touched, libmppi_hip.so is not rebuilt."""
import ctypes as C
import os
import time

import numpy as np
import pytest
import torch

from mppiisaac.backend import capi
from mppiisaac.planner.isaacgym_wrapper import IsaacGymWrapper
from mppiisaac.planner.mppi import MPPIConfig, make_config
from mppiisaac.utils.config_store import load_config
from oracle.oracle import Oracle

pytestmark = pytest.mark.gpu

LINK = """  <link name="{name}">
    <inertial><origin xyz="{cx} {cy} {cz}" rpy="0 0 0"/><mass value="{m}"/>
      <inertia ixx="{ixx}" ixy="0.001" ixz="-0.002" iyy="{iyy}" iyz="0.0015" izz="{izz}"/></inertial>
    <collision><origin xyz="0 0 {cz}" rpy="0 0 0"/><geometry><box size="0.06 0.05 {lz}"/></geometry></collision>
  </link>
"""
JOINT = """  <joint name="{name}" type="{type}">
    <parent link="{parent}"/><child link="{child}"/><origin xyz="{xyz}" rpy="{rpy}"/><axis xyz="{axis}"/>
    <limit lower="{lo}" upper="{hi}" effort="{effort}" velocity="{vel}"/>
  </joint>
"""


def write_branched_urdf(path, tail=True):
    """base - j1 -> l1 - j2 -> l2 - j4 (prismatic) -> l4 ;  l1 - j3 -> l3 - j5 -> l5 (+ a tool link welded to l5)"""
    links = "".join(LINK.format(name=n, cx=0.01 * i, cy=-0.005 * i, cz=0.1 + 0.01 * i, m=1.5 - 0.2 * i, ixx=0.02 + 0.002 * i, iyy=0.018 + 0.001 * i, izz=0.006 + 0.001 * i,
                                lz=0.2 + 0.01 * i) for i, n in enumerate(["base", "l1", "l2", "l3", "l4", "l5"]))
    links += '  <link name="tool"/>\n'
    J = [("j1", "revolute", "base", "l1", "0 0 0.25", "0 0 0.3", "0 0 1", -2.5, 2.5, 60, 2.0),
         ("j2", "revolute", "l1", "l2", "0.05 0 0.22", "0.2 0 0", "0 1 0", -1.8, 1.8, 40, 2.0),
         ("j4", "prismatic", "l2", "l4", "0 0 0.2", "0 0.1 0", "0 0 1", -0.1, 0.15, 80, 0.5),
         ("j3", "revolute", "l1", "l3", "-0.05 0.04 0.2", "0 -0.4 0.1", "1 0 0", -2.0, 2.0, 40, 2.5),
         ("j5", "continuous", "l3", "l5", "0 0 0.24", "0 0 0", "0 0.6 0.8", 0, 0, 20, 3.0)]
    if not tail:   # a second unseen tree, [-1, 0, 1, 0]: without l5 (the tool is welded to l3)
        links, J = links.replace(LINK.format(name="l5", cx=0.05, cy=-0.025, cz=0.15, m=0.5, ixx=0.03, iyy=0.023, izz=0.011, lz=0.25), ""), J[:-1]
    joints = "".join(JOINT.format(name=n, type=t, parent=p, child=c, xyz=xyz, rpy=rpy, axis=ax, lo=lo, hi=hi, effort=e, vel=v) for n, t, p, c, xyz, rpy, ax, lo, hi, e, v in J)
    joints += '  <joint name="weld" type="fixed"><parent link="%s"/><child link="tool"/><origin xyz="0 0 0.3" rpy="0 0.2 0"/></joint>\n' % ("l5" if tail else "l3")
    with open(path, "w") as f:
        f.write('<?xml version="1.0"?>\n<robot name="branched5">\n' + links + joints + "</robot>\n")


def actor_yaml(path, urdf, **extra):
    fields = dict(type="robot", name="arm5", fixed=True, collision=False, gravity=True, urdf_file=urdf, visualize_link="tool",
                  init_joint_pose=[0.3, 0, -0.4, 0, 0.05, 0, 0.5, 0, 0.2, 0])
    fields.update(extra)
    with open(path, "w") as f:
        for k, v in fields.items():
            f.write(f"{k}: {v if not isinstance(v, bool) else str(v).lower()}\n")


def program(sim, link_name, goal_actor):
    c = capi.Cost()
    c.kind, c.n_terms = capi.COST_PROGRAM, 2
    t = c.terms[0]
    t.op, t.n, t.w = capi.OP_DIST, 3, 1.0
    t.src[0], t.idx[0] = capi.SRC_RB, sim.scene.rigid_body_index("arm5", link_name)
    t.src[1], t.idx[1] = capi.SRC_ACTOR, sim.scene.actor_index(goal_actor)
    t = c.terms[1]
    t.op, t.n, t.w = capi.OP_DOF_SQ, 1, 0.05
    t.idx[0], t.idx[1], t.idx[2] = 0, sim.scene.n_dof, 0
    return c


def test_unseen_branched_urdf_is_compiled_built_and_planned_on(tmp_path, monkeypatch):
    urdf = str(tmp_path / "branched5.urdf")
    write_branched_urdf(urdf)
    actor = str(tmp_path / "arm5.yaml")
    actor_yaml(actor, urdf)
    monkeypatch.setenv("MPPI_JIT_CACHE", str(tmp_path / "jit"))   # an empty cache: the build really happens in this test
    K, H = 512, 12
    mc = MPPIConfig(num_samples=K, horizon=H, lambda_=0.05, u_min=[-1.5] * 5, u_max=[1.5] * 5, noise_sigma=(0.4 * np.eye(5)).tolist(),
                    rollout_var_discount=0.97, sample_null_action=True)
    icfg = load_config({"defaults": [{"isaacgym": "normal"}]}).isaacgym
    compiled_before = sorted(os.listdir(os.path.join(os.path.dirname(capi.LIB_PATH), "..", "assets", "compiled")))
    lib_mtime = os.path.getmtime(capi.LIB_PATH)
    t0 = time.perf_counter()
    sim = IsaacGymWrapper(icfg, actors=[actor, "goal"], init_positions=[[0.1, -0.2, 0.0]], num_envs=K,
                          mppi_config=lambda scene: make_config(mc, viz_link=scene.viz_link_index()))
    t_build = time.perf_counter() - t0
    lib = sim._lib
    info = C.create_string_buffer(512)
    capi.check(lib, lib.mppi_jit_info(info, 512))
    assert info.value.decode().startswith("built ") and "topo_m1_0_1_0_3_free" in info.value.decode(), info.value
    kinfo = C.create_string_buffer(256)
    capi.check(lib, lib.mppi_kernel_info(sim._ctx, kinfo, 256))
    assert "topology=[-1,0,1,0,3]" in kinfo.value.decode()
    assert sim.scene.n_dof == 5 and [b["parent"] for b in sim.scene.robot_model["bodies"]] == [-1, 0, 1, 0, 3]
    sim.set_actor_position_by_name([0.35, 0.1, 0.7], "goal")
    dof, root = sim._dof_state[0].cpu().numpy().copy(), sim._root_state[0].cpu().numpy().copy()
    o = Oracle("f64")
    reach = capi.Cost()      # an in-line kind: runs on the plugin's OCTET kernel (8 lanes per sample, hand-scheduled solve of this tree)
    reach.kind = capi.COST_PANDA_REACH
    reach.link[0], reach.actor[0] = sim.scene.rigid_body_index("arm5", "tool"), sim.scene.actor_index("goal")
    reach.w[0], reach.w[1] = 1.0, 0.3
    for cost, kernel in ((reach, "rollout=oct"), (program(sim, "l4", "goal"), "rollout=oct")):   # (a program on a contact-free scene: the plugin's one-lane kernel)
        capi.check(lib, lib.mppi_set_cost(sim._ctx, C.byref(cost)))
        capi.check(lib, lib.mppi_kernel_info(sim._ctx, kinfo, 256))
        assert kernel in kinfo.value.decode(), kinfo.value
        capi.check(lib, lib.mppi_sample(sim._ctx, C.c_uint32(0)))
        capi.check(lib, lib.mppi_rollout(sim._ctx))
        S = np.zeros(K, np.float32)
        capi.check(lib, lib.mppi_get_costs(sim._ctx, capi.fptr(S)))
        eps = np.zeros((H, 5, K), np.float32)
        capi.check(lib, lib.mppi_get_noise(sim._ctx, capi.fptr(eps)))
        capi.check(lib, lib.mppi_reduce(sim._ctx, None))
        capi.check(lib, lib.mppi_update(sim._ctx, None, 1))
        a = np.zeros(5, np.float32)
        capi.check(lib, lib.mppi_get_action(sim._ctx, capi.fptr(a)))
        U0 = np.zeros((H, 5))
        Uo, ao, So = o.command(sim._c_model, sim._mppi_config, cost, dof, root, U0, eps)
        rel = np.abs(S - So).max() / np.abs(So).max()
        assert rel < 1e-4, rel                          # trajectory cost 1e-4 rel
        assert np.abs(a - ao).max() < 1e-3 * 1.5        # action 1e-3 |u_max|
        U = np.zeros((H, 5), np.float32)
        capi.check(lib, lib.mppi_set_nominal(sim._ctx, capi.fptr(U)))   # both costs start from the same nominal
    # the simulator steps of the new tree (sim_step + materialise kernels of the plugin) against the oracle's step
    u = torch.tensor([[0.5, -0.8, 0.1, 0.9, -1.0]], device=sim.device)
    sim.apply_robot_cmd(u)
    for _ in range(3):
        sim.step()
    q, qd = dof[0::2].astype(np.float64), dof[1::2].astype(np.float64)
    tgt = o.cmd_map(sim._c_model, u[0].cpu().numpy())
    for _ in range(3):
        q, qd = o.step(sim._c_model, root, q, qd, tgt)
    got = sim._dof_state[7].cpu().numpy()
    assert np.abs(got[0::2] - q).max() < 1e-4 and np.abs(got[1::2] - qd).max() < 2e-3
    rb, _ = o.rigid_body_state(sim._c_model, root, q, qd)
    assert np.abs(sim._rigid_body_state[7].cpu().numpy()[:, 0:3] - rb[:, 0:3]).max() < 1e-4
    # a second context of the same tree: found among the loaded plugins, no second build; a fresh process would find it in the cache
    t0 = time.perf_counter()
    sim2 = IsaacGymWrapper(icfg, actors=[actor, "goal"], init_positions=[[0.1, -0.2, 0.0]], num_envs=64,
                           mppi_config=lambda scene: make_config(MPPIConfig(num_samples=64, horizon=H, noise_sigma=(0.4 * np.eye(5)).tolist()), viz_link=-1))
    t_again = time.perf_counter() - t0
    assert t_again < 0.5 * t_build + 2.0
    assert os.listdir(str(tmp_path / "jit")) and all(f.endswith(".so") for f in os.listdir(str(tmp_path / "jit")))
    assert sorted(os.listdir(os.path.join(os.path.dirname(capi.LIB_PATH), "..", "assets", "compiled"))) == compiled_before
    assert os.path.getmtime(capi.LIB_PATH) == lib_mtime
    print(f"\n[runtime tree] [-1,0,1,0,3]: first context {t_build:.1f} s ({info.value.decode()}), next context {t_again * 1e3:.0f} ms; "
          "steady state: the plugin holds the same template instantiations a shipped tree has (slowdown 1.0x)")
    sim.stop_sim()
    sim2.stop_sim()


def test_on_demand_builds_can_be_switched_off(tmp_path, monkeypatch):
    urdf = str(tmp_path / "branched4.urdf")
    write_branched_urdf(urdf, tail=False)       # (a tree no earlier test of this process has loaded a plugin for)
    actor = str(tmp_path / "arm4.yaml")
    actor_yaml(actor, urdf, init_joint_pose=[0.0] * 8)
    monkeypatch.setenv("MPPI_JIT_CACHE", str(tmp_path / "jit_off"))
    monkeypatch.setenv("MPPI_JIT", "0")
    icfg = load_config({"defaults": [{"isaacgym": "normal"}]}).isaacgym
    with pytest.raises(capi.MppiHipError, match="could not be built on demand.*MPPI_JIT=0"):
        IsaacGymWrapper(icfg, actors=[actor, "goal"], num_envs=64,
                        mppi_config=lambda scene: make_config(MPPIConfig(num_samples=64, horizon=8, noise_sigma=np.eye(4).tolist()), viz_link=-1))
    assert not os.path.exists(str(tmp_path / "jit_off")) or not os.listdir(str(tmp_path / "jit_off"))


def test_three_free_actors_get_kernels_with_four_free_slots(tmp_path, monkeypatch, oracle64):
    """MPPI_MAX_FREE = 4: the shipped scene kernels carry two free-actor slots (every example of the reference has at most two free
    actors; two more slots cost the register-bound scene kernels 26 state values per sample); an env with THREE free boxes gets the
    scene kernels of its tree built on demand with -DMPPI_FREE_SLOTS=4 - here the boxer with three blocks: rollouts of the octet
    kernel (helper wavefront) and the K = 1 world's step against the fp64 oracle."""
    import yaml
    from mppiisaac.utils.isaacgym_utils import CONF_DIR
    from scenes import boxer_push
    monkeypatch.setenv("MPPI_JIT_CACHE", str(tmp_path / "jit"))
    blocks = []
    base = yaml.safe_load(open(os.path.join(CONF_DIR, "actors", "block.yaml")))
    for i, pos in enumerate(([0.3, 1.6, 0.1], [-0.5, 1.9, 0.1], [0.1, 3.3, 0.1])):
        cfgb = dict(base, name=f"block{i}", init_pos=pos, noise_sigma_size=[0.0, 0.0, 0.0], noise_percentage_mass=0.0, noise_percentage_friction=0.0)
        path = str(tmp_path / f"block{i}.yaml")
        with open(path, "w") as f:
            yaml.safe_dump(cfgb, f)
        blocks.append(path)
    K, H = 256, 10
    mc = MPPIConfig(num_samples=K, horizon=H, lambda_=0.1, u_min=[-0.8, -1.5], u_max=[0.8, 1.5], noise_sigma=[[0.3, 0.0], [0.0, 0.8]], sample_null_action=True)
    icfg = load_config({"defaults": [{"isaacgym": "normal"}]}).isaacgym
    sim = IsaacGymWrapper(icfg, actors=["boxer"] + blocks + ["goal"], init_positions=[[0.0, 2.5, 0.05]], num_envs=K,
                          mppi_config=lambda scene: make_config(mc, viz_link=scene.viz_link_index()))
    lib = sim._lib
    info = C.create_string_buffer(512)
    capi.check(lib, lib.mppi_jit_info(info, 512))
    assert "topo_m1_m1_scene4" in info.value.decode(), info.value
    assert sim._c_model.n_actors == 5 and sum(1 for a in sim.env_cfg if a.type == "box" and not a.fixed) == 3
    cost = capi.Cost()
    cost.kind, cost.n_terms = capi.COST_PROGRAM, 3
    for j in range(3):
        t = cost.terms[j]
        t.op, t.n, t.w = capi.OP_DIST, 2, 1.0 + j
        t.src[0], t.idx[0] = capi.SRC_ACTOR, sim.scene.actor_index(f"block{j}")
        t.src[1], t.idx[1] = capi.SRC_ACTOR, sim.scene.actor_index("goal")
    capi.check(lib, lib.mppi_set_cost(sim._ctx, C.byref(cost)))
    dof, root = sim._dof_state[0].cpu().numpy().copy(), sim._root_state[0].cpu().numpy().copy()
    capi.check(lib, lib.mppi_sample(sim._ctx, C.c_uint32(0)))
    rng = np.random.default_rng(3)
    U = (0.4 * rng.normal(size=(H, 2))).astype(np.float32)
    U[:, 0] = np.abs(U[:, 0]) + 0.3                       # drive into the blocks
    capi.check(lib, lib.mppi_set_nominal(sim._ctx, capi.fptr(U)))
    capi.check(lib, lib.mppi_rollout(sim._ctx))
    S, eps = np.zeros(K, np.float32), np.zeros((H, 2, K), np.float32)
    capi.check(lib, lib.mppi_get_costs(sim._ctx, capi.fptr(S)))
    capi.check(lib, lib.mppi_get_noise(sim._ctx, capi.fptr(eps)))
    So, _, _ = oracle64.rollout(sim._c_model, sim._mppi_config, cost, dof, root, U, eps)
    rel = np.abs(S - So) / np.abs(So)
    print(f"\n[three free actors] {info.value.decode()}; rollouts vs fp64 oracle: within 1e-4 {np.mean(rel <= 1e-4):.4f}, 1e-3 {np.mean(rel <= 1e-3):.4f}, max {rel.max():.1e}")
    assert np.mean(rel <= 1e-3) >= 0.99 and np.median(rel) < 1e-5
    assert np.abs(So - So[0]).max() > 1e-3                # (the rollouts differ: the blocks are pushed around)
    sim.stop_sim()

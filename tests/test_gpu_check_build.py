"""The CHECK build of the library on the GPU (libmppi_hip_check.so, built next to the product library by
__graft_entry__.build(): -DMPPI_CHECK).  Every access to a sample's LDS rows is bounds-checked against the row length the kernel
allocated, and the workgroup barriers of the owner / helper-wavefront kernels carry a phase canary (both wavefronts must be at
the SAME barrier of the sequence; csrc/mppi_scene.hpp MPPI_BARRIER).  A violation traps: the launch fails.  Run here on the
cases that stress exactly that code: ragged sample counts (partly filled wavefronts, a last workgroup whose helper wavefront
has fewer live samples), the helper-wavefront kernel of the short trees, the octet kernel of the gripper scene, the
trajectory-dumping instantiations of generic mode.  Zero findings = every launch completes and the costs agree with the
product build's."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from mppiisaac.backend import capi
from scenes import boxer_push, panda_pick, point_reach

pytestmark = pytest.mark.gpu
CHECK_LIB = os.path.join(os.path.dirname(capi.__file__), "..", "..", "csrc", "libmppi_hip_check.so")


def costs(lib, m, cfg, cost, dof, root, trajectory=False):
    ctx = C.c_void_p()
    capi.check(lib, lib.mppi_create(C.byref(m), C.byref(cfg), 0, C.byref(ctx)))
    capi.check(lib, lib.mppi_set_cost(ctx, C.byref(cost)))
    d, r = np.ascontiguousarray(dof, np.float32), np.ascontiguousarray(root, np.float32)
    capi.check(lib, lib.mppi_set_state(ctx, capi.fptr(d), capi.fptr(r)))
    capi.check(lib, lib.mppi_sample(ctx, C.c_uint32(0)))
    for _ in range(2):
        capi.check(lib, lib.mppi_rollout(ctx))
    capi.check(lib, lib.mppi_synchronize(ctx))            # a trap inside the kernel surfaces here
    S = np.zeros(cfg.num_samples, np.float32)
    capi.check(lib, lib.mppi_get_costs(ctx, capi.fptr(S)))
    info = C.create_string_buffer(256)
    lib.mppi_kernel_info(ctx, info, 256)
    if trajectory:                                        # the DUMP instantiations (generic Objective mode)
        capi.check(lib, lib.mppi_rollout_trajectory(ctx))
        capi.check(lib, lib.mppi_synchronize(ctx))
    lib.mppi_destroy(ctx)
    return S, info.value.decode()


@pytest.mark.parametrize("make,K,H,kernel", [(boxer_push, 1021, 8, "scene-oct-pair"), (boxer_push, 8192, 6, "scene-oct-pair"),
                                             (panda_pick, 999, 6, "scene-oct"), (point_reach, 77, 5, None)])
def test_check_build_finds_nothing(make, K, H, kernel):
    assert torch.cuda.is_available()
    assert os.path.exists(CHECK_LIB), "libmppi_hip_check.so is missing: __graft_entry__.build() builds it next to the product library"
    chk, prod = capi.load_library(CHECK_LIB), capi.load_library()
    scene, m, cfg, cost, dof, root = make(K=K, H=H)
    if make is not point_reach:
        root = np.array(root, np.float32)
        if make is boxer_push:
            root[scene.actor_index("block"), 0:3] = [0.0, 1.9, 0.0923]     # in front of the robot: box-box contact in most samples
    Sc, info = costs(chk, m, cfg, cost, dof, root, trajectory=True)
    Sp, _ = costs(prod, m, cfg, cost, dof, root)
    if kernel:
        assert f"rollout={kernel} " in info, info
    assert np.isfinite(Sc).all()
    rel = np.abs(Sc - Sp) / np.abs(Sp)
    # (the check build's extra branches change how products and sums are contracted: last-bit differences, amplified by contact)
    assert np.median(rel) < 1e-5 and (rel <= 1e-3).mean() >= 0.99, (np.median(rel), rel.max())


def test_check_build_on_a_forest_of_moving_bases():
    """two jackals in one env (mppi_hip.h ABI 7: one floating base per tree; the one-lane scene kernels with NB + 2 base frames and
    accumulators per sample): every LDS row access of the rollout inside the rows the kernel allocated, costs as the product build's"""
    from mppiisaac.objectives import MultiJackalObjective, compile_program
    from mppiisaac.planner.mppi import MPPIConfig, make_config
    from scenes import build_scene
    assert os.path.exists(CHECK_LIB)
    chk, prod = capi.load_library(CHECK_LIB), capi.load_library()
    scene = build_scene(["jackal_a", "jackal_b", "goal"], [[0.0, 0.0, 0.1], [0.5, -2.0, 0.1]])
    m = scene.to_c()
    cfg = make_config(MPPIConfig(num_samples=100, horizon=10, noise_sigma=np.eye(4).tolist(), lambda_=0.01, u_min=[-1.5], u_max=[1.5],
                                 sample_null_action=True), viz_link=scene.viz_link_index())
    obj = MultiJackalObjective()
    cost = compile_program(obj.terms(), obj.weights, scene)
    dof, root = scene.initial_state()
    Sc, info = costs(chk, m, cfg, cost, dof, root)
    Sp, _ = costs(prod, m, cfg, cost, dof, root)
    assert "rollout=scene " in info and "-2,-2,-2,-2]" in info, info
    assert np.isfinite(Sc).all()
    np.testing.assert_allclose(Sc, Sp, rtol=1e-4)

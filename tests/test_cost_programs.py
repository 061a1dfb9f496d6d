"""MPPI_COST_PROGRAM on the host: the kernels' term interpreter (csrc/mppi_device.hpp: program_cost, compiled for the host
by tests/hostemu) against the oracle's interpreter (pinned to the reference's planners by tests/test_golden_boundary.py) and
against the in-line cost kinds it generalises.  The -m gpu tests repeat this through the C-ABI on the device."""
import ctypes as C

import numpy as np
import pytest

import mppiisaac.objectives as objectives
from mppiisaac.backend import capi
from scenes import boxer_push, build_scene, panda_pick, panda_reach, point_reach
from test_hostemu_parity import emu_rollout


class _Sim:
    def __init__(self, scene):
        self.scene = scene


@pytest.mark.parametrize("make,Obj,K,H", [(panda_reach, objectives.PandaReachObjective, 64, 12), (point_reach, objectives.PointReachObjective, 64, 10),
                                          (boxer_push, objectives.BoxerPushObjective, 16, 10), (panda_pick, objectives.PandaPickObjective, 16, 10)])
def test_program_equals_inline_kind_and_oracle(make, Obj, K, H, hostemu, oracle64):
    """the four in-kernel cost kinds restated as programs: same rollout costs as the in-line code (device arithmetic on the
    host) and as the oracle's interpreter"""
    scene, m, cfg, cost, dof, root = make(K=K, H=H)
    prog = Obj(None).program_spec(_Sim(scene))
    assert prog.kind == capi.COST_PROGRAM and prog.n_terms >= 1
    eps = oracle64.sample(cfg)
    U = 0.02 * np.random.default_rng(1).normal(size=(H, cfg.nu))
    S_kind, _, _ = emu_rollout(hostemu, m, cfg, cost, dof, root, U, eps)
    S_prog, _, _ = emu_rollout(hostemu, m, cfg, prog, dof, root, U, eps)
    np.testing.assert_allclose(S_prog, S_kind, rtol=2e-5)
    So, _, _ = oracle64.rollout(m, cfg, prog, dof, root, U, eps)
    Sk, _, _ = oracle64.rollout(m, cfg, cost, dof, root, U, eps)
    np.testing.assert_allclose(So, Sk, rtol=1e-6)                    # oracle: program == kind (fp32 weights in both)
    np.testing.assert_allclose(S_prog, So, rtol=2e-3 if make in (boxer_push, panda_pick) else 5e-5)


@pytest.mark.parametrize("case", ["boxer_reach", "heijn_reach", "heijn_push", "albert", "omni_panda_pick", "panda_stick_push"])
def test_example_objectives_as_programs_match_oracle(case, hostemu, oracle64):
    """the example objectives that have no in-line kind, on their own example scenes: rollout costs through the device
    arithmetic (host build) vs the oracle"""
    from mppiisaac.planner.mppi import MPPIConfig, make_config
    from test_golden_boundary import EXAMPLE_SCENES, OBJECTIVES
    over = {"left_wheel_joints": None}
    scene = build_scene(EXAMPLE_SCENES[case], [[0.0, 0.0, 0.2 if case == "albert" else 0.05]])
    m = scene.to_c()
    K, H, nu = 8, 8, scene.nu
    cfg = make_config(MPPIConfig(num_samples=K, horizon=H, noise_sigma=(0.2 * np.eye(nu)).tolist(), lambda_=0.1, u_min=[-0.5], u_max=[0.5],
                                 sample_null_action=True), viz_link=scene.viz_link_index())
    prog = getattr(objectives, OBJECTIVES[case])(None).program_spec(_Sim(scene))
    dof, root = scene.initial_state()
    eps = oracle64.sample(cfg)
    U = np.zeros((H, nu))
    S, _, _ = emu_rollout(hostemu, m, cfg, prog, dof, root, U, eps)
    So, _, _ = oracle64.rollout(m, cfg, prog, dof, root, U, eps)
    assert np.isfinite(So).all() and (So > 0).all()
    np.testing.assert_allclose(S, So, rtol=2e-3)


def test_program_validation_errors():
    scene, m, cfg, cost, dof, root = panda_reach(K=8, H=4)
    obj = objectives.PandaReachObjective(None)
    spec = obj.program_spec(_Sim(scene))
    assert spec.terms[0].op == capi.OP_DIST and spec.terms[0].src[0] == capi.SRC_RB and spec.terms[1].op == capi.OP_TILT
    with pytest.raises(ValueError):
        objectives.compile_program([objectives.Term(1.0, "dist", ((0.0, 0.0, 0.0), objectives.actor("goal"), 3))], {}, scene)
    with pytest.raises(ValueError):
        objectives.compile_program([objectives.Term(1.0, "dist", (objectives.actor("goal"), objectives.actor("goal"), 3))] * 17, {}, scene)
    with pytest.raises(ValueError):
        objectives.compile_program([objectives.Term(1.0, "dist", (objectives.actor("nobody"), objectives.actor("goal"), 3))], {}, scene)

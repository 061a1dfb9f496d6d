"""Reference-style Python Objectives traced into cost programs, on the GPU: the traced programs through the in-kernel interpreter
against the reference's golden costs; the planner binds them, validates them against the eager Objective and falls back to generic
mode - saying why - when an Objective cannot be traced or drifts (mppiisaac/trace.py)."""
import logging

import numpy as np
import pytest
import torch

from mppiisaac.backend import capi

pytestmark = pytest.mark.gpu


def _planner(objective, K=256, H=8, **over):
    from mppiisaac.planner.mppi_isaac import MPPIisaacPlanner
    from mppiisaac.utils.config_store import load_config
    cfg = load_config({"defaults": [{"mppi": "panda"}, {"isaacgym": "normal"}], "actors": ["panda_stick", "goal"], "initial_actor_positions": [[0.0, 0.0, 0.0]],
                       "nx": 14}, overrides={"mppi.num_samples": K, "mppi.horizon": H, "mppi.filter_u": False, **{f"mppi.{k}": v for k, v in over.items()}})
    p = MPPIisaacPlanner(cfg, objective)
    p.sim.set_actor_position_by_name([0.5, -0.4, 0.3], "goal")
    return p


Q0 = [0.0, -0.94, 0.0, -2.8, 0.0, 1.8675, 0.0]


@pytest.mark.parametrize("case", sorted(__import__("reference_style_objectives").CASES))
def test_traced_reference_style_objectives_through_the_hip_interpreter(case):
    """reference-style restatement -> trace -> MPPI_COST_PROGRAM -> mppi_eval_cost on the reference's golden inputs == the
    reference planner's compute_cost (tests/golden/objective_costs.json) to 2e-6"""
    from mppiisaac import trace
    from mppiisaac.objectives import compile_program
    from mppiisaac.planner.mppi import MPPIConfig, make_config
    from reference_style_objectives import CASES
    from scenes import build_scene
    from test_golden_boundary import EXAMPLE_SCENES, gold
    from test_gpu_parity import Ctx
    from test_trace import _StubSim
    g = gold("objective_costs.json")[case]
    scene = build_scene(EXAMPLE_SCENES[case], [[0.0, 0.0, 0.05]])
    m = scene.to_c()
    spec = compile_program(trace.trace_objective(CASES[case](None), _StubSim(scene)), {}, scene)
    c = Ctx(m, make_config(MPPIConfig(num_samples=64, horizon=4, noise_sigma=np.eye(m.nu).tolist())), spec)
    n, nd = len(g["cost"]), scene.n_dof
    dof, root = np.zeros((n, 2 * nd), np.float32), np.zeros((n, m.n_actors, 13), np.float32)
    rb, cf = np.zeros((n, m.n_rb, 13), np.float32), np.zeros((n, m.n_rb, 3), np.float32)
    rb[:, :, 6] = 1.0
    root[:, :, 6] = 1.0
    for key, val in g["inputs"].items():
        val = np.asarray(val, np.float32)
        kind, *names = key.split(":")
        if kind == "link":
            rb[:, scene.rigid_body_index(*names)] = val
        elif kind == "contact":
            cf[:, scene.rigid_body_index(*names)] = val
        elif kind == "dof_state":
            dof[:, :] = val[:, :2 * nd]
        else:
            root[:, scene.actor_index(names[0]), {"position": slice(0, 3), "orientation": slice(3, 7), "velocity": slice(7, 10)}[kind]] = val
    out = np.zeros(n, np.float32)
    c.call("mppi_eval_cost", n, capi.fptr(dof), capi.fptr(root), capi.fptr(rb), capi.fptr(cf), capi.fptr(out))
    c.close()
    np.testing.assert_allclose(out, np.asarray(g["cost"]), rtol=2e-6, atol=2e-6)


def test_a_reference_style_objective_runs_fused_and_plans_like_the_declared_one():
    """bench.py's ReferenceStyleReach (compute_cost over the getters, nothing declared): the planner traces it, recognises the shape of
    the in-line PANDA_REACH kind - the headline kernel, not the interpreter - validates it on the first command and returns the
    actions of the PandaReachObjective that declares its fused spec"""
    from bench import ReferenceStyleReach
    from mppiisaac.objectives import PandaReachObjective
    a, b = _planner(ReferenceStyleReach()), _planner(PandaReachObjective(None))
    assert a.mppi._fused_cost is not None and a.mppi._fused_cost.kind == capi.COST_PANDA_REACH
    assert bytes(a.mppi._fused_cost) == bytes(b.mppi._fused_cost)
    for _ in range(3):
        ua, ub = a.compute_action(Q0, [0.0] * 7), b.compute_action(Q0, [0.0] * 7)
        np.testing.assert_array_equal(ua.numpy(), ub.numpy())
    assert a.mppi._trace_guard[2] == 3 and a.mppi._fused_cost is not None      # three commands, the first one validated
    a.update_weights({"robot_to_goal": 2.0, "robot_ori": 0.25})                # new weights: a new trace, validated again
    assert a.mppi._fused_cost.w[0] == 2.0 and a.mppi._fused_cost.w[1] == 0.25
    a.compute_action(Q0, [0.0] * 7)
    assert a.mppi._trace_guard[2] == 1
    for p in (a, b):
        p.sim.stop_sim()


def test_an_untraceable_objective_runs_in_generic_mode_and_says_why(caplog):
    class Saturating(object):
        weights = {}

        def reset(self):
            pass

        def compute_cost(self, sim):
            ee = sim.get_actor_link_by_name("panda", "panda_ee_tip")
            goal = sim.get_actor_position_by_name("goal")
            return torch.tanh(torch.linalg.norm(ee[:, 0:3] - goal[:, 0:3], axis=1))
    with caplog.at_level(logging.WARNING, logger="mppiisaac"):
        p = _planner(Saturating())
        u = p.compute_action(Q0, [0.0] * 7)
    assert p.mppi._fused_cost is None and np.isfinite(u.numpy()).all()
    said = [r.getMessage() for r in caplog.records if "generic mode" in r.getMessage()]
    assert len(said) == 1 and "torch function 'tanh'" in said[0]
    p.sim.stop_sim()


def test_a_traced_objective_that_drifts_is_caught_and_dropped(caplog):
    """an Objective whose Python side changes behind the tracer's back (an attribute that is not `.weights`): the re-validation -
    every TRACE_RECHECK-th command - costs the same rollouts with the Objective itself, finds the program apart, and the planner
    goes on in generic mode (this command included)"""
    class Drifting(object):
        weights = {"d": 1.0}
        gain = 1.0

        def reset(self):
            pass

        def compute_cost(self, sim):
            ee = sim.get_actor_link_by_name("panda", "panda_ee_tip")
            goal = sim.get_actor_position_by_name("goal")
            return self.gain * self.weights["d"] * torch.linalg.norm(ee[:, 0:3] - goal[:, 0:3], axis=1)
    obj = Drifting()
    p = _planner(obj)
    p.mppi.TRACE_RECHECK = 4
    for _ in range(3):
        p.compute_action(Q0, [0.0] * 7)
    assert p.mppi._fused_cost is not None
    obj.gain = 3.0
    with caplog.at_level(logging.WARNING, logger="mppiisaac"):
        p.compute_action(Q0, [0.0] * 7)     # command 4: re-validated
        p.compute_action(Q0, [0.0] * 7)
    assert p.mppi._fused_cost is None
    assert any("disagrees with the Objective" in r.getMessage() for r in caplog.records)
    assert np.isfinite(p.mppi.get_costs().numpy()).all()
    p.sim.stop_sim()

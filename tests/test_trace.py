"""mppiisaac/trace.py: reference-style Python Objectives traced into cost programs."""

import numpy as np
import pytest
import torch

from reference_style_objectives import CASES
from test_golden_boundary import EXAMPLE_SCENES, ReplaySim, gold

from mppiisaac import trace
from mppiisaac.objectives import compile_program, evaluate


class _StubSim:
    """what trace_objective needs of a sim: the scene's name lookups"""
    device = "cpu"
    num_envs = 4
    env_cfg = None

    def __init__(self, scene):
        self.scene = scene


@pytest.fixture(scope="module")
def scenes():
    from scenes import build_scene
    cache = {}

    def get(case):
        if case not in cache:
            cache[case] = build_scene(EXAMPLE_SCENES[case], [[0.0, 0.0, 0.05]])
        return cache[case]
    return get


@pytest.mark.parametrize("case", sorted(CASES))
def test_reference_style_objectives_trace_to_programs_that_reproduce_the_reference(case, scenes, oracle64):
    """every example objective of the reference, restated as plain torch code over the getters: (1) the restatement itself equals the
    reference planner on the golden inputs, (2) its traced Term list, run by the torch evaluator, does too, (3) compiled into an
    MPPI_COST_PROGRAM and run by the oracle's interpreter - what the in-kernel interpreter is checked against - it does too"""
    g = gold("objective_costs.json")[case]
    obj = CASES[case](None)
    assert {k: float(v) for k, v in obj.weights.items()} == g["weights"]
    replay = ReplaySim(g["inputs"])
    np.testing.assert_allclose(obj.compute_cost(replay).numpy(), np.array(g["cost"]), rtol=1e-6, atol=1e-6)
    scene = scenes(case)
    terms = trace.trace_objective(obj, _StubSim(scene))
    assert all(isinstance(t.weight, float) for t in terms)
    np.testing.assert_allclose(evaluate(terms, {}, replay).numpy(), np.array(g["cost"]), rtol=1e-6, atol=1e-6)
    m = scene.to_c()
    spec = compile_program(terms, {}, scene)
    inp = {k: np.array(v) for k, v in g["inputs"].items()}
    got = []
    for k in range(len(g["cost"])):
        rb, root, cf = np.zeros((m.n_rb, 13)), np.zeros((m.n_actors, 13)), np.zeros((m.n_rb, 3))
        q, qd = np.zeros(16), np.zeros(16)
        for key, val in inp.items():
            kind, *names = key.split(":")
            if kind == "link":
                rb[scene.rigid_body_index(*names)] = val[k]
            elif kind == "contact":
                cf[scene.rigid_body_index(*names)] = val[k]
            elif kind == "dof_state":
                q[:scene.n_dof], qd[:scene.n_dof] = val[k][0::2][:scene.n_dof], val[k][1::2][:scene.n_dof]
            else:
                col = {"position": slice(0, 3), "orientation": slice(3, 7), "velocity": slice(7, 10)}[kind]
                root[scene.actor_index(names[0]), col] = val[k]
        got.append(oracle64.cost(m, spec, root, q, qd, rb, cf))
    np.testing.assert_allclose(got, g["cost"], rtol=1e-6, atol=1e-6)


def test_weights_are_read_at_trace_time_and_names_are_checked(scenes):
    obj = CASES["panda"](None)
    obj.weights["robot_ori"] = 0.0            # a zero weight drops the term
    terms = trace.trace_objective(obj, _StubSim(scenes("panda")))
    assert [t.op for t in terms] == ["dist"] and terms[0].weight == 1.0 and terms[0].args[2] == 3

    class Typo(CASES["panda"]):
        link = "panda_ee_tipp"
    with pytest.raises(trace.TraceError, match="panda_ee_tipp"):
        trace.trace_objective(Typo(None), _StubSim(scenes("panda")))


@pytest.mark.parametrize("body, reason", [
    ("return torch.exp(-torch.linalg.norm(ee[:, 0:3] - goal[:, 0:3], axis=1))", "torch function 'exp'"),
    ("return torch.linalg.norm(ee[:, 0:3] - goal[:, 0:3], axis=1) ** 3", "power"),
    ("return (ee[:, 0] - goal[:, 0])", "raw state component"),
    ("return torch.linalg.norm(ee[:, 0:3] - goal[:, 0:3], axis=1) + 1.5", "constant offset"),
    ("d = torch.linalg.norm(ee[:, 0:3] - goal[:, 0:3], axis=1)\n        return d if d[0] > 1 else 2 * d", "indexing a per-env scalar"),
    ("return torch.linalg.norm(ee[:, 0:3] - 2 * goal[:, 0:3], axis=1)", "norm of something other"),
    ("return torch.linalg.norm(sim._rigid_body_state[:, 3, 0:3], axis=1)", "sim._rigid_body_state"),
    ("return torch.linalg.norm(ee[:, 0:3] - goal[:, 0:3], axis=1).cpu()", "'.cpu'"),
])
def test_what_the_tracer_does_not_understand_is_refused_with_the_reason(body, reason, scenes):
    """a traced program must never be a guess: anything outside the grammar raises TraceError naming the construct (the planner then
    runs the Objective in generic mode and logs that line)"""
    src = ("import torch\nclass O:\n    weights = {}\n    def reset(self):\n        pass\n    def compute_cost(self, sim):\n"
           "        ee = sim.get_actor_link_by_name('panda', 'panda_ee_tip')\n        goal = sim.get_actor_position_by_name('goal')\n        " + body + "\n")
    ns = {}
    exec(src, ns)
    with pytest.raises(trace.TraceError) as e:
        trace.trace_objective(ns["O"](), _StubSim(scenes("panda")))
    assert reason in str(e.value), str(e.value)


def test_constants_and_operand_order_variants(scenes):
    """point - constant, constant-first differences, norms of swapped differences, keyword axes, dim=-1"""
    class O:
        weights = {}

        def reset(self):
            pass

        def compute_cost(self, sim):
            ee = sim.get_actor_link_by_name("panda", "panda_ee_tip")
            goal = sim.get_actor_position_by_name("goal")
            a = torch.norm(goal[:, :2] - ee[:, :2], dim=-1)                       # swapped: the same distance
            b = torch.linalg.norm(ee[:, 0:3] - torch.tensor([0.5, -0.4, 0.3]), dim=1)   # to a fixed point
            c = (ee[:, 0:3] - goal[:, 0:3]).norm(dim=1) * 2.0
            return 3.0 * a + b + c / 4
    terms = trace.trace_objective(O(), _StubSim(scenes("panda")))
    ee = ("link", "panda", "panda_ee_tip")
    got = sorted((t.op, t.weight, t.args) for t in terms)
    assert ("dist", 3.0, (("actor", "goal"), ee, 2)) in got or ("dist", 3.0, (ee, ("actor", "goal"), 2)) in got
    fixed = [t for t in terms if t.op == "dist" and t.weight == 1.0][0]
    assert fixed.args[0] == ee and fixed.args[2] == 3 and fixed.args[1] == pytest.approx((0.5, -0.4, 0.3), abs=1e-6)
    assert any(t.op == "dist" and t.weight == 0.5 and t.args[2] == 3 for t in terms)

"""The model blob against the URDFs, by an independent route (VERDICT r4 "what's weak" 1: oracle and kernels consume the SAME
mppi_model_t from the SAME packer, so a wrong inertia / axis / frame in urdf_compile.py or Scene.to_c would pass every parity test).

tests/golden/mass_matrices.json holds joint-space mass matrices M(q) of the ten compiled robots computed straight from the URDF
files by tools/make_mass_matrix_golden.py - its own XML walk, transform algebra and hull integration, and a different algorithm
(point Jacobians of the links' centres of mass; no spatial algebra, no body merging, no z-framing).  Here the oracle's
articulated-body algorithm on the PACKED model (assets/compiled/*.json -> Scene -> to_c) is inverted column by column at the same
joint positions: M^-1 e_i = qdd(tau = e_i) - qdd(tau = 0)  with qd = 0."""
import json
import os

import numpy as np
import pytest

from mppiisaac.planner.isaacgym_wrapper import ActorWrapper, IsaacGymConfig, Scene
from mppiisaac.utils.isaacgym_utils import load_asset
from oracle.oracle import Oracle

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "mass_matrices.json")))


@pytest.mark.parametrize("robot", GOLD["robots"], ids=lambda r: os.path.basename(r["urdf_file"]))
def test_oracle_aba_on_the_packed_model_inverts_the_urdfs_mass_matrix(robot):
    actor = ActorWrapper(type="robot", name="r", urdf_file=robot["urdf_file"], fixed=True, gravity=False, collision=False, dof_mode="effort")
    scene = Scene([actor], IsaacGymConfig(), load_asset(actor))
    model = scene.to_c()
    assert sorted(scene.dof_names) == sorted(robot["joints"])
    perm = [robot["joints"].index(n) for n in scene.dof_names]          # model DOF order -> fixture order
    o = Oracle("f64")
    n = scene.n_dof
    _, root = scene.initial_state()
    worst = 0.0
    for case in robot["cases"]:
        q = np.asarray(case["q"])[perm]
        M = np.asarray(case["M"])[np.ix_(perm, perm)]
        zero = o.forward_dynamics(model, root, q, np.zeros(n), np.zeros(n))
        Minv = np.stack([o.forward_dynamics(model, root, q, np.zeros(n), np.eye(n)[i]) - zero for i in range(n)], axis=1)
        # compare as M * Minv = 1 (scale-free: the wheel inertias of the bases are 1e-3, the arm's first joints 1e+0)
        err = np.abs(M @ Minv - np.eye(n)).max()
        worst = max(worst, err)
        assert np.allclose(Minv, Minv.T, rtol=1e-9, atol=1e-12 * np.abs(Minv).max())
    # the compiled fixtures keep 12 significant digits; hull integration by two different tetrahedralisations agrees to ~1e-10
    assert worst < 1e-7, worst

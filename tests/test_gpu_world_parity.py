"""The K = 1 WORLD of an example (reference examples/<name>/world.py: IsaacGymWrapper(num_envs=1), apply_robot_cmd + step from Python) against
the fp64 oracle, step by step along the example's own closed loop: every world step on the GPU - the quad-layout step kernel, the
light-body law with the gripper at its block included - is repeated by the oracle from the same state with the same command.  The
rollout kernels have their all-K parity tests (test_gpu_parity.py, test_gpu_state_parity.py); this is the other kernel a user's loop runs."""
import importlib.util
import logging
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("name,steps,contacts", [("panda_pick", 140, 10), ("boxer_push", 120, 10), ("panda_stick_push", 120, 0), ("omni_panda_pick", 120, 0),
                                                 ("heijn_push", 120, 10), ("anymal", 80, 10), ("multi_jackal", 80, 10), ("albert", 80, 0)])
def test_world_steps_follow_the_oracle_along_the_closed_loop(name, steps, contacts, oracle64):
    spec = importlib.util.spec_from_file_location("examples_run", os.path.join(ROOT, "mppi-isaac_amd", "examples", "run.py"))
    run = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(run)
    from mppiisaac.utils.transport import bytes_to_torch, torch_to_bytes
    logging.disable(logging.WARNING)
    try:
        cfg = run.config(name)
        planner = run.make_planner(name, cfg)
        sim = run.make_world(name, cfg)
        m = sim.scene.to_c()
        errs, contact_steps = [], 0
        for i in range(steps):
            dof = sim._dof_state[0].cpu().numpy().astype(float).copy()
            root = sim._root_state[0].cpu().numpy().astype(float).copy()
            action = bytes_to_torch(planner.compute_action_tensor(torch_to_bytes(sim._dof_state), torch_to_bytes(sim._root_state)))
            u = action.detach().cpu().numpy().astype(float).reshape(-1)
            sim.apply_robot_cmd(action.to(sim.device).reshape(1, -1))
            sim.step()
            ro, q, qd, cf = oracle64.scene_step(m, root, dof[0::2].copy(), dof[1::2].copy(), oracle64.cmd_map(m, u))   # (the command map: diff-drive wheels)
            dof1 = sim._dof_state[0].cpu().numpy().astype(float)
            root1 = sim._root_state[0].cpu().numpy().astype(float)
            assert np.isfinite(root1).all() and np.isfinite(dof1).all(), i
            errs.append(max(np.abs(ro[:, :3] - root1[:, :3]).max(), np.abs(q - dof1[0::2]).max()))
            contact_steps += int(np.abs(cf).max() > 1e-3)
        sim.stop_sim(); planner.sim.stop_sim()
    finally:
        logging.disable(logging.NOTSET)
    errs = np.array(errs)
    print(f"{name}: {steps} world steps vs the fp64 oracle from the same state and command: median {np.median(errs):.1e}, 95 % {np.percentile(errs, 95):.1e}, "
          f"max {errs.max():.1e} (positions [m] and joint positions [rad | m]); a contact force somewhere in {contact_steps} of them")
    assert contact_steps >= contacts                             # (the loop does meet its block / the ground)
    assert np.percentile(errs, 95) < 2e-4 and errs.max() < 1e-3     # measured: panda_pick 3.0e-6 / 3.6e-5, boxer_push 1.9e-5 / 4.0e-5

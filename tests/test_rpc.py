"""Planner RPC: the reference's two-process workflow - planner served, world as client -
over the in-tree transport (zerorpc itself is absent from the image)."""
import threading

import numpy as np
import pytest
import torch

from mppiisaac.utils import rpc
from mppiisaac.utils.transport import bytes_to_torch, torch_to_bytes


class FakePlanner:
    """the call surface world.py uses (reference examples/panda/world.py:36-48)"""

    def __init__(self):
        self.calls = 0

    def compute_action_tensor(self, dof_bytes, root_bytes):
        dof, root = bytes_to_torch(dof_bytes), bytes_to_torch(root_bytes)
        self.calls += 1
        return torch_to_bytes(dof[0, 0::2] + root[0, 0, 0])

    def get_rollouts(self):
        return torch_to_bytes(torch.arange(30 * 64 * 3, dtype=torch.float32).reshape(30, 64, 3))

    def big(self, n):
        return b"\x01" * n

    def fails(self):
        raise ValueError("no such actor")

    def add_to_env(self, cfgs):
        return len(cfgs)


def serve(obj):
    srv = rpc.Server(obj)
    url = srv.bind("tcp://127.0.0.1:0") if rpc.BACKEND != "zerorpc" else None
    t = threading.Thread(target=srv.run, daemon=True)
    t.start()
    return srv, url, t


@pytest.mark.skipif(rpc.BACKEND == "zerorpc", reason="real zerorpc present: its own tests apply")
def test_planner_calls_round_trip_over_tcp():
    srv, url, t = serve(FakePlanner())
    try:
        c = rpc.Client()
        c.connect(url)
        dof = torch.tensor([[0.1, 0.0, 0.2, 0.0, 0.3, 0.0]])
        root = torch.zeros((1, 2, 13))
        root[0, 0, 0] = 1.0
        out = bytes_to_torch(c.compute_action_tensor(torch_to_bytes(dof), torch_to_bytes(root)))
        np.testing.assert_allclose(out.numpy(), [1.1, 1.2, 1.3], rtol=1e-6)
        assert bytes_to_torch(c.get_rollouts()).shape == (30, 64, 3)
        assert c.add_to_env([{"type": "sphere", "name": "s0", "size": [0.1]}]) == 1     # plain python structures pass too
        assert len(c.big(8 << 20)) == 8 << 20                                            # multi-megabyte frames
        with pytest.raises(rpc.RemoteError, match="no such actor"):
            c.fails()
        with pytest.raises(rpc.RemoteError):
            c("_private")
        with pytest.raises(rpc.RemoteError):
            c.calls()                                                                    # attribute, not a method
        assert "compute_action_tensor" in c("_zerorpc_list")
        # the server survived the failures and serves a second client after the first one leaves
        c.close()
        c2 = rpc.Client(url)
        assert bytes_to_torch(c2.get_rollouts()).shape == (30, 64, 3)
        c2.close()
    finally:
        srv.close()
        t.join(timeout=2)
    assert not t.is_alive()


def test_endpoint_validation():
    if rpc.BACKEND == "zerorpc":
        pytest.skip("real zerorpc")
    with pytest.raises(ValueError):
        rpc.Server(FakePlanner()).bind("ipc:///tmp/x")
    with pytest.raises(RuntimeError):
        rpc.Client().anything()


@pytest.mark.gpu
def test_served_hip_planner_drives_a_world_client():
    """the reference workflow end to end: MPPIisaacPlanner behind the server, a K=1 world in the client, state and
    action travelling as torch.save blobs (examples/panda/planner.py:43-48, world.py:21-48)."""
    from mppiisaac.objectives import PandaReachObjective
    from mppiisaac.planner.isaacgym_wrapper import IsaacGymWrapper
    from mppiisaac.planner.mppi_isaac import MPPIisaacPlanner
    from mppiisaac.utils.config_store import load_config
    cfg = load_config({"defaults": [{"mppi": "panda"}, {"isaacgym": "normal"}], "actors": ["panda_stick", "goal"],
                       "initial_actor_positions": [[0.0, 0.0, 0.0]], "nx": 14},
                      overrides={"mppi.num_samples": 512, "mppi.horizon": 12})
    planner = MPPIisaacPlanner(cfg, PandaReachObjective(cfg))
    srv, url, t = serve(planner)
    try:
        world = IsaacGymWrapper(cfg.isaacgym, actors=cfg.actors, init_positions=cfg.initial_actor_positions, num_envs=1)
        world.set_actor_position_by_name([0.5, -0.4, 0.3], "goal")
        c = rpc.Client(url)
        ee = world.scene.rigid_body_index("panda", "panda_ee_tip")
        d0 = None
        for _ in range(60):
            a = bytes_to_torch(c.compute_action_tensor(torch_to_bytes(world._dof_state), torch_to_bytes(world._root_state)))
            world.apply_robot_cmd(a.to(world.device).reshape(1, -1))
            world.step()
            d = float(torch.linalg.norm(world._rigid_body_state[0, ee, 0:3] - torch.tensor([0.5, -0.4, 0.3], device=world.device)))
            d0 = d if d0 is None else d0
        assert d < 0.8 * d0                                  # the served planner steers the client's world to the goal
        assert bytes_to_torch(c.get_rollouts()).shape[1:] == (512, 3)
        c.close()
    finally:
        srv.close()
        t.join(timeout=2)

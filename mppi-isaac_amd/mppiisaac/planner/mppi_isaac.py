"""Planner facade with the reference's public surface (reference mppiisaac/planner/mppi_isaac.py:18-138):
`MPPIisaacPlanner(cfg, objective, prior=None)`, `compute_action`, `compute_action_tensor`, `command`,
`get_rollouts`, `update_objective`, `update_weights`, `update_mppi_params`, attributes `.sim`, `.mppi`,
`.cfg`.  The K rollout envs and the MPPI core both live in ONE HIP context (libmppi_hip.so); the
Isaac Gym + mppi_torch pair of the reference is what that library replaces."""
from typing import Callable, Optional

import torch

from mppiisaac.planner.isaacgym_wrapper import IsaacGymWrapper
from mppiisaac.planner.mppi import MPPIPlanner, make_config
from mppiisaac.utils.transport import bytes_to_array, torch_to_bytes


class MPPIisaacPlanner(object):
    def __init__(self, cfg, objective: Callable, prior: Optional[Callable] = None, *, shard: bool = False, process_group=None):
        self.cfg = cfg
        self.objective = objective
        self.done = False
        K_total = int(cfg.mppi.num_samples)
        k_off, k_loc = 0, K_total
        if shard:  # sample sharding over torch.distributed ranks
            import torch.distributed as dist
            world, rank = dist.get_world_size(process_group), dist.get_rank(process_group)
            if K_total % world:
                raise ValueError(f"num_samples={K_total} is not divisible by world_size={world}")
            k_loc = K_total // world
            k_off = rank * k_loc
        self._shard, self._pg = shard, process_group
        self._k_off, self._k_loc = k_off, k_loc
        self._build(prior)

    def _build(self, prior):
        cfg = self.cfg
        # the C config is built once the scene is known (viz link from the robot's `visualize_link`)
        probe = lambda scene: make_config(cfg.mppi, k_offset=self._k_off, k_local=self._k_loc,
                                          viz_link=scene.viz_link_index())
        self.sim = IsaacGymWrapper(cfg.isaacgym, actors=cfg.actors, init_positions=cfg.initial_actor_positions,
                                   num_envs=self._k_loc, device=cfg.mppi.device, mppi_config=probe)
        self.prior = (lambda state, t: prior.compute_command(self.sim)) if prior else None
        self.mppi = MPPIPlanner(cfg.mppi, cfg.nx, dynamics=self.dynamics, running_cost=self.running_cost,
                                prior=self.prior, sim=self.sim, shard=self._shard, process_group=self._pg)
        self._prior_obj = prior
        self.state_place_holder = torch.zeros((self._k_loc, self.cfg.nx))
        self._bind_objective()

    def _bind_objective(self):
        spec = getattr(self.objective, "fused_spec", None)
        if callable(spec):
            self.mppi.set_fused_cost(spec(self.sim))
            self.mppi.set_trace_guard(None)
            return
        # an Objective that declares nothing for this backend (the reference's contract: compute_cost / reset / weights): its
        # compute_cost is traced ONCE into a cost program (mppiisaac/trace.py) and runs inside the rollout kernel from then on,
        # validated against the eager Objective on the first command and every 64th (MPPIPlanner.command).  Untraceable, or caught
        # drifting: generic mode (reference mppi_isaac.py:57-69 on precomputed states), with the reason in the log, once.
        cost = self._traced_spec()
        self.mppi.set_fused_cost(cost)
        if getattr(self, "_trace_fresh", False):    # a new trace (another Objective, other weights): validated on its first command
            self._trace_fresh = False
            self.mppi.set_trace_guard(None)
        self.mppi.set_trace_guard(self._trace_failed if cost is not None else None)

    def _traced_spec(self):
        import os
        if os.environ.get("MPPI_TRACE_OBJECTIVE", "1") == "0" or (self._prior_obj is not None and self.cfg.mppi.use_priors):
            return None   # (a prior is evaluated at every rollout step on the stepped envs: generic mode)
        w = getattr(self.objective, "weights", None)
        key = (id(self.objective), tuple(sorted((str(k), float(v)) for k, v in w.items())) if isinstance(w, dict) else None, self.sim.generation)
        cached = getattr(self, "_trace_cache", None)
        if cached is not None and cached[0] == key:
            return cached[1]
        from mppiisaac import trace
        from mppiisaac.objectives import specialise_program
        cost = None
        if getattr(self, "_trace_dropped", None) == id(self.objective):
            pass   # (its traced program disagreed with it once: it stays in generic mode)
        else:
            try:
                self.traced_terms = trace.trace_objective(self.objective, self.sim)
                cost = specialise_program(self.traced_terms, {}, self.sim.scene)
            except (trace.TraceError, ValueError) as e:
                self.traced_terms = None
                if getattr(self, "_trace_said", None) != (id(self.objective), str(e)):
                    self._trace_said = (id(self.objective), str(e))
                    import logging
                    logging.getLogger("mppiisaac").warning("%s.compute_cost runs in generic mode (a Python call on the states of the whole "
                                                           "horizon per command) - not traceable into a cost program: %s", type(self.objective).__name__, e)
        self._trace_cache = (key, cost)
        self._trace_fresh = True
        return cost

    def _trace_failed(self, detail: str):
        """MPPIPlanner.command found the traced program and the eager Objective apart: generic mode from now on"""
        import logging
        logging.getLogger("mppiisaac").warning("%s: the cost program traced from compute_cost disagrees with the Objective (%s) - it depends on "
                                               "more than the sim's getters and its weights; generic mode from here on", type(self.objective).__name__, detail)
        self._trace_dropped = id(self.objective)
        self._trace_cache = None
        self.traced_terms = None

    def update_objective(self, objective):
        self.objective = objective
        self._bind_objective()

    # callbacks with the reference's signatures (mppi_isaac.py:57-69); used by the generic mode
    def dynamics(self, _, u, t=None):
        self.sim.apply_robot_cmd(u)
        self.sim.step()
        return (self.state_place_holder, u)

    def running_cost(self, _):
        return self.objective.compute_cost(self.sim)

    def _rebind_after_restart(self):
        """the simulator was rebuilt with a new actor list (new HIP context; IsaacGymWrapper._restart carried the
        nominal control sequence over): rebuild the MPPI driver on it.  The old context is gone - its handle must
        not be used again (in the reference the mppi object simply survives the restart)."""
        self.mppi = MPPIPlanner(self.cfg.mppi, self.cfg.nx, dynamics=self.dynamics, running_cost=self.running_cost,
                                prior=self.prior, sim=self.sim, shard=self._shard, process_group=self._pg)
        self._generation = self.sim.generation

    def compute_action(self, q, qdot, obst=None, obst_tensor=None):
        self.sim.reset_root_state()
        self.sim.reset_robot_state(q, qdot)
        # two ways of placing obstacles, as in the reference (mppi_isaac.py:75-81)
        if obst:
            self.sim.update_root_state_tensor_by_obstacles(obst)
        if obst_tensor is not None:
            self.sim.update_root_state_tensor_by_obstacles_tensor(obst_tensor)
        if self.sim.generation != getattr(self, "_generation", 0):
            self._rebind_after_restart()
        self.sim.save_root_state()
        self._bind_objective()
        return self.mppi.command(self.state_place_holder).cpu()

    def reset_rollout_sim(self, dof_state_tensor, root_state_tensor, rigid_body_state_tensor=None):
        self.sim.visualize_link_buffer = []
        # (only the numbers are needed: the payloads are viewed in place, nothing is restored onto the device the blob names)
        self.sim.set_state_from_env0(bytes_to_array(dof_state_tensor), bytes_to_array(root_state_tensor))

    def compute_action_tensor(self, dof_state_tensor, root_state_tensor):
        self.objective.reset()
        self.reset_rollout_sim(dof_state_tensor, root_state_tensor)
        return self.command()

    def command(self):
        self._bind_objective()
        return torch_to_bytes(self.mppi.command(self.state_place_holder))

    def add_to_env(self, env_cfg_additions):
        self.sim.add_to_envs(env_cfg_additions)
        self._rebind_after_restart()

    def get_rollouts(self):
        if not self.sim._visualize_link_present:
            return torch_to_bytes(torch.zeros((1, 1, 1)))
        if self.mppi._fused_cost is not None:
            return torch_to_bytes(self.mppi.get_rollouts())
        # entries: [K, 3] per simulator step (reference isaacgym_wrapper.py:651-652) or one [H, K, 3] block per simulated horizon
        buf = self.sim.visualize_link_buffer
        return torch_to_bytes(torch.cat([b if b.dim() == 3 else b.unsqueeze(0) for b in buf]))

    def update_weights(self, weights):
        self.objective.weights = weights
        self._bind_objective()

    def update_mppi_params(self, params):
        """reference mppi_isaac.py:126-138: a new MPPI core with the new noise covariance on the EXISTING simulator - the
        actor list (incl. actors added through add_to_env / obstacle updates), the current world state, the saved root
        state and the nominal control sequence survive; only the HIP context is rebuilt around the new config."""
        self.cfg.mppi.noise_sigma = params["noise_sigma"]
        self.sim._restart()
        self._rebind_after_restart()
        self._bind_objective()

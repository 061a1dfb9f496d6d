"""mppiisaac - host-side mirror of the reference package of the same name
(tud-airlab/mppi-isaac), backed by the MI355X-native HIP rollout library instead of
Isaac Gym + mppi_torch.  Import surface kept: mppiisaac.planner.mppi_isaac.MPPIisaacPlanner,
mppiisaac.planner.isaacgym_wrapper.{IsaacGymWrapper,ActorWrapper,IsaacGymConfig},
mppiisaac.utils.{config_store,conversions,transport,isaacgym_utils}."""
__version__ = "0.1.0"

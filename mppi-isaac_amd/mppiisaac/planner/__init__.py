"""Host-side mirror of the reference planner interface: MPPIisaacPlanner, MPPIPlanner, IsaacGymWrapper."""

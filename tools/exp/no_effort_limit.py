"""A/B experiment: bench.py with the joint effort limits removed (no clamp -> never a second ABA solve).
Tells how much of the rollout time is the effort-clamp re-solve.  Not a valid benchmark configuration."""
import os, sys
ROOT = os.path.join(os.path.dirname(__file__), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mppi-isaac_amd"))
from mppiisaac.planner import isaacgym_wrapper as W
orig = W.Scene.to_c
def to_c(self):
    m = orig(self)
    for i in range(m.n_bodies):
        m.bodies[i].effort = 0.0
    return m
W.Scene.to_c = to_c
import bench
bench.main()

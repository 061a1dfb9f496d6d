"""The reference-API loop with a reference-style Python Objective (bench.py ReferenceStyleReach), N iterations - to be run under
rocprofv3 --kernel-trace --stats: which kernels a generic-mode control iteration consists of, and how long each takes."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mppi-isaac_amd"))
import bench
obj = bench.ReferenceStyleReachGraphSafe() if os.environ.get("GRAPH_SAFE") else bench.ReferenceStyleReach()
hz, ms, dist = bench.facade_loop("panda_reach", obj, "cuda:0", int(os.environ.get("N", "300")), 30)
print(f"{type(obj).__name__}: {hz:.1f} Hz, {ms * 1e3:.1f} us / iteration, final distance {dist:.3f}")

"""Closed loop of the pushing example; per control iteration the clearance of every wheel/caster-box candidate pair and of
chassis-block, written to gpurun_out/push_clearance.json.  (Experiment behind DESIGN.md's disc-box table.)"""
import importlib.util
import json
import os
import sys

import numpy as np

root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(root, "tests"))
sys.path.insert(0, os.path.join(root, "mppi-isaac_amd"))
spec = importlib.util.spec_from_file_location("examples_run", os.path.join(root, "mppi-isaac_amd", "examples", "run.py"))
run = importlib.util.module_from_spec(spec)
spec.loader.exec_module(run)
from test_gpu_sampler_shards import _disc_box_clearance  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "boxer_push"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
cfg = run.config(name, filter_u=False)
planner = run.make_planner(name, cfg)
sc = planner.sim.scene
rows = []


def hook(i, sim):
    rb = sim._rigid_body_state[0].cpu().numpy().astype(np.float64)
    row = {"i": i, "chassis": rb[sc.all_shapes[0]["rb"]][:7].tolist(), "block": rb[sc.all_shapes[5]["rb"]][:7].tolist()}
    for (i0, i1), names in zip(sc.dropped_pair_shapes, sc.dropped_pairs):
        row["/".join(names)] = _disc_box_clearance(sc.all_shapes[i0], sc.all_shapes[i1], rb)
    rows.append(row)


print(run.run_world(name, cfg, planner, steps, report=False, hook=hook))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(rows, open("gpurun_out/push_clearance.json", "w"))

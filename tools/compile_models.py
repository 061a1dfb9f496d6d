#!/usr/bin/env python3
"""Compile the in-scope URDFs of the reference asset tree into the JSON model fixtures
under mppi-isaac_amd/assets/compiled/.  Runs only where /root/reference exists (this
container); the GPU box uses the committed JSON.  Usage: python tools/compile_models.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mppi-isaac_amd"))
from mppiisaac.backend.urdf_compile import compile_urdf, prune_links, save_model

REF = os.environ.get("MPPI_REFERENCE", "/root/reference")
# urdf_file value used in conf/actors/*.yaml  ->  fixture name
URDFS = {
    "point_robot.urdf": "point_robot",
    "panda_isaac/robots/franka_panda_stick.urdf": "franka_panda_stick",
    "panda_isaac/robots/franka_panda_gripper.urdf": "franka_panda_gripper",
    "panda_isaac/robots/franka_panda.urdf": "franka_panda",
    "boxer/boxer.urdf": "boxer",
    "heijn/heijn.urdf": "heijn",
    "jackal/jackal.urdf": "jackal",
    "albert/albert.urdf": "albert",
    "omni_panda/omniPandaWithGripper.urdf": "omni_panda_gripper",
    "anymal_c/urdf/anymal.urdf": "anymal",
}
# models whose URDF has more links than MPPI_MAX_LINKS: only links with collision geometry and the ones the example objectives
# name stay reported rigid bodies (reference examples/anymal/planner.py:24-41)
KEEP_LINKS = {"anymal": ("base", "face_front", "face_rear", "LF_KFE", "LH_KFE", "RH_KFE", "RF_KFE")}
out_dir = os.path.join(ROOT, "mppi-isaac_amd", "assets", "compiled")
os.makedirs(out_dir, exist_ok=True)
for rel, name in URDFS.items():
    path = os.path.join(REF, "assets", "urdf", rel)
    if not os.path.exists(path):
        print("skip (missing)", rel); continue
    m = compile_urdf(path, name=name)
    if name in KEEP_LINKS:
        m = prune_links(m, KEEP_LINKS[name])
    m["urdf_file"] = rel
    save_model(m, os.path.join(out_dir, name + ".json"))
    print(f"{name}: {len(m['links'])} links, {len(m['bodies'])} dof;",
          "masses", [round(b['inertia']['mass'], 3) for b in m['bodies']], "base", round(m['base']['inertia']['mass'], 3))

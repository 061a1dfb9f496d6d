#!/usr/bin/env python3
"""Restate the reference's actor descriptions (conf/actors/*.yaml) in this repo's own conf tree.
Input: tests/golden/actor_cfgs.json (ActorWrapper field dicts captured by tools/make_golden.py);
output: mppi-isaac_amd/conf/actors/<name>.yaml holding only the non-default fields."""
import json, os, sys
import yaml
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mppi-isaac_amd"))
from mppiisaac.planner.isaacgym_wrapper import ActorWrapper
import dataclasses

gold = json.load(open(os.path.join(ROOT, "tests", "golden", "actor_cfgs.json")))
out = os.path.join(ROOT, "mppi-isaac_amd", "conf", "actors")
os.makedirs(out, exist_ok=True)
for fname, fields in gold.items():
    dflt = dataclasses.asdict(ActorWrapper(type=fields["type"], name=fields["name"]))
    keep = {"type": fields["type"], "name": fields["name"]}
    for k, v in fields.items():
        if k in ("type", "name"):
            continue
        if v != dflt[k]:
            keep[k] = v
    with open(os.path.join(out, fname + ".yaml"), "w") as f:
        f.write(f"# actor '{fname}': restated from the reference's conf/actors/{fname}.yaml (non-default ActorWrapper fields)\n")
        yaml.safe_dump(keep, f, default_flow_style=None, sort_keys=False, width=120)
print("wrote", len(gold), "actor files")

# ---- MPPI parameter files: conf/mppi/<name>.yaml from tests/golden/mppi_cfgs.json (values of the reference's files)
mg = json.load(open(os.path.join(ROOT, "tests", "golden", "mppi_cfgs.json")))
mout = os.path.join(ROOT, "mppi-isaac_amd", "conf", "mppi")
os.makedirs(mout, exist_ok=True)
for name, vals in mg.items():
    note = ""
    if vals.get("mppi_mode", "halton-spline") == "halton-spline" and int(vals.get("horizon", 30)) < 12:
        note = "; horizon < 12: fewer than 3 spline knots, every step is sampled directly"
    with open(os.path.join(mout, name + ".yaml"), "w") as f:
        f.write(f"# MPPI parameters '{name}' (values: reference conf/mppi/{name}.yaml{note})\n")
        f.write("defaults: [base_mppi]\n")
        yaml.safe_dump(vals, f, default_flow_style=None, sort_keys=True, width=120)
print("wrote", len(mg), "mppi files")

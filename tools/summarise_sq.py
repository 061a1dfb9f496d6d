#!/usr/bin/env python3
"""Condense the two SQ counter passes of tools/pmc_sq.sh (gpurun_out/prof_<tag>/pmc_sq, pmc_sq2) into
gpurun_out/prof_<tag>/sq_summary.json and - when run in the repo with profiles/ - into profiles/<tag>_sq_summary.json +
the index profiles/sq_latest.json that bench.py's `roofline.issue` block reads.

Per kernel: launches, mean of every counter per launch, and the per-wave figures the issue roofline uses
(SQ_INSTS_* / SQ_WAVES), issue_cycles_frac = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES (share of a wave's resident cycles in
which it issues), wait_cycles_frac = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES.  Counters are summed over all SEs / XCDs by
rocprofv3; SQ_WAVE_CYCLES and friends count in units of 4 cycles on gfx950 (the quad-cycle the SQ arbitrates in), which
cancels in the fractions."""
import collections
import csv
import glob
import json
import os
import re
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "sq"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")


def short(n):
    n = re.sub(r"\(anonymous namespace\)::|void ", "", n)
    return n.split("<")[0].split("(")[0]


agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("pmc_sq", "pmc_sq2"):
    fs = glob.glob(os.path.join(src, d, "*", "*_counter_collection.csv"))
    if not fs:
        print("no counter csv under", os.path.join(src, d))
        continue
    for r in csv.DictReader(open(fs[0])):
        k = short(r["Kernel_Name"])
        if not (k.startswith("k_rollout") or k.startswith("k_combine") or k.startswith("k_sim_step") or k.startswith("k_sample")):
            continue
        if k.startswith("k_sim_step") and r.get("Grid_Size") in ("64",):
            k += "/K1"
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, v in agg.items():
    e = {"launches": max(len(x) for x in v.values())}
    for c, x in v.items():
        e[c] = sum(x) / len(x)
    w = e.get("SQ_WAVES")
    if w:
        for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_LDS_BANK_CONFLICT"):
            if c in e:
                e[c + "_per_wave"] = e[c] / w
    wc = e.get("SQ_WAVE_CYCLES")
    if wc:
        for c, name in (("SQ_ACTIVE_INST_ANY", "issue_cycles_frac"), ("SQ_WAIT_INST_ANY", "wait_cycles_frac"), ("SQ_WAIT_ANY", "wait_any_frac")):
            if c in e:
                e[name] = e[c] / wc
    out[k] = e
bench = {}
bp = os.path.join(src, "pmc_sq_bench.json")
if os.path.exists(bp):
    try:
        bench = json.loads(open(bp).read().strip().splitlines()[-1])
    except Exception:  # noqa: BLE001
        bench = {}
summary = {"tag": tag, "workload": bench.get("config", {}).get("workload"), "K": bench.get("config", {}).get("K_per_gpu"),
           "H": bench.get("config", {}).get("H"), "bench_under_rocprof_hz": bench.get("value"), "kernels": out}
json.dump(summary, open(os.path.join(src, "sq_summary.json"), "w"), indent=1, sort_keys=True)
dst = os.path.join(ROOT, "profiles")
if os.path.isdir(dst) and summary["workload"]:
    json.dump(summary, open(os.path.join(dst, f"{tag}_sq_summary.json"), "w"), indent=1, sort_keys=True)
    main = next((out[k] for k in ("k_rollout_quad", "k_rollout_scene_quad", "k_rollout", "k_rollout_scene") if k in out), None)
    lp = os.path.join(dst, "sq_latest.json")
    latest = json.load(open(lp)) if os.path.exists(lp) else {"by_workload": {}}
    latest["by_workload"][summary["workload"].split(" ")[0]] = {"tag": tag, "K": summary["K"], "H": summary["H"], "k_rollout": main}
    json.dump(latest, open(lp, "w"), indent=1, sort_keys=True)
for k, e in out.items():
    print(k, {c: (round(x, 3) if isinstance(x, float) else x) for c, x in e.items() if "per_wave" in c or "frac" in c or c in ("launches", "SQ_WAVES")})

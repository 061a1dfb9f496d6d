#!/usr/bin/env python3
"""The measurement table of DESIGN.md section 6 from ONE tag of profiles/ (the files one `tools/gpu_r6.sh <tag> ... prof` pass leaves):
    python tools/design_cards.py r06m            -> markdown on stdout
Per workload: closed-loop rate and per-iteration time (bench line), rollout kernel by hipEvents inside the timed loop and by
rocprofv3 --kernel-trace --stats, HBM traffic per launch from the PMC passes against the algorithmic bytes, SQ counters per
wavefront, and the facade rows of the headline bench line.  Every figure names the file it comes from; nothing is typed by hand."""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
tag = sys.argv[1]


def load(name):
    path = os.path.join(P, name)
    if not os.path.exists(path):
        return None
    with open(path) as f:
        txt = f.read().strip()
    if not name.endswith(".json"):
        return txt
    try:
        return json.loads(txt)                      # (a pretty-printed summary)
    except json.JSONDecodeError:
        return json.loads(txt.splitlines()[-1])     # (a bench log: the result line is the last one)


def rocprof_avg(name, needle):
    path = os.path.join(P, name)
    if not os.path.exists(path):
        return None
    with open(path) as f:
        for row in csv.DictReader(f):
            if needle in row["Name"]:
                return float(row["AverageNs"]) / 1e6, int(row["Calls"])
    return None


rows = [("panda reach 4096 x 20 (headline)", "bench.json", "", "k_rollout_quad"), ("point reach 1024 x 15", "bench_point_reach.json", None, None),
        ("pushing scene 8192 x 25", "bench_boxer_push.json", "_boxer", "k_rollout_scene_quad"), ("gripper scene 8192 x 30", "bench_panda_pick.json", "_pick", "k_rollout_scene_quad"),
        ("gripper scene 65536 x 30, one GPU", "bench_panda_pick_65536.json", None, None)]
print(f"| workload (`profiles/{tag}_*`) | closed loop | rollout kernel, hipEvents in the timed loop | rocprofv3 average (launches) | HBM per launch, PMC / algorithmic | SQ per wavefront: VALU / SALU / LDS, issuing |")
print("|---|---|---|---|---|---|")
for title, bench, sfx, kern in rows:
    b = load(f"{tag}_{bench}")
    if b is None:
        continue
    hz, ms, roll = b["value"], b["ms_per_step"], b["kernels_ms"]["k_rollout(+record tail)"]
    rp = rocprof_avg(f"{tag}{sfx}_kernel_stats.csv", kern) if sfx is not None else None
    pmc = load(f"{tag}{sfx}_pmc_summary.json") if sfx is not None else None
    sq = load(f"{tag}{sfx}_sq_summary.json") if sfx is not None else None
    traffic = "-"
    if pmc:
        k = next((v for n, v in pmc.items() if isinstance(v, dict) and "rollout" in n), {})
        t = k.get("hbm_traffic_bytes_per_launch")
        if t:
            traffic = f"{t / 1e6:.2f} MB / {b['roofline'].get('bytes_alg_per_launch', 0) / 1e6:.2f} MB"
    sqs = "-"
    if sq:
        kk = next((v for n, v in sq.get("kernels", {}).items() if "rollout" in n), None)
        if kk:
            act = kk.get("issue_cycles_frac")
            sqs = f"{kk.get('SQ_INSTS_VALU_per_wave', 0) / 1e3:.1f} k / {kk.get('SQ_INSTS_SALU_per_wave', 0) / 1e3:.1f} k / {kk.get('SQ_INSTS_LDS_per_wave', 0) / 1e3:.1f} k" + (f", {100 * act:.0f} %" if act else "")
    print(f"| {title} | **{hz:.0f} Hz**, {ms:.4f} ms | {roll:.4f} ms | " + (f"{rp[0]:.4f} ms ({rp[1]})" if rp else "-") + f" | {traffic} | {sqs} |")
b = load(f"{tag}_bench.json")
if b:
    print()
    print(f"Facade rows of the headline line (`profiles/{tag}_bench.json`): `value_facade` {b.get('value_facade', 0):.0f} Hz, `value_generic_objective` {b.get('value_generic_objective', 0):.0f} Hz"
          + (f", `value_generic_objective_untraced` {b['value_generic_objective_untraced']:.0f} Hz" if b.get("value_generic_objective_untraced") else "")
          + f"; roofline object: achieved {b['roofline']['achieved']:.1f} GB/s algorithmic of {b['roofline']['peak']:.0f} (frac {b['roofline']['frac']:.4f}), issue_frac {b['roofline'].get('issue_frac', 0):.3f};"
          f" cpu_baseline {b['cpu_baseline']['value']:.1f} {b['cpu_baseline']['unit']} on {b['cpu_baseline']['cores']} threads ({b['cpu_baseline']['kind']}).")

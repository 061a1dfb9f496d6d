#!/usr/bin/env python3
"""Condense a gpurun_out/prof_<tag>/ directory (tools/profile_bench.sh) into profiles/<tag>_*:
the rocprofv3 kernel_stats table and the per-kernel FETCH_SIZE / WRITE_SIZE means.
HBM traffic per launch = 2*FETCH_SIZE + WRITE_SIZE: on gfx950 rocprofv3's FETCH_SIZE counts 64 B per
128-B request (MI355X_MICROARCH.md, HBM section); calibrated here on k_reduce, whose only sizeable read
is the du buffer (K*H*nu*4 B) and which reports 0.507x of it.  WRITE_SIZE matches k_sample's eps write exactly."""
import collections, csv, glob, json, os, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)
stats = glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))[0]
shutil.copy(stats, os.path.join(dst, f"{tag}_kernel_stats.csv"))
shutil.copy(os.path.join(src, "trace_bench.json"), os.path.join(dst, f"{tag}_bench_under_rocprof.json"))
def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("<")[0].split("(")[0]
out = collections.defaultdict(dict)
_bench = json.load(open(os.path.join(src, "trace_bench.json")))
workload = _bench["config"]["workload"].split(" ")[0]
if not glob.glob(os.path.join(src, "pmc_fetch", "*", "*_counter_collection.csv")):  # STATS_ONLY run
    print(open(os.path.join(dst, f"{tag}_kernel_stats.csv")).read()[:1500])
    sys.exit(0)
for name, key in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    f = glob.glob(os.path.join(src, name, "*", "*_counter_collection.csv"))[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        out[k][key + "_KB_mean"] = sum(v) / len(v)
        out[k]["launches"] = len(v)
for k, v in out.items():
    if "FETCH_SIZE_KB_mean" in v and "WRITE_SIZE_KB_mean" in v:
        v["hbm_traffic_bytes_per_launch"] = int(1024 * (2 * v["FETCH_SIZE_KB_mean"] + v["WRITE_SIZE_KB_mean"]))
json.dump(out, open(os.path.join(dst, f"{tag}_pmc_summary.json"), "w"), indent=1, sort_keys=True)
main = out.get("k_rollout_quad") or out.get("k_rollout_scene_quad") or out.get("k_rollout", {})
latest_path = os.path.join(dst, "pmc_latest.json")
latest = json.load(open(latest_path)) if os.path.exists(latest_path) else {}
if "by_workload" not in latest:
    latest = {"by_workload": {}}
latest["by_workload"][workload] = {"tag": tag, "K": _bench["config"].get("K_per_gpu"), "k_rollout": main}   # bench.py reads the entry of ITS workload only
json.dump(latest, open(latest_path, "w"), indent=1, sort_keys=True)
print(open(os.path.join(dst, f"{tag}_kernel_stats.csv")).read()[:1200])
print(json.dumps(main, indent=1))

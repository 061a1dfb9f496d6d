#!/bin/bash
# SQ issue/wait breakdown of the bench kernels (one --pmc pass; gpurun allows --pmc with --kernel-trace only)
set -u
TAG=${1:-sq}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
WORKLOAD=${WORKLOAD:-panda_reach}
STEPS=${STEPS:-100}
CMD="python $REPO/bench.py --workload $WORKLOAD --steps $STEPS --warmup 10 --no-cpu-baseline"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_sq -- $CMD > $OUT/pmc_sq_bench.json 2> $OUT/pmc_sq.err
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_sq2 -- $CMD > $OUT/pmc_sq2_bench.json 2> $OUT/pmc_sq2.err
python - <<PY
import csv, glob, collections
for d in ("pmc_sq", "pmc_sq2"):
    fs = glob.glob("$OUT/%s/*/*_counter_collection.csv" % d)
    if not fs: print("no csv for", d); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"]
        if "k_rollout" in k or "k_combine" in k or "k_sim_step" in k:
            import re
            name = re.sub(r"\(anonymous namespace\)::|void ", "", k).split("<")[0].split("(")[0] + ("/K1" if r["Grid_Size"] in ("64",) else "")
            agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        print(k, {c: round(sum(x)/len(x), 1) for c, x in v.items()})
PY
tail -3 $OUT/pmc_sq.err

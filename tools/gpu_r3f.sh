#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03f
mkdir -p $OUT
cd $REPO
timeout 1500 python -m pytest tests -m gpu -q -x -s > $OUT/gpu_tests.log 2>&1; echo "pytest rc=$?" >> $OUT/gpu_tests.log
tail -4 $OUT/gpu_tests.log
grep -h "vs fp64 oracle\|vs oracle on\|shared-lane kernel vs\|pushing scene, block\|randomised actors" $OUT/gpu_tests.log | head -30
timeout 300 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -2 $OUT/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r03f/bench.json"))
print("value",d["value"],"shipped",d["value_shipped_conf"],"ms",d["ms_per_step"],"rollout",d["kernels_ms"], "cpu", d["cpu_baseline"]["value"])
PY

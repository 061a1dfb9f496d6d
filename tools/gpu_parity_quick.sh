#!/bin/bash
# quick GPU check of a contact-law change: parity tests of the contact scenes + the two contact benches.   tools/gpu_parity_quick.sh <tag>
TAG=${1:-quick}
mkdir -p gpurun_out/$TAG
python -m pytest tests/test_gpu_parity.py tests/test_gpu_state_parity.py tests/test_gpu_sampler_shards.py -m gpu -q -s 2>&1 | grep -v "^$" > gpurun_out/$TAG/parity.log
grep "vs fp64\|passed\|failed\|Error\|assert " gpurun_out/$TAG/parity.log | cut -c1-330 | head -60
for w in boxer_push panda_pick; do python bench.py --no-cpu-baseline --workload $w > gpurun_out/$TAG/bench_$w.json; done
python - <<PY
import json
for w in ("boxer_push","panda_pick"):
    d=json.loads(open("gpurun_out/$TAG/bench_%s.json"%w).read().strip().splitlines()[-1]); print(w, d["value"], d["kernels_ms"], d["config"].get("task_outcome"))
PY

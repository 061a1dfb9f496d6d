#!/bin/bash
# round 3: A/B of the LDS-parking / opaque-pick build against the committed base, then kernel stats + PMC traffic + SQ counters
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03c
mkdir -p $OUT
cd $REPO
C=mppi-isaac_amd/csrc
python tools/exp/ab_time.py panda_reach,boxer_push,panda_pick $C/libmppi_hip_base.so $C/libmppi_hip.so 2>&1 | grep -v "contact model\|amdgpu.ids" | tee $OUT/ab_base_vs_product.txt
timeout 600 python -m pytest tests -m gpu -q -x > $OUT/gpu_tests.log 2>&1; tail -2 $OUT/gpu_tests.log
WORKLOAD=boxer_push STEPS=100 bash tools/profile_bench.sh r03c_boxer > $OUT/prof_boxer.log 2>&1
WORKLOAD=panda_pick STEPS=60 bash tools/profile_bench.sh r03c_pick > $OUT/prof_pick.log 2>&1
WORKLOAD=boxer_push STEPS=60 bash tools/pmc_sq.sh r03c_boxer > $OUT/sq_boxer.log 2>&1
WORKLOAD=panda_pick STEPS=40 bash tools/pmc_sq.sh r03c_pick > $OUT/sq_pick.log 2>&1
python tools/summarise_profile.py r03c_boxer 2>&1 | tail -5
python tools/summarise_profile.py r03c_pick 2>&1 | tail -5
cat gpurun_out/prof_r03c_boxer/sq_summary.json 2>/dev/null | head -40

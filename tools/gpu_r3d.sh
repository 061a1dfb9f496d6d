#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03d
mkdir -p $OUT
cd $REPO
C=mppi-isaac_amd/csrc
python tools/exp/ab_time.py boxer_push,panda_pick $C/libmppi_hip_base.so $C/libmppi_hip.so 2>&1 | grep -v "contact model\|amdgpu.ids" | tee $OUT/ab_base_vs_product.txt
timeout 600 python -m pytest tests -m gpu -q -x > $OUT/gpu_tests.log 2>&1; tail -2 $OUT/gpu_tests.log
WORKLOAD=boxer_push STEPS=100 bash tools/profile_bench.sh r03d_boxer > $OUT/prof_boxer.log 2>&1
python tools/summarise_profile.py r03d_boxer 2>&1 | tail -6

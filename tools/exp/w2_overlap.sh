#!/bin/bash
# Experiment: how much do two resident wavefronts per SIMD overlap in the contact-scene rollout?  The product build
# needs > 256 registers (one wavefront per SIMD); MPPI_BUILD_VARIANT=w2 builds the same kernel for two.
#   MPPI_BUILD_VARIANT=w2 python __graft_entry__.py && gpurun -- bash tools/exp/w2_overlap.sh
mkdir -p gpurun_out/w2
for w in ${WORKLOADS:-boxer_push panda_pick}; do for K in 8192 16384; do for l in "" _w2; do
  MPPI_HIP_LIB=mppi-isaac_amd/csrc/libmppi_hip$l.so timeout 300 python bench.py --workload $w --k-total $K --no-cpu-baseline --steps 100 --warmup 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$w K=$K lib=$l', round(d['value'],1), 'Hz', round(d['ms_per_step'],4), 'ms')"
done; done; done 2>&1 | tee gpurun_out/w2/log.txt

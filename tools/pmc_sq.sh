#!/bin/bash
# SQ issue / wait / LDS counters of the bench kernels: two --pmc passes of the bench command (gpurun allows --pmc together
# with --kernel-trace only), condensed by tools/summarise_sq.py into gpurun_out/prof_<tag>/sq_summary.json - the file that
# is copied to profiles/<tag>_sq_summary.json and indexed by profiles/sq_latest.json (bench.py's roofline.issue reads it).
#   usage: WORKLOAD=panda_reach [KTOTAL=65536] tools/pmc_sq.sh <tag>
set -u
TAG=${1:-sq}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
WORKLOAD=${WORKLOAD:-panda_reach}
STEPS=${STEPS:-100}
EXTRA=""
if [ -n "${KTOTAL:-}" ]; then EXTRA="--k-total $KTOTAL"; fi
CMD="env MPPI_BENCH_SECOND=0 python $REPO/bench.py --workload $WORKLOAD --steps $STEPS --warmup 10 --no-cpu-baseline --no-facade $EXTRA"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_sq -- $CMD > $OUT/pmc_sq_bench.json 2> $OUT/pmc_sq.err
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_sq2 -- $CMD > $OUT/pmc_sq2_bench.json 2> $OUT/pmc_sq2.err
python $REPO/tools/summarise_sq.py $TAG
tail -3 $OUT/pmc_sq.err

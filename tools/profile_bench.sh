#!/bin/bash
# rocprofv3 passes of the bench command (run on the GPU box via gpurun):
#   1. --kernel-trace --stats            per-kernel durations
#   2. --pmc FETCH_SIZE                  HBM read bytes   (own pass, per MI355X_MICROARCH.md)
#   3. --pmc WRITE_SIZE                  HBM write bytes  (own pass)
# Summaries land in gpurun_out/prof_<tag>/ ; copy what should be judged into profiles/.
set -u
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
WORKLOAD=${WORKLOAD:-panda_reach}
STEPS=${STEPS:-300}
EXTRA=""
if [ -n "${KTOTAL:-}" ]; then EXTRA="--k-total $KTOTAL"; fi
CMD="env MPPI_BENCH_SECOND=0 python $REPO/bench.py --workload $WORKLOAD --steps $STEPS --warmup 30 --no-cpu-baseline --no-facade $EXTRA"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace_bench.json 2> $OUT/trace.err
if [ -z "${STATS_ONLY:-}" ]; then
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $CMD > $OUT/pmc_fetch_bench.json 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $CMD > $OUT/pmc_write_bench.json 2> $OUT/pmc_write.err
fi
find $OUT -name "*.csv" | head -30
for f in $(find $OUT/trace -name "*kernel_stats.csv"); do echo "== $f"; cat $f; done

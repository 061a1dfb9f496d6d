#!/bin/bash
# round 3, first GPU call: full GPU suite, bench of every workload, contact agreement at BASELINE sizes
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03a
mkdir -p $OUT
cd $REPO
timeout 1500 python -m pytest tests -m gpu -q -x -s > $OUT/gpu_tests.log 2>&1; echo "pytest rc=$?" >> $OUT/gpu_tests.log
tail -8 $OUT/gpu_tests.log
grep -h "vs fp64 oracle\|vs oracle on\|shared-lane kernel vs" $OUT/gpu_tests.log | head -30
bash tools/gpu_round.sh r03a bench 2>&1 | tail -12
timeout 600 python tools/exp/contact_agreement.py > $OUT/contact_agreement.txt 2>&1
grep -v "contact model" $OUT/contact_agreement.txt | tail -8

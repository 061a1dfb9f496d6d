#!/bin/bash
# One gpurun call's worth of work: GPU test suite, bench lines of every workload, A/B switches, rocprofv3 passes.
#   usage (on the GPU box, from the repo root): tools/gpu_round.sh <tag> [tests] [bench] [ab] [prof] [exp]
set -u
TAG=${1:-r02a}; shift
WHAT="${*:-tests bench ab prof exp}"
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
B="python bench.py --no-cpu-baseline"
if [[ $WHAT == *tests* ]]; then
  timeout 1500 python -m pytest tests -m gpu -q -s --durations=15 > $OUT/gpu_tests.log 2>&1; echo "pytest rc=$?" >> $OUT/gpu_tests.log
  tail -5 $OUT/gpu_tests.log
fi
if [[ $WHAT == *bench* ]]; then
  timeout 300 python bench.py > $OUT/bench.json 2> $OUT/bench.err
  timeout 300 python bench.py --steps 2000 --warmup 200 --no-cpu-baseline > $OUT/bench_2000.json 2>> $OUT/bench.err
  for w in point_reach boxer_push panda_pick; do timeout 300 $B --workload $w > $OUT/bench_$w.json 2>> $OUT/bench.err; done
  timeout 600 $B --workload panda_pick --k-total 65536 --steps 40 --warmup 5 > $OUT/bench_panda_pick_65536.json 2>> $OUT/bench.err
fi
if [[ $WHAT == *ab* ]]; then
  MPPI_FOLD=1 timeout 300 $B --steps 2000 --warmup 200 > $OUT/ab_fold.json 2>> $OUT/ab.err
  MPPI_BENCH_FORCE_DIST=1 timeout 300 $B --steps 2000 --warmup 200 > $OUT/ab_dist_graph.json 2>> $OUT/ab.err
  MPPI_BENCH_FORCE_DIST=1 MPPI_BENCH_GRAPH=0 timeout 300 $B --steps 2000 --warmup 200 > $OUT/ab_dist_eager.json 2>> $OUT/ab.err
  MPPI_BENCH_FORCE_DIST=1 MPPI_FOLD=1 timeout 300 $B --steps 2000 --warmup 200 > $OUT/ab_dist_graph_fold.json 2>> $OUT/ab.err
  MPPI_BENCH_FORCE_DIST=1 timeout 300 $B --workload panda_pick > $OUT/ab_dist_graph_pick.json 2>> $OUT/ab.err
  MPPI_FOLD=1 timeout 300 $B --workload panda_pick > $OUT/ab_fold_pick.json 2>> $OUT/ab.err
fi
if [[ $WHAT == *prof* ]]; then
  WORKLOAD=panda_reach STEPS=300 bash tools/profile_bench.sh ${TAG} > $OUT/prof_reach.log 2>&1
  WORKLOAD=boxer_push STEPS=100 bash tools/profile_bench.sh ${TAG}_boxer > $OUT/prof_boxer.log 2>&1
  WORKLOAD=panda_pick STEPS=60 bash tools/profile_bench.sh ${TAG}_pick > $OUT/prof_pick.log 2>&1
  WORKLOAD=panda_reach STEPS=100 bash tools/pmc_sq.sh ${TAG} > $OUT/sq_reach.log 2>&1
  WORKLOAD=boxer_push STEPS=60 bash tools/pmc_sq.sh ${TAG}_boxer > $OUT/sq_boxer.log 2>&1
  WORKLOAD=panda_pick STEPS=40 bash tools/pmc_sq.sh ${TAG}_pick > $OUT/sq_pick.log 2>&1
fi
if [[ $WHAT == *exp* ]]; then
  CLOSED_LOOP_ONLY=1 timeout 300 python tools/exp/scene_breakdown.py > $OUT/breakdown.log 2>&1
  MPPI_ROLLOUT=oct CLOSED_LOOP_ONLY=1 timeout 300 python tools/exp/scene_breakdown.py > $OUT/breakdown_oct.log 2>&1
  timeout 300 python tools/exp/wave_balance.py boxer_push panda_pick > $OUT/wave_balance.log 2>&1
  MPPI_ROLLOUT=oct timeout 300 $B --workload boxer_push > $OUT/ab_boxer_oct.json 2>> $OUT/ab.err
fi
for f in $OUT/bench*.json $OUT/ab_*.json; do [ -f $f ] && python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split("/")[-1], "Hz=%.1f ms=%.4f med=%.4f rollout_ms=%.4f tail_ms=%.4f" % (d["value"], d["ms_per_step"], d["latency_ms"]["median"], d["kernels_ms"]["k_rollout(+record tail)"], d["kernels_ms"]["k_combine_update(+world step)"]), d["config"]["parallelism"][:60])
except Exception as e:
    print(sys.argv[1], "unreadable:", e)
PY
done

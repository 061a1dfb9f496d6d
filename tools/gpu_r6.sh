#!/bin/bash
# round-6 gpurun work: tools/gpu_r6.sh <tag> [tests] [tasks] [bench] [benchall] [prof] [ab]
set -u
TAG=${1:-r06a}; shift
WHAT="${*:-tests tasks bench}"
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
if [[ $WHAT == *tasks* ]]; then
  timeout 900 python tools/task_outcomes.py ${TASK_STEPS:-900} ${TASKS:-panda_pick omni_panda_pick panda_stick_push} > $OUT/task_outcomes.txt 2> $OUT/task_outcomes.err; echo "tasks rc=$?"
  cat $OUT/task_outcomes.txt; tail -5 $OUT/task_outcomes.err
fi
if [[ $WHAT == *tests* ]]; then
  timeout 1700 python -m pytest tests -m gpu -q -s --durations=12 ${PYTEST_ARGS:-} > $OUT/gpu_tests.log 2>&1; echo "pytest rc=$?" >> $OUT/gpu_tests.log
  grep -E "passed|failed|FAILED|Error|rc=" $OUT/gpu_tests.log | tail -40
fi
if [[ $WHAT == *bench* ]]; then
  timeout 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_driver_cmd.json 2>> $OUT/bench.err
  tail -5 $OUT/bench.err
fi
if [[ $WHAT == *benchall* ]]; then
  for w in point_reach boxer_push panda_pick; do timeout 300 python bench.py --no-cpu-baseline --workload $w > $OUT/bench_$w.json 2>> $OUT/bench.err; done
  timeout 600 python bench.py --no-cpu-baseline --workload panda_pick --k-total 65536 --steps 40 --warmup 5 > $OUT/bench_panda_pick_65536.json 2>> $OUT/bench.err
fi
if [[ $WHAT == *ab* ]]; then
  timeout 600 python tools/exp/ab_time.py ${AB_ARGS:-} > $OUT/ab_time.txt 2>&1; cat $OUT/ab_time.txt | tail -30
fi
if [[ $WHAT == *prof* ]]; then
  WORKLOAD=panda_reach STEPS=300 bash tools/profile_bench.sh ${TAG} > $OUT/prof_reach.log 2>&1
  WORKLOAD=boxer_push STEPS=100 bash tools/profile_bench.sh ${TAG}_boxer > $OUT/prof_boxer.log 2>&1
  WORKLOAD=panda_pick STEPS=60 bash tools/profile_bench.sh ${TAG}_pick > $OUT/prof_pick.log 2>&1
  WORKLOAD=panda_reach STEPS=100 bash tools/pmc_sq.sh ${TAG} > $OUT/sq_reach.log 2>&1
  WORKLOAD=boxer_push STEPS=60 bash tools/pmc_sq.sh ${TAG}_boxer > $OUT/sq_boxer.log 2>&1
  WORKLOAD=panda_pick STEPS=40 bash tools/pmc_sq.sh ${TAG}_pick > $OUT/sq_pick.log 2>&1
  cd $REPO; for t in ${TAG} ${TAG}_boxer ${TAG}_pick; do python tools/summarise_profile.py $t > /dev/null 2>&1; done
  mkdir -p $OUT/profiles && cp profiles/${TAG}* $OUT/profiles/ 2>/dev/null; cp profiles/pmc_latest.json profiles/sq_latest.json $OUT/profiles/ 2>/dev/null
fi
for f in $OUT/bench*.json; do [ -f $f ] || continue; python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split("/")[-1], "Hz=%.1f ms=%.4f rollout_ms=%.4f tail_ms=%.4f facade=%s generic=%s" % (d["value"], d["ms_per_step"], d["kernels_ms"]["k_rollout(+record tail)"], d["kernels_ms"]["k_combine_update(+world step)"], d.get("value_facade"), d.get("value_generic_objective")), json.dumps(d["config"].get("task_outcome")))
except Exception as e:
    print(sys.argv[1], "unreadable:", e)
PY
done
exit 0

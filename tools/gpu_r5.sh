#!/bin/bash
# round-5 gpurun work: tools/gpu_r5.sh <tag> [tests] [bench] [facade] [benchall] [prof]
set -u
TAG=${1:-r05a}; shift
WHAT="${*:-tests bench facade}"
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
if [[ $WHAT == *tests* ]]; then
  timeout 1700 python -m pytest tests -m gpu -q -s --durations=12 ${PYTEST_ARGS:-} > $OUT/gpu_tests.log 2>&1; echo "pytest rc=$?" >> $OUT/gpu_tests.log
  tail -25 $OUT/gpu_tests.log
fi
if [[ $WHAT == *bench* ]]; then
  timeout 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_driver_cmd.json 2>> $OUT/bench.err
  python - $OUT/bench.json $OUT/bench_driver_cmd.json <<'PY'
import json, sys
for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], "value %.1f facade %s generic %s rollout_ms %.4f" % (d["value"], d.get("value_facade"), d.get("value_generic_objective"), d["kernels_ms"]["k_rollout(+record tail)"]))
        print("   facade:", json.dumps({k: v for k, v in (d["config"].get("facade") or {}).items() if k != "what"}))
    except Exception as e:
        print(f, "unreadable:", e)
PY
  tail -5 $OUT/bench.err
fi
if [[ $WHAT == *facade* ]]; then
  timeout 600 python tools/facade_profile.py > $OUT/facade_profile.txt 2> $OUT/facade_profile.err; echo "facade rc=$?"
  head -40 $OUT/facade_profile.txt; tail -5 $OUT/facade_profile.err
fi
if [[ $WHAT == *benchall* ]]; then
  for w in point_reach boxer_push panda_pick; do timeout 300 python bench.py --no-cpu-baseline --workload $w > $OUT/bench_$w.json 2>> $OUT/bench.err; done
  timeout 600 python bench.py --no-cpu-baseline --workload panda_pick --k-total 65536 --steps 40 --warmup 5 > $OUT/bench_panda_pick_65536.json 2>> $OUT/bench.err
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

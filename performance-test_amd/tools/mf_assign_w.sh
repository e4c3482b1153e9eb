#!/bin/bash
# Quality against cost of the step assignment: W lanes of the wavefront take cells per batch (ZZZ_EXPERIMENTS build);
# rounds per step from the plan (ZZZ_MF_PLAN_STATS), set-up and action time.
cd "$(dirname "$0")/../.."
export ZZZ_HIP_LIB=$PWD/performance-test_amd/libzzz_hip_exp.so
export ZZZ_MF_PLAN_STATS=1
for c in ${CASES:-p1 p3 p2}; do
for w in ${WS:-64 32 16 8 1}; do
  ZZZ_MF_ASSIGN_W=$w python performance-test_amd/tools/mf_bench.py $c > /tmp/mfw.out 2> /tmp/mfw.err
  grep "rounds per step" /tmp/mfw.err | tail -1
  tail -1 /tmp/mfw.out | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$c W $w setup_warm_ms %.2f action_ms %.4f' % (d['setup_warm_ms'], d['action_ms']))"
done; done

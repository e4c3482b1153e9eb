#!/bin/bash
# Where the matrix-free action spends its time: phases switched off one at a time in the ZZZ_EXPERIMENTS build
# (ZZZ_MF_DEBUG bits: 1 element kernel, 2 rounds, 4 staging gather, 8 write-out, 16 finish kernel; results wrong).
cd "$(dirname "$0")/../.."
export ZZZ_HIP_LIB=$PWD/performance-test_amd/libzzz_hip_exp.so
for c in ${CASES:-p1 p3}; do
for d in 0 1 2 4 8 16 3 31; do
  ZZZ_MF_DEBUG=$d python performance-test_amd/tools/mf_bench.py $c 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$c dbg $d action_ms %.4f' % d['action_ms'])"
done; done

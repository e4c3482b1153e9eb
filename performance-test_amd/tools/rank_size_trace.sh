#!/bin/bash
# Per-iteration timeline of the CG loop at the 8-GPU per-rank size (1.25 M rows) with the multi-GPU code path
# attached on ONE GPU (1-rank RCCL communicator + peer-memory mailboxes): kernel-trace of bench.py --force_comm,
# summarised by tools/trace_gaps.py into gpurun_out/rank_trace_<tag>/{classical,single_reduction}.csv
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/rank_trace_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for form in classical single_reduction; do
  for comm in --force_comm ""; do
    name=$form${comm:+_comm}
    rocprofv3 --kernel-trace --output-format csv -d $OUT/$name -o t -- python3 $R/bench.py --ndofs 1250000 $comm --cg $form --steps 2 --warmup 1 --no_cpu_baseline > $OUT/$name.json 2> $OUT/$name.log
    python3 $R/performance-test_amd/tools/trace_gaps.py $OUT/$name $OUT/$name.csv > $OUT/$name.txt
    rm -rf $OUT/$name
    echo "== $name"; cat $OUT/$name.txt; python3 -c "
import json,sys
d=json.loads(open('$OUT/$name.json').read().strip().splitlines()[-1]); print('solve ms', d['phases_ms']['ZZZ Solve'], 'iters', d['config']['krylov_iterations'], 'us/iter', 1e3*d['phases_ms']['ZZZ Solve']/d['config']['krylov_iterations'])"
  done
done

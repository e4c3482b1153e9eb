#!/bin/bash
# Per-iteration timeline of the CG loop at the per-rank size of another BASELINE config with the multi-GPU code path
# attached on ONE GPU (1-rank RCCL communicator + mailboxes): rank_size_trace_cfg.sh <tag> "<bench.py args>"
TAG=${1:-c4}
ARGS=$2
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/rank_trace_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for form in classical single_reduction; do
  name=${form}_comm
  rocprofv3 --kernel-trace --output-format csv -d $OUT/$name -o t -- python3 $R/bench.py $ARGS --force_comm --cg $form --steps 2 --warmup 1 --no_cpu_baseline --no_other_configs > $OUT/$name.json 2> $OUT/$name.log
  echo "== $TAG $name"; python3 $R/performance-test_amd/tools/trace_gaps.py $OUT/$name $OUT/$name.csv
  rm -rf $OUT/$name
done

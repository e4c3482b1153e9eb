#!/bin/bash
# Profiles of a bench command for profiles/ (run on the GPU box through gpurun):
#   1. rocprofv3 --kernel-trace --stats of `python3 bench.py [--config CFG]` (the judged command), CSV summary
#   2. two --pmc passes (FETCH_SIZE, WRITE_SIZE; own runs, kernel trace only) for the HBM traffic
# Results land in gpurun_out/profile_<tag>_<cfg>/; tools/profile_summarise.py turns them into profiles/ files.
# Usage: bash performance-test_amd/tools/profile_bench.sh <tag> [cfg = c2] [extra bench args]
TAG=${1:-r02}
CFG=${2:-c2}
shift; shift
EXTRA="$@"
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/profile_${TAG}_${CFG}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --config $CFG $EXTRA > $OUT/bench_plain.json 2> $OUT/bench_plain.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $R/bench.py --config $CFG $EXTRA --no_cpu_baseline > $OUT/bench_under_rocprof.json 2> $OUT/trace.log
find $OUT/trace -type f ! -name '*kernel_stats.csv' -delete
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -o p -- python3 $R/bench.py --config $CFG $EXTRA --steps 1 --warmup 0 --no_cpu_baseline > $OUT/pmc_$c.json 2> $OUT/pmc_$c.log
  python3 $R/performance-test_amd/tools/profile_summarise.py reduce $OUT/pmc_$c $c > $OUT/pmc_$c.reduced.json
  rm -rf $OUT/pmc_$c
done
tail -c 400 $OUT/bench_plain.json

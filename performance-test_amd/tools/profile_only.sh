#!/bin/bash
# Profiles of one record of bench.py's other_configs (`python3 bench.py --only <record>`) for profiles/:
#   1. the plain run; 2. rocprofv3 --kernel-trace --stats of the same command; 3. two --pmc passes (FETCH_SIZE, WRITE_SIZE)
# Results: gpurun_out/profile_<tag>_<record>/ ; tools/profile_summarise.py merge_only turns them into profiles/ files.
# Usage (GPU box): bash performance-test_amd/tools/profile_only.sh <tag> <record>
TAG=${1:-r04}
REC=${2:-cgpoisson_p1_c2}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/profile_${TAG}_${REC}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --only $REC > $OUT/bench_plain.json 2> $OUT/bench_plain.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $R/bench.py --only $REC > $OUT/bench_under_rocprof.json 2> $OUT/trace.log
find $OUT/trace -type f ! -name '*kernel_stats.csv' -delete
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -o p -- python3 $R/bench.py --only $REC > $OUT/pmc_$c.json 2> $OUT/pmc_$c.log
  python3 $R/performance-test_amd/tools/profile_summarise.py reduce $OUT/pmc_$c $c > $OUT/pmc_$c.reduced.json
  rm -rf $OUT/pmc_$c
done
tail -c 300 $OUT/bench_plain.json

#!/bin/bash
# per-iteration timeline at a given size for several vector-kernel grids (ZZZ_VGRID_PER = entries per thread)
R=$GRAFT_REPO_ROOT
N=${1:-1250000}
OUT=$R/gpurun_out/ab_vgrid
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for per in ${PERS:-8}; do
  export ZZZ_VGRID_PER=$per
  for mode in "--cg single_reduction --force_comm" "--cg classical --force_comm" "--cg classical"; do
    name=per${per}_$(echo $mode | tr -d ' -')
    rocprofv3 --kernel-trace --output-format csv -d $OUT/$name -o t -- python3 $R/bench.py --ndofs $N $mode --steps 2 --warmup 1 --no_cpu_baseline > $OUT/$name.json 2> $OUT/$name.log
    echo "== per=$per $mode"; python3 $R/performance-test_amd/tools/trace_gaps.py $OUT/$name $OUT/$name.csv
    rm -rf $OUT/$name
  done
done

R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/ab_tail
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for t in 0 1; do
  export ZZZ_TAIL=$t
  name=sr_comm_tail$t
  rocprofv3 --kernel-trace --output-format csv -d $OUT/$name -o t -- python3 $R/bench.py --ndofs 1250000 --force_comm --cg single_reduction --steps 2 --warmup 1 --no_cpu_baseline > $OUT/$name.json 2> $OUT/$name.log
  python3 $R/performance-test_amd/tools/trace_gaps.py $OUT/$name $OUT/$name.csv
  rm -rf $OUT/$name
done

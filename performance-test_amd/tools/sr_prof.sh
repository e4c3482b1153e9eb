cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for m in sr_comm sr_plain cl_comm; do
  case $m in
    sr_comm) args="--cg single_reduction --force_comm";;
    sr_plain) args="--cg single_reduction";;
    cl_comm) args="--cg classical --force_comm";;
  esac
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$m -o p -- python3 $R/bench.py --no_cpu_baseline $args --steps 1 --warmup 0 > $R/gpurun_out/prof_$m.log 2>&1
  f=$(find $R/gpurun_out/prof_$m -name '*kernel_stats.csv' | head -1)
  echo "== $m"; head -12 "$f"
  find $R/gpurun_out/prof_$m -type f ! -name '*kernel_stats.csv' -delete
done

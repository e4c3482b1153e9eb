set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06r
mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_gpu_block_rows.py -q -m gpu -x > $O/pytest.log 2>&1
tail -3 $O/pytest.log
for v in a flat a flat; do
  if [ $v = flat ]; then export ZZZ_HIP_LIB=$R/performance-test_amd/ab_old/libzzz_hip_flat.so; else unset ZZZ_HIP_LIB; fi
  python3 bench.py --only c5_rank --steps 3 --warmup 1 --no_cpu_baseline --no_alt_pc > $O/c5_rank_$v.json 2> $O/c5_rank_$v.log
  python3 - <<PY
import json
d=json.loads(open("$O/c5_rank_$v.json").read().strip().splitlines()[-1])
r=d["c5_rank"]
print("$v", round(r["ms_per_step"],2), {k:round(v,2) for k,v in r["phases_ms"].items()}, r["krylov_iterations"], round(r["product_ms"],4))
PY
done

set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06q
mkdir -p $O
cd $R
timeout 1800 python3 -m pytest tests/test_gpu_block_rows.py tests/test_gpu_product.py tests/test_gpu_partitions.py -q -m gpu -x > $O/pytest.log 2>&1
tail -5 $O/pytest.log
python3 bench.py --only c5_rank --steps 3 --warmup 1 --no_cpu_baseline > $O/c5_rank.json 2> $O/c5_rank.log
python3 - <<PY
import json
for n in ("c5_rank",):
    d=json.loads(open("$O/%s.json"%n).read().strip().splitlines()[-1])
    r=d[n]
    print(n, round(r["ms_per_step"],2), {k:round(v,2) for k,v in r["phases_ms"].items()}, r["krylov_iterations"], round(r["product_ms"],4), r["operator"][:40], r.get("alt_preconditioner",{}).get("ZZZ Solve ms"))
PY

R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06x
mkdir -p $O
cd $R
for i in 1 2 3; do timeout 900 python3 -m pytest tests/test_gpu_block_rows.py -q -m gpu -x -k "form_chosen or by_node or window" > $O/pytest_$i.log 2>&1; tail -2 $O/pytest_$i.log; done
timeout 1500 python3 -m pytest tests/test_gpu_block_rows.py tests/test_gpu_product.py tests/test_gpu_partitions.py -q -m gpu -x > $O/pytest_all.log 2>&1; tail -3 $O/pytest_all.log
python3 bench.py --only c5_rank --steps 3 --warmup 1 --no_cpu_baseline --no_alt_pc > $O/c5_rank.json 2> $O/c5_rank.log
python3 - <<PY
import json
d=json.loads(open("$O/c5_rank.json").read().strip().splitlines()[-1]); r=d["c5_rank"]
print(round(r["ms_per_step"],2), {k:round(v,2) for k,v in r["phases_ms"].items()}, r["krylov_iterations"], round(r["product_ms"],4))
PY

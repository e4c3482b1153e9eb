set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06s
mkdir -p $O
cd $R
timeout 1800 python3 -m pytest tests/test_gpu_assembly.py tests/test_gpu_block_rows.py tests/test_gpu_driver.py -q -m gpu -x > $O/pytest.log 2>&1
tail -5 $O/pytest.log
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5 -o t -- python3 $R/bench.py --only c5_rank --steps 3 --warmup 1 --no_cpu_baseline --no_alt_pc > $O/c5_rank.json 2> $O/c5_rank.log
python3 - <<PY
import csv,glob,json
f=glob.glob("$O/c5/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    n=r["Name"]
    if any(k in n for k in ("asm_","k_cell_","k_bw","k_row","k_adj","spmv_win")): print(f'{float(r["AverageNs"])/1e3:10.1f} us x {r["Calls"]:>5}  {n[:70]}')
d=json.loads(open("$O/c5_rank.json").read().strip().splitlines()[-1])
r=d["c5_rank"]
print(round(r["ms_per_step"],2), {k:round(v,2) for k,v in r["phases_ms"].items()}, r["krylov_iterations"], round(r["product_ms"],4))
PY

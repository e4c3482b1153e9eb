# block-row builders with one walk over the values (slots parked, 128-bit fingerprints): parity + timing at C4
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06k2
mkdir -p $O
cd $R
timeout 1500 python3 -m pytest tests/test_gpu_block_rows.py tests/test_gpu_product.py tests/test_gpu_partitions.py -q -m gpu -x > $O/pytest.log 2>&1
tail -8 $O/pytest.log
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c4 -o t -- python3 $R/bench.py --only c4_total --steps 3 --warmup 1 --no_cpu_baseline --no_alt_pc > $O/c4_total.json 2> $O/c4_total.log
python3 - <<PY
import csv,glob,json
for d in ("c4",):
    f=glob.glob("$O/%s/**/*kernel_stats.csv"%d,recursive=True)[0]
    for r in csv.DictReader(open(f)):
        n=r["Name"]
        if any(k in n for k in ("asm_","k_cell_","k_bk","k_sp_","spmv_blk","k_row","k_adj","scan")): print(d, f'{float(r["AverageNs"])/1e3:10.1f} us x {r["Calls"]:>5}  {n[:70]}')
d=json.loads(open("$O/c4_total.json").read().strip().splitlines()[-1])
r=d["c4_total"]
print(round(r["ms_per_step"],2), {k:round(v,2) for k,v in r["phases_ms"].items()}, r["krylov_iterations"], round(r["product_ms"],4), r["operator"][:40])
PY

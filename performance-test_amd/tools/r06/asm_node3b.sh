# (1) node3 kernel with one LDS round trip per pair, (2) special forms chosen at assembly time (no generic pack): parity + timing
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06i
mkdir -p $O
cd $R
timeout 1500 python3 -m pytest tests/test_gpu_assembly.py tests/test_gpu_block_rows.py tests/test_gpu_product.py -q -m gpu -x > $O/pytest.log 2>&1
tail -8 $O/pytest.log
python3 bench.py --only c4_total --steps 3 --warmup 1 --no_cpu_baseline > $O/c4_total.json 2> $O/c4_total.log
python3 bench.py --only c5_rank --steps 3 --warmup 1 --no_cpu_baseline > $O/c5_rank.json 2> $O/c5_rank.log
ZZZ_SELLP_EARLY=0 python3 bench.py --only c4_total --steps 3 --warmup 1 --no_cpu_baseline > $O/c4_total_late.json 2> $O/c4_total_late.log
python3 - <<PY
import json
for n in ("c4_total","c5_rank","c4_total_late"):
    try:
        d=json.loads(open("$O/%s.json"%n).read().strip().splitlines()[-1])
        r=d["other_configs"][n.replace("_late","")]
        print(n, round(r["ms_per_step"],2), {k:round(v,2) for k,v in r["phases_ms"].items()}, r["krylov_iterations"], round(r["product_ms"],4), r["operator"][:40])
    except Exception as e: print(n,"failed",e)
PY
cd /tmp; export TMPDIR=/tmp
export ZZZ_ASM_NODE3=1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/node3 -o t -- python3 $R/performance-test_amd/tools/asm_probe.py elasticity 1 110 4 > $O/node3.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$O/node3/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    n=r["Name"]
    if "asm_" in n or "k_cell_" in n: print(f'{float(r["AverageNs"])/1e3:10.1f} us x {r["Calls"]:>4}  {n[:70]}')
PY

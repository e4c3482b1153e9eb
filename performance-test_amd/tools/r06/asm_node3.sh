# elasticity P1 matrix assembly: thread per node (asm_matrix_p1_node3) against thread per scalar row; parity tests first
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06h
mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_gpu_assembly.py -q -m gpu -x > $O/pytest_assembly.log 2>&1
tail -5 $O/pytest_assembly.log
cd /tmp; export TMPDIR=/tmp
for k in 0 1; do
  export ZZZ_ASM_NODE3=$k
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/node3_$k -o t -- python3 $R/performance-test_amd/tools/asm_probe.py elasticity 1 110 4 > $O/node3_$k.log 2>&1
  python3 - <<PY
import csv,glob
f=glob.glob("$O/node3_$k/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    n=r["Name"]
    if "asm_" in n or "k_cell_" in n: print("NODE3=$k", f'{float(r["AverageNs"])/1e3:10.1f} us x {r["Calls"]:>4}  {n[:70]}')
PY
done
unset ZZZ_ASM_NODE3

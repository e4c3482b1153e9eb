set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06m
mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_gpu_block_rows.py -q -m gpu -x > $O/pytest.log 2>&1
tail -5 $O/pytest.log
cd /tmp; export TMPDIR=/tmp
export ZZZ_HIP_LIB=$R/performance-test_amd/libzzz_hip_exp.so
for ph in 2 3 4; do
  export ZZZ_BW_PHASES=$ph
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ph$ph -o t -- python3 $R/performance-test_amd/tools/asm_probe.py poisson 3 61 2 > $O/ph$ph.log 2>&1
  python3 - <<PY
import csv,glob
f=glob.glob("$O/ph$ph/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    n=r["Name"]
    if "k_bw_block" in n or "k_bw_values" in n: print("phases=$ph", f'{float(r["AverageNs"])/1e3:10.1f} us x {r["Calls"]:>4}  {n[:50]}')
PY
done

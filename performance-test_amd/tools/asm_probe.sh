#!/bin/bash
# asm_probe.sh "<asm_probe.py args>" "<ENV=.. ENV=..>" ... : kernel times of the assembly kernels per environment
ARGS=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for envs in "$@"; do
  OUT=$R/gpurun_out/asm_probe_$$
  mkdir -p $OUT
  env $envs rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p -o t -- python3 $R/performance-test_amd/tools/asm_probe.py $ARGS > $OUT/log 2>&1
  echo "== $envs"
  python3 - <<PY
import csv,glob
f=glob.glob("$OUT/p/**/*kernel_stats.csv",recursive=True)
for r in csv.DictReader(open(f[0])) if f else []:
    n=r["Name"]
    if "asm_matrix" in n or "asm_vector" in n or "k_cell_" in n: print(f'{float(r["AverageNs"])/1e3:10.1f} us x {r["Calls"]:>4}  {n[:60]}')
PY
  rm -rf $OUT
done

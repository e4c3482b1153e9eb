#!/bin/bash
# matrix assembly kernel time under measurement knobs: ab_asm.sh "<bench.py args>" "<ENV=.. ENV=..>" ...
ARGS=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for envs in "$@"; do
  OUT=$R/gpurun_out/ab_asm_$$
  mkdir -p $OUT
  env $envs rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p -o t -- python3 $R/bench.py $ARGS --steps 2 --warmup 1 --no_cpu_baseline --no_other_configs > $OUT/bench.json 2> $OUT/bench.log
  echo "== $envs"
  python3 - <<PY
import csv,glob,json
f=glob.glob("$OUT/p/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    n=r["Name"]
    if "asm_matrix" in n or "asm_vector" in n or "k_cell_" in n: print(f'{float(r["AverageNs"])/1e3:10.1f} us x {r["Calls"]:>4}  {n[:60]}')
d=json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1]); print({k[:16]:round(v,2) for k,v in d["phases_ms"].items()})
PY
  rm -rf $OUT
done

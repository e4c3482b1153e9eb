#!/bin/bash
# per-kernel averages of the pattern build (create_matrix) of a config: rocprofv3 --kernel-trace --stats of a bench run
CFG=${1:-c2}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/stats_pattern_$CFG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p -o t -- python3 $R/bench.py --config $CFG --steps ${STEPS:-3} --warmup 1 --no_cpu_baseline --no_other_configs > $OUT/bench.json 2> $OUT/bench.log
python3 - <<PY
import csv,glob
f=glob.glob("$OUT/p/**/*kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows:
    n=r["Name"]
    if any(k in n for k in ("spmv_","k_update","k_sr_","k_init","k_extract")): continue
    print(f'{float(r["AverageNs"])/1e3:10.1f} us x {r["Calls"]:>5}  {n[:110]}')
PY
python3 -c "
import json; d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]); print(d['phases_ms'])"
rm -rf $OUT/p

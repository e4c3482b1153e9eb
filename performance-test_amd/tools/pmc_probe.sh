#!/bin/bash
# Derived-metric probe of the CG kernels (one rocprofv3 --pmc pass per metric group; kernel trace only).
# Every pass runs under `timeout` (a counter group the tool cannot collect aborts it and the abort handler can hang).
# Usage: CFG=c5_rank bash performance-test_amd/tools/pmc_probe.sh "MemUnitBusy MemUnitStalled" "L2CacheHit" ...
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "$@"; do
  i=$((i+1))
  timeout ${PMC_TIMEOUT:-240} rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/pmcp_$i -o p -- python3 $R/bench.py --config ${CFG:-c2} --steps 1 --warmup 0 --no_cpu_baseline --no_other_configs --no_alt_pc > /dev/null 2> $R/gpurun_out/pmcp_$i.log
  python3 - "$R/gpurun_out/pmcp_$i" <<'PY'
import csv,glob,sys,collections,re
acc=collections.defaultdict(lambda: collections.defaultdict(lambda:[0.0,0]))
for f in glob.glob(sys.argv[1]+"/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"]
        k="spmv" if ("spmv_tile" in n or "spmv_sellp" in n or "spmv_one" in n) else "update_p" if "k_update_p" in n else "update_xr" if "k_update_xr" in n else None
        if not k: continue
        if int(r["End_Timestamp"])-int(r["Start_Timestamp"])<30000: continue
        a=acc[k][r["Counter_Name"]]; a[0]+=float(r["Counter_Value"]); a[1]+=1
for k in acc:
    print(k, {c:round(v[0]/v[1],3) for c,v in acc[k].items()})
PY
  rm -rf $R/gpurun_out/pmcp_$i
done

#!/bin/bash
# rocprofv3 --kernel-trace --stats of the assembly phases of one case (tools/ab_lib.py --child): per-kernel averages
# Usage (on the GPU box): bash performance-test_amd/tools/stats_child.sh [case = c2]
CASE=${1:-c2}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/stats_child && mkdir -p $R/gpurun_out/stats_child
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/stats_child -o t -- python3 $R/performance-test_amd/tools/ab_lib.py --child $CASE > /dev/null 2> $R/gpurun_out/stats_child/log
python3 - <<PY
import csv, glob
f = glob.glob("$R/gpurun_out/stats_child/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:22]:
    print(r["Name"][:70].ljust(70), r["Calls"].rjust(5), f"{float(r['AverageNs']) / 1e3:10.1f} us")
PY

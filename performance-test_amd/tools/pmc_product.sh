#!/bin/bash
# SQ / TA / TCP / TCC counters of the CG product kernel on one case (one rocprofv3 --pmc pass per group; kernel trace only;
# every pass under `timeout`).  Usage: bash performance-test_amd/tools/pmc_product.sh <tag> <case> ["group" ...]
# Writes gpurun_out/pmcprod_<tag>_<case>.json (per kernel: average counter values per dispatch) and prints it.
TAG=$1; CASE=$2; shift; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
if [ $# -eq 0 ]; then
  set -- "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES" \
         "SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
         "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL" \
         "SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_INT32 SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE GRBM_TA_BUSY" \
         "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum" \
         "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
         "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE"
fi
i=0
for grp in "$@"; do
  i=$((i+1))
  D=$R/gpurun_out/pmcprod_${TAG}_${CASE}_$i
  timeout ${PMC_TIMEOUT:-200} rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $D -o p -- python3 $R/performance-test_amd/tools/prod_probe.py $CASE ${REPS:-12} 1 > $D.log 2>&1
  tail -1 $D.log | cut -c1-300
done
python3 - "$R/gpurun_out" "$TAG" "$CASE" <<'PY'
import csv, glob, sys, collections, json, os, shutil
root, tag, case = sys.argv[1:4]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
dur = collections.defaultdict(lambda: [0.0, 0])
for d in sorted(glob.glob(f"{root}/pmcprod_{tag}_{case}_*")):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if "spmv_" not in n:
                continue
            k = n[:n.find("(")] if "(" in n else n
            k = k[-90:]
            a = acc[k][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
    shutil.rmtree(d, ignore_errors=True)
out = {k: {c: v[0] / v[1] for c, v in cs.items()} for k, cs in acc.items()}
json.dump(out, open(f"{root}/pmcprod_{tag}_{case}.json", "w"), indent=1)
for k, cs in out.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:40s} {v:16.1f}")
PY

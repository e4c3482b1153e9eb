#!/bin/bash
# SQ / TCC counters of the assembly kernels (one rocprofv3 --pmc pass per group; kernel trace only; every pass under `timeout`).
# Usage: bash performance-test_amd/tools/pmc_asm.sh <tag> "<asm_probe.py args>" ["group" ...]; writes gpurun_out/pmcasm_<tag>.json
TAG=$1; ARGS=$2; shift; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
if [ $# -eq 0 ]; then
  set -- "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
         "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_INT32 SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" \
         "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE"
fi
i=0
for grp in "$@"; do
  i=$((i+1))
  D=$R/gpurun_out/pmcasm_${TAG}_$i
  timeout ${PMC_TIMEOUT:-150} rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $D -o p -- python3 $R/performance-test_amd/tools/asm_probe.py $ARGS > $D.log 2>&1
  tail -1 $D.log | cut -c1-200
done
python3 - "$R/gpurun_out" "$TAG" <<'PY'
import csv, glob, sys, collections, json, os, shutil
root, tag = sys.argv[1:3]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
dur = collections.defaultdict(lambda: [0.0, 0])
for d in sorted(glob.glob(f"{root}/pmcasm_{tag}_*")):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            k = next((x for x in ("asm_matrix_p1", "asm_vector_p1", "asm_matrix_pk_pos", "asm_vector_pk", "k_cell_geom", "k_cell_load_p1") if x in n), None)
            if not k:
                continue
            a = acc[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
            t = dur[k]; t[0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); t[1] += 1
    shutil.rmtree(d, ignore_errors=True)
out = {k: dict({c: v[0] / v[1] for c, v in cs.items()}, avg_us_under_pmc=dur[k][0] / max(dur[k][1], 1) / 1e3) for k, cs in acc.items()}
json.dump(out, open(f"{root}/pmcasm_{tag}.json", "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
PY

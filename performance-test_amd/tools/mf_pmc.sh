#!/bin/bash
# SQ counters of the matrix-free kernels (one rocprofv3 --pmc pass per group; kernel trace only): tools/mf_bench.py <case>.
# usage: CASE=p1 bash performance-test_amd/tools/mf_pmc.sh   (on the GPU box)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/mf_pmc
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA" "GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/mf_pmc/$tag -o p -- python3 $R/performance-test_amd/tools/mf_bench.py ${CASE:-p1} > /dev/null 2> $R/gpurun_out/mf_pmc/$tag.log
done
python3 - <<'PY'
import csv, glob, os, collections
R = os.environ['GRAFT_REPO_ROOT']
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
dur = collections.defaultdict(float); nd = collections.defaultdict(int)
for f in sorted(glob.glob(R + '/gpurun_out/mf_pmc/*/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        k = 'k_mf_action' if 'k_mf_action' in k else 'k_mf_finish' if 'k_mf_finish' in k else None
        if not k:
            continue
        acc[k][r['Counter_Name']] += float(r['Counter_Value']); n[(k, r['Counter_Name'])] += 1
        dur[k] += int(r['End_Timestamp']) - int(r['Start_Timestamp']); nd[k] += 1
for k, v in acc.items():
    print(k, 'avg_us %.1f' % (dur[k] / nd[k] / 1e3), {c: round(x / n[(k, c)]) for c, x in v.items()})
PY
rm -rf $R/gpurun_out/mf_pmc

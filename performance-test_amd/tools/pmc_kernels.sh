cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
  tag=$(echo $set | tr ' ' '_')
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_pat/$tag -o p -- python3 $R/performance-test_amd/tools/ab_lib.py --child ${CASE:-c2} > /dev/null 2> $R/gpurun_out/pmc_pat/$tag.log
done
python3 - <<'PY'
import csv, glob, os, collections
R=os.environ['GRAFT_REPO_ROOT']
for f in sorted(glob.glob(R+'/gpurun_out/pmc_pat/*/**/*counter_collection.csv', recursive=True)):
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'][:60]
        if 'row_pattern' in k or 'asm_' in k or 'onesweep' in k or 'row_copy' in k:
            acc[k][r['Counter_Name']]+=float(r['Counter_Value']); n[(k,r['Counter_Name'])]+=1
    for k,v in acc.items():
        print(k, {c: x/n[(k,c)] for c,x in v.items()})
PY

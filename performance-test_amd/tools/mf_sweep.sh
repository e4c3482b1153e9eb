#!/bin/bash
# Sweep of the matrix-free plan geometry (cells per block, workgroup size, LDS budget): tools/mf_bench.py per setting.
cd "$(dirname "$0")/../.."
out=gpurun_out/mf_sweep.log
mkdir -p gpurun_out
: > $out
for cfg in "p1 4096 512 64" "p1 2048 256 40" "p1 4096 256 64" "p1 1024 256 24" "p1 2048 128 40" "p1 3072 256 64" \
           "p3 512 256 64" "p3 1024 256 100" "p3 768 256 64" "p3 256 256 40" "p3 1024 512 100" "p3 512 128 64" "p2 1024 256 64" "p2 2048 256 64" "p2 1024 512 64"; do
  set -- $cfg
  ZZZ_MF_NC=$2 ZZZ_MF_T=$3 ZZZ_MF_LDS_KB=$4 python performance-test_amd/tools/mf_bench.py $1 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
p = d.get('plan', {})
print('$cfg', 'nc', p.get('cells_per_block'), 'nloc_max', p.get('nloc_max'), 'shared', p.get('shared_dofs'), 'setup_ms %.1f' % d['setup_warm_ms'], 'action_ms %.4f' % d['action_ms'], 'GB/s %.0f' % d.get('action_GBs', 0), 'Gdof/s %.2f' % d['Gdofs'], 'cg_s %.4f' % d['cg_s'])
" >> $out
done
cat $out

#!/bin/bash
# A/B of the x windows of the operator stream: ZZZ_SELLP_WIN = doubles of LDS per workgroup (0: off)
for cfg in "$@"; do for w in 0 2048 0 2048 3072; do
ZZZ_SELLP_WIN=$w python bench.py --config $cfg --steps 2 --warmup 1 --no_cpu_baseline --no_alt_pc 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$cfg win=$w product ms', round(d['roofline']['avg_launch_ms'],5), 'frac', round(d['roofline']['frac'],3), 'A ms', round(d['phases_ms']['ZZZ Assemble matrix'],2), 'solve', round(d['phases_ms']['ZZZ Solve'],2), 'its', d['config']['krylov_iterations'], 'step', round(d['ms_per_step'],1))"
done; done

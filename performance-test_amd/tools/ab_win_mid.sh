#!/bin/bash
# x windows at sizes around the default's threshold (elasticity P1, one GPU): product time with the default, without, forced
for nd in 600000 1000000 1500000 2500000; do for w in default 0 2048; do
if [ $w = default ]; then unset ZZZ_SELLP_WIN; else export ZZZ_SELLP_WIN=$w; fi
python bench.py --problem_type elasticity --scaling_type strong --ndofs $nd --steps 2 --warmup 1 --no_cpu_baseline --no_alt_pc --no_other_configs 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('ndofs $nd win=$w nnz', d['config']['nnz_rank0'], 'product ms', round(d['roofline']['avg_launch_ms'],5), 'windows' if 'x windows' in d['config']['spmv_operator'] else 'no windows')"
done; done

#!/bin/bash
# Jacobi against Chebyshev-Jacobi (library defaults; and with Gershgorin's bound alone) on the other BASELINE shapes
mkdir -p gpurun_out
out=gpurun_out/ab_pc_configs.log
: > $out
run() { echo "== $*" >> $out; python bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_other_configs --no_alt_pc "$@" 2>&1 | tail -1 | python -c '
import json,sys
d=json.loads(sys.stdin.read())
s=d["phases_ms"]["ZZZ Solve"]; it=d["config"]["krylov_iterations"]
print({"solve_ms":s,"its":it,"ms_per_step":d["ms_per_step"],"rel":d["config"]["relative_residual"]})' >> $out 2>&1; }
for cfg in c1 c2 c4_total c5_rank; do
  run --config $cfg
  run --config $cfg --pc chebyshev_jacobi
  ZZZ_CHEB_ESTEIG=-1 run --config $cfg --pc chebyshev_jacobi --pc_esteig -1
done
run --ndofs 1250000 --force_comm --cg classical --pc chebyshev_jacobi
cat $out

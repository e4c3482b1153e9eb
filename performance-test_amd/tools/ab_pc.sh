#!/bin/bash
# A/B of the preconditioner on one GPU: Jacobi against the Chebyshev-Jacobi polynomial at several degrees / ratios.
# usage: tools/ab_pc.sh [extra bench.py flags]   (writes gpurun_out/ab_pc.log)
mkdir -p gpurun_out
out=gpurun_out/ab_pc.log
: > $out
run() { echo "== $*" >> $out; python bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_other_configs "$@" 2>&1 | tail -1 | python -c '
import json,sys
d=json.loads(sys.stdin.read())
print({"value":d["value"],"solve_ms":d["phases_ms"]["ZZZ Solve"],"its":d["config"]["krylov_iterations"],"rel":d["config"]["relative_residual"],"cg":d["config"].get("cg_form")})' >> $out 2>&1; }
run "$@"
for dr in "1 30" "2 10" "2 30" "3 30" "3 60" "4 60" "5 100" "6 100" "8 200"; do set -- $dr; run --pc chebyshev_jacobi --pc_degree $1 --pc_ratio $2 "${EXTRA[@]}"; done
cat $out

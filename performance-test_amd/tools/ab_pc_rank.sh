#!/bin/bash
# Jacobi (both CG forms) against the Chebyshev-Jacobi polynomial at the 8-GPU per-rank size (1.25 M rows) with the
# communication path attached on one GPU (--force_comm: 1-rank communicator + peer-memory mailboxes).
mkdir -p gpurun_out
out=gpurun_out/ab_pc_rank.log
: > $out
run() { echo "== $*" >> $out; python bench.py --ndofs 1250000 --force_comm --steps 3 --warmup 1 --no_cpu_baseline --no_other_configs "$@" 2>&1 | tail -1 | python -c '
import json,sys
d=json.loads(sys.stdin.read())
s=d["phases_ms"]["ZZZ Solve"]; it=d["config"]["krylov_iterations"]
print({"solve_ms":s,"its":it,"us_per_it":1e3*s/it,"rel":d["config"]["relative_residual"]})' >> $out 2>&1; }
run --cg classical
run --cg single_reduction
for dr in "2 30" "3 60" "4 60" "5 100"; do set -- $dr; run --cg classical --pc chebyshev_jacobi --pc_degree $1 --pc_ratio $2; run --cg single_reduction --pc chebyshev_jacobi --pc_degree $1 --pc_ratio $2; done
cat $out

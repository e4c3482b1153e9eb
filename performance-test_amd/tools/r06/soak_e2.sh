#!/bin/bash
# the two-ranks-on-one-GPU hazard (a device-wide wait inside a lazily built product form while the other rank polls): elasticity
# P2 and Poisson P3 on 2 / 3 ranks, several times each
cd $GRAFT_REPO_ROOT
fail=0
run() { timeout 300 ./performance-test_amd/dolfinx-scaling-test "$@" -ksp_type cg -pc_type jacobi -ksp_rtol 1e-8 2>&1; }
for rep in 1 2 3 4; do
for cfg in "elasticity 2 2400000" "poisson 3 5000000" "elasticity 1 1500000"; do
  set -- $cfg
  for extra in "--ngpus 2 --comm local" "--ngpus 3 --comm local --allreduce comm" "--ngpus 2 --comm local -ksp_cg_single_reduction"; do
    b=$(run --problem_type $1 --order $2 --scaling_type strong --ndofs $3 $extra) || { echo "FAIL $cfg $extra: $(echo "$b" | tail -2)"; fail=1; continue; }
    echo "rep $rep $cfg [$extra]: its $(echo "$b" | grep "Krylov iterations" | awk '{print $NF}') solve $(echo "$b" | grep "^ZZZ Solve" | awk '{print $NF}')"
  done
done
done
exit $fail

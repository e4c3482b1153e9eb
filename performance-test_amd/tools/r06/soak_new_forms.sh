#!/bin/bash
# Round 6: the driver on sizes where the new product forms engage (block rows from 100 000 nodes, block windows from 2 000 000
# rows), one rank against two and three ranks on one GPU (host-mediated communicator; --allreduce peer and comm).
cd $GRAFT_REPO_ROOT
fail=0
run() { ./performance-test_amd/dolfinx-scaling-test "$@" -ksp_type cg -pc_type jacobi -ksp_rtol 1e-8 2>&1; }
for cfg in "poisson 3 5000000" "poisson 2 4500000" "elasticity 1 1500000" "elasticity 2 2400000" "elasticity 3 2100000"; do
  set -- $cfg
  a=$(run --problem_type $1 --order $2 --scaling_type strong --ndofs $3) || { echo "FAIL $cfg x1: $(echo "$a" | tail -2)"; fail=1; continue; }
  ia=$(echo "$a" | grep "Krylov iterations" | awk '{print $NF}'); na=$(echo "$a" | grep "Solution norm" | awk '{print $NF}')
  sa=$(echo "$a" | grep "^ZZZ Solve" | awk '{print $NF}')
  for extra in "--ngpus 2 --comm local" "--ngpus 3 --comm local --allreduce comm" "--ngpus 2 --comm local -ksp_cg_single_reduction"; do
    b=$(run --problem_type $1 --order $2 --scaling_type strong --ndofs $3 $extra) || { echo "FAIL $cfg $extra: $(echo "$b" | tail -2)"; fail=1; continue; }
    ib=$(echo "$b" | grep "Krylov iterations" | awk '{print $NF}'); nb=$(echo "$b" | grep "Solution norm" | awk '{print $NF}')
    ok=$(python3 -c "print(int(abs($ia-$ib)<=2 and abs($na-$nb)<=1e-6*abs($na)))")
    echo "$cfg [$extra]: its $ia / $ib  norm $na / $nb  solve(1 rank) $sa s  ok=$ok"
    [ "$ok" = 1 ] || fail=1
  done
done
exit $fail

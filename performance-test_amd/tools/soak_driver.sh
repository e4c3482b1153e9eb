#!/bin/bash
# One-off robustness sweep of the driver: every problem type and order, three sizes, one rank and two ranks
# (host-mediated communicator on one GPU).  Prints iteration counts and norms; exits non-zero on any failure
# or on a 1-rank / 2-rank mismatch.
cd $GRAFT_REPO_ROOT
fail=0
for pt in poisson elasticity cgpoisson; do for o in 1 2 3; do for n in 2000 150000 2000000; do
  [ $pt = elasticity ] && [ $o = 3 ] && [ $n = 2000000 ] && n=600000
  a=$(./performance-test_amd/dolfinx-scaling-test --problem_type $pt --order $o --scaling_type strong --ndofs $n -ksp_type cg -pc_type jacobi -ksp_rtol 1e-8 2>&1) || { echo "FAIL $pt P$o $n x1"; fail=1; continue; }
  b=$(./performance-test_amd/dolfinx-scaling-test --problem_type $pt --order $o --scaling_type strong --ndofs $n --ngpus 2 --comm local -ksp_type cg -pc_type jacobi -ksp_rtol 1e-8 2>&1) || { echo "FAIL $pt P$o $n x2: $(echo "$b" | tail -2)"; fail=1; continue; }
  ia=$(echo "$a" | grep "Krylov iterations" | awk '{print $NF}'); na=$(echo "$a" | grep "Solution norm" | awk '{print $NF}')
  ib=$(echo "$b" | grep "Krylov iterations" | awk '{print $NF}'); nb=$(echo "$b" | grep "Solution norm" | awk '{print $NF}')
  ok=$(python3 -c "print(int(abs($ia-$ib)<=2 and abs($na-$nb)<=1e-6*abs($na)))")
  echo "$pt P$o ndofs~$n: its $ia / $ib  norm $na / $nb  ok=$ok"
  [ "$ok" = 1 ] || fail=1
done; done; done
exit $fail

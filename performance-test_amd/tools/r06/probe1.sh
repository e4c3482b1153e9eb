set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06a
mkdir -p $O
EXE=$R/performance-test_amd/dolfinx-scaling-test
ARGS="--problem_type poisson --order 1 --scaling_type strong --ndofs 10000000 -ksp_type cg -pc_type jacobi -ksp_rtol 1e-8"
$EXE $ARGS > $O/driver_c2_1.log 2>&1
$EXE $ARGS > $O/driver_c2_2.log 2>&1
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --hip-trace --kernel-trace --stats --output-format csv -d $O/trace -- $EXE $ARGS > $O/driver_c2_trace.log 2>&1
ls -R $O/trace | head -30

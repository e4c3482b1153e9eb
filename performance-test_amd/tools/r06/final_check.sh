set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06t
mkdir -p $O
cd $R
timeout 1500 python3 -m pytest tests -q -m gpu -x --durations=15 > $O/pytest_gpu.log 2>&1
tail -25 $O/pytest_gpu.log
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
bash performance-test_amd/tools/r06/soak_new_forms.sh > $O/soak_new_forms.txt 2>&1; echo soak rc=$?; tail -16 $O/soak_new_forms.txt

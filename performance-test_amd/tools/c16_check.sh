cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -x -q -m gpu -k "packed or variants or golden or oracle_on_host" 2>&1 | tail -8
python performance-test_amd/tools/ab_spmv.py
python performance-test_amd/tools/ab_spmv.py 1250000
bash performance-test_amd/tools/sweep_configs.sh

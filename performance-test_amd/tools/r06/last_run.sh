R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06y
mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_gpu_block_rows.py -q -m gpu -x > $O/pytest.log 2>&1
tail -4 $O/pytest.log
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -c 300 $O/bench_default.json

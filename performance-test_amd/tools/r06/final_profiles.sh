#!/bin/bash
# Round 6, final: the profile set of profile_all.sh, the driver's command (python bench.py), the soak runs.
R=$GRAFT_REPO_ROOT
T=$R/performance-test_amd/tools
bash $T/r06/profile_all.sh > $R/gpurun_out/r06_final_profile_all.log 2>&1
mkdir -p $R/gpurun_out/r06z
cd $R
python3 bench.py > gpurun_out/r06z/bench_default.json 2> gpurun_out/r06z/bench_default.err
tail -c 400 gpurun_out/r06z/bench_default.json
bash $T/r06/soak_new_forms.sh > gpurun_out/r06z/soak_new_forms.txt 2>&1; echo soak rc=$?
bash $T/r06/soak_e2.sh > gpurun_out/r06z/soak_e2.txt 2>&1; echo soak_e2 rc=$?

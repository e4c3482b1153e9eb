#!/bin/bash
# Round 6 profiles (GPU box): kernel statistics + PMC passes of the C2 headline command and of the c4_total / c5_rank records,
# SQ counters of the new product kernels.  Results under gpurun_out/; merged into profiles/ by tools/profile_summarise.py here.
R=$GRAFT_REPO_ROOT
T=$R/performance-test_amd/tools
bash $T/profile_only.sh r06 c5_rank > /dev/null 2>&1
bash $T/profile_only.sh r06 c4_total > /dev/null 2>&1
bash $T/profile_bench.sh r06 c2 > /dev/null 2>&1
PMC_TIMEOUT=150 bash $T/pmc_product.sh r06 c4 > $R/gpurun_out/pmcprod_r06_c4.log 2>&1
PMC_TIMEOUT=150 bash $T/pmc_product.sh r06 c5rank > $R/gpurun_out/pmcprod_r06_c5rank.log 2>&1
PMC_TIMEOUT=150 bash $T/pmc_product.sh r06 c2 > $R/gpurun_out/pmcprod_r06_c2.log 2>&1
ls $R/gpurun_out | grep r06 | head -30

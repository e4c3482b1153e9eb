#!/bin/bash
# A/B of two BUILDS of libzzz_hip.so on one record of bench.py's other_configs, alternating on one box:
#   ab_product.sh <other libzzz_hip.so> <record> [record ...]   (the other build lacks newer entry points: the Python mirror
#   tolerates that only for records that do not call them -- set ZZZ_AB_OLD=1 to skip zzz_spmv_values_info)
cd "$(dirname "$0")/../.."
OTHER=$1; shift
for rec in "$@"; do
  for rep in 1 2; do
    for lib in new old; do
      if [ $lib = old ]; then export ZZZ_HIP_LIB=$OTHER ZZZ_AB_OLD=1; else unset ZZZ_HIP_LIB ZZZ_AB_OLD; fi
      python bench.py --only $rec 2>&1 | tail -1 | python -c "
import sys, json
d = list(json.loads(sys.stdin.read()).values())[0]
print('$rec $lib', round(d['ms_per_step'], 1), 'ms/step product_ms', round(d['product_ms'], 4), 'bytes', d['product_bytes_per_launch'], 'its', d['krylov_iterations'])"
    done
  done
done

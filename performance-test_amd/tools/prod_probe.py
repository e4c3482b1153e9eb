#!/usr/bin/env python3
"""One process = one device-generated matrix + N launches of the CG product (zzz_spmv_time), for rocprofv3 passes and quick
timing: prod_probe.py <case> [reps] [rounds]   (cases as in ab_sellp.py; prints median / min ms and the stream's form)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zzz  # noqa: E402
from ab_sellp import CASES  # noqa: E402

case = sys.argv[1] if len(sys.argv) > 1 else "c2"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
problem, order, ndofs, bs = CASES[case]
nx, ny, nz, r = zzz.mesh_size(ndofs, True, 1, bs, order)
form = zzz.FORM_POISSON if problem == "poisson" else zzz.FORM_ELASTICITY
with zzz.Context(0) as ctx:
    ctx.cube_generate(problem, order, nx << r, ny << r, nz << r, 1, 0)
    ctx.pattern_build()
    ctx.assemble_matrix(form)
    ctx.assemble_vector(form)
    ctx.cg_solve(max_it=3)  # p, w hold something non-trivial; builds the stream and its dictionaries
    t = [ctx.spmv_time(reps=reps, variant=int(os.environ.get("PROBE_VARIANT", "-1"))) for _ in range(rounds)]
    info = ctx.spmv_info_raw()
    vi = ctx.spmv_values_info()
    nrows, _, nnz = ctx.csr_sizes()
    byts = info[6] + 16 * nrows
    print(f"PROBE {case} rows {nrows} nnz {nnz} stream {info[6] / 1e6:.1f} MB ({vi['form']}, {vi['distinct_values']} values) "
          f"product median {1e3 * np.median(t):.1f} us min {1e3 * min(t):.1f} us -> {byts / np.median(t) / 1e9:.3f} TB/s "
          f"= {byts / np.median(t) / 8e9:.3f} of peak", flush=True)

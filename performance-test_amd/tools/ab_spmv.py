#!/usr/bin/env python3
"""A/B timing of the CG SpMV kernel variants on the BASELINE 10 M-dof matrix, interleaved rounds in
one process (cdna_hip_programming.md rule 24).  Usage: python performance-test_amd/tools/ab_spmv.py [ndofs]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zzz  # noqa: E402

ndofs = int(sys.argv[1]) if len(sys.argv) > 1 else 10000000
nx, ny, nz, r = zzz.mesh_size(ndofs, True, 1, 1, 1)
P = zzz.Part("poisson", 1, nx << r, ny << r, nz << r)
res = {}
os.environ["ZZZ_SPMV_VARIANT"] = "9"  # builds the operator stream too
for tile in (2048,):
    os.environ["ZZZ_SPMV_TILE"] = str(tile)
    with zzz.Context(0) as ctx:
        ctx.upload_part(P)
        ctx.pattern_build()
        ctx.assemble_matrix(zzz.FORM_POISSON)
        ctx.assemble_vector(zzz.FORM_POISSON)
        ctx.cg_solve(max_it=3)  # fills p with something non-trivial
        nrows, _, nnz = ctx.csr_sizes()
        alg = 12 * nnz + 4 * (nrows + 1) + 16 * nrows
        for rnd in range(6):
            for var in (0, 1, 8, 9, 16, 17):  # bit 0 nt, bit 1 pipelined, bit 3 operator stream, 16 = int32 columns
                ms = ctx.spmv_time(reps=30, variant=var)
                res.setdefault((tile, var), []).append(ms)
for (tile, var), v in sorted(res.items()):
    v = np.array(v)
    print(f"tile {tile} variant {var} (nt={var & 1}, pipe={(var >> 1) & 1}, stream={(var >> 3) & 1}): median {np.median(v):.4f} ms  min {v.min():.4f} ms"
          f"  -> {alg / np.median(v) / 1e6:.0f} GB/s algorithmic")

#!/usr/bin/env python3
"""A/B of the CG operator forms on device-generated BASELINE-sized matrices, interleaved rounds in one
process: CSR tile kernel (variant 1/0) against the sliced-ELL operator stream (variant 9/8), natural and
length-sorted row order; checks that both give the same bits.
Usage: python performance-test_amd/tools/ab_sellp.py [case ...]   (cases: c2 c1 rank c4 c5rank p2 e3)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zzz  # noqa: E402

CASES = {
    "c2": ("poisson", 1, 10000000, 1),
    "c1": ("poisson", 1, 500000, 1),
    "rank": ("poisson", 1, 1250000, 1),
    "c4": ("elasticity", 1, 4000000, 3),
    "c5rank": ("poisson", 3, 6250000, 1),
    "c5": ("poisson", 3, 50000000, 1),
    "p2": ("poisson", 2, 5000000, 1),
    "e3": ("elasticity", 3, 1000000, 3),
    "e2": ("elasticity", 2, 2000000, 3),
}


def run(case, mode):
    problem, order, ndofs, bs = CASES[case]
    nx, ny, nz, r = zzz.mesh_size(ndofs, True, 1, bs, order)
    os.environ["ZZZ_SPMV_VARIANT"] = "9"
    os.environ["ZZZ_SELLP"] = str(mode)
    out = {}
    with zzz.Context(0) as ctx:
        ctx.cube_generate(problem, order, nx << r, ny << r, nz << r, 1, 0)
        ctx.pattern_build()
        form = zzz.FORM_POISSON if problem == "poisson" else zzz.FORM_ELASTICITY
        ctx.assemble_matrix(form)
        ctx.assemble_vector(form)
        nrows, _, nnz = ctx.csr_sizes()
        info = (zzz.C.c_int64 * 8)()
        ctx._ck(ctx.L.zzz_spmv_info(ctx.h, info))
        alg = 12 * nnz + 4 * (nrows + 1) + 16 * nrows
        xv = np.random.default_rng(5).standard_normal(nrows)
        os.environ["ZZZ_SPMV_VARIANT"] = "9"
        ys = {}
        for var in (1, 9):
            ctx.spmv_time(reps=1, variant=var)
        # bit parity of the two operator forms through zzz_spmv (uses the context's variant = 9 -> stream)
        y_stream = ctx.spmv(xv)
        ctx.cg_solve(max_it=3)
        for rnd in range(5):
            for var in (1, 9) if nnz * 12 > 3e8 else (0, 8):
                out.setdefault(var, []).append(ctx.spmv_time(reps=30, variant=var))
        it, rn, r0 = ctx.cg_solve(rtol=1e-8)
        print(f"[{case} mode {mode}] rows {nrows} nnz {nnz} ({nnz / nrows:.1f}/row) stream: form {info[5]} bytes {info[6] / 1e6:.1f} MB "
              f"entries {info[7]} ({info[7] / nnz:.3f} of the pattern); iterations {it}")
        for var, v in sorted(out.items()):
            v = np.array(v)
            streamed = (info[6] if var & 8 else 10 * nnz + 4 * nrows) + 16 * nrows
            print(f"   variant {var:2d} ({'stream' if var & 8 else 'tile  '}): median {np.median(v) * 1e3:8.1f} us  min {v.min() * 1e3:8.1f} us"
                  f"  -> {alg / np.median(v) / 1e6:7.0f} GB/s algorithmic, {streamed / np.median(v) / 1e6:7.0f} GB/s streamed")
    return y_stream


def main():
    cases = sys.argv[1:] or ["c2"]
    for case in cases:
        y_ref = None
        for mode in (2, 3):
            y = run(case, mode)
            if y_ref is None:
                # reference bits: the tile kernel on the same matrix
                os.environ["ZZZ_SPMV_VARIANT"] = "1"
                problem, order, ndofs, bs = CASES[case]
                nx, ny, nz, r = zzz.mesh_size(ndofs, True, 1, bs, order)
                with zzz.Context(0) as ctx:
                    ctx.cube_generate(problem, order, nx << r, ny << r, nz << r, 1, 0)
                    ctx.pattern_build()
                    ctx.assemble_matrix(zzz.FORM_POISSON if problem == "poisson" else zzz.FORM_ELASTICITY)
                    nrows, _, nnz = ctx.csr_sizes()
                    y_ref = ctx.spmv(np.random.default_rng(5).standard_normal(nrows))
            same = np.array_equal(y, y_ref)
            print(f"   [{case} mode {mode}] stream == tile kernel bit for bit: {same}; max |diff| {np.abs(y - y_ref).max():.3e}")


if __name__ == "__main__":
    main()

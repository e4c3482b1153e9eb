#!/usr/bin/env python3
"""A/B of the pipelined product (zzz_sellp_pipe.hip) against the generic one, alternating in ONE process, and their products
compared bit for bit: ab_pipe.py [case ...]   (cases as in ab_sellp.py)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zzz  # noqa: E402
from ab_sellp import CASES  # noqa: E402

VARIANTS = os.environ.get("AB_VARIANTS", "0,1").split(",")
for case in sys.argv[1:] or ["c2"]:
    problem, order, ndofs, bs = CASES[case]
    nx, ny, nz, r = zzz.mesh_size(ndofs, True, 1, bs, order)
    form = zzz.FORM_POISSON if problem == "poisson" else zzz.FORM_ELASTICITY
    res, ys, its = {}, {}, {}
    for rnd in range(2):
        for v in VARIANTS:
            os.environ["ZZZ_SELLP_PIPE"] = v
            with zzz.Context(0) as ctx:
                ctx.cube_generate(problem, order, nx << r, ny << r, nz << r, 1, 0)
                ctx.pattern_build()
                ctx.assemble_matrix(form)
                ctx.assemble_vector(form)
                ctx.cg_solve(max_it=3)
                for k in range(3):
                    res.setdefault(v, []).append(ctx.spmv_time(reps=30))
                if rnd == 0:
                    nrows = ctx.csr_sizes()[0]
                    ys[v] = ctx.spmv(np.cos(0.37 * np.arange(nrows)))
                    its[v] = ctx.cg_solve(max_it=40)[1]
                    info = ctx.spmv_info_raw()
                    vi = ctx.spmv_values_info()
    byts = info[6] + 16 * nrows
    for v in VARIANTS:
        t = np.median(res[v])
        print(f"[{case}] pipe={v}: median {1e3 * t:.1f} us  min {1e3 * min(res[v]):.1f} us -> {byts / t / 1e9:.3f} TB/s = {byts / t / 8e9:.3f} of peak"
              f"  ({vi['form']}, stream {info[6] / 1e6:.1f} MB)")
    print(f"[{case}] products bit-identical: {all(np.array_equal(ys[VARIANTS[0]], ys[v]) for v in VARIANTS)}; residuals after 40 iterations: {[its[v] for v in VARIANTS]}", flush=True)

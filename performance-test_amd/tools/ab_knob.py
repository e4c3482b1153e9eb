#!/usr/bin/env python3
"""A/B of one environment knob of the product kernel, both settings alternating in ONE process (boxes and even
consecutive runs on one box differ by up to 10 %): usage ab_knob.py KNOB v1 v2 [case]   (cases as in ab_sellp.py)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zzz  # noqa: E402
from ab_sellp import CASES  # noqa: E402

knob, v1, v2 = sys.argv[1:4]
case = sys.argv[4] if len(sys.argv) > 4 else "c2"
problem, order, ndofs, bs = CASES[case]
nx, ny, nz, r = zzz.mesh_size(ndofs, True, 1, bs, order)
form = zzz.FORM_POISSON if problem == "poisson" else zzz.FORM_ELASTICITY
res = {}
for rnd in range(3):
    for v in (v1, v2):
        os.environ[knob] = v
        with zzz.Context(0) as ctx:
            ctx.cube_generate(problem, order, nx << r, ny << r, nz << r, 1, 0)
            ctx.pattern_build()
            ctx.assemble_matrix(form)
            ctx.assemble_vector(form)
            ctx.cg_solve(max_it=3)
            for k in range(3):
                res.setdefault(v, []).append(ctx.spmv_time(reps=30))
            res["b" + v] = ctx.spmv_info_raw()[6]
for v in (v1, v2):
    t = np.median(res[v])
    print(f"[{case}] {knob}={v}: median {1e3 * t:.1f} us  min {1e3 * min(res[v]):.1f} us  stream {res['b' + v] / 1e6:.1f} MB")

#!/usr/bin/env python3
"""A/B of one environment knob that zzz_cg_solve reads at solve time: both settings alternate on ONE assembled
system in ONE process; reports the wall time of `ZZZ Solve` and checks that the iterates agree bit for bit.
Usage: ab_solve.py KNOB v1 v2 [case] [single]   (cases as in ab_sellp.py)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zzz  # noqa: E402
from ab_sellp import CASES  # noqa: E402

knob, v1, v2 = sys.argv[1:4]
case = sys.argv[4] if len(sys.argv) > 4 else "c2"
single = len(sys.argv) > 5 and sys.argv[5] == "single"
problem, order, ndofs, bs = CASES[case]
nx, ny, nz, r = zzz.mesh_size(ndofs, True, 1, bs, order)
form = zzz.FORM_POISSON if problem == "poisson" else zzz.FORM_ELASTICITY
res, sol = {}, {}
with zzz.Context(0) as ctx:
    ctx.cube_generate(problem, order, nx << r, ny << r, nz << r, 1, 0)
    ctx.pattern_build()
    ctx.assemble_matrix(form)
    ctx.assemble_vector(form)
    ctx.cg_solve(max_it=20)
    for rnd in range(4):
        for v in (v1, v2):
            os.environ[knob] = v
            t0 = time.perf_counter()
            it, rn, r0 = ctx.cg_solve(rtol=1e-8, single_reduction=single)
            res.setdefault(v, []).append((time.perf_counter() - t0, it))
            if v not in sol:
                sol[v] = ctx.vec_download(zzz.VEC_U)
for v in (v1, v2):
    t = np.array([a for a, _ in res[v]])
    it = res[v][0][1]
    print(f"[{case}] {knob}={v}: solve median {1e3 * np.median(t):.2f} ms  min {1e3 * t.min():.2f} ms; {it} iterations, "
          f"{1e6 * np.median(t) / it:.2f} us per iteration")
print("solutions identical bit for bit:", np.array_equal(sol[v1], sol[v2]), " max |diff|", np.abs(sol[v1] - sol[v2]).max())

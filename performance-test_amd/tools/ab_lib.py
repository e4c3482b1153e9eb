#!/usr/bin/env python3
"""A/B of two BUILDS of libzzz_hip.so on the assembly phases: child processes alternate between the two libraries
(ZZZ_HIP_LIB), each times pattern build, matrix assembly and vector assembly on one device-generated system.
Usage: ab_lib.py <other libzzz_hip.so> [case ...]   (cases as in ab_sellp.py; the in-tree build is "new")"""
import os
import subprocess
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))


def child(case):
    import zzz
    from ab_sellp import CASES
    problem, order, ndofs, bs = CASES[case]
    nx, ny, nz, r = zzz.mesh_size(ndofs, True, 1, bs, order)
    form = zzz.FORM_POISSON if problem == "poisson" else zzz.FORM_ELASTICITY
    with zzz.Context(0) as ctx:
        ctx.cube_generate(problem, order, nx << r, ny << r, nz << r, 1, 0)
        t = {"pattern": [], "matrix": [], "vector": []}
        for rep in range(6):
            for name, fn in (("pattern", ctx.pattern_build), ("matrix", lambda: ctx.assemble_matrix(form)),
                             ("vector", lambda: ctx.assemble_vector(form))):
                ctx.sync()
                t0 = time.perf_counter()
                fn()
                ctx.sync()
                t[name].append(time.perf_counter() - t0)
        nrm = ctx.vec_norm(zzz.VEC_B)
        rowptr, cols, vals = ctx.csr_download()
        print("RES", " ".join(f"{1e3 * np.median(v[1:]):.4f}" for v in t.values()), repr(nrm), repr(float(np.abs(vals).sum())))


def main():
    other = os.path.abspath(sys.argv[1])
    cases = sys.argv[2:] or ["c2"]
    for case in cases:
        res = {"old": [], "new": []}
        for rnd in range(3):
            for tag in ("old", "new"):
                env = dict(os.environ)
                if tag == "old":
                    env["ZZZ_HIP_LIB"] = other
                else:
                    env.pop("ZZZ_HIP_LIB", None)
                out = subprocess.run([sys.executable, __file__, "--child", case], env=env, capture_output=True, text=True, timeout=900)
                line = [ln for ln in out.stdout.splitlines() if ln.startswith("RES")]
                if not line:
                    print(out.stdout[-2000:], out.stderr[-2000:])
                    sys.exit(1)
                res[tag].append(line[0].split()[1:])
        for tag in ("old", "new"):
            a = np.array([[float(x) for x in r[:3]] for r in res[tag]])
            print(f"[{case}] {tag}: pattern {np.median(a[:, 0]):.3f} ms  matrix (+ stream packing) {np.median(a[:, 1]):.3f} ms  "
                  f"vector {np.median(a[:, 2]):.3f} ms   |b| {res[tag][0][3]}  sum|A| {res[tag][0][4]}")
        print(f"[{case}] identical |b| and sum|A|:", res["old"][0][3:] == res["new"][0][3:])


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(sys.argv[2])
    else:
        main()

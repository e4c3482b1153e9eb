"""Timing probe of an LDS x window for the product (ZZZ_EXP_WIN): product time with and without, per BASELINE shape.
The probe kernel's result is wrong by construction; only its duration is read."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import zzz  # noqa: E402

for problem, order, ndofs, strong, nproc in (("poisson", 3, 6250000, True, 1), ("elasticity", 1, 500000, False, 8),
                                             ("poisson", 2, 5000000, True, 1), ("poisson", 1, 10000000, True, 1)):
    bs = 3 if problem == "elasticity" else 1
    nx, ny, nz, r = zzz.mesh_size(ndofs, strong, nproc, bs, order)
    os.environ["ZZZ_SELLP_WIN"] = "0"  # the probe runs on window-free streams
    with zzz.Context(0) as c:
        c.cube_generate(problem, order, nx << r, ny << r, nz << r, 1, 0)
        c.pattern_build()
        c.assemble_matrix(zzz.FORM_ELASTICITY if bs == 3 else zzz.FORM_POISSON)
        os.environ.pop("ZZZ_EXP_WIN", None)
        base = min(c.spmv_time(20) for _ in range(3))
        out = {}
        for w in (1024, 2048, 3072, 4096, 6144):
            os.environ["ZZZ_EXP_WIN"] = str(w)
            out[w] = min(c.spmv_time(20) for _ in range(3))
        # partial windows: only some slots of every chunk gather from LDS
        for w, k in ((2048, 4), (2048, 5), (3072, 5), (3072, 6)):
            os.environ["ZZZ_EXP_WIN"], os.environ["ZZZ_EXP_WIN_SLOTS"] = str(w), str(k)
            out[(w, k)] = min(c.spmv_time(20) for _ in range(3))
        os.environ.pop("ZZZ_EXP_WIN_SLOTS", None)
        os.environ.pop("ZZZ_EXP_WIN", None)
        print(problem, order, ndofs, "product ms", round(base, 4), "with window", {w: round(v, 4) for w, v in out.items()}, flush=True)

#!/usr/bin/env python3
"""Times the matrix-free action (csrc/zzz_matfree.hip) and the cg.h solve on it at the cgpoisson sizes of BASELINE's
configs: P1 10 M dofs (c2's mesh) and P3 6.2 M dofs (c5's per-GPU share).  Knobs through the environment:
ZZZ_MF_NC / ZZZ_MF_T / ZZZ_MF_LDS_KB (plan geometry), ZZZ_MF_LEGACY=1 (the two-pass kernels of rounds 1-3).
usage: mf_bench.py [p1|p2|p3|small] ..."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import zzz  # noqa: E402

CASES = {"p1": (1, 10000000), "p2": (2, 5000000), "p3": (3, 6250000), "small": (1, 500000), "p3s": (3, 500000)}


def run(name):
    order, ndofs = CASES[name]
    nx, ny, nz, r = zzz.mesh_size(ndofs, True, 1, 1, order)
    nx, ny, nz = nx << r, ny << r, nz << r
    out = {"case": name, "order": order, "mesh": [nx, ny, nz], "env": {k: v for k, v in os.environ.items() if k.startswith("ZZZ_MF")}}
    with zzz.Context(0) as ctx:
        info = ctx.cube_generate("poisson", order, nx, ny, nz)
        n = int(info[2])
        out["dofs"] = n
        out["cells"] = int(info[1])
        legacy = os.environ.get("ZZZ_MF_LEGACY", "0") != "0"
        t0 = time.perf_counter()
        if legacy:
            ctx.pattern_build()
        else:
            ctx.matfree_setup()
        ctx.sync()
        out["setup_ms"] = (time.perf_counter() - t0) * 1e3
        if not legacy:
            t0 = time.perf_counter()
            ctx.matfree_setup()
            ctx.sync()
            out["setup_warm_ms"] = (time.perf_counter() - t0) * 1e3
            out["plan"] = ctx.matfree_info()
        if not legacy:
            ctx.pattern_build()  # assemble_vector wants the adjacency
        ctx.assemble_vector(zzz.FORM_POISSON)
        ms = ctx.action_time(20)
        out["action_ms"] = ms
        if not legacy:
            out["action_GBs"] = out["plan"]["bytes_per_action"] / ms / 1e6
        t0 = time.perf_counter()
        k, rr, rr0 = ctx.cg_solve(variant=zzz.CG_CGH, pc=zzz.PC_NONE, op=zzz.OP_MATFREE, rtol=1e-6, max_it=100, profile=True)
        ctx.sync()
        dt = time.perf_counter() - t0
        out["cg_iterations"] = k
        out["cg_s"] = dt
        out["Gdofs"] = k * n / dt / 1e9  # src/cgpoisson_problem.cpp:236-241
        out["rel_res2"] = rr / rr0 if rr0 else None
        out["unorm"] = ctx.vec_norm(zzz.VEC_U)
    print(json.dumps(out))


if __name__ == "__main__":
    for a in sys.argv[1:] or ["p1", "p3"]:
        run(a)

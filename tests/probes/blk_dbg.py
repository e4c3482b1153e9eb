import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "performance-test_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, zzz
def p(*a):
    print(*a, flush=True)
blk = sys.argv[1]
os.environ["ZZZ_SELLP_BLK"] = blk
if os.environ.get("DBG_SELLP2"): os.environ["ZZZ_SELLP"] = "2"
dims = tuple(int(v) for v in sys.argv[2:5])
P = zzz.Part("elasticity", 1, *dims)
with zzz.Context(0) as ctx:
    p("ctx")
    ctx.upload_part(P); p("upload")
    ctx.pattern_build(); p("pattern")
    ctx.assemble_matrix(zzz.FORM_ELASTICITY); ctx.sync(); p("matrix")
    vi = ctx.spmv_values_info(); p("info", vi)
    x = np.random.default_rng(3).standard_normal(P.n_owned * 3)
    y = ctx.spmv(x); p("spmv", np.abs(y).max())
    rp, cl, v = ctx.csr_download()
    import scipy.sparse as sp
    A = sp.csr_matrix((v, cl, rp), shape=(len(rp) - 1, len(rp) - 1))
    p("diff vs scipy", np.abs(A @ x - y).max())
    t = ctx.spmv_time(reps=5); p("time", t)
    order = 1
    ctx.assemble_vector(zzz.FORM_ELASTICITY); p("vector")
    it, rn, r0 = ctx.cg_solve(pc=zzz.PC_JACOBI, rtol=1e-8); p("solve", it, rn / r0)
    it, rn, r0 = ctx.cg_solve(pc=zzz.PC_JACOBI, rtol=1e-8, profile=True); p("solve profile", it, rn / r0, ctx.profile())
    it, rn, r0 = ctx.cg_solve(pc=zzz.PC_JACOBI, rtol=1e-8, single_reduction=True); p("solve sr", it, rn / r0)
    it, rn, r0 = ctx.cg_solve(pc=zzz.PC_CHEBYSHEV_JACOBI, rtol=1e-8); p("solve cheb", it, rn / r0)

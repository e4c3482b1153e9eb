import sys, os, numpy as np
sys.path.insert(0, "/root/repo/performance-test_amd"); sys.path.insert(0, "/root/repo/oracle")
import zzz, zzz_oracle as zo
zo.set_num_threads(8)
order = int(sys.argv[1]); dims = [int(a) for a in sys.argv[2:5]]
P = zzz.Part("poisson", order, *dims)
bc = P.bc_marker()
rng = np.random.default_rng(5)
v = rng.standard_normal(P.n_owned)
oy = zo.action_poisson(order, P.x, P.cells, P.cell_dofs, bc, v)
with zzz.Context(0) as ctx:
    ctx.upload_part(P)
    ctx.matfree_setup()
    print(ctx.matfree_info())
    y0 = None
    for rep in range(6):
        y = ctx.action(v)
        if y0 is None:
            y0 = y
        assert np.array_equal(y, y0), "not reproducible"
        err = np.abs(y - oy)
        bad = np.nonzero(err > 1e-11 * np.abs(oy).max())[0]
        print("rep", rep, "max err", err.max() / np.abs(oy).max(), "bad", bad.size, bad[:10], err[bad[:10]], oy[bad[:10]])

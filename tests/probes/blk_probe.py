"""Round 6 probe: the block-row product of block size 3 (csrc/zzz_sellp_blk.hip) against the oracle's serial CSR loop and against
the generic stream; timing of both.  Runs on the GPU box:  python tests/probes/blk_probe.py [n ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "performance-test_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np  # noqa: E402
import zzz  # noqa: E402
import zzz_oracle as zo  # noqa: E402


def run(n, order=1, check=True):
    dims = n if isinstance(n, tuple) else (n, n, n)
    P = zzz.Part("elasticity", order, *dims)
    out = {}
    for blk in ("2", "0"):
        os.environ["ZZZ_SELLP_BLK"] = blk
        with zzz.Context(0) as ctx:
            ctx.upload_part(P)
            ctx.pattern_build()
            ctx.assemble_matrix(zzz.FORM_ELASTICITY)
            ctx.assemble_vector(zzz.FORM_ELASTICITY)
            x = np.random.default_rng(3).standard_normal(P.n_owned * 3)
            y = ctx.spmv(x)
            vi = ctx.spmv_values_info()
            t = ctx.spmv_time(reps=50)
            ctx.sync()
            t0 = time.perf_counter()
            it, rn, r0 = ctx.cg_solve(pc=zzz.PC_JACOBI, rtol=1e-8, profile=True)
            ctx.sync()
            ts = time.perf_counter() - t0
            pm = ctx.profile()
            it2, _, _ = ctx.cg_solve(pc=zzz.PC_JACOBI, rtol=1e-8, single_reduction=True)
            u = ctx.vec_download(zzz.VEC_U)
            out[blk] = (y, vi, t, it, ts, pm, it2, u)
            if blk == "2" and check:
                rp, cl, v = ctx.csr_download()
                zo.set_num_threads(8)
                oy = zo.spmv(rp.astype(np.int64), cl, v, x)
                print(f"  n={dims} order={order}: product == serial CSR loop: {np.array_equal(oy, y)}  max|diff|={np.abs(oy - y).max():.3e}")
    y1, vi1, t1, it1, ts1, pm1, sr1, u1 = out["2"]
    y0, vi0, t0_, it0, ts0, pm0, sr0, u0 = out["0"]
    print(f"n={dims} order={order} rows={P.n_owned * 3}: block rows {vi1['block_rows']} form {vi1['block_form']} table {vi1['block_table_entries']} chunks {vi1['block_chunks']} "
          f"bytes {vi1['bytes_per_product']} (generic {vi0['bytes_per_product']});  same bits as generic: {np.array_equal(y0, y1)}")
    print(f"   product ms: block {t1:.4f}  generic {t0_:.4f};  in-solve {pm1[0]:.4f} / {pm0[0]:.4f};  solve {ts1 * 1e3:.1f} ms ({it1} its, sr {sr1}) / "
          f"{ts0 * 1e3:.1f} ms ({it0} its, sr {sr0});  |u1-u0|/|u0| = {np.linalg.norm(u1 - u0) / np.linalg.norm(u0):.2e}")


if __name__ == "__main__":
    sizes = [int(a) for a in sys.argv[1:]] or [6, 17, 40, 64, 109]
    run((5, 3, 4))
    run((4, 4, 3), order=2)
    for n in sizes:
        run(n, check=n <= 64 or n == 109)

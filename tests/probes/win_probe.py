"""Round 6 probe: the block-window product of long scalar rows (csrc/zzz_sellp_win.hip) against the oracle's serial CSR loop and
against the generic stream; timing of both.  GPU box:  python tests/probes/win_probe.py [order n ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "performance-test_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np  # noqa: E402
import zzz  # noqa: E402
import zzz_oracle as zo  # noqa: E402


def run(order, n, check=True):
    dims = n if isinstance(n, tuple) else (n, n, n)
    out = {}
    for win in ("2", "0"):
        os.environ["ZZZ_SELLP_BWIN"] = win
        with zzz.Context(0) as ctx:
            info = ctx.cube_generate("poisson", order, *dims, 1, 0)
            nrows = int(info[0])
            ctx.pattern_build()
            ctx.assemble_matrix(zzz.FORM_POISSON)
            ctx.assemble_vector(zzz.FORM_POISSON)
            x = np.random.default_rng(3).standard_normal(nrows)
            y = ctx.spmv(x)
            vi = ctx.spmv_values_info()
            t = ctx.spmv_time(reps=30)
            ctx.sync()
            t0 = time.perf_counter()
            it, rn, r0 = ctx.cg_solve(pc=zzz.PC_JACOBI, rtol=1e-8, profile=True)
            ctx.sync()
            ts = time.perf_counter() - t0
            pm = ctx.profile()
            its, _, _ = ctx.cg_solve(pc=zzz.PC_JACOBI, rtol=1e-8, single_reduction=True)
            u = ctx.vec_download(zzz.VEC_U)
            out[win] = (y, vi, t, it, ts, pm, its, u)
            if win == "2" and check:
                rp, cl, v = ctx.csr_download()
                zo.set_num_threads(8)
                oy = zo.spmv(rp.astype(np.int64), cl, v, x)
                print(f"  P{order} {dims}: product == serial CSR loop: {np.array_equal(oy, y)}  max|diff|={np.abs(oy - y).max():.3e}", flush=True)
    y1, vi1, t1, it1, ts1, pm1, sr1, u1 = out["2"]
    y0, vi0, t0_, it0, ts0, pm0, sr0, u0 = out["0"]
    print(f"P{order} {dims} rows={len(y1)}: form {vi1.get('special_form')} window entries {vi1['block_table_entries']} chunks {vi1['block_chunks']} "
          f"blocks {vi1['block_form']} bytes {vi1['bytes_per_product']} (generic {vi0['bytes_per_product']});  same bits as generic: {np.array_equal(y0, y1)}")
    print(f"   product ms: window {t1:.4f}  generic {t0_:.4f};  in-solve {pm1[0]:.4f} / {pm0[0]:.4f};  solve {ts1 * 1e3:.1f} ms ({it1} its, sr {sr1}) / "
          f"{ts0 * 1e3:.1f} ms ({it0} its, sr {sr0});  |u1-u0|/|u0| = {np.linalg.norm(u1 - u0) / np.linalg.norm(u0):.2e}", flush=True)


if __name__ == "__main__":
    args = [int(a) for a in sys.argv[1:]]
    cases = list(zip(args[0::2], args[1::2])) or [(3, 4), (2, 7), (3, 12), (3, 24)]
    for order, n in cases:
        run(order, n, check=n <= 40 or n == 61)

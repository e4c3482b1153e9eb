"""Shared helpers, fixtures and constants of the GPU parity tests (tests/test_gpu_*.py: split by subject in round 6 from
the single test_gpu_parity.py of rounds 1-5).  Bars: see tests/test_gpu_assembly.py."""
import glob
import os
import numpy as np
import pytest
import zzz
import zzz_oracle as zo
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = sorted(glob.glob(os.path.join(GOLD, "*_p[123]_*.npz")))
SUPPORTED_ORDERS = (1, 2, 3)


def in_tools_build(fn):
    """The measurement-only code paths (ZZZ_TAIL, ZZZ_CG_FUSED=2, pipelined / 4096-nonzero tiles, the measurement knobs) are
    compiled under -DZZZ_EXPERIMENTS into libzzz_hip_exp.so (`make exp`), not into the product library: a test of them
    re-runs itself in a child process that loads that build through ZZZ_HIP_LIB."""
    import functools
    import subprocess
    import sys

    @functools.wraps(fn)
    def wrapper(*a, **k):
        exp = os.path.join(zzz.PKG, "libzzz_hip_exp.so")
        if os.environ.get("ZZZ_HIP_LIB") == exp:
            return fn(*a, **k)
        if not os.path.exists(exp):
            pytest.skip("libzzz_hip_exp.so (make -C performance-test_amd exp) is absent")
        out = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider",
                              f"{os.path.abspath(sys.modules[fn.__module__].__file__)}::{fn.__name__}"], env=dict(os.environ, ZZZ_HIP_LIB=exp),
                             capture_output=True, text=True, timeout=1800)
        assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-1000:]
    return wrapper


@pytest.fixture(scope="module")
def ctx():
    assert zzz.device_count() >= 1, "no GPU visible: these tests must not pass on a fallback"
    zo.set_num_threads(1)
    with zzz.Context(0) as c:
        yield c


def _upload_arrays(ctx, d, order, bs, nblock):
    ctx.upload_mesh(d["x"], d["cells"])
    ctx.upload_dofmap(order, bs, d["cell_dofs"], nblock, 0)
    ctx.upload_bc(np.nonzero(d["bc"])[0].astype(np.int32))
    if bs == 1:
        ctx.upload_facets(d["facets"])
        ctx.upload_coeff(zzz.COEFF_G, d["g"])
    ctx.upload_coeff(zzz.COEFF_F, d["f"])


def _internal_system(rp, cl, v, perm, bs):
    """P A P^T: the caller-ordered CSR (rp, cl, v) in the library's internal order (perm[i] = caller block index of
    internal block i; ghost columns, if any, keep their places), columns ascending within a row"""
    import scipy.sparse as sp

    n = rp.shape[0] - 1
    ncol = max(n, int(cl.max()) + 1)
    sperm = (perm.astype(np.int64)[:, None] * bs + np.arange(bs)).reshape(-1)  # scalar internal -> caller
    inv = np.arange(ncol, dtype=np.int64)
    inv[sperm] = np.arange(n)
    # keep structural zeros: carry the entries as (value, position) through scipy by their indices
    A = sp.csr_matrix((np.arange(1, v.size + 1, dtype=np.float64), cl, rp), shape=(n, ncol))
    B = sp.csr_matrix(A[sperm])  # rows in internal order
    B = sp.csr_matrix((B.data, inv[B.indices], B.indptr), shape=(n, ncol))
    B.sort_indices()
    return B.indptr.astype(np.int64), B.indices.astype(np.int32), v[(B.data - 1).astype(np.int64)], sperm


def _spmv_variant_case(variant, tile):
    old = {k: os.environ.get(k) for k in ("ZZZ_SPMV_VARIANT", "ZZZ_SPMV_TILE")}
    os.environ["ZZZ_SPMV_VARIANT"], os.environ["ZZZ_SPMV_TILE"] = str(variant), str(tile)
    try:
        zo.set_num_threads(1)
        for problem, order, dims in (("poisson", 1, (11, 9, 10)), ("elasticity", 2, (3, 3, 4)), ("poisson", 3, (3, 4, 3))):
            P = zzz.Part(problem, order, *dims)
            with zzz.Context(0) as c:
                c.upload_part(P)
                c.pattern_build()
                c.assemble_matrix(P.form)
                c.assemble_vector(P.form)
                rp, cl, v = c.csr_download()
                rng = np.random.default_rng(variant)
                xv = rng.standard_normal(P.n_owned * P.bs)
                np.testing.assert_array_equal(c.spmv(xv), zo.spmv(rp.astype(np.int64), cl, v, xv))
                it, rn, r0 = c.cg_solve(pc=zzz.PC_JACOBI, rtol=1e-8)
                oit, ou, _, _ = zo.pcg(rp.astype(np.int64), cl, v, c.vec_download(zzz.VEC_B), rtol=1e-8)
                assert abs(it - oit) <= 2
                assert np.linalg.norm(c.vec_download(zzz.VEC_U) - ou) <= 1e-6 * np.linalg.norm(ou)
    finally:
        for k, val in old.items():
            if val is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = val


# (problem, order, n): cubes of n^3 cells either side of every size rule that picks a form of the product by itself
_SWEEP = [("poisson", 1, 40), ("poisson", 1, 56), ("poisson", 1, 66), ("poisson", 1, 84), ("poisson", 1, 124), ("poisson", 1, 150),
          ("elasticity", 1, 30), ("elasticity", 1, 50), ("elasticity", 1, 56), ("elasticity", 1, 66),
          ("poisson", 2, 16), ("poisson", 2, 24), ("poisson", 2, 32), ("poisson", 3, 8), ("poisson", 3, 12), ("poisson", 3, 18)]




__all__ = ['CASES', 'GOLD', 'SUPPORTED_ORDERS', '_SWEEP', '_internal_system', '_spmv_variant_case', '_upload_arrays', 'ctx', 'glob', 'in_tools_build', 'np', 'os', 'pytest', 'zo', 'zzz']

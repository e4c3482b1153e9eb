"""The reference tensors the HIP kernels contract with (performance-test_amd/csrc/element_tables.inc,
exact monomial integration at 50 digits) against the oracle (Gauss-Legendre quadrature, long double) and the
golden vectors (Gauss-Jacobi quadrature, barycentric basis): three independent derivations, no GPU needed.
The contraction below is the one the kernels perform (Ae = |detJ| sum_ab (K K^T)_ab S^ab, etc.)."""
import os
import re

import numpy as np
import pytest

import zzz_oracle as zo

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = os.path.join(ROOT, "performance-test_amd", "csrc", "element_tables.inc")


def _tables(order):
    src = open(INC).read()
    m = re.search(r"ZZZ_TAB_P%d\[(\d+)\] = \{(.*?)\};" % order, src, re.S)
    vals = np.array([float(v) for v in m.group(2).replace("\n", " ").split(",") if v.strip()])
    nd = zo.ndofs_cell(order)
    assert vals.size == int(m.group(1)) == 14 * nd * nd
    S = vals[:9 * nd * nd].reshape(3, 3, nd, nd)
    M = vals[9 * nd * nd:10 * nd * nd].reshape(nd, nd)
    F = vals[10 * nd * nd:].reshape(4, nd, nd)
    return nd, S, M, F


def _geom(xc):
    J = (xc[1:] - xc[0]).T  # J[a][al] = dx_a/dX_al
    K = np.linalg.inv(J)    # K[al][a] = dX_al/dx_a
    return abs(np.linalg.det(J)), K


@pytest.mark.parametrize("order", [1, 2, 3])
def test_tables_contract_to_the_oracle_element_tensors(order):
    zo.set_num_threads(1)
    nd, S, M, F = _tables(order)
    rng = np.random.default_rng(100 + order)
    for xc in (np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1.0]]), rng.random((4, 3)), rng.random((4, 3))[[1, 0, 2, 3]]):
        adet, K = _geom(xc)
        # Poisson a: |detJ| sum_{al,be} (K K^T)[al,be] S[al,be]
        GG = K @ K.T
        A = adet * np.einsum("ab,abij->ij", GG, S)
        Ao = zo.tabulate("poisson_a", order, xc)
        assert np.abs(A - Ao).max() <= 1e-13 * np.abs(Ao).max()
        # elasticity: D[c,d,i,j] = |detJ| sum K[al,c] K[be,d] S[al,be,i,j];  mu(delta_cd tr + D[d,c]) + lambda D[c,d]
        D = adet * np.einsum("ac,bd,abij->cdij", K, K, S)
        mu, lm = 1e6 / 2.6, 1e6 * 0.3 / (1.3 * 0.4)
        tr = D[0, 0] + D[1, 1] + D[2, 2]
        E = np.zeros((nd, 3, nd, 3))
        for c in range(3):
            for d in range(3):
                E[:, c, :, d] = mu * ((tr if c == d else 0) + D[d, c]) + lm * D[c, d]
        Eo = zo.tabulate("elasticity_a", order, xc)
        assert np.abs(E.reshape(3 * nd, 3 * nd) - Eo).max() <= 1e-12 * np.abs(Eo).max()
        # mass and facet mass through the L kernels
        Mo = np.array([zo.tabulate("poisson_L", order, xc, w=np.r_[np.eye(nd)[j], np.zeros(nd)]) for j in range(nd)])
        assert np.abs(adet * M - Mo).max() <= 1e-13 * np.abs(Mo).max()
        FV = [(1, 2, 3), (0, 2, 3), (0, 1, 3), (0, 1, 2)]
        for lf in range(4):
            p = xc[list(FV[lf])]
            scale = np.linalg.norm(np.cross(p[1] - p[0], p[2] - p[0]))
            Fo = np.array([zo.tabulate("poisson_L_facet", order, xc, w=np.r_[np.zeros(nd), np.eye(nd)[j]], facet=lf)
                           for j in range(nd)])
            assert np.abs(scale * F[lf] - Fo).max() <= 1e-13 * max(np.abs(Fo).max(), 1e-300)


def test_tables_against_golden_element_tensors():
    e = np.load(os.path.join(ROOT, "tests", "golden", "element_tensors.npz"))
    for order in (1, 2, 3):
        nd, S, M, F = _tables(order)
        for nm in ("ref", "tet"):
            adet, K = _geom(e[nm])
            A = adet * np.einsum("ab,abij->ij", K @ K.T, S)
            G = e[f"poisson_a_p{order}_{nm}"]
            assert np.abs(A - G).max() <= 1e-13 * np.abs(G).max()
            assert np.abs(adet * M - e[f"mass_p{order}_{nm}"]).max() <= 1e-13 * np.abs(M).max() * adet


def test_p1_tables_closed_forms():
    nd, S, M, F = _tables(1)
    np.testing.assert_allclose(120 * M, np.ones((4, 4)) + np.eye(4), atol=1e-14)
    g = np.array([[-1, -1, -1], [1, 0, 0], [0, 1, 0], [0, 0, 1.0]])
    np.testing.assert_allclose(6 * S, np.einsum("ia,jb->abij", g, g), atol=1e-14)
    ex = (np.ones((4, 4)) + np.eye(4)) / 24
    ex[3, :] = 0
    ex[:, 3] = 0
    np.testing.assert_allclose(F[3], ex, atol=1e-15)


def test_generator_is_reproducible(tmp_path):
    """The committed .inc is what tools/gen_element_tables.py writes (run it into a scratch copy)."""
    import importlib.util
    import shutil

    tool = os.path.join(ROOT, "performance-test_amd", "tools", "gen_element_tables.py")
    scratch = tmp_path / "performance-test_amd"
    (scratch / "tools").mkdir(parents=True)
    (scratch / "csrc").mkdir()
    shutil.copy(tool, scratch / "tools" / "gen_element_tables.py")
    spec = importlib.util.spec_from_file_location("gen_tab", str(scratch / "tools" / "gen_element_tables.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.main()
    assert open(scratch / "csrc" / "element_tables.inc").read() == open(INC).read()

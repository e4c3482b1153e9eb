"""Oracle (oracle/zzz_oracle.c) against the committed golden vectors (tests/golden/*.npz).

The golden vectors come from an independent numpy/scipy restatement (tests/golden/make_golden.py);
the reference itself holds none (parity unpinned).  Bars: indices bit-exact; values
1e-12 relative (integrals are exact in both, only round-off differs); solutions 1e-8 relative in l2
(north_star asks 1e-6); iteration counts within +-2 (round-off decides the last borderline step).
"""
import glob
import os

import numpy as np
import pytest

import zzz_oracle as zo

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = sorted(glob.glob(os.path.join(GOLD, "*_p[123]_*.npz")))


@pytest.fixture(autouse=True)
def _one_thread():
    zo.set_num_threads(1)


def test_fixture_inventory():
    assert len(CASES) == 6
    assert os.path.exists(os.path.join(GOLD, "element_tensors.npz"))


@pytest.mark.parametrize("fn", CASES, ids=[os.path.basename(c)[:-4] for c in CASES])
def test_oracle_matches_golden(fn):
    d = np.load(fn)
    form = zo.FORM_ELASTICITY if str(d["problem"]) == "elasticity" else zo.FORM_POISSON
    order, bs, nblock = int(d["order"]), int(d["bs"]), int(d["nblock"])
    rowptr, cols = zo.pattern(nblock, d["cell_dofs"], bs)
    np.testing.assert_array_equal(rowptr, d["rowptr"])
    np.testing.assert_array_equal(cols, d["cols"])
    vals = zo.assemble_matrix(form, order, d["x"], d["cells"], d["cell_dofs"], d["bc"], rowptr, cols)
    assert np.abs(vals - d["vals"]).max() <= 1e-12 * np.abs(d["vals"]).max()
    g = d["g"] if form == zo.FORM_POISSON else None
    facets = d["facets"] if form == zo.FORM_POISSON else None
    b = zo.assemble_vector(form, order, d["x"], d["cells"], d["cell_dofs"], d["f"], g, facets, d["bc"])
    assert np.abs(b - d["b"]).max() <= 1e-12 * np.abs(d["b"]).max()

    it, u, rn, r0 = zo.pcg(rowptr, cols, vals, b, pc=zo.PC_JACOBI, rtol=1e-8)
    assert abs(it - int(d["it_pcg"])) <= 2
    assert np.linalg.norm(u - d["u_pcg"]) <= 1e-8 * np.linalg.norm(d["u_pcg"])
    assert rn <= 1e-8 * r0

    k, u2, _ = zo.cg(rowptr, cols, vals, b, kmax=2000, rtol=1e-8)
    assert abs(k - int(d["it_cg"])) <= 2
    assert np.linalg.norm(u2 - d["u_cg"]) <= 1e-8 * np.linalg.norm(d["u_cg"])
    # true residual of the cg.h solution: north_star's 1e-8 relative residual
    r = b - zo.spmv(rowptr, cols, vals, u2)
    assert np.linalg.norm(r) <= 1.05e-8 * np.linalg.norm(b)

    k6, u6, _ = zo.cg(rowptr, cols, vals, b, kmax=100, rtol=1e-6)  # src/cgpoisson_problem.cpp:233
    assert abs(k6 - int(d["it_cg6"])) <= 2
    if k6 < 100:
        assert np.linalg.norm(u6 - d["u_cg6"]) <= 1e-6 * np.linalg.norm(d["u_cg6"])


@pytest.mark.parametrize("fn", CASES, ids=[os.path.basename(c)[:-4] for c in CASES])
def test_oracle_feed_matches_golden(fn):
    """BC location (topological), exterior facets and coefficient interpolation restated by the
    oracle on the golden's mesh reproduce the golden's (geometric) ones exactly."""
    d = np.load(fn)
    el = str(d["problem"]) == "elasticity"
    order, bs, nblock = int(d["order"]), int(d["bs"]), int(d["nblock"])
    bcm = zo.locate_bc(1 if el else 0, order, d["x"], d["cells"], d["cell_dofs"], nblock)
    np.testing.assert_array_equal(np.repeat(bcm, bs), d["bc"])
    if el:
        assert np.abs(zo.interpolate(2, d["dof_x"]) - d["f"]).max() < 1e-15
    else:
        np.testing.assert_array_equal(zo.exterior_facets(d["cells"]), d["facets"])
        assert np.abs(zo.interpolate(0, d["dof_x"]) - d["f"]).max() < 1e-14
        assert np.abs(zo.interpolate(1, d["dof_x"]) - d["g"]).max() < 1e-15


def test_element_tensors_golden():
    e = np.load(os.path.join(GOLD, "element_tensors.npz"))
    for order in (1, 2, 3):
        nd = zo.ndofs_cell(order)
        for nm in ("ref", "tet"):
            xc = e[nm]
            A = zo.tabulate("poisson_a", order, xc)
            assert np.abs(A - e[f"poisson_a_p{order}_{nm}"]).max() <= 1e-13 * np.abs(A).max()
            E = zo.tabulate("elasticity_a", order, xc)
            assert np.abs(E - e[f"elasticity_a_p{order}_{nm}"]).max() <= 1e-13 * np.abs(E).max()
            M = np.array([zo.tabulate("poisson_L", order, xc, w=np.r_[np.eye(nd)[j], np.zeros(nd)]) for j in range(nd)])
            assert np.abs(M - e[f"mass_p{order}_{nm}"]).max() <= 1e-13 * np.abs(M).max()
            for lf in range(4):
                F = np.array([zo.tabulate("poisson_L_facet", order, xc, w=np.r_[np.zeros(nd), np.eye(nd)[j]], facet=lf)
                              for j in range(nd)])
                assert np.abs(F - e[f"facet_mass{lf}_p{order}_{nm}"]).max() <= 1e-13 * max(np.abs(F).max(), 1e-300)
            # M form = action(a, un)
            un = np.linspace(-1, 2, nd)
            assert np.abs(zo.tabulate("poisson_M", order, xc, w=un) - A @ un).max() <= 1e-13 * np.abs(A).max()

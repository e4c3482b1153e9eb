"""Mathematical pins of the oracle (the reference pins nothing, so these do):
analytic element matrices, entity-count formulas of src/mesh.cpp:44-74, nnz closed forms, symmetry,
null spaces, patch test and manufactured-solution convergence rates h^(k+1)."""
import numpy as np
import pytest

import zzz_oracle as zo

REF = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]], float)


@pytest.fixture(autouse=True)
def _one_thread():
    zo.set_num_threads(1)


def test_mesh_size_search_matches_survey_appendix_b():
    # src/mesh.cpp:78-151 evaluated for every BASELINE.json config
    assert zo.mesh_size(500000, True, 1, 1, 1) == (78, 78, 79, 0)
    assert zo.mesh_size(10000000, True, 1, 1, 1) == (108, 103, 111, 1)
    assert zo.mesh_size(10000000, True, 8, 1, 1) == (108, 103, 111, 1)
    assert zo.mesh_size(500000, False, 8, 3, 1) == (109, 109, 109, 0)
    assert zo.mesh_size(500000, False, 1, 3, 1) == (54, 54, 54, 0)
    assert zo.mesh_size(50000000, True, 8, 1, 3) == (122, 122, 123, 0)
    assert zo.mesh_size(50000, False, 1, 1, 1) == (36, 32, 40, 0)
    assert zo.num_pdofs(108, 103, 111, 1, 1) == 10016937
    assert zo.num_pdofs(122, 122, 123, 0, 3) == 49834930
    assert zo.num_pdofs(78, 78, 79, 0, 1) == 499280
    assert zo.num_pdofs(2, 2, 2, 0, 5) == -1  # reference throws "Order not supported"


def test_p1_analytic_element_matrices():
    A = zo.tabulate("poisson_a", 1, REF)
    np.testing.assert_allclose(6 * A, [[3, -1, -1, -1], [-1, 1, 0, 0], [-1, 0, 1, 0], [-1, 0, 0, 1]], atol=1e-15)
    M = np.array([zo.tabulate("poisson_L", 1, REF, w=np.r_[np.eye(4)[j], np.zeros(4)]) for j in range(4)])
    np.testing.assert_allclose(120 * M, np.ones((4, 4)) + np.eye(4), atol=1e-14)
    for lf in range(4):
        F = np.array([zo.tabulate("poisson_L_facet", 1, REF, w=np.r_[np.zeros(4), np.eye(4)[j]], facet=lf) for j in range(4)])
        area = np.sqrt(3) / 2 if lf == 0 else 0.5
        expect = area / 12 * (np.ones((4, 4)) + np.eye(4))
        expect[lf, :] = 0
        expect[:, lf] = 0
        np.testing.assert_allclose(F, expect, atol=1e-15)


@pytest.mark.parametrize("order", [1, 2, 3])
def test_element_invariants(order):
    rng = np.random.default_rng(order)
    xc = rng.random((4, 3))
    nd = zo.ndofs_cell(order)
    A = zo.tabulate("poisson_a", order, xc)
    assert np.abs(A - A.T).max() < 1e-13 * np.abs(A).max()
    assert np.abs(A.sum(axis=1)).max() < 1e-12 * np.abs(A).max()  # constants in the kernel
    # vertex permutation invariance up to the induced dof permutation is covered by the global tests;
    # here: orientation flip (negative det J) leaves the P1 matrix unchanged (|det J| scaling)
    if order == 1:
        A2 = zo.tabulate("poisson_a", 1, xc[[1, 0, 2, 3]])
        np.testing.assert_allclose(A2, A[np.ix_([1, 0, 2, 3], [1, 0, 2, 3])], rtol=1e-12, atol=1e-14)
    E = zo.tabulate("elasticity_a", order, xc)
    assert np.abs(E - E.T).max() < 1e-12 * np.abs(E).max()
    # six rigid-body modes of src/elasticity_problem.cpp:43-71 are annihilated
    X = zo.ref_nodes(order)
    P = (1 - X.sum(1))[:, None] * xc[0] + X @ xc[1:]
    modes = np.zeros((6, 3 * nd))
    for k in range(3):
        modes[k, k::3] = 1
    modes[3, 0::3], modes[3, 1::3] = -P[:, 1], P[:, 0]
    modes[4, 0::3], modes[4, 2::3] = P[:, 2], -P[:, 0]
    modes[5, 2::3], modes[5, 1::3] = P[:, 1], -P[:, 2]
    assert np.abs(E @ modes.T).max() < 1e-10 * np.abs(E).max()
    # mass matrix integrates constants: sum = volume
    M = np.array([zo.tabulate("poisson_L", order, xc, w=np.r_[np.eye(nd)[j], np.zeros(nd)]) for j in range(nd)])
    vol = abs(np.linalg.det(xc[1:] - xc[0])) / 6
    assert abs(M.sum() - vol) < 1e-13


@pytest.mark.parametrize("order", [1, 2, 3])
def test_counts_and_nnz_closed_forms(order):
    i, j, k = 3, 4, 2
    P = zo.Problem("poisson", order, i, j, k).assemble()
    V, E, F, Cn = zo.num_entities(i, j, k, 0)
    assert P.cells.shape[0] == Cn and P.x.shape[0] == V
    assert P.nblock == zo.num_pdofs(i, j, k, 0, order)
    assert P.counts[1] == (E if order >= 2 else 0) and P.counts[2] == (F if order == 3 else 0)
    s1, s2, s3 = i * j * k, i * j + i * k + j * k, i + j + k
    nnz = {1: 15 * s1 + 7 * s2 + 3 * s3 + 1, 2: 230 * s1 + 46 * s2 + 8 * s3 + 1, 3: 1311 * s1 + 153 * s2 + 15 * s3 + 1}
    assert P.cols.shape[0] == nnz[order]
    assert len(P.facets) == 4 * s2
    assert np.diff(P.rowptr).max() <= {1: 15, 2: 65, 3: 175}[order]


@pytest.mark.parametrize("problem,order", [("poisson", 1), ("poisson", 3), ("elasticity", 1), ("elasticity", 2)])
def test_global_matrix_invariants(problem, order):
    import scipy.sparse as sp

    P = zo.Problem(problem, order, 2, 3, 2).assemble()
    A = sp.csr_matrix((P.vals, P.cols, P.rowptr), shape=(P.n, P.n))
    assert abs(A - A.T).max() < 1e-9 * abs(A).max()
    bc = P.bc.astype(bool)
    # BC rows/cols are identity
    Ad = A.toarray()
    assert np.array_equal(Ad[bc][:, bc], np.eye(bc.sum()))
    assert np.abs(Ad[bc][:, ~bc]).max() == 0 and np.abs(Ad[~bc][:, bc]).max() == 0
    assert np.all(P.b[bc] == 0)
    # without BCs: Poisson rows sum to zero, elasticity annihilates rigid modes
    nobc = np.zeros_like(P.bc)
    v0 = zo.assemble_matrix(P.form, order, P.x, P.cells, P.cell_dofs, nobc, P.rowptr, P.cols)
    A0 = sp.csr_matrix((v0, P.cols, P.rowptr), shape=(P.n, P.n))
    if problem == "poisson":
        assert np.abs(A0 @ np.ones(P.n)).max() < 1e-12 * np.abs(v0).max()
        # sum of the load vector without BCs = int f_h + int_{dOmega} g_h; compare with a fine quadrature of f_h via mass
        b0 = zo.assemble_vector(P.form, order, P.x, P.cells, P.cell_dofs, P.f, P.g, P.facets, nobc)
        ones_g = zo.assemble_vector(P.form, order, P.x, P.cells, P.cell_dofs, np.zeros(P.n), np.ones(P.n), P.facets, nobc)
        assert abs(ones_g.sum() - 6.0) < 1e-12  # surface area of the unit cube
        ones_f = zo.assemble_vector(P.form, order, P.x, P.cells, P.cell_dofs, np.ones(P.n), np.zeros(P.n), P.facets, nobc)
        assert abs(ones_f.sum() - 1.0) < 1e-12  # volume
        assert np.isfinite(b0).all()
    else:
        X = P.dof_x
        mode = np.zeros(P.n)
        mode[0::3], mode[1::3] = -X[:, 1], X[:, 0]
        assert np.abs(A0 @ mode).max() < 1e-9 * np.abs(v0).max()
        assert np.abs(A0 @ np.tile([1.0, 0, 0], P.nblock)).max() < 1e-9 * np.abs(v0).max()


def test_p1_structural_zero_half():
    """on the Kuhn mesh the P1 Laplacian is a 7-point stencil inside a
    15-point pattern; the explicit zeros stay in the pattern (ADD_VALUES inserts them)."""
    P = zo.Problem("poisson", 1, 6, 6, 6).assemble()
    frac = (np.abs(P.vals) < 1e-13).mean()
    assert 0.35 < frac < 0.7  # BC rows/cols of this small mesh add zeros


def test_patch_test_linear_solution():
    """u = 1 + 2y - 3z is in every P_k space and satisfies -lap u = 0 with natural data g = du/dn."""
    import scipy.sparse as sp

    for order in (1, 2, 3):
        P = zo.Problem("poisson", order, 2, 2, 2)
        rowptr, cols = zo.pattern(P.nblock, P.cell_dofs, 1)
        nobc = np.zeros(P.n, np.uint8)
        v0 = zo.assemble_matrix(0, order, P.x, P.cells, P.cell_dofs, nobc, rowptr, cols)
        A0 = sp.csr_matrix((v0, cols, rowptr), shape=(P.n, P.n))
        u = 1 + 2 * P.dof_x[:, 1] - 3 * P.dof_x[:, 2]
        # residual A u must equal the boundary flux vector: int g v ds with g = grad u . n
        X = P.dof_x
        g = np.zeros(P.n)
        # piecewise-constant flux per face; evaluate by splitting facets per cube face
        facets = P.facets
        xc = P.x[P.cells[facets[:, 0]]]
        FV = np.array([[1, 2, 3], [0, 2, 3], [0, 1, 3], [0, 1, 2]])
        rhs = np.zeros(P.n)
        for val, sel in ((-2.0, lambda c: np.abs(c[:, 1]) < 1e-12), (2.0, lambda c: np.abs(c[:, 1] - 1) < 1e-12),
                         (3.0, lambda c: np.abs(c[:, 2]) < 1e-12), (-3.0, lambda c: np.abs(c[:, 2] - 1) < 1e-12)):
            cen = np.array([xc[i][FV[facets[i, 1]]].mean(0) for i in range(len(facets))])
            m = sel(cen)
            rhs += val * zo.assemble_vector(0, order, P.x, P.cells, P.cell_dofs, np.zeros(P.n), np.ones(P.n),
                                            np.ascontiguousarray(facets[m]), nobc)
        assert np.abs(A0 @ u - rhs).max() < 1e-12 * np.abs(v0).max()


@pytest.mark.parametrize("order", [1, 2, 3])
def test_mms_convergence_rate(order):
    """-lap u = f, u = sin(pi x) cos(pi y) cos(pi z): Dirichlet 0 on x=0,1 (the reference's BC set,
    src/poisson_problem.cpp:58-77), natural 0 elsewhere.  Nodal l2 error ~ h^(k+1)."""
    errs = []
    ns = {1: (4, 8), 2: (3, 6), 3: (2, 4)}[order]
    for n in ns:
        P = zo.Problem("poisson", order, n, n, n)
        X = P.dof_x
        ue = np.sin(np.pi * X[:, 0]) * np.cos(np.pi * X[:, 1]) * np.cos(np.pi * X[:, 2])
        P.f = 3 * np.pi**2 * ue
        P.g = np.zeros(P.n)
        P.assemble()
        it, u, _, _ = zo.pcg(P.rowptr, P.cols, P.vals, P.b, rtol=1e-12)
        errs.append(np.sqrt(np.mean((u - ue) ** 2)))
    rate = np.log2(errs[0] / errs[1])
    assert rate > order + 1 - 0.45, (errs, rate)


def test_matrix_free_action_equals_assembled():
    """cgpoisson's operator (src/cgpoisson_problem.cpp:193-230) == assembled A with BC rows zeroed,
    on vectors whose BC entries are zero (the Krylov vectors of that solver)."""
    for order in (1, 2, 3):
        P = zo.Problem("poisson", order, 2, 2, 3).assemble()
        rng = np.random.default_rng(order)
        v = rng.standard_normal(P.n)
        v[P.bc.astype(bool)] = 0
        y = zo.action_poisson(order, P.x, P.cells, P.cell_dofs, P.bc, v)
        ya = zo.spmv(P.rowptr, P.cols, P.vals, v)
        ya[P.bc.astype(bool)] = 0
        assert np.abs(y - ya).max() < 1e-12 * np.abs(P.vals).max()


def test_cg_h_semantics():
    """src/cg.h:38-86: returns k = operator applications in the loop; stops on <r,r>/<r0,r0> < rtol^2
    strictly, tested before the p update; kmax caps."""
    P = zo.Problem("poisson", 1, 4, 4, 4).assemble()
    k, u, rn = zo.cg(P.rowptr, P.cols, P.vals, P.b, kmax=1000, rtol=1e-8)
    assert rn < 1e-16 and k > 5
    k2, u2, rn2 = zo.cg(P.rowptr, P.cols, P.vals, P.b, kmax=k - 1, rtol=1e-8)
    assert k2 == k - 1 and rn2 >= 1e-16
    k3, u3, _ = zo.cg(P.rowptr, P.cols, P.vals, P.b, kmax=5, rtol=1e-8)
    assert k3 == 5
    # the test is relative to the INITIAL residual (rnorm0, src/cg.h:53,78): a warm start from a
    # partly converged x needs further iterations for the same rtol and stays a solution
    k4, u4, _ = zo.cg(P.rowptr, P.cols, P.vals, P.b, x0=u3, kmax=1000, rtol=1e-8)
    assert 1 < k4 <= k and np.linalg.norm(u4 - u) < 1e-6 * np.linalg.norm(u)


@pytest.mark.parametrize("problem,order,dims", [("poisson", 1, (8, 7, 9)), ("poisson", 2, (4, 4, 5)),
                                                ("elasticity", 1, (5, 4, 6))])
@pytest.mark.parametrize("norm", [zo.NORM_PRECONDITIONED, zo.NORM_UNPRECONDITIONED, zo.NORM_NATURAL])
def test_single_reduction_pcg_equals_classical(problem, order, dims, norm):
    """KSPCG with -ksp_cg_single_reduction is the same iteration in exact arithmetic: the restatement
    must reproduce the classical iteration count and iterate to round-off."""
    P = zo.Problem(problem, order, *dims)
    P.assemble()
    it0, u0, rn0, r00 = zo.pcg(P.rowptr, P.cols, P.vals, P.b, norm_type=norm, rtol=1e-10)
    it1, u1, rn1, r01 = zo.pcg_single_reduction(P.rowptr, P.cols, P.vals, P.b, norm_type=norm, rtol=1e-10)
    assert it0 == it1 and r00 == r01
    assert np.linalg.norm(u1 - u0) <= 1e-11 * np.linalg.norm(u0)
    assert abs(rn1 - rn0) <= 0.05 * rn0  # the recurrences drift apart by round-off (elasticity: 5e-3)
    # max_it and trivial right-hand side
    it2, _, _, _ = zo.pcg_single_reduction(P.rowptr, P.cols, P.vals, P.b, rtol=1e-30, max_it=3)
    assert it2 == 3
    it3, u3, rn3, _ = zo.pcg_single_reduction(P.rowptr, P.cols, P.vals, np.zeros_like(P.b))
    assert it3 == 0 and rn3 == 0.0 and not u3.any()


@pytest.mark.parametrize("problem,order,dims", [("poisson", 1, (12, 11, 13)), ("poisson", 3, (3, 3, 4)),
                                                ("elasticity", 2, (3, 3, 3))])
def test_chebyshev_jacobi_restatement(problem, order, dims):
    """The polynomial preconditioner restated in the oracle against an independent numpy / scipy statement: the
    Gershgorin bound, z = p_k(D^-1 A) D^-1 r built from the three-term recurrence of the Chebyshev polynomials (not
    from the oracle's rho recurrence), symmetry and positivity of the preconditioner, the solution of the direct
    solver, degree 1 = Jacobi's iteration count, higher degrees fewer iterations."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spl

    P = zo.Problem(problem, order, *dims)
    P.assemble()
    n = P.b.shape[0]
    A = sp.csr_matrix((P.vals, P.cols, P.rowptr), shape=(n, n))
    dinv = 1.0 / A.diagonal()
    hi = float((abs(A).sum(axis=1).A1 * abs(dinv)).max())
    uref = spl.spsolve(A.tocsc(), P.b)
    itj, uj, _, _ = zo.pcg(P.rowptr, P.cols, P.vals, P.b, rtol=1e-10)
    its = {}
    for degree, ratio in ((1, 30.0), (2, 10.0), (3, 60.0), (6, 100.0)):
        it, u, rn, r0, est = zo.pcg_chebyshev(P.rowptr, P.cols, P.vals, P.b, degree=degree, ratio=ratio, rtol=1e-10)
        its[degree] = it
        assert abs(est - hi) <= 1e-13 * hi
        assert rn <= 1e-10 * r0
        assert np.linalg.norm(u - uref) <= 1e-7 * np.linalg.norm(uref)
    assert abs(its[1] - itj) <= 1 and its[6] < its[3] < its[2] < its[1]

    # the Lanczos estimate of the largest eigenvalue of D^-1 A (PETSc's -ksp_chebyshev_esteig): from below, within a few
    # per cent after 10 iterations from the noise vector, so that 1.1 x the estimate is an upper bound; it replaces
    # Gershgorin's bound where that is loose (P2 / P3 / elasticity) and the solve then needs fewer products
    lmax = float(spl.eigsh(sp.diags(np.sqrt(dinv)) @ A @ sp.diags(np.sqrt(dinv)), k=1, which="LA",
                           return_eigenvectors=False)[0])
    ritz = zo.esteig(P.rowptr, P.cols, P.vals, 10)
    assert 0.95 * lmax <= ritz <= lmax * (1 + 1e-12) and lmax <= hi * (1 + 1e-12)
    assert zo.esteig(P.rowptr, P.cols, P.vals, 30) >= ritz   # Ritz values grow with the Krylov space
    it_e, u_e, rn_e, r0_e, bound = zo.pcg_chebyshev(P.rowptr, P.cols, P.vals, P.b, degree=3, ratio=60.0, rtol=1e-10, est_its=10)
    assert abs(bound - min(hi, 1.1 * ritz)) <= 1e-13 * hi and lmax < bound
    assert np.linalg.norm(u_e - uref) <= 1e-7 * np.linalg.norm(uref) and it_e <= its[3]
    if hi > 1.5 * lmax:
        assert it_e < 0.8 * its[3]
    noise = np.array([zo.lib().zo_noise(i) for i in range(4096)])
    assert abs(noise.mean()) < 0.02 and 0.27 < noise.std() < 0.31 and -0.5 <= noise.min() and noise.max() < 0.5

    # the polynomial itself: e_k = (I - M_k D^-1 A) e_0 must be the scaled Chebyshev polynomial T_k((theta - t) / delta) /
    # T_k(theta / delta) of t = D^-1 A; checked through one application (max_it = 1 from r = b: x_1 = alpha z, alpha > 0)
    for degree, ratio in ((2, 10.0), (4, 30.0)):
        lo = hi / ratio
        theta, delta = 0.5 * (hi + lo), 0.5 * (hi - lo)
        B = sp.diags(dinv) @ A
        g0 = dinv * P.b
        # Chebyshev three-term recurrence on vectors: y_j = T_j((theta I - B) / delta) g0
        y_prev, y = g0, (theta * g0 - B @ g0) / delta
        t_prev, t = 1.0, theta / delta
        for _ in range(degree - 1):
            y_prev, y = y, 2.0 * (theta * y - B @ y) / delta - y_prev
            t_prev, t = t, 2.0 * (theta / delta) * t - t_prev
        resid = y / t                       # (I - B M_k) g0  with M_k = p_k(B)
        z_expected = spl.spsolve(sp.csc_matrix(B), g0 - resid)   # p_k(B) g0 = B^-1 (g0 - resid)
        it, x1, _, _, _ = zo.pcg_chebyshev(P.rowptr, P.cols, P.vals, P.b, degree=degree, ratio=ratio, rtol=1e-30, max_it=1)
        assert it == 1
        alpha = float(x1 @ z_expected) / float(z_expected @ z_expected)
        assert alpha > 0 and np.linalg.norm(x1 - alpha * z_expected) <= 1e-9 * np.linalg.norm(x1)
        assert abs(alpha - float(P.b @ z_expected) / float(z_expected @ (A @ z_expected))) <= 1e-9 * alpha


def test_chunked_spmv_is_the_serial_spmv_to_roundoff():
    """zo_spmv_chunked restates the GPU's multi-lane row sums: lanes = 1 IS zo_spmv, more lanes change
    only the association of the additions."""
    P = zo.Problem("elasticity", 2, 3, 3, 3)
    P.assemble()
    x = np.random.default_rng(0).standard_normal(P.rowptr.shape[0] - 1)
    y1 = zo.spmv(P.rowptr, P.cols, P.vals, x)
    np.testing.assert_array_equal(zo.spmv_chunked(P.rowptr, P.cols, P.vals, x, 1), y1)
    for lanes in (2, 4, 8, 16):
        y = zo.spmv_chunked(P.rowptr, P.cols, P.vals, x, lanes)
        assert 0 < np.abs(y - y1).max() <= 2e-15 * np.abs(y1).max()


def test_near_nullspace_restatement():
    """zo_near_nullspace (build_near_nullspace, src/elasticity_problem.cpp:36-94): orthonormal, spans the rigid-body
    motions (every translation and infinitesimal rotation field is reproduced by its projection), and the unconstrained
    elasticity matrix annihilates it."""
    P = zo.Problem("elasticity", 2, 3, 4, 3)
    B, dev = zo.near_nullspace(P.dof_x)
    assert dev <= 1e-13 and np.abs(B @ B.T - np.eye(6)).max() <= 1e-13
    x = P.dof_x
    w = np.array([0.3, -1.1, 0.7])
    field = (np.array([0.5, 0.25, -2.0]) + np.cross(w, x)).reshape(-1)
    assert np.abs(B.T @ (B @ field) - field).max() <= 1e-12 * np.abs(field).max()
    rp, cl = zo.pattern(P.nblock, P.cell_dofs, 3)
    v = zo.assemble_matrix(1, 2, P.x, P.cells, P.cell_dofs, np.zeros(P.n, np.uint8), rp, cl)
    for k in range(6):
        assert np.abs(zo.spmv(rp, cl, v, B[k])).max() <= 1e-9 * np.abs(v).max()

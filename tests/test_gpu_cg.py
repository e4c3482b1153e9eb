"""GPU parity tests, part 3: the Krylov loops (src/cg.h, KSPCG + PCJACOBI, single reduction, Chebyshev-Jacobi) and the
matrix-free operator of cgpoisson."""
from _gpu_helpers import *  # noqa: F401,F403 -- helpers, fixtures (ctx), np / os / zzz / zo

pytestmark = pytest.mark.gpu  # noqa: F405


def test_solver_edge_cases(ctx):
    P = zzz.Part("poisson", 1, 5, 5, 5)
    ctx.upload_part(P)
    ctx.pattern_build()
    ctx.assemble_matrix(zzz.FORM_POISSON)
    ctx.assemble_vector(zzz.FORM_POISSON)
    b = ctx.vec_download(zzz.VEC_B)
    # max_it cap: returns max_it like KSP (diverged_its) / cg.h (kmax)
    it, _, _ = ctx.cg_solve(pc=zzz.PC_JACOBI, rtol=1e-14, max_it=3)
    assert it == 3
    k, _, _ = ctx.cg_solve(variant=zzz.CG_CGH, pc=zzz.PC_NONE, rtol=1e-14, max_it=4)
    assert k == 4
    # zero right-hand side: PETSc converges at iteration 0 (0 <= atol)
    ctx.vec_upload(zzz.VEC_B, np.zeros_like(b))
    it, rn, r0 = ctx.cg_solve(pc=zzz.PC_JACOBI, rtol=1e-8)
    assert it == 0 and rn == 0.0
    assert np.all(ctx.vec_download(zzz.VEC_U) == 0)
    # ... while linalg::cg has no guard for rnorm0 == 0 (src/cg.h:53-83): alpha = 0/0, every comparison with NaN is
    # false, the loop runs kmax times and x ends up NaN -- reproduced literally, it is not an error
    ctx.vec_upload(zzz.VEC_U, np.zeros_like(b))
    k, rr, rr0 = ctx.cg_solve(variant=zzz.CG_CGH, pc=zzz.PC_NONE, rtol=1e-6, max_it=7)
    assert k == 7 and rr0 == 0.0
    assert np.all(np.isnan(ctx.vec_download(zzz.VEC_U)))
    zo.set_num_threads(1)
    rp, cl, v = ctx.csr_download()
    ok, ox = zo.cg(rp.astype(np.int64), cl, v, np.zeros_like(b), kmax=7, rtol=1e-6)[:2]
    assert ok == 7 and np.all(np.isnan(ox))
    # KSPConvergedDefault's divergence test: norm >= divtol x initial norm -> KSP_DIVERGED_DTOL (both CG forms)
    ctx.vec_upload(zzz.VEC_B, b)
    # As in the reference (solver_function returns solver.solve()'s count whatever the reason, src/poisson_problem.cpp:172-178)
    # that is not an error: the solve returns its iteration count and the reason is there to be read; it only fails
    # under -ksp_error_if_not_converged.
    for sr in (False, True):
        it_d, _, _ = ctx.cg_solve(pc=zzz.PC_JACOBI, rtol=1e-8, dtol=0.5, single_reduction=sr)
        assert ctx.cg_reason() == -4 and 0 <= it_d < 100
        with pytest.raises(zzz.ZzzError, match="DTOL"):
            ctx.cg_solve(pc=zzz.PC_JACOBI, rtol=1e-8, dtol=0.5, single_reduction=sr, error_if_not_converged=True)
        it, _, _ = ctx.cg_solve(pc=zzz.PC_JACOBI, rtol=1e-8, single_reduction=sr)  # default divtol 1e4: converges
        assert 0 < it < 100 and ctx.cg_reason() == 2
        it3, _, _ = ctx.cg_solve(pc=zzz.PC_JACOBI, rtol=1e-8, max_it=3, single_reduction=sr)  # KSP_DIVERGED_ITS: no error either
        assert it3 == 3 and ctx.cg_reason() == -3
        with pytest.raises(zzz.ZzzError, match="KSP_DIVERGED_ITS"):
            ctx.cg_solve(pc=zzz.PC_JACOBI, rtol=1e-8, max_it=3, single_reduction=sr, error_if_not_converged=True)
    # the same limits with the polynomial preconditioner: max_it, divergence test, zero right-hand side, history
    cheb = dict(pc=zzz.PC_CHEBYSHEV_JACOBI)
    it3, _, _ = ctx.cg_solve(rtol=1e-14, max_it=3, **cheb)
    assert it3 == 3 and ctx.cg_reason() == -3
    with pytest.raises(zzz.ZzzError, match="KSP_DIVERGED_ITS"):
        ctx.cg_solve(rtol=1e-14, max_it=3, error_if_not_converged=True, **cheb)
    it_d, _, _ = ctx.cg_solve(rtol=1e-8, dtol=0.5, **cheb)
    assert ctx.cg_reason() == -4 and 0 <= it_d < 100
    it, rn, r0 = ctx.cg_solve(rtol=1e-8, **cheb)
    hist = ctx.cg_history(it + 1)
    assert 0 < it < 40 and ctx.cg_reason() == 2 and hist[0] == r0 and hist[-1] == rn and rn <= 1e-8 * r0
    ctx.vec_upload(zzz.VEC_B, np.zeros_like(b))
    it, rn, r0 = ctx.cg_solve(rtol=1e-8, **cheb)
    assert it == 0 and rn == 0.0 and np.all(ctx.vec_download(zzz.VEC_U) == 0)
    ctx.vec_upload(zzz.VEC_B, b)
    # argument errors surface as ZzzError, not crashes
    with pytest.raises(zzz.ZzzError):
        ctx.cg_solve(variant=zzz.CG_CGH, pc=zzz.PC_JACOBI)
    with pytest.raises(zzz.ZzzError):
        ctx.assemble_matrix(zzz.FORM_ELASTICITY)  # bs mismatch
    with pytest.raises(zzz.ZzzError):
        ctx.upload_dofmap(4, 1, P.cell_dofs, P.n_owned, 0)  # order 4: reference throws too


@pytest.mark.parametrize("order,dims", [(1, (7, 6, 5)), (2, (4, 3, 4)), (3, (3, 2, 3))])
def test_matrix_free_operator_and_cg(ctx, order, dims):
    """cgpoisson: the matrix-free action (src/cgpoisson_problem.cpp:193-230) and linalg::cg on it
    with the reference's arguments kmax=100, rtol=1e-6 (:233)."""
    P = zzz.Part("poisson", order, *dims)
    ctx.upload_part(P)
    ctx.pattern_build()
    ctx.assemble_vector(zzz.FORM_POISSON)
    bc = P.bc_marker()
    rng = np.random.default_rng(order)
    v = rng.standard_normal(P.n_owned)
    y = ctx.action(v)
    oy = zo.action_poisson(order, P.x, P.cells, P.cell_dofs, bc, v)
    assert np.abs(y - oy).max() <= 1e-12 * np.abs(oy).max()
    assert np.all(y[bc.astype(bool)] == 0)
    b = ctx.vec_download(zzz.VEC_B)
    ctx.vec_upload(zzz.VEC_U, np.zeros_like(b))
    k, rr, rr0 = ctx.cg_solve(variant=zzz.CG_CGH, pc=zzz.PC_NONE, op=zzz.OP_MATFREE, rtol=1e-6, max_it=100)
    u = ctx.vec_download(zzz.VEC_U)
    ok, ou = zo.cg_matfree_poisson(order, P.x, P.cells, P.cell_dofs, bc, b, kmax=100, rtol=1e-6)
    assert abs(k - ok) <= 2 and k <= 100
    assert np.linalg.norm(u - ou) <= 1e-6 * np.linalg.norm(ou)


@pytest.mark.parametrize("order,dims", [(1, (9, 8, 10)), (2, (5, 4, 6)), (3, (3, 4, 3))])
def test_jacobi_pcg_on_the_matrix_free_operator(order, dims):
    """KSPCG + PCJACOBI with the operator never assembled (op = ZZZ_OP_MATFREE; driver: --operator matfree): the diagonal
    comes from the element matrices in the matrix-free kernel's pass and must be the assembled matrix's (1.0 on constrained
    rows); the solve must be the assembled one's -- the oracle's PCG on the oracle's matrix -- to the usual bars."""
    P = zzz.Part("poisson", order, *dims)
    bc = P.bc_marker()
    orp, ocl = zo.pattern(P.n_owned, P.cell_dofs, 1)
    ov = zo.assemble_matrix(0, order, P.x, P.cells, P.cell_dofs, bc, orp, ocl)
    ob = zo.assemble_vector(0, order, P.x, P.cells, P.cell_dofs, P.f, P.g, P.facets, bc)
    odiag = np.array([ov[orp[i]:orp[i + 1]][ocl[orp[i]:orp[i + 1]] == i][0] for i in range(P.n_owned)])
    oit, ou, orn, or0 = zo.pcg(orp, ocl, ov, ob, rtol=1e-8)
    with zzz.Context(0) as c:
        # nothing but mesh, dofmap, Dirichlet set and the right-hand side: no pattern, no matrix
        c.upload_part(P)
        d = c.matfree_diagonal()
        assert np.abs(d - odiag).max() <= 1e-12 * np.abs(odiag).max()
        assert np.all(d[bc.astype(bool)] == 1.0)
        np.testing.assert_array_equal(d, c.matfree_diagonal())  # the same bits every time
        c.vec_upload(zzz.VEC_B, ob)
        it, rn, r0 = c.cg_solve(pc=zzz.PC_JACOBI, op=zzz.OP_MATFREE, rtol=1e-8)
        u = c.vec_download(zzz.VEC_U)
        assert abs(it - oit) <= 2
        assert np.linalg.norm(u - ou) <= 1e-6 * np.linalg.norm(ou)
        assert abs(r0 - or0) <= 1e-11 * or0
        assert np.linalg.norm(ob - zo.spmv(orp, ocl, ov, u)) <= 1e-7 * np.linalg.norm(ob)
        # ... and against the library's own assembled solve
        c.pattern_build()
        c.assemble_matrix(zzz.FORM_POISSON)
        c.assemble_vector(zzz.FORM_POISSON)
        ita, _, _ = c.cg_solve(pc=zzz.PC_JACOBI, rtol=1e-8)
        ua = c.vec_download(zzz.VEC_U)
        assert abs(it - ita) <= 2 and np.linalg.norm(u - ua) <= 1e-7 * np.linalg.norm(ua)
        # -pc_type none on the same operator
        itn, _, _ = c.cg_solve(pc=zzz.PC_NONE, op=zzz.OP_MATFREE, rtol=1e-8)
        itna, _, _ = c.cg_solve(pc=zzz.PC_NONE, rtol=1e-8)
        assert abs(itn - itna) <= 2
        # what stays with the assembled operator says so
        for bad in (dict(pc=zzz.PC_CHEBYSHEV_JACOBI), dict(single_reduction=True)):
            with pytest.raises(zzz.ZzzError):
                c.cg_solve(op=zzz.OP_MATFREE, rtol=1e-8, **bad)


def test_jacobi_pcg_on_the_matrix_free_operator_partitioned():
    """The same across three z-slabs on one GPU (host-mailbox communicator): halo of p before every action, all-reduced
    scalars; iteration count and solution of the single-rank assembled solve."""
    import threading

    problem, order, dims, nparts = "poisson", 2, (4, 4, 9), 3
    G = zzz.Part(problem, order, *dims)
    with zzz.Context(0) as c0:
        c0.upload_part(G)
        c0.pattern_build()
        c0.assemble_matrix(G.form)
        c0.assemble_vector(G.form)
        it0, _, _ = c0.cg_solve(pc=zzz.PC_JACOBI, rtol=1e-8)
        u0 = c0.vec_download(zzz.VEC_U)
        _, _, v0 = c0.csr_download()
        rp0, cl0, _ = c0.csr_download()
    d0 = np.array([v0[rp0[i]:rp0[i + 1]][cl0[rp0[i]:rp0[i + 1]] == i][0] for i in range(G.n_owned)])
    grp = zzz.LocalGroup(nparts)
    out = [None] * nparts
    err = []

    def run(rank):
        try:
            P = zzz.Part(problem, order, *dims, nparts, rank)
            with zzz.Context(0) as c:
                c.comm_init_local(grp.h, rank)
                c.upload_part(P)
                c.upload_halo(P)
                c.pattern_build()
                c.assemble_vector(P.form)
                it, rn, r0 = c.cg_solve(pc=zzz.PC_JACOBI, op=zzz.OP_MATFREE, rtol=1e-8)
                out[rank] = (it, P.own_offset, c.vec_download(zzz.VEC_U), c.matfree_diagonal())
        except Exception as e:  # noqa: BLE001
            err.append((rank, repr(e)))

    th = [threading.Thread(target=run, args=(r,)) for r in range(nparts)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    grp.close()
    assert not err, err
    assert len({o[0] for o in out}) == 1 and abs(out[0][0] - it0) <= 2
    u = np.concatenate([o[2] for o in out])
    d = np.concatenate([o[3] for o in out])
    assert np.linalg.norm(u - u0) <= 1e-7 * np.linalg.norm(u0)
    assert np.abs(d - d0).max() <= 1e-12 * np.abs(d0).max()


@pytest.mark.parametrize("problem,order,dims", [("poisson", 1, (20, 18, 19)), ("elasticity", 1, (8, 7, 9)), ("poisson", 3, (5, 4, 6))])
def test_inverse_diagonal_as_codes_keeps_every_bit(problem, order, dims):
    """KSPCG + PCJACOBI with Jacobi's inverse diagonal read as 16-bit codes into a table of its distinct values and
    z = D^-1 r recomputed where it is used instead of stored (ZZZ_CG_DINV_CODES; default for vectors of 32 MB and more):
    the same doubles multiplied in the same places -- iteration count, residual norms and solution identical bit for bit
    to the run on the plain array; an odd number of rows and a partitioned run (all-reduced scalars) included."""
    P = zzz.Part(problem, order, *dims)
    res = {}
    old = os.environ.get("ZZZ_CG_DINV_CODES")
    try:
        for knob in ("0", "2"):
            os.environ["ZZZ_CG_DINV_CODES"] = knob
            with zzz.Context(0) as c:
                c.upload_part(P)
                c.pattern_build()
                c.assemble_matrix(P.form)
                c.assemble_vector(P.form)
                out = []
                for sr in (False, True):  # (-ksp_cg_single_reduction: k_sr_update with the codes, round 5)
                    for norm in (zzz.NORM_PRECONDITIONED, zzz.NORM_UNPRECONDITIONED, zzz.NORM_NATURAL):
                        it, rn, r0 = c.cg_solve(pc=zzz.PC_JACOBI, norm=norm, rtol=1e-9, single_reduction=sr)
                        out.append((it, rn, r0, c.vec_download(zzz.VEC_U), c.cg_info()["dinv_codes"], c.cg_history(it + 1)))
                # what does not take the coded path says so in cg_info and still works
                for sr in (False, True):
                    itn, _, _ = c.cg_solve(pc=zzz.PC_NONE, rtol=1e-9, single_reduction=sr)
                    out.append((itn, 0.0, 0.0, c.vec_download(zzz.VEC_U), c.cg_info()["dinv_codes"], c.cg_history(itn + 1)))
                itn, _, _ = c.cg_solve(pc=zzz.PC_CHEBYSHEV_JACOBI, rtol=1e-9, single_reduction=True)
                out.append((itn, 0.0, 0.0, c.vec_download(zzz.VEC_U), c.cg_info()["dinv_codes"], c.cg_history(itn + 1)))
                res[knob] = out
    finally:
        if old is None:
            os.environ.pop("ZZZ_CG_DINV_CODES", None)
        else:
            os.environ["ZZZ_CG_DINV_CODES"] = old
    for a, b in zip(res["0"], res["2"]):
        assert a[0] == b[0] and a[1] == b[1] and a[2] == b[2]
        np.testing.assert_array_equal(a[3], b[3])
        np.testing.assert_array_equal(a[5], b[5])  # the residual history, every iteration
    assert all(o[4] == 0 for o in res["0"]) and all(o[4] > 0 for o in res["2"][:6]) and all(o[4] == 0 for o in res["2"][6:])


def test_rccl_path_single_rank(ctx):
    """The multi-GPU code path (reduce -> ncclAllReduce -> scalar kernels, halo with no neighbour)
    on a 1-rank communicator must reproduce the single-GPU solve exactly."""
    P = zzz.Part("poisson", 1, 8, 8, 8)
    ctx.upload_part(P)
    ctx.pattern_build()
    ctx.assemble_matrix(zzz.FORM_POISSON)
    ctx.assemble_vector(zzz.FORM_POISSON)
    it0, rn0, _ = ctx.cg_solve(pc=zzz.PC_JACOBI, rtol=1e-8)
    u0 = ctx.vec_download(zzz.VEC_U)
    n0 = ctx.vec_norm(zzz.VEC_U)
    with zzz.Context(0) as c2:
        c2.comm_init(1, 0, zzz.comm_unique_id())
        c2.upload_part(P)
        c2.upload_halo(P)
        c2.pattern_build()
        c2.assemble_matrix(zzz.FORM_POISSON)
        c2.assemble_vector(zzz.FORM_POISSON)
        it1, rn1, _ = c2.cg_solve(pc=zzz.PC_JACOBI, rtol=1e-8)
        u1 = c2.vec_download(zzz.VEC_U)
        assert it1 == it0 and rn1 == rn0
        np.testing.assert_array_equal(u1, u0)
        assert c2.vec_norm(zzz.VEC_U) == n0
        # the same through the peer-memory all-reduce (attach agrees through the RCCL communicator)
        assert c2.comm_p2p_attach(c2.comm_p2p_export())
        for sr in (False, True):
            it2, rn2, _ = c2.cg_solve(pc=zzz.PC_JACOBI, rtol=1e-8, single_reduction=sr)
            assert abs(it2 - it0) <= (1 if sr else 0)
            if not sr:
                assert rn2 == rn0
                np.testing.assert_array_equal(c2.vec_download(zzz.VEC_U), u0)
        assert c2.vec_norm(zzz.VEC_U) == pytest.approx(n0, rel=1e-9)
        c2.comm_p2p_disable()
        it3, rn3, _ = c2.cg_solve(pc=zzz.PC_JACOBI, rtol=1e-8)
        assert it3 == it0 and rn3 == rn0


@in_tools_build
def test_allreduce_folded_into_the_producers_tail_keeps_every_bit():
    """ZZZ_TAIL=1 (csrc/zzz_tail.h): the scalar all-reduce of a multi-GPU iteration done by the last-arriving workgroup of
    the product / of k_update_xr instead of a kernel of its own -- same summation tree, same mailbox protocol: identical
    iteration counts, norm histories and solutions, in both CG forms, with more partials than one pass of the tree
    (> 512 workgroups).  An A/B variant, off by default (measured 1 us slower per iteration at the 8-GPU per-rank size)."""
    P = zzz.Part("poisson", 1, 60, 60, 61)
    res = {}
    try:
        for tail in ("0", "1"):
            os.environ["ZZZ_TAIL"] = tail
            with zzz.Context(0) as c:
                c.comm_init(1, 0, zzz.comm_unique_id())
                c.upload_part(P)
                c.upload_halo(P)
                assert c.comm_p2p_attach(c.comm_p2p_export())
                c.pattern_build()
                c.assemble_matrix(zzz.FORM_POISSON)
                c.assemble_vector(zzz.FORM_POISSON)
                for sr in (False, True):
                    it, rn, r0 = c.cg_solve(pc=zzz.PC_JACOBI, rtol=1e-8, single_reduction=sr)
                    res[(tail, sr)] = (it, rn, r0, c.cg_history(it + 1), c.vec_download(zzz.VEC_U))
    finally:
        os.environ.pop("ZZZ_TAIL", None)
    for sr in (False, True):
        a, b = res[("0", sr)], res[("1", sr)]
        assert a[0] == b[0] and a[1] == b[1] and a[2] == b[2] and a[0] > 50
        np.testing.assert_array_equal(a[3], b[3])
        np.testing.assert_array_equal(a[4], b[4])


@pytest.mark.parametrize("problem,order,dims", [("poisson", 1, (12, 10, 14)), ("poisson", 2, (6, 5, 7)),
                                                ("poisson", 3, (4, 3, 5)), ("elasticity", 1, (6, 6, 6)),
                                                ("elasticity", 2, (3, 3, 4))])
@pytest.mark.parametrize("norm", [zzz.NORM_PRECONDITIONED, zzz.NORM_UNPRECONDITIONED, zzz.NORM_NATURAL])
def test_single_reduction_cg(ctx, problem, order, dims, norm):
    """-ksp_cg_single_reduction (KSPCGUseSingleReduction) against its oracle restatement and against the
    classical iteration: same iteration count (+-2), same solution, same norm history."""
    P = zo.Problem(problem, order, *dims)
    P.assemble()
    G = zzz.Part(problem, order, *dims)
    ctx.upload_part(G)
    ctx.pattern_build()
    ctx.assemble_matrix(G.form)
    ctx.assemble_vector(G.form)
    rowptr, cols, vals = ctx.csr_download()
    b = ctx.vec_download(zzz.VEC_B)
    ito, uo, rno, r0o = zo.pcg_single_reduction(rowptr.astype(np.int64), cols, vals, b, norm_type=norm, rtol=1e-9)
    it, rn, r0 = ctx.cg_solve(pc=zzz.PC_JACOBI, norm=norm, rtol=1e-9, single_reduction=True)
    u = ctx.vec_download(zzz.VEC_U)
    hist = ctx.cg_history(it + 1)
    assert abs(it - ito) <= 2
    assert abs(r0 - r0o) <= 1e-12 * r0o and rn <= 1e-9 * r0
    assert hist.shape[0] == it + 1 and hist[0] == r0 and hist[-1] == rn
    assert np.linalg.norm(u - uo) <= 1e-7 * np.linalg.norm(uo)
    itc, rnc, r0c = ctx.cg_solve(pc=zzz.PC_JACOBI, norm=norm, rtol=1e-9)
    uc = ctx.vec_download(zzz.VEC_U)
    assert abs(it - itc) <= 2 and r0 == pytest.approx(r0c, rel=1e-13)
    assert np.linalg.norm(u - uc) <= 1e-7 * np.linalg.norm(uc)
    # true residual of the single-reduction solution
    r = b - zo.spmv(rowptr.astype(np.int64), cols, vals, u)
    if norm == zzz.NORM_UNPRECONDITIONED:
        assert np.linalg.norm(r) <= 1.1e-9 * np.linalg.norm(b)
    # the option is KSPCG/assembled-operator only
    with pytest.raises(zzz.ZzzError):
        ctx.cg_solve(variant=zzz.CG_CGH, pc=zzz.PC_NONE, single_reduction=True)
    with pytest.raises(zzz.ZzzError):
        ctx.cg_solve(op=zzz.OP_MATFREE, pc=zzz.PC_NONE, single_reduction=True)


@pytest.mark.parametrize("problem,order,dims", [("poisson", 1, (24, 22, 23)), ("poisson", 2, (8, 7, 9)),
                                                ("poisson", 3, (5, 4, 6)), ("elasticity", 1, (8, 8, 9)),
                                                ("elasticity", 2, (4, 3, 5))])
@pytest.mark.parametrize("degree,ratio,esteig", [(1, 30.0, -1), (2, 10.0, 0), (3, 30.0, -1), (3, 60.0, 0), (5, 60.0, 12)])
def test_chebyshev_jacobi_preconditioner(ctx, problem, order, dims, degree, ratio, esteig):
    """ZZZ_PC_CHEBYSHEV_JACOBI (fewer all-reduces per solve) against its oracle restatement
    zo.pcg_chebyshev: same spectrum bound, iteration count +-2, solution 1e-6, residual within rtol; degree 1 is
    Jacobi scaled by a constant (the same iteration as PC_JACOBI); degree >= 2 takes fewer iterations than Jacobi."""
    P = zo.Problem(problem, order, *dims)
    G = zzz.Part(problem, order, *dims)
    ctx.upload_part(G)
    ctx.pattern_build()
    ctx.assemble_matrix(G.form)
    ctx.assemble_vector(G.form)
    rowptr, cols, vals = ctx.csr_download()
    b = ctx.vec_download(zzz.VEC_B)
    # spectrum bound: Gershgorin's alone (esteig < 0) or min(Gershgorin, 1.1 x the Lanczos estimate of `esteig` Jacobi-PCG
    # iterations on the noise vector; 0 = the default, 10) -- the same hash of the caller's row number on both sides
    est_its = 0 if esteig < 0 else (esteig or 10)
    ito, uo, rno, r0o, est = zo.pcg_chebyshev(rowptr.astype(np.int64), cols, vals, b, degree=degree, ratio=ratio, rtol=1e-9,
                                              est_its=est_its)
    import scipy.sparse as sp

    A = sp.csr_matrix((vals, cols, rowptr.astype(np.int64)), shape=(b.shape[0], b.shape[0]))
    gersh = float((abs(A).sum(axis=1).A1 / np.abs(A.diagonal())).max())
    assert est <= gersh * (1 + 1e-14)
    if est_its and (order > 1 or problem == "elasticity"):
        assert est < 0.8 * gersh   # where Gershgorin's bound is loose the estimate takes over
    kw = dict(pc_degree=degree, pc_ratio=ratio, pc_esteig_its=esteig)
    it, rn, r0 = ctx.cg_solve(pc=zzz.PC_CHEBYSHEV_JACOBI, rtol=1e-9, **kw)
    u = ctx.vec_download(zzz.VEC_U)
    assert ctx.cg_reason() == 2
    assert abs(ctx.cg_info()["pc_spectrum_bound"] - est) <= 2e-6 * est
    assert abs(it - ito) <= 2
    assert abs(r0 - r0o) <= 1e-11 * r0o and rn <= 1e-9 * r0
    assert np.linalg.norm(u - uo) <= 1e-6 * np.linalg.norm(uo)
    r = b - zo.spmv(rowptr.astype(np.int64), cols, vals, u)
    assert np.linalg.norm(r) <= 1e-6 * np.linalg.norm(b)
    itj, _, _ = ctx.cg_solve(pc=zzz.PC_JACOBI, rtol=1e-9)
    uj = ctx.vec_download(zzz.VEC_U)
    assert np.linalg.norm(u - uj) <= 1e-6 * np.linalg.norm(uj)
    if degree == 1:
        assert abs(it - itj) <= 1
    else:
        assert it < itj
    # run-to-run: every bit
    it2, rn2, _ = ctx.cg_solve(pc=zzz.PC_CHEBYSHEV_JACOBI, rtol=1e-9, **kw)
    assert it2 == it and rn2 == rn and np.array_equal(ctx.vec_download(zzz.VEC_U), u)
    assert np.array_equal(ctx.vec_download(zzz.VEC_B), b)   # the estimate's right-hand side never replaces b
    # the polynomial's terms as launches of their own (the tile kernel's form) instead of the product's epilogue: the
    # same iteration up to the grouping of the partial sums of <r,z>
    os.environ["ZZZ_CHEB_FUSED"] = "0"
    try:
        it3, rn3, r03 = ctx.cg_solve(pc=zzz.PC_CHEBYSHEV_JACOBI, rtol=1e-9, **kw)
        u3 = ctx.vec_download(zzz.VEC_U)
    finally:
        del os.environ["ZZZ_CHEB_FUSED"]
    assert abs(it3 - it) <= 1 and abs(r03 - r0) <= 1e-13 * r0 and np.linalg.norm(u3 - u) <= 1e-8 * np.linalg.norm(u)
    os.environ["ZZZ_SELLP"] = "0"   # and on the CSR tile kernel (the knob is read when a context is created)
    try:
        with zzz.Context(0) as c:
            c.upload_part(G)
            c.pattern_build()
            c.assemble_matrix(G.form)
            c.assemble_vector(G.form)
            it4, rn4, r04 = c.cg_solve(pc=zzz.PC_CHEBYSHEV_JACOBI, rtol=1e-9, **kw)
            u4 = c.vec_download(zzz.VEC_U)
            assert c.spmv_operator_form() == 0
    finally:
        del os.environ["ZZZ_SELLP"]
    assert abs(it4 - it) <= 1 and np.linalg.norm(u4 - u) <= 1e-8 * np.linalg.norm(u)
    # KSPCG with the assembled operator only
    # -ksp_cg_single_reduction with the polynomial: one reduction point per k products; the same iteration to round-off
    its, rns, r0s = ctx.cg_solve(pc=zzz.PC_CHEBYSHEV_JACOBI, rtol=1e-9, single_reduction=True, **kw)
    us = ctx.vec_download(zzz.VEC_U)
    assert abs(its - it) <= 2 and abs(r0s - r0) <= 1e-12 * r0 and rns <= 1e-9 * r0s and ctx.cg_reason() == 2
    assert np.linalg.norm(us - u) <= 1e-7 * np.linalg.norm(u)
    assert abs(ctx.cg_info()["pc_spectrum_bound"] - est) <= 2e-6 * est
    os.environ["ZZZ_CHEB_FUSED"] = "0"
    try:
        its2, _, _ = ctx.cg_solve(pc=zzz.PC_CHEBYSHEV_JACOBI, rtol=1e-9, single_reduction=True, **kw)
        us2 = ctx.vec_download(zzz.VEC_U)
    finally:
        del os.environ["ZZZ_CHEB_FUSED"]
    assert abs(its2 - its) <= 1 and np.linalg.norm(us2 - us) <= 1e-8 * np.linalg.norm(us)
    for bad in (dict(variant=zzz.CG_CGH), dict(op=zzz.OP_MATFREE), dict(pc_degree=-1), dict(pc_esteig_its=65)):
        with pytest.raises(zzz.ZzzError):
            ctx.cg_solve(pc=zzz.PC_CHEBYSHEV_JACOBI, **bad)


def test_single_reduction_cg_breakdown_and_limits(ctx):
    """max_it reached, zero right-hand side and immediate convergence behave as in the classical path"""
    G = zzz.Part("poisson", 1, 6, 6, 6)
    ctx.upload_part(G)
    ctx.pattern_build()
    ctx.assemble_matrix(G.form)
    ctx.assemble_vector(G.form)
    it, rn, r0 = ctx.cg_solve(pc=zzz.PC_JACOBI, rtol=1e-14, max_it=5, single_reduction=True)
    itc, rnc, _ = ctx.cg_solve(pc=zzz.PC_JACOBI, rtol=1e-14, max_it=5)
    assert it == 5 and itc == 5 and rn == pytest.approx(rnc, rel=1e-9)
    ctx.vec_upload(zzz.VEC_B, np.zeros(G.n_owned))
    it, rn, r0 = ctx.cg_solve(pc=zzz.PC_JACOBI, single_reduction=True)
    assert it == 0 and rn == 0.0 and np.all(ctx.vec_download(zzz.VEC_U) == 0.0)


@in_tools_build
def test_fused_direction_kernel_keeps_every_bit():
    """Two kernels per iteration (the product fused with p = z + b p and the pending x update, chosen for
    cache-resident loops) against the three-kernel form: the same operations on the same operands, so the
    iteration count, the whole residual history and the solution are bit-identical -- for KSPCG with each norm
    type, for src/cg.h, on natural and length-sorted streams."""
    keys = ("ZZZ_CG_FUSED", "ZZZ_SELLP", "ZZZ_SELLP_WIN")
    old = {k: os.environ.get(k) for k in keys}
    try:
        os.environ["ZZZ_SELLP_WIN"] = "0"  # the fused kernel gathers from memory: an A/B variant of window-free streams
        for problem, order, dims, sellp in (("poisson", 1, (17, 15, 19), "1"), ("elasticity", 1, (7, 6, 8), "1"),
                                            ("poisson", 2, (7, 6, 5), "3"), ("poisson", 3, (4, 4, 5), "2")):
            P = zzz.Part(problem, order, *dims)
            res = {}
            for fused in ("0", "2"):
                os.environ["ZZZ_CG_FUSED"], os.environ["ZZZ_SELLP"] = fused, sellp
                with zzz.Context(0) as c:
                    c.upload_part(P)
                    c.pattern_build()
                    c.assemble_matrix(P.form)
                    c.assemble_vector(P.form)
                    out = []
                    for kw in (dict(pc=zzz.PC_JACOBI, rtol=1e-8), dict(pc=zzz.PC_NONE, norm=zzz.NORM_UNPRECONDITIONED, rtol=1e-7),
                               dict(pc=zzz.PC_JACOBI, norm=zzz.NORM_NATURAL, rtol=1e-8), dict(pc=zzz.PC_JACOBI, rtol=1e-30, max_it=9),
                               dict(variant=zzz.CG_CGH, pc=zzz.PC_NONE, rtol=1e-6, max_it=100)):
                        if kw.get("variant") == zzz.CG_CGH:
                            c.vec_upload(zzz.VEC_U, np.zeros(P.n_owned * P.bs))
                        it, rn, r0 = c.cg_solve(**kw)
                        assert c.cg_fused() == (fused == "2")
                        out.append((it, rn, r0, c.cg_history(it + 1), c.vec_download(zzz.VEC_U)))
                    res[fused] = out
            for a, b in zip(res["0"], res["2"]):
                assert a[0] == b[0] and a[1] == b[1] and a[2] == b[2]
                np.testing.assert_array_equal(a[3], b[3])
                np.testing.assert_array_equal(a[4], b[4])
    finally:
        for k, val in old.items():
            if val is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = val


@pytest.mark.parametrize("order,dims,nc,t", [
    (1, (26, 23, 19), 512, 256), (1, (26, 23, 19), 1024, 128), (1, (40, 31, 37), 0, 0), (1, (9, 8, 7), 256, 64 * 4),
    (2, (14, 13, 11), 256, 256), (2, (20, 17, 19), 0, 0), (3, (9, 8, 7), 256, 128), (3, (13, 12, 14), 0, 0),
])
def test_matrix_free_cell_blocks_against_oracle(ctx, order, dims, nc, t):
    """The one-pass matrix-free kernel (csrc/zzz_matfree.hip) on plans of MANY cell blocks (the small cases of
    test_matrix_free_operator_and_cg fit one block): dofs shared between blocks, partial sums finished in block order,
    rounds of the in-LDS accumulation.  y = action(x) against the oracle's serial assembly of form M
    (src/cgpoisson_problem.cpp:193-230), constrained rows zero, bit-identical from call to call and from plan to plan."""
    zo.set_num_threads(8)
    env = {"ZZZ_MF_NC": str(nc), "ZZZ_MF_T": str(t)} if nc else {}
    old = {k: os.environ.get(k) for k in ("ZZZ_MF_NC", "ZZZ_MF_T")}
    try:
        for k in old:
            os.environ.pop(k, None)
        os.environ.update(env)
        P = zzz.Part("poisson", order, *dims)
        ctx.upload_part(P)
        ctx.matfree_setup()
        info = ctx.matfree_info()
        assert info["valid"] == 1 and info["blocks"] > 1 and info["shared_dofs"] > 0
        if nc:
            assert info["cells_per_block"] <= nc and info["threads"] == t
        bc = P.bc_marker()
        v = np.random.default_rng(order).standard_normal(P.n_owned)
        oy = zo.action_poisson(order, P.x, P.cells, P.cell_dofs, bc, v)
        y = ctx.action(v)
        assert np.abs(y - oy).max() <= 1e-12 * np.abs(oy).max()
        assert np.all(y[bc.astype(bool)] == 0)
        for _ in range(4):
            np.testing.assert_array_equal(ctx.action(v), y)
        ctx.matfree_setup()  # a second plan of the same mesh is the same plan
        np.testing.assert_array_equal(ctx.action(v), y)
        # Dirichlet set changed: the plan follows (it carries the markers)
        ctx.upload_bc(np.zeros(0, np.int32))
        oy0 = zo.action_poisson(order, P.x, P.cells, P.cell_dofs, np.zeros_like(bc), v)
        y0 = ctx.action(v)
        assert np.abs(y0 - oy0).max() <= 1e-12 * np.abs(oy0).max()
        # symmetry and constants in the kernel of the unconstrained operator
        w = np.random.default_rng(7).standard_normal(P.n_owned)
        assert abs(w @ y0 - v @ ctx.action(w)) <= 1e-10 * abs(w @ y0)
        assert np.abs(ctx.action(np.ones(P.n_owned))).max() <= 1e-9 * np.abs(y0).max()
        # the two-pass form of rounds 1-3 (the fallback for meshes the plan cannot hold) computes the same operator
        ctx.upload_bc(np.nonzero(bc)[0].astype(np.int32))
        ctx.pattern_build()
        os.environ["ZZZ_MF_LEGACY"] = "1"
        yl = ctx.action(v)
        del os.environ["ZZZ_MF_LEGACY"]
        assert np.abs(yl - y).max() <= 1e-12 * np.abs(oy).max()
        np.testing.assert_array_equal(ctx.action(v), y)
    finally:
        os.environ.pop("ZZZ_MF_LEGACY", None)
        for k, val in old.items():
            if val is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = val

"""GPU parity tests of round 6: the block-row product of block size 3 (csrc/zzz_sellp_blk.hip; PETSc MatMult inside
KSPSolve, src/elasticity_problem.cpp:250-259) and the pattern build on meshes whose cell array ends on a page boundary.

Bars as in test_gpu_parity.py: the product BIT-EXACT against the oracle's serial CSR loop (zo.spmv), iteration counts within
+-2, solutions 1e-6 (1e-9 between two forms of the library's own product)."""
import os

import numpy as np
import pytest

import zzz
import zzz_oracle as zo

pytestmark = pytest.mark.gpu


class _Env:
    def __init__(self, **kw):
        self.kw = {k: str(v) for k, v in kw.items()}

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kw}
        os.environ.update(self.kw)

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _run(P, x, **env):
    with _Env(**env):
        with zzz.Context(0) as c:
            c.upload_part(P)
            c.pattern_build()
            c.assemble_matrix(P.form)
            c.assemble_vector(P.form)
            y = c.spmv(x)
            vi = c.spmv_values_info()
            it, rn, r0 = c.cg_solve(pc=zzz.PC_JACOBI, rtol=1e-8)
            u = c.vec_download(zzz.VEC_U)
            its, _, _ = c.cg_solve(pc=zzz.PC_JACOBI, rtol=1e-8, single_reduction=True)
            us = c.vec_download(zzz.VEC_U)
            itc, _, _ = c.cg_solve(pc=zzz.PC_CHEBYSHEV_JACOBI, rtol=1e-8)  # (epilogue products: the generic kernel serves them)
            uc = c.vec_download(zzz.VEC_U)
            csr = c.csr_download()
    return dict(y=y, vi=vi, it=it, u=u, its=its, us=us, itc=itc, uc=uc, csr=csr, rel=rn / r0)


@pytest.mark.parametrize("order,dims", [(1, (1, 1, 1)), (1, (2, 1, 3)), (1, (5, 3, 4)), (1, (9, 11, 7)), (1, (16, 16, 16)),
                                        (1, (32, 32, 32)), (1, (64, 32, 16)), (1, (44, 40, 36)), (2, (4, 4, 3)), (2, (8, 8, 8)), (3, (3, 2, 2)),
                                        (3, (5, 5, 4))])
def test_block_row_product_is_the_serial_csr_loop(order, dims):
    """Elasticity P1-P3: the product in block-row form == zo.spmv bit for bit == the generic stream's product
    (ZZZ_SELLP_BLK=0); CG in the classical and the single-reduction form agrees with the generic kernel's solve
    (iterations +-1, solution 1e-9) and with the oracle (iterations +-2, 1e-6); Chebyshev-Jacobi (its products carry an
    epilogue and stay on the generic kernel) is unchanged."""
    zo.set_num_threads(4)
    P = zzz.Part("elasticity", order, *dims)
    x = np.random.default_rng(11).standard_normal(P.n_owned * 3)
    a = _run(P, x, ZZZ_SELLP_BLK=2, ZZZ_SELLP=2)
    b = _run(P, x, ZZZ_SELLP_BLK=0, ZZZ_SELLP=2)
    assert not b["vi"]["block_rows"]
    if order > 1 and dims != (4, 4, 3) and not a["vi"]["block_rows"]:
        # (P2 / P3 elasticity on all but tiny meshes: more than 2 048 distinct values -- declined, the generic stream serves it)
        np.testing.assert_array_equal(a["y"], b["y"])
        return
    assert a["vi"]["block_rows"], a["vi"]
    assert a["vi"]["block_table_entries"] >= 2 and a["vi"]["block_chunks"] >= 1
    # few distinct blocks (dyadic coordinates, tiny meshes): the table's rows in LDS; else rows of offsets + a value dictionary
    assert a["vi"]["block_form"] == (1 if a["vi"]["block_table_entries"] <= 2200 else 2)
    if dims == (44, 40, 36):
        assert a["vi"]["block_form"] == 2
    rp, cl, v = a["csr"]
    oy = zo.spmv(rp.astype(np.int64), cl, v, x)
    np.testing.assert_array_equal(a["y"], oy)
    np.testing.assert_array_equal(a["y"], b["y"])
    for k in ("it", "its", "itc"):
        assert abs(a[k] - b[k]) <= 2, (k, a[k], b[k])  # (the workgroups' partial sums of <p, A p> are added in another order)
    for k in ("u", "us", "uc"):
        assert np.linalg.norm(a[k] - b[k]) <= 1e-9 * np.linalg.norm(b[k]), k
    assert a["rel"] <= 1e-8
    if P.n_owned <= 40000:
        ob = zo.assemble_vector(P.form, order, P.x, P.cells, P.cell_dofs, P.f, P.g, None, P.bc_marker())
        oit, ou, _, _ = zo.pcg(rp.astype(np.int64), cl, v, ob, rtol=1e-8)
        assert abs(a["it"] - oit) <= 2 and np.linalg.norm(a["u"] - ou) <= 1e-6 * np.linalg.norm(ou)


def test_block_row_form_declines_what_it_cannot_hold():
    """Columns beyond 16-bit codes (a random numbering kept as it is), or a mesh with more distinct values than the LDS dictionary
    holds (the unstructured ring-with-spurs mesh): the generic stream serves the product, same bits.  30^3 sub-cubes (~9 800 bit
    patterns of blocks, coordinates i / 30) take the form with the rows in memory."""
    zo.set_num_threads(8)
    P = zzz.Part("elasticity", 1, 30, 30, 30)
    x = np.random.default_rng(12).standard_normal(P.n_owned * 3)
    a = _run(P, x, ZZZ_SELLP_BLK=2, ZZZ_SELLP=2)
    assert a["vi"]["block_rows"] and a["vi"]["block_form"] == 2 and a["vi"]["block_table_entries"] > 2200
    rp, cl, v = a["csr"]
    np.testing.assert_array_equal(a["y"], zo.spmv(rp.astype(np.int64), cl, v, x))
    U = zzz.Part.spoke("elasticity", 1, 6)
    xu = np.random.default_rng(14).standard_normal(U.n_owned * 3)
    w = _run(U, xu, ZZZ_SELLP_BLK=2, ZZZ_SELLP=2)
    assert not w["vi"]["block_rows"]
    rp, cl, v = w["csr"]
    np.testing.assert_array_equal(w["y"], zo.spmv(rp.astype(np.int64), cl, v, xu))
    Q = zzz.Part("elasticity", 1, 41, 41, 40).renumbered("random", seed=3)
    xq = np.random.default_rng(13).standard_normal(Q.n_owned * 3)
    q = _run(Q, xq, ZZZ_SELLP_BLK=2, ZZZ_SELLP=2, ZZZ_RENUMBER=0)
    assert not q["vi"]["block_rows"]
    rp, cl, v = q["csr"]
    np.testing.assert_array_equal(q["y"], zo.spmv(rp.astype(np.int64), cl, v, xq))
    # ... and put into the library's own order the same feed qualifies again
    r = _run(Q, xq, ZZZ_SELLP_BLK=2, ZZZ_SELLP=2)
    assert r["vi"]["block_rows"]
    assert np.abs(r["y"] - q["y"]).max() <= 1e-12 * np.abs(q["y"]).max()


@pytest.mark.parametrize("nparts", [2, 3])
def test_block_rows_on_a_partition(nparts):
    """z-slab partitions through the host-mediated communicator on one GPU: interior / boundary SLICE lists of the block-row
    kernel (64 nodes each; the generic kernel's groups are 256 rows), halo overlap, the all-reduced scalars.  Product of a global
    vector to round-off of the single-rank one (ghost columns sort last locally), solve +-1 iteration / 1e-9."""
    import threading

    dims = (12, 12, 21)
    G = zzz.Part("elasticity", 1, *dims)
    xg = np.random.default_rng(5).standard_normal(G.n_owned * 3)
    g = _run(G, xg, ZZZ_SELLP_BLK=2, ZZZ_SELLP=2)
    assert g["vi"]["block_rows"]
    grp = zzz.LocalGroup(nparts)
    out, err = [None] * nparts, []

    def run(rank):
        try:
            P = zzz.Part("elasticity", 1, *dims, nparts, rank)
            with zzz.Context(0) as c:
                c.comm_init_local(grp.h, rank)
                c.upload_part(P)
                c.upload_halo(P)
                c.pattern_build()
                c.assemble_matrix(P.form)
                c.assemble_vector(P.form)
                lo, hi = P.own_offset * 3, (P.own_offset + P.n_owned) * 3
                y = c.spmv(xg[lo:hi])
                blk = c.spmv_values_info()["block_rows"]
                it, rn, r0 = c.cg_solve(pc=zzz.PC_JACOBI, rtol=1e-8)
                u = c.vec_download(zzz.VEC_U)
                its, _, _ = c.cg_solve(pc=zzz.PC_JACOBI, rtol=1e-8, single_reduction=True)
                out[rank] = (P.own_offset, y, blk, it, u, its, c.vec_download(zzz.VEC_U))
        except Exception as e:  # noqa: BLE001
            err.append((rank, repr(e)))

    with _Env(ZZZ_SELLP_BLK=2, ZZZ_SELLP=2):
        th = [threading.Thread(target=run, args=(r,)) for r in range(nparts)]
        for t in th:
            t.start()
        for t in th:
            t.join(timeout=300)
    grp.close()
    assert not err, err
    assert all(o is not None and o[2] for o in out), [o and o[2] for o in out]
    y = np.concatenate([o[1] for o in out])
    u = np.concatenate([o[4] for o in out])
    us = np.concatenate([o[6] for o in out])
    assert np.abs(y - g["y"]).max() <= 1e-13 * np.abs(g["y"]).max()
    assert len({o[3] for o in out}) == 1 and abs(out[0][3] - g["it"]) <= 1
    assert len({o[5] for o in out}) == 1 and abs(out[0][5] - g["its"]) <= 1
    assert np.linalg.norm(u - g["u"]) <= 1e-9 * np.linalg.norm(g["u"])
    assert np.linalg.norm(us - g["us"]) <= 1e-9 * np.linalg.norm(g["us"])


@pytest.mark.parametrize("problem,order,dims", [("poisson", 1, (64, 32, 32)), ("elasticity", 1, (32, 64, 32)),
                                                ("poisson", 1, (64, 64, 64)), ("poisson", 2, (16, 32, 32))])
def test_pattern_build_when_the_connectivity_ends_on_a_page_boundary(problem, order, dims):
    """6 * 2^k cells: the connectivity array is a whole number of 2 MiB pages and any read one entry past it faults.  The
    sort-free adjacency build did exactly that for the empty range of the last run (rounds 3-5; every mesh of 64 x 32 x 32
    or 64^3 sub-cubes crashed in zzz_csr_pattern_build).  Pattern against the oracle's, index for index."""
    zo.set_num_threads(8)
    P = zzz.Part(problem, order, *dims)
    with zzz.Context(0) as c:
        c.upload_part(P)
        c.pattern_build()
        rp, cl, _ = c.csr_download(values=False)
    orp, ocl = zo.pattern(P.n_owned, P.cell_dofs, P.bs)
    np.testing.assert_array_equal(rp, orp)
    np.testing.assert_array_equal(cl, ocl)


# ---- long scalar rows: the block-window product (csrc/zzz_sellp_win.hip) ------------------------------------------------------
def _run_poisson(P, x, **env):
    with _Env(**env):
        with zzz.Context(0) as c:
            c.upload_part(P)
            c.pattern_build()
            c.assemble_matrix(P.form)
            c.assemble_vector(P.form)
            y = c.spmv(x)
            vi = c.spmv_values_info()
            it, rn, r0 = c.cg_solve(pc=zzz.PC_JACOBI, rtol=1e-8)
            u = c.vec_download(zzz.VEC_U)
            its, _, _ = c.cg_solve(pc=zzz.PC_JACOBI, rtol=1e-8, single_reduction=True)
            us = c.vec_download(zzz.VEC_U)
            itc, _, _ = c.cg_solve(pc=zzz.PC_CHEBYSHEV_JACOBI, rtol=1e-8)
            uc = c.vec_download(zzz.VEC_U)
            rp, cl, v = c.csr_download()
            # a second set of values on the same pattern: the structure (row order, windows) is kept, the value codes refreshed
            v2 = -3.0 * v  # (as few distinct values as before: the block dictionaries must still hold them)
            c.csr_upload_values(v2)
            y2 = c.spmv(x)
            vi2 = c.spmv_values_info()
    return dict(y=y, vi=vi, it=it, u=u, its=its, us=us, itc=itc, uc=uc, csr=(rp, cl, v), rel=rn / r0, y2=y2, v2=v2, vi2=vi2)


@pytest.mark.parametrize("order,dims", [(3, (4, 3, 5)), (3, (12, 12, 12)), (3, (20, 6, 9)), (2, (10, 9, 11)), (2, (24, 24, 24))])
def test_block_window_product_is_the_serial_csr_loop(order, dims):
    """Poisson P2 / P3 with the product forced into the block-window form (rows in Morton blocks of 4 096, x from a window in LDS,
    values from a dictionary in LDS): == zo.spmv bit for bit == the generic stream's product (ZZZ_SELLP_BWIN=0), also after new
    values arrive on the same pattern; CG (classical, single reduction) agrees with the generic kernel's solve and the oracle."""
    zo.set_num_threads(4)
    P = zzz.Part("poisson", order, *dims)
    x = np.random.default_rng(21).standard_normal(P.n_owned)
    a = _run_poisson(P, x, ZZZ_SELLP_BWIN=2, ZZZ_SELLP=2)
    b = _run_poisson(P, x, ZZZ_SELLP_BWIN=0, ZZZ_SELLP=2)
    assert a["vi"]["row_windows"] and not b["vi"]["row_windows"], (a["vi"], b["vi"])
    assert a["vi"]["block_table_entries"] >= P.n_owned and a["vi"]["block_chunks"] >= 1 and a["vi"]["block_form"] == -(-P.n_owned // 4096)
    rp, cl, v = a["csr"]
    np.testing.assert_array_equal(a["y"], zo.spmv(rp.astype(np.int64), cl, v, x))
    np.testing.assert_array_equal(a["y"], b["y"])
    assert a["vi2"]["row_windows"]
    np.testing.assert_array_equal(a["y2"], zo.spmv(rp.astype(np.int64), cl, a["v2"], x))
    for k in ("it", "its", "itc"):
        assert abs(a[k] - b[k]) <= 2, (k, a[k], b[k])
    for k in ("u", "us", "uc"):
        assert np.linalg.norm(a[k] - b[k]) <= 1e-9 * np.linalg.norm(b[k]), k
    assert a["rel"] <= 1e-8
    if P.n_owned <= 60000:
        ob = zo.assemble_vector(P.form, order, P.x, P.cells, P.cell_dofs, P.f, P.g, P.facets, P.bc_marker())
        oit, ou, _, _ = zo.pcg(rp.astype(np.int64), cl, v, ob, rtol=1e-8)
        assert abs(a["it"] - oit) <= 2 and np.linalg.norm(a["u"] - ou) <= 1e-6 * np.linalg.norm(ou)


def test_block_window_form_on_a_mesh_that_is_no_lattice_and_on_short_rows():
    """The ring-with-spurs mesh (P3): rows ordered by their nodes' coordinates whatever the mesh; if a block's distinct values
    exceed the dictionary the generic stream serves the product -- the same bits either way.  P1 (15 entries per row) never
    takes the form."""
    zo.set_num_threads(8)
    U = zzz.Part.spoke("poisson", 3, 3)
    xu = np.random.default_rng(22).standard_normal(U.n_owned)
    w = _run_poisson(U, xu, ZZZ_SELLP_BWIN=2, ZZZ_SELLP=2)
    rp, cl, v = w["csr"]
    np.testing.assert_array_equal(w["y"], zo.spmv(rp.astype(np.int64), cl, v, xu))
    np.testing.assert_array_equal(w["y2"], zo.spmv(rp.astype(np.int64), cl, w["v2"], xu))
    P1 = zzz.Part("poisson", 1, 20, 18, 19)
    x1 = np.random.default_rng(23).standard_normal(P1.n_owned)
    q = _run_poisson(P1, x1, ZZZ_SELLP_BWIN=2, ZZZ_SELLP=2)
    assert not q["vi"]["row_windows"]


@pytest.mark.parametrize("nparts", [2, 3])
def test_block_windows_on_a_partition(nparts):
    """z-slab partitions through the host-mediated communicator on one GPU: interior / boundary BLOCK lists of the block-window
    kernel, halo overlap, all-reduced scalars; product of a global vector to round-off of the single-rank one, solves +-1 / 1e-9."""
    import threading

    dims = (5, 5, 12)
    G = zzz.Part("poisson", 3, *dims)
    xg = np.random.default_rng(6).standard_normal(G.n_owned)
    g = _run_poisson(G, xg, ZZZ_SELLP_BWIN=2, ZZZ_SELLP=2)
    assert g["vi"]["row_windows"]
    grp = zzz.LocalGroup(nparts)
    out, err = [None] * nparts, []

    def run(rank):
        try:
            P = zzz.Part("poisson", 3, *dims, nparts, rank)
            with zzz.Context(0) as c:
                c.comm_init_local(grp.h, rank)
                c.upload_part(P)
                c.upload_halo(P)
                c.pattern_build()
                c.assemble_matrix(P.form)
                c.assemble_vector(P.form)
                lo, hi = P.own_offset, P.own_offset + P.n_owned
                y = c.spmv(xg[lo:hi])
                win = c.spmv_values_info()["row_windows"]
                it, rn, r0 = c.cg_solve(pc=zzz.PC_JACOBI, rtol=1e-8)
                u = c.vec_download(zzz.VEC_U)
                its, _, _ = c.cg_solve(pc=zzz.PC_JACOBI, rtol=1e-8, single_reduction=True)
                out[rank] = (P.own_offset, y, win, it, u, its, c.vec_download(zzz.VEC_U))
        except Exception as e:  # noqa: BLE001
            err.append((rank, repr(e)))

    with _Env(ZZZ_SELLP_BWIN=2, ZZZ_SELLP=2):
        th = [threading.Thread(target=run, args=(r,)) for r in range(nparts)]
        for t in th:
            t.start()
        for t in th:
            t.join(timeout=300)
    grp.close()
    assert not err, err
    assert all(o is not None and o[2] for o in out), [o and o[2] for o in out]
    y = np.concatenate([o[1] for o in out])
    u = np.concatenate([o[4] for o in out])
    us = np.concatenate([o[6] for o in out])
    assert np.abs(y - g["y"]).max() <= 1e-13 * np.abs(g["y"]).max()
    assert len({o[3] for o in out}) == 1 and abs(out[0][3] - g["it"]) <= 1
    assert len({o[5] for o in out}) == 1 and abs(out[0][5] - g["its"]) <= 1
    assert np.linalg.norm(u - g["u"]) <= 1e-9 * np.linalg.norm(g["u"])
    assert np.linalg.norm(us - g["us"]) <= 1e-9 * np.linalg.norm(g["us"])


@pytest.mark.parametrize("order,dims", [(2, (8, 8, 8)), (3, (5, 5, 4))])
def test_block_windows_serve_the_vector_valued_spaces_the_block_rows_decline(order, dims):
    """Elasticity P2 / P3: more distinct values than the block-row form's dictionary holds -- their scalar rows (80-170 entries)
    then take the block-window form like any long rows (ordered by their nodes).  == zo.spmv bit for bit."""
    zo.set_num_threads(4)
    P = zzz.Part("elasticity", order, *dims)
    x = np.random.default_rng(31).standard_normal(P.n_owned * 3)
    a = _run(P, x, ZZZ_SELLP_BLK=2, ZZZ_SELLP_BWIN=2, ZZZ_SELLP=2)
    b = _run(P, x, ZZZ_SELLP_BLK=0, ZZZ_SELLP_BWIN=0, ZZZ_SELLP=2)
    assert a["vi"]["block_rows"] or a["vi"]["row_windows"], a["vi"]
    assert not b["vi"]["block_rows"] and not b["vi"]["row_windows"]
    rp, cl, v = a["csr"]
    np.testing.assert_array_equal(a["y"], zo.spmv(rp.astype(np.int64), cl, v, x))
    np.testing.assert_array_equal(a["y"], b["y"])
    for k in ("it", "its", "itc"):
        assert abs(a[k] - b[k]) <= 2, (k, a[k], b[k])
    for k in ("u", "us", "uc"):
        assert np.linalg.norm(a[k] - b[k]) <= 1e-9 * np.linalg.norm(b[k]), k


@pytest.mark.parametrize("problem,order,dims,knobs", [("elasticity", 1, (20, 18, 22), dict(ZZZ_SELLP_BLK=2)),
                                                       ("poisson", 3, (8, 7, 9), dict(ZZZ_SELLP_BWIN=2)),
                                                       ("elasticity", 2, (6, 5, 6), dict(ZZZ_SELLP_BWIN=2))])
def test_form_chosen_at_assembly_time_equals_form_chosen_at_the_first_product(problem, order, dims, knobs):
    """Last third of round 6: an unpartitioned matrix gets its special form inside zzz_assemble and then no generic operator
    stream at all (sell_update); ZZZ_SELLP_EARLY=0 packs the stream first and builds the form at the first product as before.
    Same form, same product bit for bit (= the serial CSR loop), same solves; values uploaded afterwards take the same path;
    the Chebyshev epilogue and the plain product API run on a matrix that has no generic stream."""
    zo.set_num_threads(4)
    P = zzz.Part(problem, order, *dims)
    x = np.sin(0.41 * np.arange(P.n_owned * P.bs)) + 0.3
    res = {}
    for early in ("1", "0"):
        with _Env(ZZZ_SELLP_EARLY=early, **knobs):
            with zzz.Context(0) as c:
                c.upload_part(P)
                c.pattern_build()
                c.assemble_matrix(P.form)
                c.assemble_vector(P.form)
                raw_before = c.spmv_info_raw()
                y = c.spmv(x)
                vi = c.spmv_values_info()
                it, rn, r0 = c.cg_solve(pc=zzz.PC_JACOBI, rtol=1e-9)
                u = c.vec_download(zzz.VEC_U)
                itc, _, _ = c.cg_solve(pc=zzz.PC_CHEBYSHEV_JACOBI, rtol=1e-9)
                uc = c.vec_download(zzz.VEC_U)
                rp, cl, v = c.csr_download()
                v2 = -2.5 * v
                c.csr_upload_values(v2)
                y2 = c.spmv(x)
                vi2 = c.spmv_values_info()
                res[early] = dict(y=y, vi=vi, it=it, u=u, itc=itc, uc=uc, y2=y2, vi2=vi2, entries=int(raw_before[7]))
    a, b = res["1"], res["0"]
    assert a["vi"]["special_form"] == b["vi"]["special_form"] != "", (a["vi"], b["vi"])
    assert a["vi2"]["special_form"] == a["vi"]["special_form"]
    assert a["entries"] == 0 and b["entries"] > 0  # (no generic stream was packed / one was)
    np.testing.assert_array_equal(a["y"], zo.spmv(rp.astype(np.int64), cl, v, x))
    np.testing.assert_array_equal(a["y"], b["y"])
    np.testing.assert_array_equal(a["y2"], zo.spmv(rp.astype(np.int64), cl, v2, x))
    np.testing.assert_array_equal(a["y2"], b["y2"])
    assert a["it"] == b["it"] and a["itc"] == b["itc"]
    np.testing.assert_array_equal(a["u"], b["u"])
    np.testing.assert_array_equal(a["uc"], b["uc"])


@pytest.mark.parametrize("dims", [(1, 1, 1), (3, 2, 4), (12, 11, 13), (31, 17, 9)])
def test_elasticity_p1_matrix_by_node_equals_matrix_by_row(dims):
    """asm_matrix_p1_node3 (a thread per node: one geometry and one block-column search per (node, cell) pair) writes the values
    asm_matrix_p1<3> (a thread per scalar row) writes, bit for bit, boundary rows and columns included (fem::assemble_matrix +
    set_diagonal, src/elasticity_problem.cpp:199-213; against the oracle: the golden and medium-size cases of test_gpu_assembly.py
    run the by-node kernel)."""
    P = zzz.Part("elasticity", 1, *dims)
    vals = {}
    for k in ("1", "0"):
        with _Env(ZZZ_ASM_NODE3=k):
            with zzz.Context(0) as c:
                c.upload_part(P)
                c.pattern_build()
                c.assemble_matrix(P.form)
                vals[k] = c.csr_download()
    for q in range(3):
        np.testing.assert_array_equal(vals["1"][q], vals["0"][q])

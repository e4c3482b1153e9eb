"""Worker of test_peer_memory_allreduce_between_processes: one rank = one PROCESS, both on GPU 0.
Each rank solves the same complete problem; the 2-rank peer-only communicator doubles every dot
product (exactly, in fp64), so the iteration must reproduce the un-communicated solve bit for bit
-- if and only if the mailbox transport (hipIpcOpenMemHandle mapping, system-scope stores, tags)
delivers every value of every round."""
import numpy as np


def run(rank, nranks, conn, single_reduction):
    try:
        import zzz

        P = zzz.Part("poisson", 1, 14, 12, 13)
        with zzz.Context(0) as c:
            c.upload_part(P)
            c.pattern_build()
            c.assemble_matrix(P.form)
            c.assemble_vector(P.form)
            it0, rn0, r00 = c.cg_solve(pc=zzz.PC_JACOBI, rtol=1e-9, single_reduction=single_reduction)
            u0 = c.vec_download(zzz.VEC_U)
            nrm0 = c.vec_norm(zzz.VEC_U)
            c.comm_init_peer_only(nranks, rank)
            conn.send(c.comm_p2p_export())
            enabled = c.comm_p2p_attach(conn.recv())
            if not enabled:
                conn.send(("disabled",))
                return
            res = []
            for _ in range(3):
                it, rn, r0 = c.cg_solve(pc=zzz.PC_JACOBI, rtol=1e-9, single_reduction=single_reduction)
                res.append((it, rn / r0, bool(np.array_equal(c.vec_download(zzz.VEC_U), u0))))
            nrm = c.vec_norm(zzz.VEC_U)  # la::norm over both (replicated) ranks: sqrt(2) x the single-rank norm
            conn.send(("ok", it0, rn0 / r00, res, nrm, nrm0))
    except Exception as e:  # noqa: BLE001
        conn.send(("error", repr(e)))


def run_partition(rank, nranks, conn, problem, order, dims, desert=False):
    """Worker of test_peer_memory_halo_between_processes: one rank = one PROCESS, all on GPU 0, each with its z-slab of
    one partitioned problem and a communicator that has NO transport but the peer memory (zzz_comm_init_peer_only):
    the forward halo of every product travels as device stores into the neighbour's window (mapped here through
    hipIpcOpenMemHandle), the scalars through the mailboxes."""
    try:
        import zzz

        P = zzz.Part(problem, order, *dims, nranks, rank)
        with zzz.Context(0) as c:
            c.comm_init_peer_only(nranks, rank)
            conn.send(c.comm_p2p_export())
            enabled = c.comm_p2p_attach(conn.recv())
            if not enabled:
                conn.send(("disabled",))
                return
            if rank % 2 == 0:
                c.upload_part(P)
                c.upload_halo(P)
            else:
                c.cube_generate(problem, order, *dims, nranks, rank)
            c.pattern_build()
            c.assemble_matrix(P.form)
            c.assemble_vector(P.form)
            info = c.comm_info()
            if desert:
                # rank 1 leaves before the first exchange: rank 0 must get an error within the bound of its wait
                if rank == 1:
                    conn.send(("ok", {"deserted": True}))
                    return
                import time

                t0 = time.perf_counter()
                try:
                    c.spmv(np.ones(P.n_owned * P.bs))
                    verdict = "no error"
                except zzz.ZzzError as e:
                    verdict = repr(e)
                conn.send(("ok", {"deserted": False, "verdict": verdict, "seconds": time.perf_counter() - t0}))
                return
            lo, hi = P.own_offset * P.bs, (P.own_offset + P.n_owned) * P.bs
            x = np.sin(0.37 * np.arange(lo, hi))           # a known global vector: halo + product
            y = c.spmv(x)
            out = {"info": info, "offset": P.own_offset, "y": y}
            for name, kw in (("jacobi", dict(pc=zzz.PC_JACOBI)), ("sr", dict(pc=zzz.PC_JACOBI, single_reduction=True)),
                             ("cheb", dict(pc=zzz.PC_CHEBYSHEV_JACOBI)),
                             ("cheb_sr", dict(pc=zzz.PC_CHEBYSHEV_JACOBI, single_reduction=True))):
                it, rn, r0 = c.cg_solve(rtol=1e-9, **kw)
                out[name] = (it, rn / r0, c.vec_download(zzz.VEC_U), c.vec_norm(zzz.VEC_U))
            # the same solve with the window switched off has no transport left: an error, not a hang
            c.comm_p2p_halo(False)
            try:
                c.cg_solve(pc=zzz.PC_JACOBI, rtol=1e-9)
                out["no_transport"] = "solved"
            except zzz.ZzzError as e:
                out["no_transport"] = repr(e)
            conn.send(("ok", out))
    except Exception as e:  # noqa: BLE001
        conn.send(("error", repr(e)))

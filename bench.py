#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path on MI355X.

Metric (BASELINE.json): DoF/s for `ZZZ Assemble matrix` + `ZZZ Assemble vector` + `ZZZ Solve`,
and the CG-SpMV achieved HBM GB/s against the 8 TB/s peak.

One "step" = one pass of the hot path over the synthetic cube problem: build the sparsity pattern
(create_matrix, inside the reference's `ZZZ Assemble` umbrella), assemble A, assemble b, solve
A u = b with Jacobi-preconditioned CG to rtol 1e-8 (the reference run
`--problem_type poisson --order 1 --scaling_type strong --ndofs 10000000 -ksp_type cg
-pc_type jacobi -ksp_rtol 1e-8`, BASELINE.json configs[1]; mesh 108x103x111 refined once ==
216x206x222, 10 016 937 dofs, src/mesh.cpp:78-151).  Inputs (mesh, dofmap, Dirichlet set, nodal
coefficients) are resident in HBM before the timed region.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

For N > 1 the cube is cut into N z-slabs, one per GPU (strong scaling: the total problem is
fixed); inside libzzz_hip the halo exchange runs on RCCL and the CG scalars are reduced through
peer-memory mailboxes over xGMI (ncclAllReduce as fallback); torch.distributed (gloo) is plumbing
only: unique-id broadcast, mailbox-handle all_gather, votes, barrier, max over ranks.  The first
warm-up step also times the two CG forms (classical / -ksp_cg_single_reduction) on both all-reduce
transports; the timed steps use the fastest (config.cg_form_tuning_s, config.workload).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "performance-test_amd"))

import numpy as np  # noqa: E402

import zzz  # noqa: E402  (ctypes mirror of include/zzz_abi.h; loads libzzz_hip.so, fails loudly if absent)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


class deadline:
    """Bound on a step that other ranks take part in (communicator bootstrap, gloo collectives, a warm-up or timed
    region whose collectives could block on a rank that has died): when it does not finish in time the process
    prints why and ends with a NON-ZERO status by a plain exit -- never a hang, never a re-exec."""

    def __init__(self, seconds, what, last_words=None):
        import threading

        self.what, self.seconds, self.last_words = what, seconds, last_words
        self.t = threading.Timer(seconds, self._fire)
        self.t.daemon = True

    def _fire(self):
        sys.stderr.write(f"bench.py: '{self.what}' did not finish within {self.seconds} s "
                         f"(rank {os.environ.get('RANK', '0')}); giving up\n")
        sys.stderr.flush()
        if self.last_words is not None:
            # an extra beside a measurement that is already complete: its line is printed, not lost -- but a solve that
            # hangs is a library bug or a dead peer, so the status is non-zero all the same
            self.last_words()
            sys.stdout.flush()
        os._exit(3)

    def __enter__(self):
        self.t.start()
        return self

    def __exit__(self, *a):
        self.t.cancel()
        return False


def spmv_algorithmic_bytes(n, nnz):
    # fp64 values + int32 columns + int32 row pointers + x read once + y written
    return 12 * nnz + 4 * (n + 1) + 16 * n


def pmc_traffic(nrows, nnz, streamed=None):
    """HBM bytes per SpMV launch from the committed rocprofv3 PMC passes (profiles/): FETCH_SIZE and
    WRITE_SIZE are collected in separate runs of this same command (they cannot share a pass and PMC
    collection cannot run inside a timed benchmark), corrected as MI355X_MICROARCH.md prescribes for
    gfx950 (FETCH_SIZE counts half of a coalesced stream: x2; KiB units).  Only reported when a
    profile of the same matrix (rows, nonzeros) and the same operator stream (bytes per product) exists; the latest
    round wins."""
    import glob

    best = (None, None, {})
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_*.json"))):
        try:
            d = json.load(open(path))
            if d["rows"] != nrows or d["nnz"] != nnz:
                continue
            k = d["kernels"]["spmv"]
            # ... and of the same operator stream: a profile taken before the stream's format changed does not count
            if streamed is not None and k.get("bytes_streamed") != streamed:
                continue
            # (traffic_bytes: the calibrated count of tools/profile_summarise.py -- the x2 of FETCH_SIZE does not hold for
            # every access pattern; older profiles carry the two counters only)
            best = (k.get("traffic_bytes", 2 * k["FETCH_SIZE_KiB"] * 1024 + k["WRITE_SIZE_KiB"] * 1024), os.path.relpath(path, ROOT),
                    {"traffic_low": k.get("traffic_low", (k["FETCH_SIZE_KiB"] + k["WRITE_SIZE_KiB"]) * 1024),
                     "traffic_high": k.get("traffic_high", (2 * k["FETCH_SIZE_KiB"] + k["WRITE_SIZE_KiB"]) * 1024),
                     "traffic_kernel": k.get("kernel"), "kernel_rocprof_us": k.get("kernel_rocprof_us"),
                     "traffic_dispatches": k.get("dispatches")})
        except Exception:
            continue
    return best


# BASELINE.json configs as one-GPU workloads (--config): what each one asks of ONE GPU
CONFIGS = {
    # configs[0]: the reference's CPU-runnable case
    "c1": dict(problem_type="poisson", order=1, scaling_type="strong", ndofs=500000, mesh_nproc=1,
               note="BASELINE configs[0] (500 k dofs) on one GPU"),
    # configs[1] (and configs[2] at N > 1): the headline
    "c2": dict(problem_type="poisson", order=1, scaling_type="strong", ndofs=10000000, mesh_nproc=1,
               note="BASELINE configs[1]"),
    # configs[3]: elasticity P1 weak, 500 k dofs per GPU x 8 GPUs = 109^3 sub-cubes, 3 993 000 dofs: the TOTAL problem on one GPU
    "c4_total": dict(problem_type="elasticity", order=1, scaling_type="weak", ndofs=500000, mesh_nproc=8,
                     note="BASELINE configs[3]: the whole 8-GPU weak-scaling problem (mesh of 8 processes) on one GPU"),
    # configs[4]: Poisson P3 strong 50 M dofs over 8 GPUs: one GPU's share of the rows
    # ... and the whole of it on ONE GPU: 49 834 930 dofs, 2 406 964 246 nonzeros (64-bit row pointers, operator stream)
    "c5": dict(problem_type="poisson", order=3, scaling_type="strong", ndofs=50000000, mesh_nproc=1,
               note="BASELINE configs[4] whole on one GPU (2.4 G nonzeros): the largest single-GPU configuration"),
    "c5_rank": dict(problem_type="poisson", order=3, scaling_type="strong", ndofs=6250000, mesh_nproc=1,
                    note="BASELINE configs[4]: the per-GPU share (50 M / 8 dofs) of the P3 problem on one GPU"),
}


def cpu_baseline(P, ctx, iters_gpu):
    """The oracle (CPU restatement, kind 'port') timed on this box's host cores on the same workload: full matrix +
    vector assembly, then the WHOLE Jacobi-PCG solve to the same tolerance (its own iteration count is reported; at C2
    that is ~7 s on the driver's 32 cores, so nothing is extrapolated)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ.setdefault("OMP_PROC_BIND", "spread")  # before libgomp starts: spread threads over the sockets
    os.environ.setdefault("OMP_PLACES", "cores")
    import zzz_oracle as zo

    rowptr32, cols, _ = ctx.csr_download(values=False)
    rowptr = rowptr32.astype(np.int64)  # the oracle's own pattern builder is serial; the GPU's pattern is
    bc = P.bc_marker()                  # bit-identical (tests), so the CPU baseline is not charged for building it
    # threads: the count that streams this matrix fastest on this box (more is not better once the
    # memory channels are saturated or the container's CPU quota is exceeded)
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    xs = np.ones(rowptr.shape[0] - 1)
    ones = np.ones(cols.shape[0])
    best = (None, 1)
    for c in sorted({1, 4, 8, 16, 32, 64, 96, 128, avail}):
        if c > avail:
            continue
        zo.set_num_threads(c)
        zo.pcg(rowptr, cols, ones, xs, rtol=1e-30, max_it=6)  # whole iterations (SpMV, dots, axpys), not SpMV alone
        dt = zo.last_pcg_loop_seconds
        if best[0] is None or dt < best[0]:
            best = (dt, c)
    cores = best[1]
    del ones
    zo.set_num_threads(cores)
    t0 = time.perf_counter()
    vals = zo.assemble_matrix(P.form, P.order, P.x, P.cells, P.cell_dofs, bc, rowptr, cols)
    t1 = time.perf_counter()
    b = zo.assemble_vector(P.form, P.order, P.x, P.cells, P.cell_dofs, P.f, P.g,
                           P.facets if P.form == 0 else None, bc)
    t2 = time.perf_counter()
    t3 = time.perf_counter()
    it_cpu = zo.pcg(rowptr, cols, vals, b, rtol=1e-8, max_it=10000)[0]
    t_solve = time.perf_counter() - t3        # KSPSolve as a whole (working copies, set-up, loop)
    t_iter = zo.last_pcg_loop_seconds / max(it_cpu, 1)  # iteration loop alone; NUMA-aware working copies inside
    # "1 MPI rank" (BASELINE configs[0] is the reference's 1-rank case): Krylov iteration time on one thread
    zo.set_num_threads(1)
    zo.pcg(rowptr, cols, vals, b, rtol=1e-8, max_it=4)
    t_iter1 = zo.last_pcg_loop_seconds / 4
    zo.set_num_threads(cores)
    t_total = (t2 - t0) + t_solve
    n = P.n_owned * P.bs
    # BASELINE configs[0]: "--ndofs 500000, CPU reference path, 1 MPI rank" -- the whole of it on ONE thread (~10 s)
    c1 = None
    try:
        zo.set_num_threads(1)
        nx1, ny1, nz1, r1 = zzz.mesh_size(500000, True, 1, 1, 1)
        P1 = zzz.Part("poisson", 1, nx1 << r1, ny1 << r1, nz1 << r1)
        ta = time.perf_counter()
        rp1, cl1 = zo.pattern(P1.n_owned, P1.cell_dofs, 1)
        tb = time.perf_counter()
        bc1 = P1.bc_marker()
        v1 = zo.assemble_matrix(0, 1, P1.x, P1.cells, P1.cell_dofs, bc1, rp1, cl1)
        b1 = zo.assemble_vector(0, 1, P1.x, P1.cells, P1.cell_dofs, P1.f, P1.g, P1.facets, bc1)
        tc = time.perf_counter()
        it1 = zo.pcg(rp1, cl1, v1, b1, rtol=1e-8, max_it=10000)[0]
        td = time.perf_counter()
        c1 = {"dofs": int(P1.n_owned), "threads": 1, "create_matrix_s": tb - ta, "assemble_s": tc - tb, "solve_s": td - tc,
              "krylov_iterations": int(it1), "dofs_per_s": P1.n_owned / (td - tb),
              "dofs_per_s_with_create_matrix": P1.n_owned / (td - ta)}
        del P1, rp1, cl1, v1, b1
    except Exception as e:  # noqa: BLE001 -- a baseline beside the baseline must not cost the line
        c1 = {"error": repr(e)}
    zo.set_num_threads(cores)
    return {
        "value": n / t_total, "unit": "DoF/s", "cores": cores, "kind": "port",
        "sample": (f"oracle/zzz_oracle.c with OpenMP on {cores} threads, same {n}-dof problem, the whole of it: matrix "
                   f"assembly {t1 - t0:.2f} s + vector assembly {t2 - t1:.2f} s + Jacobi-PCG to 1e-8 in {it_cpu} iterations "
                   f"({t_solve:.2f} s, {t_iter * 1e3:.1f} ms per iteration; the GPU solve took {iters_gpu})"),
        "assemble_s": t2 - t0, "ms_per_cg_iteration": t_iter * 1e3, "solve_s": t_solve, "krylov_iterations": it_cpu,
        "ms_per_cg_iteration_1_thread": t_iter1 * 1e3,
        # BASELINE configs[0] on this box's CPU, one thread: DoF/s of ZZZ Assemble matrix + vector + ZZZ Solve
        "c1_1_thread_dofs_per_s": (c1 or {}).get("dofs_per_s"), "c1_1_thread": c1,
    }


def physical_bytes_per_product(ctx, nrows, nnz, single_reduction=False):
    """Bytes one launch of the CG product addresses: the operator stream (or the packed CSR + row pointers) + x + y (+ r)"""
    sinfo = ctx.spmv_info_raw()
    extra = 8 * nrows if single_reduction else 0
    if sinfo[5]:
        return sinfo[6] + 16 * nrows + extra, sinfo
    return (10 if sinfo[0] else 12) * nnz + 4 * (nrows + 1) + 16 * nrows + extra, sinfo


def run_other_config(name, steps=3, unstructured=None):
    """One of the other BASELINE configurations as a one-GPU workload, OUTSIDE the headline's timed region: the same
    step (pattern + A + b + Jacobi-CG to 1e-8) on a device-generated feed, one warm-up + `steps` timed steps, so that
    the driver's record carries a number for every config, not only builder-run profiles.
    unstructured = target dofs: `--mesh_type unstructured` instead (src/mesh.cpp:209-453, host/spoke_mesh.cpp: a mesh that is
    no lattice; P1 Poisson, the whole exterior boundary constrained), fed through the upload entry points."""
    extra = {}
    if unstructured:
        c = dict(problem_type="poisson", order=1, scaling_type="strong", ndofs=unstructured, mesh_nproc=1,
                 note="--mesh_type unstructured: the ring-with-spurs mesh, every block cut m x m x m; no lattice")
        bs, form = 1, zzz.FORM_POISSON
        t0 = time.perf_counter()
        m = zzz.host().zzzh_spoke_size(int(unstructured), 1)
        P = zzz.Part.spoke("poisson", 1, m)
        extra["feed_s"] = time.perf_counter() - t0
        extra["mesh"] = f"119 blocks x {m}^3 sub-blocks x 6 tetrahedra, {P.nverts} vertices, {P.ncells} cells"
    else:
        c = CONFIGS[name]
        bs = 3 if c["problem_type"] == "elasticity" else 1
        nx, ny, nz, r = zzz.mesh_size(c["ndofs"], c["scaling_type"] == "strong", c["mesh_nproc"], bs, c["order"])
        nx, ny, nz = nx << r, ny << r, nz << r
        form = zzz.FORM_ELASTICITY if bs == 3 else zzz.FORM_POISSON
    with zzz.Context(0) as ctx:
        if unstructured:
            ctx.upload_part(P)
            info = [P.n_owned, P.ncells]
            del P
        else:
            info = ctx.cube_generate(c["problem_type"], c["order"], nx, ny, nz, 1, 0)
        ph = {"pattern": [], "assemble_matrix": [], "assemble_vector": [], "solve": []}
        it = 0
        t_all = 0.0
        for k in range(steps + 1):
            ctx.sync()
            t0 = time.perf_counter()
            ctx.pattern_build()
            ctx.sync()
            t1 = time.perf_counter()
            ctx.assemble_matrix(form)
            ctx.sync()
            t2 = time.perf_counter()
            ctx.assemble_vector(form)
            ctx.sync()
            t3 = time.perf_counter()
            it, rn, r0 = ctx.cg_solve(variant=zzz.CG_PETSC, pc=zzz.PC_JACOBI, rtol=1e-8, max_it=10000, profile=True)
            ctx.sync()
            t4 = time.perf_counter()
            if k == 0:
                continue  # warm-up
            for key, dt in zip(("pattern", "assemble_matrix", "assemble_vector", "solve"), (t1 - t0, t2 - t1, t3 - t2, t4 - t3)):
                ph[key].append(dt)
            t_all += t4 - t0
        nrows, _, nnz = ctx.csr_sizes()
        spmv_ms, spmv_n = ctx.profile()
        bytes_pl, sinfo = physical_bytes_per_product(ctx, nrows, nnz)
        ctx_windows = ctx.spmv_x_windows()[0] > 0
        values = ctx.spmv_values_info()
        unorm = ctx.vec_norm(zzz.VEC_U)
        if name in ("c4_total", "c5_rank"):
            # beside the measurement: one solve with the library's polynomial preconditioner on the same system (its terms ride on
            # the same product kernel as epilogues)
            ctx.sync()
            tc = time.perf_counter()
            itc, rnc, r0c = ctx.cg_solve(variant=zzz.CG_PETSC, pc=zzz.PC_CHEBYSHEV_JACOBI, rtol=1e-8, max_it=10000)
            ctx.sync()
            extra["alt_preconditioner"] = {"pc_type": "chebyshev_jacobi (degree 3, ratio 60)", "ZZZ Solve ms": (time.perf_counter() - tc) * 1e3,
                                           "krylov_iterations": itc, "relative_residual": rnc / r0c if r0c else 0.0}
        if unstructured:
            # what the structured feed's luck was worth: entries kept in the stream, and the matrix-free action here
            extra["stream_entries_over_pattern"] = sinfo[7] / nnz
            _, ikind = ctx.internal_order()
            extra["internal_numbering"] = {0: "the caller's order kept", 1: "lattice order", 2: "coordinate-bin order"}[ikind]
            ctx.matfree_setup()
            extra["matfree_action_ms"] = ctx.action_time(20)
            extra["matfree_plan"] = ctx.matfree_info()
    ms = t_all / steps * 1e3
    phys = bytes_pl / (spmv_ms * 1e-3) / 1e9 if spmv_ms > 0 else 0.0
    return {"workload": f"--problem_type {c['problem_type']} --order {c['order']} --scaling_type {c['scaling_type']} "
                        f"--ndofs {c['ndofs']} -ksp_type cg -pc_type jacobi -ksp_rtol 1e-08 [{c['note']}]",
            "dofs": int(info[0]), "nnz": nnz, "steps": steps, "ms_per_step": ms, "value_dofs_per_s": int(info[0]) / (ms * 1e-3),
            "phases_ms": {k: float(np.mean(v)) * 1e3 for k, v in ph.items()},
            "krylov_iterations": it, "relative_residual": rn / r0 if r0 else 0.0, "solution_norm": unorm,
            "product_ms": spmv_ms, "product_launches_timed": spmv_n, "product_bytes_per_launch": bytes_pl,
            "product_GBs": phys, "roofline_frac": phys / HBM_PEAK_GBS,
            # the same launch time against SURVEY 8(d)'s reference-format byte count (12 B per pattern entry + 4 (n + 1) + 16 n: what
            # a plain CSR product would have to move): above 1 means the form in use moves fewer bytes than CSR could at the peak
            "reference_format_bytes": spmv_algorithmic_bytes(nrows, nnz),
            "reference_format_over_peak": (spmv_algorithmic_bytes(nrows, nnz) / (spmv_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if spmv_ms > 0 else 0.0,
            "operator": (f"block rows (block size 3: one lane per node, 16-bit codes into a table of {values['block_table_entries']} distinct "
                         f"3 x 3 blocks, " + ("the table's rows in LDS" if values["block_form"] == 1 else
                                               "rows of value offsets in memory and the values in LDS")
                         + f"; {values['block_chunks']} chunks of 16 block slots per node; csrc/zzz_sellp_blk.hip)") if values.get("block_rows") else
                        (f"block windows (long scalar rows: {values['block_form']} blocks of 4 096 rows in the Morton order of their nodes, "
                         f"x from a window in LDS ({values['block_table_entries']} window entries in all), values as 16-bit codes into block "
                         f"dictionaries in LDS; {values['block_chunks']} chunks of 8 entries per row; csrc/zzz_sellp_win.hip)")
                        if values.get("row_windows") else
                        ("sliced-ELL operator stream" + (" with x windows in LDS" if ctx_windows else "")
                         + (", values as 16-bit codes into per-slice dictionaries" if values["form"] == "slice dictionaries" else
                            f", values as codes into a {values['form']} of {values['distinct_values']} distinct values"
                            if values["form"] != "doubles" else "")) if sinfo[5] else "CSR tile kernel",
            **extra}


def run_matfree_operator(name, steps=3):
    """A BASELINE configuration with `--operator matfree` (an extension, not the reference's path): the same KSPCG + PCJACOBI
    to 1e-8, but the operator is never assembled -- every product is the matrix-free action of cgpoisson
    (csrc/zzz_matfree.hip) and Jacobi's diagonal comes from the element matrices.  The step: dof -> cell adjacency (the
    right-hand side's assembly walks it), the action's plan (what takes the matrix's place), b, solve.  Beside the
    assembled record of the same name so that the two can be compared: from P2 up recomputing beats streaming."""
    c = CONFIGS[name]
    nx, ny, nz, r = zzz.mesh_size(c["ndofs"], c["scaling_type"] == "strong", c["mesh_nproc"], 1, c["order"])
    nx, ny, nz = nx << r, ny << r, nz << r
    with zzz.Context(0) as ctx:
        info = ctx.cube_generate(c["problem_type"], c["order"], nx, ny, nz, 1, 0)
        ph = {"adjacency": [], "plan": [], "assemble_vector": [], "solve": []}
        it, t_all = 0, 0.0
        for k in range(steps + 1):
            ctx.sync()
            t0 = time.perf_counter()
            ctx.pattern_build()
            ctx.sync()
            t1 = time.perf_counter()
            ctx.matfree_setup()
            ctx.sync()
            t2 = time.perf_counter()
            ctx.assemble_vector(zzz.FORM_POISSON)
            ctx.sync()
            t3 = time.perf_counter()
            it, rn, r0 = ctx.cg_solve(variant=zzz.CG_PETSC, pc=zzz.PC_JACOBI, op=zzz.OP_MATFREE, rtol=1e-8, max_it=10000,
                                      profile=True)
            ctx.sync()
            t4 = time.perf_counter()
            if k == 0:
                continue
            for key, dt in zip(("adjacency", "plan", "assemble_vector", "solve"), (t1 - t0, t2 - t1, t3 - t2, t4 - t3)):
                ph[key].append(dt)
            t_all += t4 - t0
        act_ms, act_n = ctx.profile()
        plan = ctx.matfree_info()
        unorm = ctx.vec_norm(zzz.VEC_U)
    ms = t_all / steps * 1e3
    return {"workload": f"--problem_type {c['problem_type']} --order {c['order']} --scaling_type {c['scaling_type']} "
                        f"--ndofs {c['ndofs']} --operator matfree -ksp_type cg -pc_type jacobi -ksp_rtol 1e-08 [{c['note']}; "
                        "extension: no matrix, the reference multiplies with the assembled one]",
            "dofs": int(info[0]), "steps": steps, "ms_per_step": ms, "value_dofs_per_s": int(info[0]) / (ms * 1e-3),
            "phases_ms": {k: float(np.mean(v)) * 1e3 for k, v in ph.items()},
            "krylov_iterations": it, "relative_residual": rn / r0 if r0 else 0.0, "solution_norm": unorm,
            "action_ms": act_ms, "action_launches_timed": act_n, "us_per_iteration": float(np.mean(ph["solve"])) / max(it, 1) * 1e6,
            "bytes_addressed_per_action": plan["bytes_per_action"], "plan": plan}


def run_cgpoisson(order, ndofs, note, solves=3):
    """--problem_type cgpoisson (src/cgpoisson_problem.cpp): linalg::cg(u, b, action, 100, 1e-6) on the matrix-free
    operator, the reference's only caller of src/cg.h.  Gdof/s as the reference prints it (:236-241: iterations x global
    dofs / time of the cg call); the action kernel's time from HIP events around its launches inside the solve and from
    back-to-back launches; bytes: what the kernel addresses (plan streams + vectors) and the algorithmic minimum of
    SURVEY 8(d)'s vector-assembly count (connectivity and dofmap once, geometry once, un and y once)."""
    nx, ny, nz, r = zzz.mesh_size(ndofs, True, 1, 1, order)
    nx, ny, nz = nx << r, ny << r, nz << r
    nd = {1: 4, 2: 10, 3: 20}[order]
    with zzz.Context(0) as ctx:
        info = ctx.cube_generate("poisson", order, nx, ny, nz, 1, 0)
        n, ncells = int(info[0]), int(info[1])
        nverts = (nx + 1) * (ny + 1) * (nz + 1)
        ctx.sync()
        t0 = time.perf_counter()
        ctx.matfree_setup()
        ctx.sync()
        setup_cold = time.perf_counter() - t0
        t0 = time.perf_counter()
        ctx.matfree_setup()
        ctx.sync()
        setup_warm = time.perf_counter() - t0
        plan = ctx.matfree_info()
        ctx.pattern_build()  # the right-hand side's assembly walks the dof -> cell adjacency
        ctx.assemble_vector(zzz.FORM_POISSON)
        ts, it = [], 0
        for k in range(solves + 1):
            ctx.vec_upload(zzz.VEC_U, np.zeros(n))
            ctx.sync()
            t0 = time.perf_counter()
            it, rr, rr0 = ctx.cg_solve(variant=zzz.CG_CGH, pc=zzz.PC_NONE, op=zzz.OP_MATFREE, rtol=1e-6, max_it=100, profile=True)
            ctx.sync()
            if k:
                ts.append(time.perf_counter() - t0)
        act_ms, act_n = ctx.profile()
        act_b2b = ctx.action_time(20)
        unorm = ctx.vec_norm(zzz.VEC_U)
    t_cg = float(np.mean(ts))
    alg = 16 * ncells + 24 * nverts + (4 * nd * ncells if order > 1 else 0) + 16 * n
    return {"workload": f"--problem_type cgpoisson --order {order} --scaling_type strong --ndofs {ndofs} [{note}]",
            "dofs": n, "cells": ncells, "cg_iterations": it, "cg_s": t_cg, "Gdof_per_s": it * n / t_cg / 1e9,
            "relative_residual_squared": rr / rr0 if rr0 else 0.0, "solution_norm": unorm,
            "action_ms": act_ms, "action_launches_timed": act_n, "action_ms_back_to_back": act_b2b,
            "us_per_iteration": t_cg / max(it, 1) * 1e6,
            "algorithmic_bytes": alg, "bytes_addressed": plan["bytes_per_action"],
            "frac_of_peak_algorithmic": alg / (act_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if act_ms > 0 else None,
            "frac_of_peak_addressed": plan["bytes_per_action"] / (act_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if act_ms > 0 else None,
            "plan": plan, "setup_ms": setup_warm * 1e3, "setup_ms_cold": setup_cold * 1e3,
            "kernel": "k_mf_action + k_mf_finish (one pass over cell blocks in LDS; csrc/zzz_matfree.hip)"}


def run_rank_size(problem, order, nx, ny, nz, what):
    """One rank's share of a multi-GPU BASELINE configuration as a problem of its own on ONE GPU, with the multi-GPU code path
    attached (1-rank communicator, peer-memory mailboxes: every scalar all-reduce and the halo entry points run, over no
    links): the slab's mesh, nz / N layers of the cube (no ghost columns: the slab's neighbours are not there).  What a rank's
    kernels take per Krylov iteration -- an UPPER bound on what N GPUs can gain: xGMI latency comes on top."""
    form = zzz.FORM_ELASTICITY if problem == "elasticity" else zzz.FORM_POISSON
    zzz.comm_load()
    with zzz.Context(0) as ctx:
        ctx.comm_init(1, 0, zzz.comm_unique_id())
        p2p = os.environ.get("ZZZ_P2P", "1") != "0" and ctx.comm_p2p_attach(ctx.comm_p2p_export())
        info = ctx.cube_generate(problem, order, nx, ny, nz, 1, 0)
        ctx.pattern_build()
        ctx.assemble_matrix(form)
        ctx.assemble_vector(form)
        best = None
        for sr in (True, False):
            ctx.cg_solve(variant=zzz.CG_PETSC, pc=zzz.PC_JACOBI, rtol=1e-8, max_it=10000, single_reduction=sr)
            ts = []
            for _ in range(2):
                ctx.sync()
                t0 = time.perf_counter()
                it, rn, r0 = ctx.cg_solve(variant=zzz.CG_PETSC, pc=zzz.PC_JACOBI, rtol=1e-8, max_it=10000, single_reduction=sr)
                ctx.sync()
                ts.append(time.perf_counter() - t0)
            t = min(ts)
            if best is None or t < best[0]:
                best = (t, it, "single_reduction" if sr else "classical")
        nrows = ctx.csr_sizes()[0]
    t, it, form_name = best
    return {"what": what, "mesh": f"{nx}x{ny}x{nz} sub-cubes", "rows": nrows, "dofs": int(info[0]), "krylov_iterations": it,
            "solve_ms": t * 1e3, "us_per_iteration": t / max(it, 1) * 1e6, "cg_form": form_name,
            "scalar_allreduce": "peer-memory mailboxes" if p2p else "ncclAllReduce"}


def run_driver_one_shot(args, timeout=900):
    """ONE run of the drop-in driver binary (performance-test_amd/dolfinx-scaling-test: the reference's CLI and timer surface,
    src/main.cpp:57-74,152-211): every phase once, as the reference runs them, its "Summary of timings" parsed.  A child
    process (this one's GPU context stays as it is), outside every timed region."""
    import re
    import subprocess

    exe = os.path.join(zzz.PKG, "dolfinx-scaling-test")
    if not os.path.exists(exe):
        return {"error": "driver binary not built"}
    r = subprocess.run([exe] + args, capture_output=True, text=True, timeout=timeout)
    if r.returncode != 0:
        return {"error": f"exit status {r.returncode}: {r.stderr[-300:]}"}
    rec = {"command": "dolfinx-scaling-test " + " ".join(args), "timers_ms": {}}
    for line in r.stdout.splitlines():
        m = re.match(r"^(ZZZ [^|]+?)\s*\|\s*(\d+)\s+([0-9.eE+-]+)\s+([0-9.eE+-]+)\s*$", line)
        if m:
            rec["timers_ms"][m.group(1)] = float(m.group(4)) * 1e3
        m = re.match(r"^\*\*\* Number of Krylov iterations: (\d+)", line)
        if m:
            rec["krylov_iterations"] = int(m.group(1))
        m = re.match(r"^\*\*\* Solution norm:\s+([0-9.eE+-]+)", line)
        if m:
            rec["solution_norm"] = float(m.group(1))
        m = re.match(r"^\s*Total degrees of freedom:\s+(\d+)", line)
        if m:
            rec["dofs"] = int(m.group(1))
    t = rec["timers_ms"]
    if "dofs" in rec and t.get("ZZZ Solve"):
        rec["dofs_per_s"] = {k: rec["dofs"] / (v * 1e-3) for k, v in t.items()
                             if k in ("ZZZ Assemble", "ZZZ Assemble matrix", "ZZZ Assemble vector", "ZZZ Solve") and v > 0}
    rec["note"] = ("one process, every phase ONCE (first-time allocations, code objects loaded at zzz_ctx_create); the `ZZZ Assemble` "
                   "umbrella holds create_matrix + matrix + vector (src/poisson_problem.cpp:49,122-157)")
    return rec


def full_pattern_product(a, nx, ny, nz):
    """The product on the FULL pattern (ZZZ_SELLP_DROP=0: every structural entry of the reference's matrix streamed, exact
    zeros included), so that the kernel's share of `roofline.frac` can be told from the mesh's: 46 % of C2's entries are
    exact zeros only because of the Kuhn lattice."""
    old = os.environ.get("ZZZ_SELLP_DROP")
    os.environ["ZZZ_SELLP_DROP"] = "0"
    try:
        with zzz.Context(0) as ctx:
            ctx.cube_generate(a.problem_type, a.order, nx, ny, nz, 1, 0)
            ctx.pattern_build()
            form = zzz.FORM_ELASTICITY if a.problem_type == "elasticity" else zzz.FORM_POISSON
            ctx.assemble_matrix(form)
            ctx.assemble_vector(form)
            nrows, _, nnz = ctx.csr_sizes()
            ms = ctx.spmv_time(40)
            streamed, sinfo = physical_bytes_per_product(ctx, nrows, nnz)
    finally:
        if old is None:
            del os.environ["ZZZ_SELLP_DROP"]
        else:
            os.environ["ZZZ_SELLP_DROP"] = old
    gbs = streamed / (ms * 1e-3) / 1e9
    return {"what": "the same product kernel on the full pattern (ZZZ_SELLP_DROP=0): every structural entry streamed",
            "entries_streamed": int(sinfo[7]), "pattern_entries": nnz, "bytes_per_launch": streamed, "avg_launch_ms": ms,
            "launches_timed": 40, "achieved": gbs, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
            "reference_format_bytes": spmv_algorithmic_bytes(nrows, nnz),
            "reference_format_over_peak": spmv_algorithmic_bytes(nrows, nnz) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default=None, choices=sorted(CONFIGS),
                    help="a BASELINE.json configuration as a one-GPU workload (default: c2 = the headline); sets "
                         "--problem_type/--order/--scaling_type/--ndofs")
    ap.add_argument("--ndofs", type=int, default=10000000)
    ap.add_argument("--problem_type", default="poisson")
    ap.add_argument("--order", type=int, default=1)
    ap.add_argument("--scaling_type", default="strong")
    ap.add_argument("--mesh_nproc", type=int, default=0,
                    help="number of processes the mesh-size search is run for (src/mesh.cpp:87-90; weak scaling: "
                         "ndofs x processes); 0 = the number of GPUs of this run")
    ap.add_argument("--pc", default="jacobi", choices=["jacobi", "none", "chebyshev_jacobi"],
                    help="BASELINE's metric is quoted with jacobi; chebyshev_jacobi is the library's polynomial "
                         "preconditioner (fewer iterations and all-reduces, more products), an A/B line only")
    ap.add_argument("--pc_degree", type=int, default=0)
    ap.add_argument("--pc_ratio", type=float, default=0.0)
    ap.add_argument("--pc_esteig", type=int, default=0, help="Lanczos steps of the spectrum estimate (0: 10, < 0: Gershgorin alone)")
    ap.add_argument("--rtol", type=float, default=1e-8)
    ap.add_argument("--numbering", default="native", choices=["native", "rcm", "random", "reverse"],
                    help="N=1: how the CALLER numbers dofs, vertices and cells of the host feed before upload: native = "
                         "the structured generator's own order; rcm / random / reverse = what a DOLFINx-style feed may look "
                         "like (src/mesh.cpp:153-162,182-186).  The library renumbers internally (zzz_renumber.hip), so "
                         "`ZZZ Solve` should not depend on this; ZZZ_RENUMBER=0 shows what the caller's order would cost")
    ap.add_argument("--only", default=None, help="run only this record of other_configs (c1, c4_total, c5_rank, c5_whole, "
                    "unstructured_p1, cgpoisson_p1_c2, cgpoisson_p3_c5rank) and print it")
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--no_alt_pc", action="store_true", help="skip the Chebyshev-Jacobi solve beside the measurement")
    ap.add_argument("--no_other_configs", action="store_true",
                    help="default run only: skip the compact records beside the headline (c1, c4_total, c5_rank, c5 whole, "
                         "cgpoisson P1 / P3, the product on the full pattern; after and outside the headline's timed region)")
    ap.add_argument("--force_dist", action="store_true",
                    help="N=1 only: take the whole N > 1 code path (gloo process group, unique-id broadcast, 1-rank RCCL "
                         "communicator, mailbox handle all_gather, warm-up vote, tuning) -- what a 1-GPU box can run of it")
    ap.add_argument("--force_comm", action="store_true",
                    help="N=1 only: attach a 1-rank RCCL communicator to time the multi-GPU code path's fixed costs")
    ap.add_argument("--cg", default="auto", choices=["auto", "classical", "single_reduction"],
                    help="KSPCG form: classical (PETSc default, two reductions per iteration) or "
                         "-ksp_cg_single_reduction (one); auto = classical on one GPU, single_reduction on N > 1")
    a = ap.parse_args()
    if a.only:
        zzz.hip()
        rec = {"unstructured_p1": lambda: run_other_config(None, steps=2, unstructured=5000000),
               "cgpoisson_p1_c2": lambda: run_cgpoisson(1, 10000000, "the mesh of BASELINE configs[1]"),
               "cgpoisson_p3_c5rank": lambda: run_cgpoisson(3, 6250000, "the per-GPU share of BASELINE configs[4]"),
               "c5_rank_matfree_operator": lambda: run_matfree_operator("c5_rank"),
               "c5_whole_matfree_operator": lambda: run_matfree_operator("c5", steps=2),
               "c5_whole": lambda: run_other_config("c5", steps=2)}.get(a.only, lambda: run_other_config(a.only))()
        print(json.dumps({a.only: rec}))
        return
    cfg_note = None
    if a.config:
        c = CONFIGS[a.config]
        a.problem_type, a.order, a.scaling_type, a.ndofs = c["problem_type"], c["order"], c["scaling_type"], c["ndofs"]
        a.mesh_nproc = c["mesh_nproc"] if a.gpus == 1 else 0
        cfg_note = c["note"]

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run --nproc-per-node N")
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")

    # Load libzzz_hip (and with it /opt/rocm's HIP runtime) and, for N > 1, librccl BEFORE torch is
    # imported: torch bundles its own HIP/RCCL copies, and the two sets must never resolve into each
    # other.  torch is used for gloo plumbing only and never touches the GPU in this process.
    zzz.hip()
    uid_bytes = None
    multi = world > 1 or a.force_dist
    # The drop-in surface as a user of dolfinx-scaling-test sees it: one-shot runs of the driver binary, as child processes,
    # BEFORE this process allocates anything on the GPU (a child started after this process has freed tens of GB pays the
    # driver's clearing of that memory inside whatever phase allocates first: 0.7 s under `ZZZ Assemble matrix`, measured) and
    # outside every timed region.
    one_shot = {}
    if world == 1 and not multi and not a.force_comm and a.config is None and not a.no_other_configs \
            and (a.problem_type, a.order, a.ndofs) == ("poisson", 1, 10000000):
        for key, dargs in (("driver_one_shot", ["--problem_type", "poisson", "--order", "1", "--scaling_type", "strong", "--ndofs", "10000000",
                                                "-ksp_type", "cg", "-pc_type", "jacobi", "-ksp_rtol", "1e-8"]),
                           ("driver_one_shot_c4_total", ["--problem_type", "elasticity", "--order", "1", "--scaling_type", "strong", "--ndofs",
                                                         "4000000", "-ksp_type", "cg", "-pc_type", "jacobi", "-ksp_rtol", "1e-8"])):
            try:
                one_shot[key] = run_driver_one_shot(dargs)
            except Exception as e:  # noqa: BLE001
                one_shot[key] = {"error": repr(e)}
    if multi or a.force_comm:
        zzz.comm_load()  # EVERY rank binds /opt/rocm's librccl.so.1 before torch's bundled copy can be loaded
    if multi and rank == 0:
        uid_bytes = zzz.comm_unique_id()  # ncclGetUniqueId
    dist = None
    if multi:
        import torch
        import torch.distributed as dist

        import datetime

        if a.force_dist and world == 1:
            # started by hand, not by torch.distributed.run: the rendezvous of a single rank
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(29400 + os.getpid() % 500))
        with deadline(600, "gloo process group"):
            dist.init_process_group(backend="gloo", init_method="env://", world_size=world, rank=rank,
                                    timeout=datetime.timedelta(seconds=600))
        # every rank must have bound the SAME librccl (torch ships its own copy: load order decides)
        paths = [None] * world
        with deadline(120, "all_gather of the bound librccl paths"):
            dist.all_gather_object(paths, zzz.comm_library_path())
        if len(set(paths)) != 1:
            if rank == 0:
                sys.stderr.write("bench.py: the ranks bound different RCCL libraries: " + repr(paths) + "\n")
            sys.exit(4)

    def barrier():
        if dist is not None:
            with deadline(900, "barrier"):
                dist.barrier()

    bs = 3 if a.problem_type == "elasticity" else 1
    strong = a.scaling_type == "strong"
    nx, ny, nz, r = zzz.mesh_size(a.ndofs, strong, a.mesh_nproc or world, bs, a.order)
    nx, ny, nz = nx << r, ny << r, nz << r
    form = zzz.FORM_ELASTICITY if a.problem_type == "elasticity" else zzz.FORM_POISSON
    # host feed (C++ generator + upload) only when the CPU baseline needs the host arrays; otherwise the
    # feed is generated on the device (zzz_cube_generate) -- identical problem, no PCIe traffic
    if a.numbering != "native":
        if multi:
            raise SystemExit("--numbering applies to single-GPU runs (the multi-GPU feed is generated on the device)")
        a.no_cpu_baseline = True  # the CPU port is timed on the native order only (it has no renumbering of its own)
    need_host_arrays = not multi and (not a.no_cpu_baseline or a.numbering != "native")
    P = zzz.Part(a.problem_type, a.order, nx, ny, nz, world, rank) if need_host_arrays else None
    if a.numbering != "native":
        pattern = None
        if a.numbering == "rcm":
            # the dof graph for scipy's reverse Cuthill-McKee: the block pattern, built once on the GPU from the native feed
            with zzz.Context(local_rank) as c0:
                c0.upload_mesh(P.x, P.cells)
                c0.upload_dofmap(P.order, 1, P.cell_dofs, P.n_owned, P.n_ghost)  # block graph: one dof per block
                c0.pattern_build()
                rp0, cl0, _ = c0.csr_download(values=False)
            pattern = (rp0, cl0)
        P = P.renumbered(a.numbering, seed=1, pattern=pattern)
    # ZZZ_BENCH_DEVICE: all ranks on one device (only for the two-processes-on-one-GPU probe of the RCCL path)
    ctx = zzz.Context(int(os.environ.get("ZZZ_BENCH_DEVICE", local_rank)))
    if multi:
        import torch

        uid = torch.zeros(128, dtype=torch.uint8)
        if rank == 0:
            uid = torch.frombuffer(bytearray(uid_bytes), dtype=torch.uint8).clone()
        with deadline(300, "unique-id broadcast"):
            dist.broadcast(uid, src=0)
        with deadline(600, "ncclCommInitRank / ncclCommSplit"):
            ctx.comm_init(world, rank, bytes(uid.numpy().tobytes()))
    # CG scalar all-reduces through peer-memory mailboxes when every rank can map every peer (the library
    # tests the transport and makes the ranks agree); otherwise ncclAllReduce.  ZZZ_P2P=0 keeps RCCL.
    p2p = False
    if multi and os.environ.get("ZZZ_P2P", "1") != "0":
        mine = torch.frombuffer(bytearray(ctx.comm_p2p_export()), dtype=torch.uint8).clone()
        allh = [torch.zeros(zzz.P2P_HANDLE_BYTES, dtype=torch.uint8) for _ in range(world)]
        with deadline(300, "mailbox handle all_gather"):
            dist.all_gather(allh, mine)
        with deadline(300, "peer-memory mailbox attach (8 test rounds + agreement)"):
            p2p = ctx.comm_p2p_attach(b"".join(bytes(h.numpy().tobytes()) for h in allh))
    if not multi and a.force_comm:
        ctx.comm_init(1, 0, zzz.comm_unique_id())
        if os.environ.get("ZZZ_P2P", "1") != "0":
            p2p = ctx.comm_p2p_attach(ctx.comm_p2p_export())
    if P is not None:
        ctx.upload_part(P)
        if a.force_comm:
            ctx.upload_halo(P)
        ndofs_global, ncells_global = P.global_dofs_total, P.global_cells
    else:
        info = ctx.cube_generate(a.problem_type, a.order, nx, ny, nz, world, rank)  # includes the halo plan
        ndofs_global, ncells_global = int(info[0]), int(info[1])
    # The first pass over the assembly phases, COLD: what a one-shot run of the reference driver (every phase once,
    # src/main.cpp:152-170) would print under `ZZZ Assemble` -- first-time device allocations included.  Reported as
    # phases_ms_cold; the timed steps below are warm.
    cold = {}
    ctx.sync()
    t0 = time.perf_counter()
    ctx.pattern_build()  # also needed up front so that sizes are known; rebuilt inside every timed step
    ctx.sync()
    cold["create_matrix (sparsity pattern, adjacency, tiles)"] = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    ctx.assemble_matrix(form)
    ctx.sync()
    cold["ZZZ Assemble matrix"] = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    ctx.assemble_vector(form)
    ctx.sync()
    cold["ZZZ Assemble vector"] = (time.perf_counter() - t0) * 1e3
    cold["ZZZ Assemble (pattern + matrix + vector)"] = sum(cold.values())
    nrows, ncols, nnz = ctx.csr_sizes()
    pc = {"jacobi": zzz.PC_JACOBI, "none": zzz.PC_NONE, "chebyshev_jacobi": zzz.PC_CHEBYSHEV_JACOBI}[a.pc]
    pc_kw = dict(pc_degree=a.pc_degree, pc_ratio=a.pc_ratio, pc_esteig_its=a.pc_esteig) if pc == zzz.PC_CHEBYSHEV_JACOBI else {}
    single_reduction = a.cg == "single_reduction" or (a.cg == "auto" and (multi or a.force_comm))

    def step(profile=False):
        t = {}
        ctx.sync()
        tp = time.perf_counter()
        ctx.pattern_build()  # fem::petsc::create_matrix (src/poisson_problem.cpp:122-123), inside `ZZZ Assemble`
        ctx.sync()
        t["pattern"] = time.perf_counter() - tp
        t0 = time.perf_counter()
        ctx.assemble_matrix(form)
        ctx.sync()
        t["assemble_matrix"] = time.perf_counter() - t0
        t1 = time.perf_counter()
        ctx.assemble_vector(form)
        ctx.sync()
        t["assemble_vector"] = time.perf_counter() - t1
        t2 = time.perf_counter()
        it, rn, r0 = ctx.cg_solve(variant=zzz.CG_PETSC, pc=pc, rtol=a.rtol, max_it=10000, profile=profile,
                                  single_reduction=single_reduction, **pc_kw)
        ctx.sync()
        t["solve"] = time.perf_counter() - t2
        t["iters"] = it
        t["rel"] = rn / r0 if r0 else 0.0
        return t

    tuning = None
    guard = deadline(2400, "warm-up steps") if dist is not None else None
    if guard:
        guard.__enter__()
    for w in range(a.warmup):
        ok = 1
        try:
            step()
        except zzz.ZzzError:
            if not (p2p and dist is not None):
                raise
            ok = 0
        if p2p and dist is not None:
            # the warm-up doubles as a probe of the peer-memory all-reduce under the real load: a time-out
            # on any rank sends every rank back to ncclAllReduce, and the warm-up step is repeated
            flag = torch.tensor([ok], dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0:
                ctx.comm_p2p_disable()
                p2p = False
                step()
        if w == 0 and a.cg == "auto" and (dist is not None or a.force_comm):
            # N > 1: which CG form and which all-reduce transport are faster depends on the all-reduce latency
            # of this node, which only a run on it can tell: time one solve of each combination on the
            # assembled warm-up system (untimed region), MAX over ranks, and keep the fastest for the timed steps.
            forms = (True, False)
            # pm: 2 = scalars through the mailboxes AND the halo through the peer-memory window, 1 = scalars only (halo on
            # the communicator's send / recv), 0 = both on the communicator
            window = p2p and ctx.comm_p2p_halo(True)
            combos = [(sr, pm) for pm in (((2, 1, 0) if window else (1, 0)) if p2p else (0,)) for sr in forms]
            tuning = {}
            for sr, pm in combos:
                if p2p:
                    (ctx.comm_p2p_enable if pm else ctx.comm_p2p_disable)()
                    ctx.comm_p2p_halo(pm == 2)
                barrier()
                ctx.sync()
                t0 = time.perf_counter()
                try:
                    ctx.cg_solve(variant=zzz.CG_PETSC, pc=pc, rtol=a.rtol, max_it=10000, single_reduction=sr, **pc_kw)
                    ctx.sync()
                    dt = time.perf_counter() - t0
                except zzz.ZzzError:
                    dt = float("inf")
                if dist is not None:
                    tt = torch.tensor([dt], dtype=torch.float64)
                    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                    dt = float(tt[0])
                tuning[(sr, pm)] = dt
            best = min(tuning, key=lambda k: (tuning[k], k))
            single_reduction, use_pm = best
            if p2p:
                (ctx.comm_p2p_enable if use_pm else ctx.comm_p2p_disable)()
                ctx.comm_p2p_halo(use_pm == 2)
                p2p = bool(use_pm)
    if guard:
        guard.__exit__()

    def timed_region():
        """EXACTLY a.steps steps between barrier + device sync on both sides; None if the peer-memory all-reduce
        failed on some rank (every rank then fails within one round of it and all meet at the closing vote)"""
        barrier()
        ctx.sync()
        t_begin = time.perf_counter()
        phases, ok = [], 1
        try:
            for _ in range(a.steps):
                phases.append(step(profile=True))
            ctx.sync()
        except zzz.ZzzError:
            if not (p2p and dist is not None):
                raise
            ok = 0
        if p2p and dist is not None:
            vote = torch.tensor([ok], dtype=torch.int32)
            dist.all_reduce(vote, op=dist.ReduceOp.MIN)  # takes the place of the closing barrier
            if int(vote.item()) == 0:
                return None, None
        else:
            barrier()
        return phases, time.perf_counter() - t_begin

    guard = deadline(2400, "timed region") if dist is not None else None
    if guard:
        guard.__enter__()
    phases, elapsed = timed_region()
    if guard:
        guard.__exit__()
    if phases is None:  # never seen; kept so that a transport problem costs a repeat, not the measurement
        ctx.comm_p2p_disable()
        p2p = False
        phases, elapsed = timed_region()
    if dist is not None:
        import torch

        tt = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt[0])

    spmv_ms, spmv_n = ctx.profile()
    with deadline(300, "solution norm (all-reduce)"):
        unorm = ctx.vec_norm(zzz.VEC_U)
    # what every rank's multi-GPU path did: one entry per rank in the JSON line, so that a first N > 1 record can be
    # read without a second run
    rank_info = None
    if multi or a.force_comm:
        mine = ctx.comm_info()
        mine.update(rows=nrows, nnz=nnz, krylov_iterations=phases[-1]["iters"], spmv_avg_ms=spmv_ms,
                    solve_s=float(np.mean([p["solve"] for p in phases])), librccl=zzz.comm_library_path())
        rank_info = [mine]
        if dist is not None:
            rank_info = [None] * world
            with deadline(300, "all_gather of the per-rank diagnostics"):
                dist.all_gather_object(rank_info, mine)
    ms_per_step = elapsed / a.steps * 1e3
    iters = phases[-1]["iters"]

    out = None
    if rank == 0:
        alg_bytes = spmv_algorithmic_bytes(nrows, nnz) + (8 * nrows if single_reduction else 0)  # + read of r
        achieved = alg_bytes / (spmv_ms * 1e-3) / 1e9 if spmv_ms > 0 else 0.0
        streamed, sinfo = physical_bytes_per_product(ctx, nrows, nnz, single_reduction)
        kernel_name = ("spmv_tile_kernel (CG SpMV + <p,Ap> partials)" if not sinfo[5] else
                       "spmv_blk3_kernel (CG SpMV of a block-size-3 matrix in block-row form, one lane per node, + <p,Ap> partials)"
                       if ctx.spmv_values_info().get("block_rows") else
                       "spmv_win_kernel (CG SpMV of long scalar rows, x from LDS windows, + <p,Ap> partials)"
                       if ctx.spmv_values_info().get("row_windows") else
                       "spmv_one_kernel (CG SpMV on the operator stream's one-chunk slices, two rows per lane, + <p,Ap> partials)"
                       if ctx.spmv_values_info()["one_chunk_kernel"] else
                       "spmv_sellp_kernel (CG SpMV on the sliced-ELL operator stream + <p,Ap> partials)")
        traffic, traffic_src, traffic_more = pmc_traffic(nrows, nnz, streamed)
        phys = streamed / (spmv_ms * 1e-3) / 1e9 if spmv_ms > 0 else 0.0
        avg = lambda k: float(np.mean([p[k] for p in phases]))  # noqa: E731
        out = {
            "metric": "DoF/s for ZZZ Assemble + ZZZ Solve; CG-SpMV achieved HBM GB/s vs peak",
            "value": ndofs_global / (elapsed / a.steps),
            "unit": "DoF/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": (f"--problem_type {a.problem_type} --order {a.order} --scaling_type {a.scaling_type} "
                             f"--ndofs {a.ndofs} -ksp_type cg -pc_type {a.pc} -ksp_rtol {a.rtol:g}"
                             + (" -ksp_cg_single_reduction" if single_reduction else "")
                             + (f" [mesh of {a.mesh_nproc} processes]" if a.mesh_nproc and a.mesh_nproc != world else "")),
                "baseline_config": (a.config or "c2") + (": " + cfg_note if cfg_note else ""),
                "mesh": f"{nx}x{ny}x{nz} sub-cubes x 6 tetrahedra", "dofs": ndofs_global,
                "cells": ncells_global, "nnz_rank0": nnz, "rows_rank0": nrows,
                "partition": f"{world} z-slab(s)", "krylov_iterations": iters,
                "relative_residual": phases[-1]["rel"], "solution_norm": unorm,
            },
            "phases_ms": {"create_matrix (sparsity pattern, adjacency, tiles)": avg("pattern") * 1e3,
                          "ZZZ Assemble matrix": avg("assemble_matrix") * 1e3,
                          "ZZZ Assemble vector": avg("assemble_vector") * 1e3, "ZZZ Solve": avg("solve") * 1e3},
            "phases_ms_cold": cold,
            "dofs_per_s": {"ZZZ Assemble (pattern + matrix + vector)":
                           ndofs_global / (avg("pattern") + avg("assemble_matrix") + avg("assemble_vector")),
                           "ZZZ Assemble matrix": ndofs_global / avg("assemble_matrix"),
                           "ZZZ Assemble vector": ndofs_global / avg("assemble_vector"),
                           "ZZZ Solve": ndofs_global / avg("solve"),
                           "iterations x dofs / ZZZ Solve": iters * ndofs_global / avg("solve")},
            # PHYSICAL roofline of the dominant kernel: the bytes the product ADDRESSES per launch (operator stream or
            # packed CSR + row pointers + x + y (+ r)) / its HIP-event time / 8 TB/s.  The operator stream leaves out the
            # pattern's exact zeros and nearly all column indices, so SURVEY 8(d)'s reference-format byte count
            # (12 B per pattern entry ...) is NOT what moves; it is kept as `algorithmic_equivalent_*`, the rate a
            # plain CSR product would need to finish in the same time (it may exceed the peak: it is not a traffic claim).
            "roofline": {"bound": "hbm", "kernel": kernel_name,
                         "achieved": phys, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": phys / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "bytes_per_launch": streamed, "avg_launch_ms": spmv_ms, "launches_timed": spmv_n,
                         "traffic_over_bytes": (traffic / streamed) if traffic else None,
                         # the same PMC passes read both ways, per launch of THIS kernel name (tools/profile_summarise.py):
                         # low = FETCH_SIZE + WRITE_SIZE as counted, high = 2 x FETCH_SIZE + WRITE_SIZE (the guide's gfx950
                         # correction; `traffic` is this one); kernel_rocprof_us = its average duration in the kernel-trace pass
                         **traffic_more,
                         "algorithmic_equivalent_bytes_per_launch": alg_bytes,
                         "algorithmic_equivalent_GBs": achieved,
                         "algorithmic_equivalent_over_peak": achieved / HBM_PEAK_GBS},
        }
        # one whole PCG iteration, physically: the product's bytes + the two vector kernels' (k_update_xr, k_update_p: three
        # reads and two writes of a vector each) over solve time / iterations
        dinv_codes = ctx.cg_info()["dinv_codes"] > 0  # Jacobi's inverse diagonal as 16-bit codes, z recomputed: 68 B per row
        it_bytes = streamed + (96 if single_reduction else (68 if dinv_codes else 80)) * nrows
        it_s = avg("solve") / max(iters, 1)
        out["roofline"]["iteration"] = {"what": "one whole Jacobi-PCG iteration: bytes its kernels address / (ZZZ Solve / iterations)",
                                        "bytes": it_bytes, "us": it_s * 1e6, "achieved": it_bytes / it_s / 1e9, "unit": "GB/s",
                                        "frac": it_bytes / it_s / 1e9 / HBM_PEAK_GBS}
        # (the same as scalars: a record that keeps scalar keys only still carries them)
        out["roofline"]["iteration_frac"] = out["roofline"]["iteration"]["frac"]
        out["roofline"]["iteration_us"] = out["roofline"]["iteration"]["us"]
        if out["roofline"]["frac"] > 1.0:
            # a stream that stays in L2 / the Infinity Cache between launches can be read faster than HBM delivers: said,
            # not asserted (the line and the other ranks' barrier must not die of it)
            out["roofline"]["frac_above_one"] = True
            sys.stderr.write("bench.py: the product's byte rate exceeds the HBM peak (cache-resident operator stream)\n")
        c16 = ctx.spmv_info()
        if sinfo[5]:
            out["config"]["spmv_operator"] = (f"sliced-ELL operator stream, {'length-sorted' if sinfo[5] == 2 else 'natural'} row order: "
                                              f"{sinfo[7]} entries ({sinfo[7] / nnz:.3f} of the {nnz}-entry pattern; exact zeros "
                                              f"dropped, chunks of 8 padded), {sinfo[6]} B per product")
            vi = ctx.spmv_values_info()
            out["config"]["spmv_values"] = vi
            if vi["form"] == "slice dictionaries":
                out["config"]["spmv_operator"] += ("; values as 16-bit codes into per-slice dictionaries (bit-identical products; as doubles "
                                                   f"the stream would be {vi['bytes_per_product_as_doubles']} B per product)")
            elif vi["form"] != "doubles":
                out["config"]["spmv_operator"] += (f"; values as 16-bit codes into a {vi['form']} of the matrix's "
                                                   f"{vi['distinct_values']} distinct values (bit-identical products; as doubles the "
                                                   f"stream would be {vi['bytes_per_product_as_doubles']} B per product)")
            xw = ctx.spmv_x_windows()
            if xw[0]:
                out["config"]["spmv_operator"] += (f"; x windows: per group of 256 rows the columns it reaches are loaded into LDS "
                                                   f"({xw[0]} doubles per workgroup at most, {xw[1]} B per product mostly from L2; not "
                                                   f"counted in the bytes above) and gathered from there")
        else:
            out["config"]["spmv_operator"] = "CSR tile kernel"
        out["config"]["spmv_column_stream"] = ("(CSR tile kernel not in use)" if sinfo[5] else
                                               f"16-bit band codes ({c16[1]} offset bits), {c16[2]} of {c16[3]} tiles on int32 columns"
                                               if c16[0] else "int32")
        iperm, ikind = ctx.internal_order()
        out["config"]["numbering"] = {
            "caller": a.numbering,
            "internal": {0: "the caller's order kept", 1: "lattice order computed by the library (zzz_renumber.hip)",
                         2: "coordinate-bin order computed by the library"}[ikind],
            "dofs_moved": int(np.count_nonzero(iperm != np.arange(iperm.size)))}
        if rank_info:
            out["config"]["ranks"] = rank_info
        if tuning:
            tname = lambda sr, pm: (("single_reduction" if sr else "classical")  # noqa: E731
                                    + {0: "+ncclAllReduce", 1: "+peer_memory", 2: "+peer_memory+peer_halo"}[int(pm)])
            out["config"]["cg_form_tuning_us_per_iteration"] = {
                tname(sr, pm): (1e6 * v / max(iters, 1) if v != float("inf") else None) for (sr, pm), v in tuning.items()}
            out["config"]["cg_form_tuning_s"] = {tname(sr, pm): v for (sr, pm), v in tuning.items()}
        if multi or a.force_comm:
            out["config"]["scalar_allreduce"] = ("peer-memory mailboxes over xGMI (one kernel: reduce + exchange)" if p2p
                                                 else "ncclAllReduce")
            # the combination the timed steps ran, and how it was chosen: --cg given = taken as asked, no tuning solves
            out["config"]["cg_form"] = {"form": "single_reduction" if single_reduction else "classical",
                                        "scalar_allreduce": "peer_memory" if p2p else "ncclAllReduce",
                                        "halo": ("peer-memory window (device stores into the neighbour's memory)"
                                                 if (rank_info and rank_info[0].get("halo_own_communicator") == 2)
                                                 else "communicator send / recv"),
                                        "chosen_by": ("--cg " + a.cg) if a.cg != "auto" else
                                                     ("warm-up tuning (fastest of %d combinations)" % len(tuning) if tuning else "default")}
        out["config"]["feed"] = "host arrays uploaded (zzz_*_upload)" if P is not None else "generated on the device (zzz_cube_generate)"
        if not multi and not a.no_cpu_baseline:
            if nnz > 2**31 - 1:
                # the oracle's assembly takes the 32-bit row pointers of DOLFINx; a matrix beyond them (c5 whole: 2.4 G
                # nonzeros, ~60 GB of host working copies) is not timed on the CPU -- said here rather than failing
                out["cpu_baseline"] = {"value": None, "unit": "DoF/s", "cores": 0, "kind": "port",
                                       "sample": f"not run: {nnz} nonzeros exceed the oracle's 32-bit row pointers; see "
                                                 "--config c5_rank for the per-GPU size of the same problem"}
            else:
                out["cpu_baseline"] = cpu_baseline(P, ctx, iters)
                out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
                out["gpu_over_cpu_zzz_solve"] = out["cpu_baseline"]["solve_s"] / avg("solve")
    # beside the measurement (after it, untimed): the same assembled system solved once with the library's polynomial
    # preconditioner -- fewer iterations and all-reduces for more products, i.e. what the N > 1 runs are bound by
    alt_pc = None
    if pc == zzz.PC_JACOBI and not a.no_alt_pc:
        try:
            def line_without_it():
                if rank == 0:
                    out["alt_preconditioner"] = {"error": "did not finish within 600 s"}
                    print(json.dumps(out))

            with deadline(600, "Chebyshev-Jacobi solve beside the measurement", last_words=line_without_it):
                ctx.cg_solve(variant=zzz.CG_PETSC, pc=zzz.PC_CHEBYSHEV_JACOBI, rtol=a.rtol, max_it=10000,
                             single_reduction=single_reduction)
                barrier()
                ctx.sync()
                t0 = time.perf_counter()
                ita, rna, r0a = ctx.cg_solve(variant=zzz.CG_PETSC, pc=zzz.PC_CHEBYSHEV_JACOBI, rtol=a.rtol, max_it=10000,
                                             single_reduction=single_reduction)
                ctx.sync()
                dt = time.perf_counter() - t0
                if dist is not None:
                    tt = torch.tensor([dt], dtype=torch.float64)
                    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                    dt = float(tt[0])
                alt_pc = {"pc_type": "chebyshev_jacobi (degree 3, ratio 60: the library's defaults)", "ZZZ Solve ms": dt * 1e3,
                          "krylov_iterations": ita, "relative_residual": rna / r0a if r0a else 0.0,
                          "products_per_iteration": 3, "allreduces_per_iteration": 1 if single_reduction else 2,
                          "cg_form": "single_reduction" if single_reduction else "classical",
                          "jacobi ZZZ Solve ms": float(np.mean([p["solve"] for p in phases])) * 1e3,
                          "note": "one solve of the timed steps' system, outside the timed region; not part of value"}
        except zzz.ZzzError as e:
            alt_pc = {"error": repr(e)}
        if rank == 0:
            out["alt_preconditioner"] = alt_pc
    ctx.close()
    if rank == 0 and not multi and not a.force_comm and a.config is None and not a.no_other_configs \
            and (a.problem_type, a.order, a.ndofs) == ("poisson", 1, 10000000):
        # the other BASELINE configs, compactly, so that the driver's record (not only builder-run profiles) has them
        try:
            out["roofline"]["full_pattern"] = full_pattern_product(a, nx, ny, nz)
            out["roofline"]["full_pattern_frac"] = out["roofline"]["full_pattern"]["frac"]
            out["roofline"]["full_pattern_ms"] = out["roofline"]["full_pattern"]["avg_launch_ms"]
        except Exception as e:  # noqa: BLE001 -- the headline line must still be printed
            out["roofline"]["full_pattern"] = {"error": repr(e)}
        out["other_configs"] = {}
        out["other_configs"].update(one_shot)
        for name in ("c1", "c4_total", "c5_rank", "c5"):
            try:
                out["other_configs"]["c5_whole" if name == "c5" else name] = run_other_config(name, steps=2 if name == "c5" else 3)
            except Exception as e:  # noqa: BLE001 -- the headline line must still be printed
                out["other_configs"]["c5_whole" if name == "c5" else name] = {"error": repr(e)}
        try:
            out["other_configs"]["unstructured_p1"] = run_other_config(None, steps=2, unstructured=5000000)
        except Exception as e:  # noqa: BLE001
            out["other_configs"]["unstructured_p1"] = {"error": repr(e)}
        for key, order, nd_, note in (("cgpoisson_p1_c2", 1, 10000000, "the mesh of BASELINE configs[1]"),
                                      ("cgpoisson_p3_c5rank", 3, 6250000, "the per-GPU share of BASELINE configs[4]")):
            try:
                out["other_configs"][key] = run_cgpoisson(order, nd_, note)
            except Exception as e:  # noqa: BLE001
                out["other_configs"][key] = {"error": repr(e)}
        for key, name, st in (("c5_rank_matfree_operator", "c5_rank", 3), ("c5_whole_matfree_operator", "c5", 2)):
            try:
                out["other_configs"][key] = run_matfree_operator(name, steps=st)
            except Exception as e:  # noqa: BLE001
                out["other_configs"][key] = {"error": repr(e)}
        # what ONE rank of the multi-GPU configurations does per Krylov iteration (no 8-GPU node is needed to know the bound)
        rs = {}
        lay = lambda n_, parts: -(-n_ // parts)  # noqa: E731 -- layers of the thickest z-slab
        mx4 = zzz.mesh_size(500000, False, 8, 3, 1)
        mx5 = zzz.mesh_size(50000000, True, 1, 1, 3)
        for key, prob, order, mesh, single_ms, what in (
                ("c3_n2", "poisson", 1, (nx, ny, lay(nz, 2)), avg("solve") * 1e3, "BASELINE configs[2] on 2 GPUs: one z-slab"),
                ("c3_n4", "poisson", 1, (nx, ny, lay(nz, 4)), avg("solve") * 1e3, "BASELINE configs[2] on 4 GPUs: one z-slab"),
                ("c3_n8", "poisson", 1, (nx, ny, lay(nz, 8)), avg("solve") * 1e3, "BASELINE configs[2] on 8 GPUs: one z-slab"),
                ("c4_n8", "elasticity", 1, (mx4[0] << mx4[3], mx4[1] << mx4[3], lay(mx4[2] << mx4[3], 8)),
                 (out["other_configs"].get("c4_total", {}).get("phases_ms") or {}).get("solve"), "BASELINE configs[3]: one of 8 z-slabs"),
                ("c5_n8", "poisson", 3, (mx5[0] << mx5[3], mx5[1] << mx5[3], lay(mx5[2] << mx5[3], 8)),
                 (out["other_configs"].get("c5_whole", {}).get("phases_ms") or {}).get("solve"), "BASELINE configs[4]: one of 8 z-slabs")):
            try:
                rec = run_rank_size(prob, order, mesh[0], mesh[1], mesh[2], what)
                nr = int(key.split("_n")[1])
                if single_ms:
                    rec["single_gpu_solve_ms"] = single_ms
                    # strong scaling: one GPU's solve over one rank's; weak (c4): the total problem's solve over one rank's x N
                    rec["projected_speedup_upper_bound"] = single_ms / rec["solve_ms"]
                    rec["projected_speedup_note"] = (f"single-GPU ZZZ Solve / one rank's ZZZ Solve at N = {nr}: kernels only, no xGMI "
                                                     "latency, no load imbalance -- a bound, not a measurement of N GPUs")
                rs[key] = rec
            except Exception as e:  # noqa: BLE001
                rs[key] = {"error": repr(e)}
        rs["scalar_allreduce_default"] = ("peer-memory mailboxes over xGMI (one kernel: reduce the workgroups' partials, store {values, tag} "
                                          "into every peer's mailbox, poll the own one); ncclAllReduce is the fallback, chosen per node by "
                                          "bench.py's warm-up tuning: an RCCL all-reduce of 1-3 doubles is two extra launches and ~20 us on "
                                          "one node, the mailbox kernel ~6 us (DESIGN.md section 5)")
        out["other_configs"]["rank_sizes"] = rs
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()

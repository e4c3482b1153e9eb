/*
 * zzz_abi.h -- C-ABI of libzzz_hip.so: the MI355X (gfx950) implementation of the hot path of
 * FEniCS/performance-test (`ZZZ Assemble matrix`, `ZZZ Assemble vector`, `ZZZ Solve`).
 *
 * Plain C, POD arguments, opaque context handle.  No exceptions cross this boundary: every
 * function returns 0 on success or a ZZZ_ERR_* code; zzz_last_error() gives the message.
 * The host owns host arrays; every upload copies; the library owns all device memory behind the
 * context; downloads write into caller-provided buffers.  One driving host thread per context;
 * one context per GPU (one process or thread per GPU).  Work is enqueued asynchronously on the
 * context's HIP stream except the *_download functions, zzz_sync and zzz_cg_solve.
 *
 * Each entry point names the reference interface it replaces (paths relative to the reference
 * repository, FEniCS/performance-test @ 2025-10-24).
 *
 * Data model at the boundary (one mesh partition = one context, DOLFINx's per-rank view):
 *   - block dofs [0, n_owned) are owned, [n_owned, n_owned + n_ghost) are ghosts
 *     (the la::Vector layout, src/cgpoisson_problem.cpp:212-215); scalar dof = bs*block + comp.
 *   - the partition holds every cell that touches an owned dof (one ghost-cell layer), so each
 *     owned matrix row and vector entry is complete locally and MatAssemblyBegin/End's row
 *     exchange (src/poisson_problem.cpp:132-133) and b.scatter_rev (:154) have nothing to ship.
 *   - cell_dofs follows the Basix local ordering of the P_k gll_warped tetrahedron
 *     (src/poisson_problem.cpp:35-38): vertices 0-3; edges e0=(2,3) e1=(1,3) e2=(1,2) e3=(0,3)
 *     e4=(0,2) e5=(0,1), k-1 dofs each, counted from the edge's first local vertex; faces
 *     f0=(1,2,3) f1=(0,2,3) f2=(0,1,3) f3=(0,1,2).  Edge orientation is resolved by the dofmap
 *     (the caller permutes an edge's sub-dofs when the global direction opposes the local one),
 *     so the kernels apply no dof transformation -- DOLFINx's convention for Lagrange.
 *   - the matrix is scalar CSR (PETSc AIJ): fp64 values, int32 columns (local scalar indices,
 *     ascending within a row), int32 row pointers; rows = owned scalar dofs.
 */
#ifndef ZZZ_ABI_H
#define ZZZ_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct zzz_ctx zzz_ctx;

/* error classes */
enum
{
  ZZZ_OK = 0,
  ZZZ_ERR_ARG = 1,     /* bad argument / call order (std::runtime_error in the reference) */
  ZZZ_ERR_HIP = 2,     /* HIP runtime error */
  ZZZ_ERR_RCCL = 3,    /* RCCL error / librccl not loadable */
  ZZZ_ERR_NO_GPU = 4,  /* no usable device: the library has no CPU fallback */
  ZZZ_ERR_LIMIT = 5,   /* size exceeds an int32 index range */
  ZZZ_ERR_DIVERGED = 6 /* the solve did not converge AND zzz_solver_opts.error_if_not_converged was set */
};

/* forms: form_Poisson_{a,L,M}{1,2,3}, form_Elasticity_{a,L}{1,2,3}
 * (src/poisson_problem.cpp:110-119, src/elasticity_problem.cpp:184-191) */
enum
{
  ZZZ_FORM_POISSON = 0,
  ZZZ_FORM_ELASTICITY = 1
};

enum
{
  ZZZ_COEFF_F = 0, /* "w0" of L (src/poisson_problem.cpp:117, src/elasticity_problem.cpp:189) */
  ZZZ_COEFF_G = 1  /* "w1" of the Poisson L (src/poisson_problem.cpp:117) */
};

enum
{
  ZZZ_VEC_B = 0, /* right-hand side b */
  ZZZ_VEC_U = 1  /* solution u (u->x() of the reference) */
};

/* Krylov options: what "-ksp_type cg -pc_type jacobi -ksp_rtol 1e-8" selects through
 * solver.set_from_options() (src/poisson_problem.cpp:169), or the arguments of linalg::cg
 * (src/cg.h:39-40). */
enum
{
  ZZZ_PC_NONE = 0,
  ZZZ_PC_JACOBI = 1,
  /* polynomial preconditioner: z = p_k(D^-1 A) D^-1 r, k steps of the Chebyshev iteration for D^-1 A started from zero
   * (PETSc: -pc_type ksp -ksp_ksp_type chebyshev -ksp_ksp_max_it k -ksp_pc_type jacobi [EXT]).  The "stronger
   * preconditioner" of README.md:61-62,108-110 that needs only the product and no reduction inside its application:
   * fewer CG iterations, i.e. fewer all-reduces per solve, for more products -- for multi-GPU runs.  Spectrum
   * bounds [hi / pc_ratio, hi] with hi = min(Gershgorin's bound of D^-1 A, 1.1 x a Lanczos estimate: pc_esteig_its).
   * ZZZ_CG_PETSC + ZZZ_OP_CSR only (classical or single-reduction form). */
  ZZZ_PC_CHEBYSHEV_JACOBI = 2
};
enum
{
  ZZZ_NORM_PRECONDITIONED = 0, /* PETSc KSPCG default: ||M^-1 r|| */
  ZZZ_NORM_UNPRECONDITIONED = 1,
  ZZZ_NORM_NATURAL = 2
};
enum
{
  ZZZ_CG_PETSC = 0, /* KSPCG: zero initial guess, test dp <= max(rtol*dp0, atol) */
  ZZZ_CG_CGH = 1    /* src/cg.h:38-86: x is the initial guess, test <r,r>/<r0,r0> < rtol^2 */
};
enum
{
  ZZZ_OP_CSR = 0,    /* assembled operator (poisson, elasticity) */
  ZZZ_OP_MATFREE = 1 /* cgpoisson's action(a, un) (src/cgpoisson_problem.cpp:193-230) */
};

typedef struct
{
  int32_t variant;  /* ZZZ_CG_PETSC | ZZZ_CG_CGH */
  int32_t pc;       /* ZZZ_PC_* (ZZZ_CG_CGH requires ZZZ_PC_NONE; with ZZZ_OP_MATFREE: none or jacobi) */
  int32_t norm;     /* ZZZ_NORM_* (ZZZ_CG_PETSC only) */
  int32_t op;       /* ZZZ_OP_* */
  int32_t max_it;   /* -ksp_max_it (PETSc default 10000) / kmax */
  int32_t profile;  /* != 0: record HIP events around the SpMV launches (see zzz_profile_get) */
  int32_t single_reduction; /* != 0: PETSc's -ksp_cg_single_reduction (KSPCGUseSingleReduction): the same CG
                             * with the recurrences s = A z, w = s + b w, <p,w> by recurrence, so that one
                             * iteration needs ONE fused reduction of (<r,z>, <z,s>, norm) instead of two.
                             * ZZZ_CG_PETSC + ZZZ_OP_CSR only. */
  int32_t error_if_not_converged; /* PETSc's -ksp_error_if_not_converged: != 0 makes a solve that ends with a negative
                                   * KSPConvergedReason (DTOL, NaN/Inf, max_it) fail with ZZZ_ERR_DIVERGED.  0 (the
                                   * default, as in PETSc): the solve returns ZZZ_OK with its iteration count, like
                                   * solver.solve() in solver_function (src/poisson_problem.cpp:172-178), and the
                                   * reason is read from zzz_cg_info */
  double rtol;      /* -ksp_rtol / rtol */
  double atol;      /* -ksp_atol (PETSc default 1e-50); unused by ZZZ_CG_CGH */
  double dtol;      /* -ksp_divtol: KSPConvergedDefault stops with KSP_DIVERGED_DTOL once the norm reaches
                     * dtol x the initial norm (zzz_cg_info then reports reason -4, as KSPGetConvergedReason would);
                     * <= 0 selects PETSc's default 1e4 (KSPCreate sets
                     * divtol = 1.e4); unused by ZZZ_CG_CGH (src/cg.h has no such test) */
  /* (added in round 3 behind the fields above, whose layout is unchanged) */
  int32_t pc_degree; /* ZZZ_PC_CHEBYSHEV_JACOBI: Chebyshev steps per application (0 selects 3) */
  int32_t pc_esteig_its; /* ZZZ_PC_CHEBYSHEV_JACOBI: Jacobi-PCG iterations on a noise vector whose Lanczos coefficients give
                          * the estimate of the largest eigenvalue of D^-1 A (PETSc's -ksp_chebyshev_esteig, safety factor
                          * 1.1); the bound used is min(Gershgorin's, 1.1 x estimate).  0 selects 10, < 0 Gershgorin alone */
  double pc_ratio;   /* ZZZ_PC_CHEBYSHEV_JACOBI: upper / lower bound of the targeted spectrum (<= 1 selects 60) */
} zzz_solver_opts;

/* ---- library / device ------------------------------------------------------------------ */

/* Number of visible GPUs; 0 when there is none (then zzz_ctx_create fails with ZZZ_ERR_NO_GPU). */
int zzz_device_count(void);
/* hipMemGetInfo of a device in bytes (for the driver's --memory_profiling log, src/mem.cpp:18-38). */
int zzz_device_memory(int device, size_t* free_bytes, size_t* total_bytes);

/* Creates a context on `device`.  Replaces MPI_Init/PetscInitialize as far as this path needs
 * them (src/main.cpp:245-258). */
int zzz_ctx_create(int device, zzz_ctx** out);
void zzz_ctx_destroy(zzz_ctx* ctx);

/* Message of the last error on ctx (or of the last context-less error when ctx == NULL).
 * The pointer stays valid until the next call on that context. */
const char* zzz_last_error(const zzz_ctx* ctx);

/* Blocks until all enqueued work is done (the driver calls it before stopping a ZZZ timer). */
int zzz_sync(zzz_ctx* ctx);

/* ---- problem data (feed) --------------------------------------------------------------- */

/* mesh::Geometry::x() and the geometry dofmap of the mesh handed to problem()
 * (src/poisson_problem.h:19-23; created at src/mesh.cpp:184-186).
 * x: nverts*3 row-major; cell_verts: ncells*4 local vertex indices. */
int zzz_mesh_upload(zzz_ctx* ctx, int64_t nverts, const double* x, int64_t ncells, const int32_t* cell_verts);

/* fem::create_functionspace's DofMap + IndexMap (src/poisson_problem.cpp:43-44,
 * src/elasticity_problem.cpp:108-111).  order 1..3 (form_*.at(order-1),
 * src/poisson_problem.cpp:117); bs 1 or 3; cell_dofs: ncells*nd block indices, nd = 4/10/20. */
int zzz_dofmap_upload(zzz_ctx* ctx, int order, int bs, const int32_t* cell_dofs, int64_t n_owned, int64_t n_ghost);

/* fem::DirichletBC(u0 == 0, bdofs) (src/poisson_problem.cpp:53-77, src/elasticity_problem.cpp:119-145).
 * bc_dofs: local SCALAR dof indices, owned and ghost, all with value 0. */
int zzz_bc_upload(zzz_ctx* ctx, int64_t nbc, const int32_t* bc_dofs);

/* Exterior facets integrated by the `ds` term of L (src/Poisson.py:32); what
 * create_entities(2)/create_connectivity(2,3) prepare (src/main.cpp:147-148).
 * pairs: nfacets*(cell, local facet 0..3); local facet f is opposite local vertex f. */
int zzz_facets_upload(zzz_ctx* ctx, int64_t nfacets, const int32_t* cell_facet_pairs);

/* Nodal values of a coefficient Function (f->interpolate / g->interpolate,
 * src/poisson_problem.cpp:83-106, src/elasticity_problem.cpp:153-176): (n_owned+n_ghost)*bs. */
int zzz_coeff_upload(zzz_ctx* ctx, int which, const double* values);

/* The whole feed above for the reference's structured cube problems, generated on the device in
 * closed form instead of uploaded: create_cube_mesh's box of nx*ny*nz sub-cubes x 6 tetrahedra
 * (src/mesh.cpp:184-186) cut into `nparts` z-slabs, the P_order function space, the Dirichlet set and
 * the interpolated coefficients of problem() (src/poisson_problem.cpp:33-106,
 * src/elasticity_problem.cpp:101-176) and the halo plan.  Equivalent to uploading the arrays of
 * zzzh_part_create() (include/zzz_host.h); problem = ZZZ_FORM_POISSON | ZZZ_FORM_ELASTICITY.
 * info (optional, 6 entries): global scalar dofs, global cells, owned block dofs, ghost block dofs,
 * global block index of local dof 0, local cells. */
int zzz_cube_generate(zzz_ctx* ctx, int problem, int order, int64_t nx, int64_t ny, int64_t nz, int nparts, int part,
                      int64_t* info);

/* ---- matrix ---------------------------------------------------------------------------- */

/* fem::petsc::create_matrix(*a) (src/poisson_problem.cpp:122-123): sparsity pattern of the
 * owned rows + the dof->cell adjacency the assembly kernels walk.  Outside `ZZZ Assemble
 * matrix`, inside `ZZZ Assemble`, as in the reference. */
int zzz_csr_pattern_build(zzz_ctx* ctx);

/* Sizes of the local matrix: owned scalar rows, local scalar columns (owned+ghost), nonzeros. */
int zzz_csr_sizes(const zzz_ctx* ctx, int64_t* nrows, int64_t* ncols, int64_t* nnz);

/* Copies the CSR arrays to the host (parity checks; MatView-like).  NULL pointers are skipped.
 * rowptr: nrows+1, cols/vals: nnz. */
int zzz_csr_download(zzz_ctx* ctx, int32_t* rowptr, int32_t* cols, double* vals);
/* The row pointers in 64 bits (PetscInt of the reference's CI build): the only form for matrices of 2^31 nonzeros
 * or more (Poisson P3 at 50 M dofs: 2.4 G), where zzz_csr_download(rowptr != NULL) fails with ZZZ_ERR_LIMIT. */
int zzz_csr_rowptr64_download(zzz_ctx* ctx, int64_t* rowptr);

/* Replaces the matrix values (testing the solver on a given operator). vals: nnz. */
int zzz_csr_upload_values(zzz_ctx* ctx, const double* vals);

/* The `ZZZ Assemble matrix` block (src/poisson_problem.cpp:125-139,
 * src/elasticity_problem.cpp:199-213): tabulate_tensor of form a per cell, constrained rows and
 * columns zeroed, ADD into A, then set_diagonal = 1.0 on constrained rows. */
int zzz_assemble_matrix(zzz_ctx* ctx, int form);

/* The `ZZZ Assemble vector` block (src/poisson_problem.cpp:146-157,
 * src/elasticity_problem.cpp:220-231): cell (+ exterior-facet) integrals of L into b,
 * apply_lifting (identically zero: u0 == 0), bc->set. */
int zzz_assemble_vector(zzz_ctx* ctx, int form);

/* ---- vectors --------------------------------------------------------------------------- */

/* Host <-> device copies of b or u; n_owned*bs entries (owned part). */
int zzz_vec_download(zzz_ctx* ctx, int which, double* out);
int zzz_vec_upload(zzz_ctx* ctx, int which, const double* in);

/* la::norm(*u->x()) (src/main.cpp:229): l2 norm over owned entries, summed over all ranks. */
int zzz_vec_norm(zzz_ctx* ctx, int which, double* out);

/* y = A x on host vectors of the owned size (ghost values of x are exchanged first when a
 * communicator is attached).  MatMult; for parity checks of the SpMV kernel alone.
 * Non-finite x: the product runs on an operator stream that leaves out the entries whose assembled value is exactly
 * zero (what MAT_IGNORE_ZERO_ENTRIES does to an AIJ matrix [EXT]) and pads aligned slices with +0.0 entries pointing
 * at a neighbouring column of the same slice.  For finite x that changes no bit of y.  For x with Inf/NaN it does:
 * PETSc's MatMult on the full pattern yields NaN in every row that has a STRUCTURAL entry in such a column (0 * NaN);
 * this product yields NaN in every row with a NONZERO entry there, may miss rows whose only coupling to that column
 * is an exact zero, and may add rows of the same 64-row slice through a padding entry.  The CG iterations see a
 * non-finite vector only after they have broken down (reported as KSP_DIVERGED_NANORINF either way).  ZZZ_SELLP_DROP=0
 * with ZZZ_SELLP_FORMS=5 (no aligned slices) keeps every structural entry and no padding on a real column: NaN then propagates exactly
 * as in the serial CSR loop (tested). */
int zzz_spmv(zzz_ctx* ctx, const double* x, double* y);

/* Measurement aid: HIP-event time of `reps` back-to-back launches of the CG SpMV kernel
 * (w = A p with the <p,w> partials) on the context's stream; variant < 0 keeps the configured
 * kernel variant; otherwise a bit set for A/B runs: bit 0 non-temporal loads, bit 1 pipelined CSR
 * tiles, bit 3 the operator stream of zzz_sellp.hip when one was built (default 9 = stream + non-temporal),
 * bit 4 int32 instead of packed 16-bit columns in the tile kernel. */
int zzz_spmv_time(zzz_ctx* ctx, int reps, int variant, double* avg_ms);

/* y = action(x): the matrix-free operator lambda of cgpoisson (src/cgpoisson_problem.cpp:193-230):
 * assemble_vector of form M = action(a, un) with un = x, rows of constrained dofs zeroed, ghosts
 * of x updated first.  Needs mesh, dofmap and Dirichlet dofs only: no pattern, no matrix (the reference's cgpoisson
 * creates neither, src/cgpoisson_problem.cpp:47-247).  Builds the operator's plan on first use (zzz_matfree_setup). */
int zzz_action(zzz_ctx* ctx, const double* x, double* y);

/* What cgpoisson sets up once before its solve: fem::create_form of M (src/cgpoisson_problem.cpp:133-145), the
 * coefficient storage of `un` (:182) and the Scatterer with its buffers (:187-190).  Here: the plan of the one-pass
 * matrix-free kernel (csrc/zzz_matfree.hip) -- cells cut into blocks by the Morton order of their centroids, block-local
 * dof lists and 16-bit indices, the order in which a block adds its element vectors, P2/P3 geometry factors.  Optional:
 * zzz_action and zzz_cg_solve(op = ZZZ_OP_MATFREE) build it on first use.  Invalidated by mesh, dofmap and bc uploads. */
int zzz_matfree_setup(zzz_ctx* ctx);
/* info[0] plan valid, [1] cell blocks, [2] cells per block, [3] threads per workgroup, [4] most dofs a block touches,
 * [5] dofs shared between blocks, [6] their partial sums per action, [7] bytes one action addresses (plan + vectors). */
int zzz_matfree_info(zzz_ctx* ctx, int64_t info[8]);
/* diag[n_owned] = the diagonal of the operator zzz_action applies, as the assembled matrix holds it (MatGetDiagonal after
 * fem::set_diagonal: 1.0 on constrained rows, src/poisson_problem.cpp:146-157) -- computed from the element matrices in
 * the matrix-free kernel's pass, nothing assembled.  It is what PCJACOBI uses when zzz_cg_solve runs KSPCG on
 * op = ZZZ_OP_MATFREE (an extension: the reference's KSP path always multiplies with the assembled AIJ matrix,
 * src/poisson_problem.cpp:159-181; same mathematics, the operator recomputed per product instead of streamed). */
int zzz_matfree_diagonal(zzz_ctx* ctx, double* diag);
/* Measurement aid, as zzz_spmv_time: HIP-event time of `reps` back-to-back actions w = action(p) with the <p,w>
 * partials (what one iteration of linalg::cg launches at src/cg.h:62,65). */
int zzz_action_time(zzz_ctx* ctx, int reps, double* avg_ms);

/* `ZZZ Create near-nullspace`: build_near_nullspace of src/elasticity_problem.cpp:36-94 (called at :233-244) -- the six
 * rigid-body modes of the vector-valued space at the dof coordinates (tabulate_dof_coordinates: every dof's reference
 * node pushed through its cell's affine map), orthonormalised in basis order as la::orthonormalize does, checked as
 * la::is_orthonormal does (error "Space not orthonormal" otherwise; *max_deviation = largest |<x_i,x_j> - delta_ij|).
 * Inner products over the owned entries, summed over the ranks (collective with a communicator attached).  The
 * reference passes the basis to MatSetNearNullSpace for GAMG; Jacobi-CG does not consume it. */
int zzz_near_nullspace_build(zzz_ctx* ctx, double* max_deviation);
/* mode k (0..5) of that basis, owned part, 3 * n_owned entries in the caller's numbering */
int zzz_near_nullspace_download(zzz_ctx* ctx, int k, double* out);

/* ---- the reference's native partition ----------------------------------------------------- */

/* Global indices of the local block dofs (index_map.local_to_global: owned, then ghosts) and of the local mesh
 * vertices: what two ranks use to recognise a shared dof or vertex.  Needed by zzz_ghost_layer_build only. */
int zzz_global_ids_upload(zzz_ctx* ctx, const int64_t* dof_global, const int64_t* vert_global);
/* ... of the local block dofs as they are now (after zzz_ghost_layer_build: with the new ghosts) */
int zzz_global_ids_download(zzz_ctx* ctx, int64_t* dof_global);

/* Collective.  For a feed partitioned as the reference partitions it -- mesh::create_cell_partitioner(
 * GhostMode::none), src/mesh.cpp:182-183: every rank holds its own cells only, so the matrix rows and vector
 * entries of dofs on the partition interface are incomplete locally and the reference completes them in every
 * assembly by MatAssemblyBegin/End (src/poisson_problem.cpp:132-133) and b.scatter_rev(std::plus) (:154).
 * Call after mesh, dofmap, Dirichlet dofs, exterior facets, coefficients, the forward-scatter plan
 * (zzz_halo_upload) and the global indices have been uploaded: the ranks exchange ONCE the cells that touch a
 * neighbour's dofs, every rank appends what it receives as ghost cells (new ghost dofs join the forward scatter
 * of their owners) and from then on assembles complete owned rows with no exchange per assembly: A and b equal
 * the reference's after its MatAssembly / scatter_rev.  zzz_local_sizes reports the extended sizes. */
int zzz_ghost_layer_build(zzz_ctx* ctx);
/* sizes = {vertices, cells, owned block dofs, ghost block dofs, cells the caller uploaded, neighbours} */
int zzz_local_sizes(const zzz_ctx* ctx, int64_t sizes[6]);

/* The library keeps the owned dofs in an INTERNAL locality order of its own (computed from geometry when the dofmap
 * is uploaded: lexicographic by lattice cell, entity type by entity type -- csrc/zzz_renumber.hip), so that its speed does
 * not depend on the numbering DOLFINx's partitioner and reordering happened to produce (src/mesh.cpp:153-162,182-186).
 * Every index and vector of this ABI is in the CALLER's numbering; the translation is the library's business.  Two
 * things are visible all the same: (1) MatMult sums a row in ascending INTERNAL column order, so zzz_spmv (and the CG
 * iterates) equal the serial CSR loop bit for bit on the internally ordered system P A P^T, and the caller-ordered loop
 * only to round-off -- exactly as PETSc's result depends on the local numbering of the run; (2) this function, which
 * returns P: perm[i] = caller index of internal owned block dof i (the identity when the caller's order was kept:
 * structured feeds of this repository, meshes that are not a lattice, ZZZ_RENUMBER=0).  kind (optional): low bits 0 caller's
 * dof order kept, 1 lattice order, 2 coordinate-bin order (ZZZ_RENUMBER=2 only); bit 4 (16): the CELLS are kept in an
 * internal order as well (simplex type by simplex type, lattice cube by lattice cube) -- invisible at this ABI except
 * that an entry of A or b is then the sum of its cells' contributions in that order instead of the caller's cell
 * order (the same terms; differences of the last bits).  No reference counterpart. */
int zzz_internal_order_download(zzz_ctx* ctx, int32_t* perm /* n_owned */, int32_t* kind);

/* ---- solve ----------------------------------------------------------------------------- */

/* solver_function(u, b) (src/poisson_problem.cpp:164-179; src/cgpoisson_problem.cpp:178-244;
 * called under `ZZZ Solve` at src/main.cpp:208-211).  Solves A u = b with the vectors held by
 * the context; returns the Krylov iteration count like KSPGetIterationNumber / cg()'s k.
 * rnorm[0] = final norm, rnorm[1] = initial norm (variant CGH: <r,r> and <r0,r0>). */
int zzz_cg_solve(zzz_ctx* ctx, const zzz_solver_opts* opts, int* iters, double* rnorm);

/* Residual-norm history of the last solve (KSPGetResidualHistory): copies min(n, iters+1). */
int zzz_cg_history(zzz_ctx* ctx, int n, double* out);

/* About the last zzz_cg_solve: info[0] = 1 when the iteration ran as two kernels (product fused with the
 * direction update p = z + b p, x += a p: the A/B variant ZZZ_CG_FUSED=2), 0 for the three-kernel form; bit 1 (value 2)
 * when Jacobi's inverse diagonal was read as 16-bit codes into a table of its distinct values and z = D^-1 r recomputed
 * instead of stored (68 instead of 80 B per row and iteration in the two vector kernels; the same bits), info[0] >> 8 = those
 * distinct values;
 * info[1] = its iteration count (same iterates, bit for bit, either way); info[2] = how it ended, in
 * KSPConvergedReason's numbering: 2 KSP_CONVERGED_RTOL, 3 KSP_CONVERGED_ATOL, -3 KSP_DIVERGED_ITS (max_it / kmax
 * reached), -4 KSP_DIVERGED_DTOL, -9 KSP_DIVERGED_NANORINF; info[3] = with ZZZ_PC_CHEBYSHEV_JACOBI the upper bound of
 * the spectrum of D^-1 A the polynomial was built on, x 1e6 (0 otherwise). */
int zzz_cg_info(zzz_ctx* ctx, int64_t info[4]);

/* Average duration (ms) and count of the SpMV launches event-timed during the last
 * zzz_cg_solve with opts.profile != 0. */
int zzz_profile_get(zzz_ctx* ctx, double* spmv_avg_ms, int64_t* spmv_count);

/* Storage the CG SpMV streams for the current matrix (no reference counterpart: MatMult on AIJ reads
 * 12 B per nonzero, src/poisson_problem.cpp:168-177 [EXT]).  info[0] = 1 when the column stream is the
 * 16-bit band code (10 B per nonzero), 0 for int32 columns; info[1] = offset bits of the code;
 * info[2] = tiles left on int32 columns; info[3] = number of tiles; info[4] = lanes per row of the row sums
 * (1: a row's products are added in column order like the scalar CPU loop; L > 1, chosen for rows of
 * >= 128 nonzeros on average: lane j adds the j-th contiguous chunk of ceil(len/L) products in column
 * order, chunk sums combined pairwise ((c0+c1)+(c2+c3))+...); info[5] = 1 when the SpMV runs on the sliced-ELL
 * copy (chosen for matrices small enough to stay cache-resident between CG iterations; same column-order row
 * sums, bit-identical results) instead of the CSR tile kernel; info[6..7] = 0. * When the stream carries x windows (block size 3: the columns a group of 256 rows reaches are loaded into LDS once per
 * group, the stream's codes are window indices) info[3] = -(LDS doubles per workgroup) and info[2] = bytes of x those
 * loads move per product. */
int zzz_spmv_info(zzz_ctx* ctx, int64_t info[8]);
/* The values of the operator stream (csrc/zzz_sellp.hip): info[0] = 0 when the product reads them as doubles, 1 / 2 when it
 * reads 16-bit codes into a dictionary of the matrix's DISTINCT values held in memory / copied into LDS by every workgroup
 * (matrices of regular meshes hold few: ~1 300 at 10 M-dof P1 Poisson; at most 65 535 qualify; same doubles, same
 * order of operations, bit-identical products); info[1] = distinct values (+0.0 included); info[2] = bytes per product of
 * the stream in the form in use; info[3] = bytes per product with the values as doubles; info[4] = 1 when the stream is
 * one of one-chunk slices and the product runs on the kernel for those (csrc/zzz_sellp_pipe.hip: two rows per lane), else 0;
 * info[5] = workgroups per CU of the product's persistent grid.  0s when the product does not run on the stream.
 * zzz_spmv_values_info writes info[0..3] only (its signature of round 4: a caller built against that header passes four
 * entries); everything from info[4] on comes through zzz_spmv_values_info2, which writes the first min(n, 10) entries:
 * info[6] = 1 when the product of a block-size-3 matrix runs in block-row form (csrc/zzz_sellp_blk.hip: one lane per node,
 * 16-bit codes into a table of the matrix's distinct 3 x 3 blocks in LDS), info[7] = entries of that table (zero block
 * included), info[8] = chunks of 16 block slots per node, info[9] = 1 when the table's rows (nine doubles each) sit in LDS, 2 when
 * they are rows of nine 16-bit value offsets in memory and the VALUES sit in LDS (more than 2 200 distinct blocks; then
 * info[7] still counts the blocks); info[2] is then the bytes of THAT form. */
int zzz_spmv_values_info(zzz_ctx* ctx, int64_t info[4]);
int zzz_spmv_values_info2(zzz_ctx* ctx, int n, int64_t* info);
/* Version of this header's ABI: bumped whenever an existing entry point changes what it reads or writes (6: round 6). */
#define ZZZ_ABI_VERSION 6
int zzz_abi_version(void);

/* ---- multi-GPU (one context per GPU; RCCL over xGMI) --------------------------------------- */

#define ZZZ_UNIQUE_ID_BYTES 128

/* dlopen librccl.so.1 now (it is otherwise loaded by the first zzz_comm_* call).  For processes that also
 * hold another copy of RCCL -- a Python process that imports torch, which bundles its own -- so that every
 * rank binds the SAME library: call it on every rank before that other copy is loaded.  No reference
 * counterpart (MPI_Init, src/main.cpp:42 [EXT], is the closest). */
int zzz_comm_load(void);
/* File the bound librccl was loaded from (dladdr), "" before zzz_comm_load.  Every rank of a job must report the
 * same library: bench.py compares them before the first collective and exits non-zero on a mismatch. */
const char* zzz_comm_library_path(void);
/* What the multi-GPU path of this context does, for diagnostics (bench.py prints it per rank): info = {ranks,
 * rank, neighbours, bytes sent per forward scatter, bytes received, 1 = halo on its own communicator + stream
 * (ncclCommSplit) / 2 = halo through peer memory (zzz_comm_p2p_halo), 1 = CG scalars through the peer-memory mailboxes, 1 = halo overlapped with the interior rows,
 * interior / boundary work items (groups of 256 rows or tiles) of the product, 1 = host-mediated local backend, mean exposed
 * halo wait per product of the last profiled solve in ns (main stream idle between its interior rows and the halo's arrival)}. */
int zzz_comm_info(zzz_ctx* ctx, int64_t info[12]);

/* ncclGetUniqueId on the root; ship the bytes to the other ranks out of band (the driver's
 * threads share memory; bench.py broadcasts them).  Replaces MPI_COMM_WORLD bootstrap. */
int zzz_comm_unique_id(void* id /* ZZZ_UNIQUE_ID_BYTES */);

/* ncclCommInitRank: attaches this context as `rank` of `nranks`. */
int zzz_comm_init(zzz_ctx* ctx, int nranks, int rank, const void* id);

/* Host-mediated communicator for single-process runs: the `nranks` contexts are driven by `nranks`
 * threads of one process (they may share a GPU); collectives meet at a barrier and exchange through
 * host memory.  Functionally identical to the RCCL path, far slower: meant for validating
 * partitioned runs on a single-GPU machine.  Every rank must make the same sequence of calls. */
int zzz_local_group_create(int nranks, void** group);
void zzz_local_group_destroy(void* group);
/* Break the group: every rank waiting in (or later entering) one of its collectives returns ZZZ_ERR_RCCL instead
 * of waiting for a rank that has failed.  The library does this itself when a rank fails INSIDE a collective. */
void zzz_local_group_abort(void* group);
int zzz_comm_init_local(zzz_ctx* ctx, void* group, int rank);

/* Optional: carry the CG's scalar all-reduces (MPI_Allreduce inside la::inner_product /
 * la::squared_norm, src/cg.h:53,65,74) through peer memory over xGMI instead of ncclAllReduce: each
 * rank exports a small device mailbox, every rank attaches all of them (hipIpcOpenMemHandle between
 * processes, peer access between contexts of one process).  Call after zzz_comm_init /
 * zzz_comm_init_local, on every rank: export, exchange the handles out of band in rank order (as the
 * unique id), attach.  Attach runs test rounds and makes the ranks agree; *enabled = 0 means the
 * communicator's own all-reduce stays in use (never an error).  The mailbox sums in rank order on
 * every rank (bit-identical everywhere); RCCL sums in its own order: the same iteration to round-off. */
#define ZZZ_P2P_HANDLE_BYTES 128
/* A communicator with no transport of its own: only the peer-memory all-reduce and halo below work on it.  For
 * replicated runs and for exercising the peer-memory transport between processes. */
int zzz_comm_init_peer_only(zzz_ctx* ctx, int nranks, int rank);
int zzz_comm_p2p_export(zzz_ctx* ctx, void* handle /* ZZZ_P2P_HANDLE_BYTES */);
int zzz_comm_p2p_attach(zzz_ctx* ctx, const void* handles /* nranks x ZZZ_P2P_HANDLE_BYTES */, int* enabled);
int zzz_comm_p2p_disable(zzz_ctx* ctx);
/* back on after zzz_comm_p2p_disable, only if attach had succeeded; every rank must make the same call */
int zzz_comm_p2p_enable(zzz_ctx* ctx, int* enabled);
/* The same allocation carries a halo window (ZZZ_P2P_HALO_MB, default 64): with the mailboxes enabled, the forward
 * scatter of the product's input vector (common::Scatterer::scatter_fwd inside MatMult, src/poisson_problem.cpp:177;
 * src/cgpoisson_problem.cpp:225-229) is then plain device stores into the NEIGHBOUR's window over xGMI plus an arrival
 * tag, and a copy out of the own window on the receiving side -- no ncclSend / ncclRecv, no communicator in the solve
 * loop at all (a peer-only communicator becomes a complete transport).  Plans that do not fit the window (more than 32
 * neighbours, or a message beyond window / 2 / ranks) keep the communicator's send / recv.  on = 0 keeps the halo on
 * the communicator while the all-reduces stay on the mailboxes (A/B; every rank must make the same call);
 * *in_use = 1 when the next exchange goes through the window.
 * COLLECTIVE, all four: with the mailboxes attached, zzz_comm_p2p_attach, zzz_comm_p2p_enable, zzz_comm_p2p_halo and
 * zzz_halo_upload end in one agreement of all ranks on the exchange's transport (both ends of an exchange must use the
 * same one).  Every rank of the communicator must make the same calls the same number of times, in the same order -- as
 * MPI ranks must for MPI_Comm_dup; the agreement goes through the communicator's own all-reduce (blocking, no time-out)
 * and, on a peer-only communicator, through the mailboxes (a rank that waits ten minutes for the others gives up with
 * ZZZ_ERR_RCCL). */
int zzz_comm_p2p_halo(zzz_ctx* ctx, int on, int* in_use);

/* The forward scatter of common::Scatterer / IndexMap (src/cgpoisson_problem.cpp:187-190,
 * 225-229): for neighbour k, this rank sends x[send_idx[send_off[k]..send_off[k+1])] (owned
 * block dofs) and receives recv_cnt[k] block values into the next ghost slots; the ghost block
 * range is ordered by neighbour, then by the sender's send order. */
int zzz_halo_upload(zzz_ctx* ctx, int nneigh, const int32_t* neigh_rank, const int64_t* send_off,
                    const int32_t* send_idx, const int64_t* recv_cnt);

#ifdef __cplusplus
}
#endif
#endif /* ZZZ_ABI_H */

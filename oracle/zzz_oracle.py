"""ctypes binding of oracle/libzzz_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module
(see the header of zzz_oracle.c, including its PARITY UNPINNED statement).  The product path
(performance-test_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
last_pcg_loop_seconds = 0.0

i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")


def build(force=False):
    so = os.path.join(_HERE, "libzzz_oracle.so")
    src = os.path.join(_HERE, "zzz_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libzzz_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.zo_num_pdofs.restype = C.c_int64
        L.zo_num_pdofs.argtypes = [C.c_int64] * 3 + [C.c_int, C.c_int]
        L.zo_num_entities.argtypes = [C.c_int64] * 3 + [C.c_int, i64p]
        L.zo_mesh_size.argtypes = [C.c_int64, C.c_int, C.c_int64, C.c_int64, C.c_int, i64p]
        L.zo_box_mesh.argtypes = [C.c_int64] * 3 + [f64p, i32p]
        L.zo_ndofs_cell.argtypes = [C.c_int]
        L.zo_ref_nodes.argtypes = [C.c_int, f64p]
        L.zo_tabulate_poisson_a.argtypes = [C.c_int, f64p, f64p]
        L.zo_tabulate_poisson_L_cell.argtypes = [C.c_int, f64p, f64p, f64p]
        L.zo_tabulate_poisson_L_facet.argtypes = [C.c_int, f64p, f64p, C.c_int, f64p]
        L.zo_tabulate_poisson_M.argtypes = [C.c_int, f64p, f64p, f64p]
        L.zo_tabulate_elasticity_a.argtypes = [C.c_int, f64p, f64p]
        L.zo_tabulate_elasticity_L.argtypes = [C.c_int, f64p, f64p, f64p]
        L.zo_build_dofmap.restype = C.c_int64
        L.zo_build_dofmap.argtypes = [C.c_int, C.c_int64, C.c_int64, i32p, f64p, i32p, C.c_void_p, i64p]
        L.zo_exterior_facets.restype = C.c_int64
        L.zo_exterior_facets.argtypes = [C.c_int64, i32p, C.c_void_p]
        L.zo_locate_bc.argtypes = [C.c_int, C.c_int, C.c_int64, C.c_int64, i32p, f64p, i32p, C.c_int64, u8p]
        L.zo_interpolate.argtypes = [C.c_int, C.c_int64, f64p, f64p]
        L.zo_pattern.restype = C.c_int64
        L.zo_pattern.argtypes = [C.c_int64, C.c_int64, C.c_int, C.c_int, i32p, i64p, C.c_void_p]
        L.zo_assemble_matrix.argtypes = [C.c_int, C.c_int, f64p, C.c_int64, i32p, i32p, u8p, C.c_int64, i64p, i32p, f64p]
        L.zo_assemble_vector.argtypes = [C.c_int, C.c_int, f64p, C.c_int64, i32p, i32p, f64p, f64p, C.c_int64, i32p,
                                         u8p, C.c_int64, f64p]
        L.zo_action_poisson.argtypes = [C.c_int, f64p, C.c_int64, i32p, i32p, u8p, C.c_int64, f64p, f64p]
        L.zo_spmv.argtypes = [C.c_int64, i64p, i32p, f64p, f64p, f64p]
        L.zo_cg.argtypes = [C.c_int64, i64p, i32p, f64p, f64p, f64p, C.c_int, C.c_double, f64p]
        L.zo_cg_matfree_poisson.argtypes = [C.c_int, f64p, C.c_int64, i32p, i32p, u8p, C.c_int64, f64p, f64p,
                                            C.c_int, C.c_double]
        L.zo_pcg.argtypes = [C.c_int64, i64p, i32p, f64p, f64p, f64p, C.c_int, C.c_int, C.c_double, C.c_double,
                             C.c_int, f64p]
        L.zo_pcg_sr.argtypes = L.zo_pcg.argtypes
        L.zo_pcg_cheb.argtypes = [C.c_int64, i64p, i32p, f64p, f64p, f64p, C.c_int, C.c_int, C.c_double, C.c_double,
                                  C.c_double, C.c_int, f64p]
        L.zo_esteig.restype = C.c_double
        L.zo_esteig.argtypes = [C.c_int64, i64p, i32p, f64p, C.c_int, C.c_int64]
        L.zo_noise.restype = C.c_double
        L.zo_noise.argtypes = [C.c_int64]
        L.zo_spmv_chunked.argtypes = [C.c_int64, i64p, i32p, f64p, f64p, f64p, C.c_int]
        L.zo_norm2.restype = C.c_double
        L.zo_norm2.argtypes = [C.c_int64, f64p]
        L.zo_set_num_threads.argtypes = [C.c_int]
        _LIB = L
    return _LIB


FORM_POISSON, FORM_ELASTICITY = 0, 1
PC_NONE, PC_JACOBI = 0, 1
NORM_PRECONDITIONED, NORM_UNPRECONDITIONED, NORM_NATURAL = 0, 1, 2


def set_num_threads(n):
    lib().zo_set_num_threads(int(n))


def num_threads():
    return int(lib().zo_num_threads())


def num_pdofs(i, j, k, r, order):
    return int(lib().zo_num_pdofs(i, j, k, r, order))


def num_entities(i, j, k, r):
    out = np.zeros(4, np.int64)
    lib().zo_num_entities(i, j, k, r, out)
    return tuple(int(v) for v in out)


def mesh_size(target_dofs, strong, nproc, dofs_per_node, order):
    """(Nx, Ny, Nz, r) of src/mesh.cpp:78-151"""
    out = np.zeros(4, np.int64)
    lib().zo_mesh_size(int(target_dofs), 1 if strong else 0, int(nproc), int(dofs_per_node), int(order), out)
    return tuple(int(v) for v in out)


def box_mesh(nx, ny, nz):
    nv = (nx + 1) * (ny + 1) * (nz + 1)
    nc = 6 * nx * ny * nz
    x = np.zeros((nv, 3))
    cells = np.zeros((nc, 4), np.int32)
    lib().zo_box_mesh(nx, ny, nz, x, cells)
    return x, cells


def ndofs_cell(order):
    return int(lib().zo_ndofs_cell(order))


def ref_nodes(order):
    X = np.zeros((ndofs_cell(order), 3))
    lib().zo_ref_nodes(order, X)
    return X


def tabulate(form, order, cd, w=None, facet=None):
    """form in {'poisson_a','poisson_L','poisson_L_facet','poisson_M','elasticity_a','elasticity_L'}"""
    L = lib()
    nd = ndofs_cell(order)
    cd = np.ascontiguousarray(cd, np.float64).reshape(12)
    if w is not None:
        w = np.ascontiguousarray(w, np.float64)
    if form == "poisson_a":
        A = np.zeros((nd, nd))
        L.zo_tabulate_poisson_a(order, cd, A)
        return A
    if form == "elasticity_a":
        A = np.zeros((3 * nd, 3 * nd))
        L.zo_tabulate_elasticity_a(order, cd, A)
        return A
    if form == "poisson_L":
        b = np.zeros(nd)
        L.zo_tabulate_poisson_L_cell(order, cd, w, b)
        return b
    if form == "poisson_L_facet":
        b = np.zeros(nd)
        L.zo_tabulate_poisson_L_facet(order, cd, w, int(facet), b)
        return b
    if form == "poisson_M":
        b = np.zeros(nd)
        L.zo_tabulate_poisson_M(order, cd, w, b)
        return b
    if form == "elasticity_L":
        b = np.zeros(3 * nd)
        L.zo_tabulate_elasticity_L(order, cd, w, b)
        return b
    raise ValueError(form)


def build_dofmap(order, x, cells):
    nv, nc = x.shape[0], cells.shape[0]
    nd = ndofs_cell(order)
    cell_dofs = np.zeros((nc, nd), np.int32)
    counts = np.zeros(3, np.int64)
    ndofs = lib().zo_build_dofmap(order, nv, nc, cells, x, cell_dofs, None, counts)
    dof_x = np.zeros((ndofs, 3))
    lib().zo_build_dofmap(order, nv, nc, cells, x, cell_dofs, dof_x.ctypes.data_as(C.c_void_p), counts)
    return int(ndofs), cell_dofs, dof_x, tuple(int(c) for c in counts)


def exterior_facets(cells):
    nc = cells.shape[0]
    n = lib().zo_exterior_facets(nc, cells, None)
    out = np.zeros((n, 2), np.int32)
    lib().zo_exterior_facets(nc, cells, out.ctypes.data_as(C.c_void_p))
    return out


def locate_bc(kind, order, x, cells, cell_dofs, ndofs):
    m = np.zeros(ndofs, np.uint8)
    lib().zo_locate_bc(kind, order, x.shape[0], cells.shape[0], cells, x, cell_dofs, ndofs, m)
    return m


def interpolate(which, dof_x):
    n = dof_x.shape[0]
    out = np.zeros(3 * n if which == 2 else n)
    lib().zo_interpolate(which, n, dof_x, out)
    return out


def pattern(nblock, cell_dofs, bs):
    nc, nd = cell_dofs.shape
    rowptr = np.zeros(nblock * bs + 1, np.int64)
    nnz = lib().zo_pattern(nblock, nc, nd, bs, cell_dofs, rowptr, None)
    cols = np.zeros(nnz, np.int32)
    lib().zo_pattern(nblock, nc, nd, bs, cell_dofs, rowptr, cols.ctypes.data_as(C.c_void_p))
    return rowptr, cols


def assemble_matrix(form, order, x, cells, cell_dofs, bc_scalar, rowptr, cols):
    vals = np.zeros(cols.shape[0])
    rc = lib().zo_assemble_matrix(form, order, x, cells.shape[0], cells, cell_dofs, bc_scalar, rowptr.shape[0] - 1,
                                  rowptr, cols, vals)
    if rc != 0:
        raise RuntimeError("entry missing from sparsity pattern")
    return vals


def assemble_vector(form, order, x, cells, cell_dofs, f, g, facets, bc_scalar):
    n = bc_scalar.shape[0]
    b = np.zeros(n)
    if g is None:
        g = np.zeros(1)
    if facets is None:
        facets = np.zeros((0, 2), np.int32)
    lib().zo_assemble_vector(form, order, x, cells.shape[0], cells, cell_dofs, f, g, facets.shape[0],
                             np.ascontiguousarray(facets).reshape(-1) if facets.size else np.zeros(2, np.int32),
                             bc_scalar, n, b)
    return b


def near_nullspace(dof_x):
    """build_near_nullspace (src/elasticity_problem.cpp:36-94): (basis [6][3 n], largest deviation from orthonormality)"""
    n = dof_x.shape[0]
    B = np.zeros((6, 3 * n))
    lib().zo_near_nullspace.restype = C.c_double
    lib().zo_near_nullspace.argtypes = [C.c_int64, np.ctypeslib.ndpointer(np.float64, flags="C"), np.ctypeslib.ndpointer(np.float64, flags="C")]
    dev = lib().zo_near_nullspace(n, np.ascontiguousarray(dof_x, np.float64), B)
    return B, float(dev)


def action_poisson(order, x, cells, cell_dofs, bc, u):
    y = np.zeros_like(u)
    lib().zo_action_poisson(order, x, cells.shape[0], cells, cell_dofs, bc, u.shape[0], u, y)
    return y


def spmv(rowptr, cols, vals, x):
    y = np.zeros(rowptr.shape[0] - 1)
    lib().zo_spmv(y.shape[0], rowptr, cols, vals, x, y)
    return y


def spmv_chunked(rowptr, cols, vals, x, lanes):
    """y = A x in the summation order of the GPU's multi-lane row phase (lanes = 1: spmv)"""
    y = np.zeros(rowptr.shape[0] - 1)
    lib().zo_spmv_chunked(y.shape[0], rowptr, cols, vals, x, y, int(lanes))
    return y


def cg(rowptr, cols, vals, b, x0=None, kmax=50, rtol=1e-8):
    """src/cg.h:38-86; returns (iterations, x, <r,r>/<r0,r0>)"""
    x = np.zeros_like(b) if x0 is None else np.array(x0, np.float64)
    rn = np.zeros(1)
    k = lib().zo_cg(b.shape[0], rowptr, cols, vals, b, x, int(kmax), float(rtol), rn)
    return int(k), x, float(rn[0])


def cg_matfree_poisson(order, x, cells, cell_dofs, bc, b, kmax=100, rtol=1e-6):
    u = np.zeros_like(b)
    k = lib().zo_cg_matfree_poisson(order, x, cells.shape[0], cells, cell_dofs, bc, b.shape[0], b, u, int(kmax),
                                    float(rtol))
    return int(k), u


def pcg(rowptr, cols, vals, b, pc=PC_JACOBI, norm_type=NORM_PRECONDITIONED, rtol=1e-8, atol=1e-50, max_it=10000):
    """PETSc KSPCG restatement; returns (iterations, x, final_norm, initial_norm)"""
    x = np.zeros_like(b)
    rn = np.zeros(3)
    it = lib().zo_pcg(b.shape[0], rowptr, cols, vals, b, x, pc, norm_type, rtol, atol, max_it, rn)
    global last_pcg_loop_seconds
    last_pcg_loop_seconds = float(rn[2])  # the iteration loop alone (setup copies excluded)
    return int(it), x, float(rn[0]), float(rn[1])


def pcg_single_reduction(rowptr, cols, vals, b, pc=PC_JACOBI, norm_type=NORM_PRECONDITIONED, rtol=1e-8, atol=1e-50,
                         max_it=10000):
    """KSPCG with -ksp_cg_single_reduction restated; returns (iterations, x, final_norm, initial_norm)"""
    x = np.zeros_like(b)
    rn = np.zeros(3)
    it = lib().zo_pcg_sr(b.shape[0], rowptr, cols, vals, b, x, pc, norm_type, rtol, atol, max_it, rn)
    return int(it), x, float(rn[0]), float(rn[1])


def pcg_chebyshev(rowptr, cols, vals, b, degree=2, ratio=10.0, rtol=1e-8, atol=1e-50, max_it=10000, est_its=0):
    """KSPCG with the Chebyshev-Jacobi polynomial preconditioner restated; returns (iterations, x, final_norm,
    initial_norm, spectrum bound).  est_its = 0: Gershgorin's bound; > 0: min(Gershgorin, 1.1 x the Lanczos estimate of
    est_its Jacobi-PCG iterations on the noise vector)"""
    x = np.zeros_like(b)
    rn = np.zeros(3)
    it = lib().zo_pcg_cheb(b.shape[0], rowptr, cols, vals, b, x, degree, est_its, ratio, rtol, atol, max_it, rn)
    return int(it), x, float(rn[0]), float(rn[1]), float(rn[2])


def esteig(rowptr, cols, vals, est_its=10, offset=0):
    """largest Ritz value of D^-1 A after est_its Jacobi-PCG iterations on the noise vector (zo_esteig)"""
    return float(lib().zo_esteig(rowptr.shape[0] - 1, rowptr, cols, vals, est_its, offset))


def norm2(x):
    return float(lib().zo_norm2(x.shape[0], x))


class Problem:
    """The reference's problem() factories restated end-to-end on one rank
    (src/poisson_problem.cpp:29-182, src/elasticity_problem.cpp:97-264)."""

    def __init__(self, problem_type, order, nx, ny, nz):
        self.problem_type, self.order = problem_type, order
        self.bs = 3 if problem_type == "elasticity" else 1
        self.form = FORM_ELASTICITY if problem_type == "elasticity" else FORM_POISSON
        self.x, self.cells = box_mesh(nx, ny, nz)
        self.nblock, self.cell_dofs, self.dof_x, self.counts = build_dofmap(order, self.x, self.cells)
        self.n = self.nblock * self.bs
        bcm = locate_bc(1 if problem_type == "elasticity" else 0, order, self.x, self.cells, self.cell_dofs,
                        self.nblock)
        self.bc_block = bcm
        self.bc = np.repeat(bcm, self.bs).astype(np.uint8)
        if problem_type == "elasticity":
            self.f = interpolate(2, self.dof_x)
            self.g = None
            self.facets = None
        else:
            self.f = interpolate(0, self.dof_x)
            self.g = interpolate(1, self.dof_x)
            self.facets = exterior_facets(self.cells)
        self.rowptr = self.cols = self.vals = self.b = None

    def assemble(self):
        self.rowptr, self.cols = pattern(self.nblock, self.cell_dofs, self.bs)
        self.vals = assemble_matrix(self.form, self.order, self.x, self.cells, self.cell_dofs, self.bc, self.rowptr,
                                    self.cols)
        self.b = assemble_vector(self.form, self.order, self.x, self.cells, self.cell_dofs, self.f, self.g,
                                 self.facets, self.bc)
        return self

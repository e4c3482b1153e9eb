"""ctypes mirror of the two C ABIs of this package (include/zzz_abi.h, include/zzz_host.h).

Used by tests/ and bench.py to drive the same entry points the C++ driver
(host/main.cpp, the `dolfinx-scaling-test` binary) calls.  No compute happens in Python and there
is no CPU fallback: `Context()` raises when the HIP library or a GPU is missing.
Never imports anything from oracle/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROOT = os.path.dirname(PKG)

FORM_POISSON, FORM_ELASTICITY = 0, 1
COEFF_F, COEFF_G = 0, 1
VEC_B, VEC_U = 0, 1
PC_NONE, PC_JACOBI, PC_CHEBYSHEV_JACOBI = 0, 1, 2
NORM_PRECONDITIONED, NORM_UNPRECONDITIONED, NORM_NATURAL = 0, 1, 2
CG_PETSC, CG_CGH = 0, 1
OP_CSR, OP_MATFREE = 0, 1
ERR_NO_GPU = 4

ABI_SYMBOLS = [
    "zzz_device_count", "zzz_device_memory", "zzz_ctx_create", "zzz_ctx_destroy", "zzz_last_error", "zzz_sync", "zzz_mesh_upload",
    "zzz_dofmap_upload", "zzz_bc_upload", "zzz_facets_upload", "zzz_coeff_upload", "zzz_cube_generate",
    "zzz_csr_pattern_build",
    "zzz_csr_sizes", "zzz_csr_download", "zzz_csr_rowptr64_download", "zzz_csr_upload_values", "zzz_assemble_matrix", "zzz_assemble_vector",
    "zzz_vec_download", "zzz_vec_upload", "zzz_vec_norm", "zzz_spmv", "zzz_spmv_time", "zzz_spmv_values_info", "zzz_spmv_values_info2", "zzz_abi_version", "zzz_action", "zzz_matfree_setup", "zzz_matfree_info", "zzz_matfree_diagonal", "zzz_action_time", "zzz_near_nullspace_build", "zzz_near_nullspace_download", "zzz_cg_solve", "zzz_cg_history",
    "zzz_profile_get", "zzz_cg_info", "zzz_internal_order_download", "zzz_global_ids_upload", "zzz_global_ids_download", "zzz_ghost_layer_build", "zzz_local_sizes", "zzz_spmv_info", "zzz_comm_load", "zzz_comm_library_path", "zzz_comm_info", "zzz_comm_unique_id", "zzz_comm_init", "zzz_halo_upload", "zzz_local_group_create",
    "zzz_local_group_destroy", "zzz_local_group_abort", "zzz_comm_init_local", "zzz_comm_init_peer_only", "zzz_comm_p2p_export", "zzz_comm_p2p_attach", "zzz_comm_p2p_disable", "zzz_comm_p2p_enable", "zzz_comm_p2p_halo",
]
HOST_SYMBOLS = [
    "zzzh_num_pdofs", "zzzh_num_entities", "zzzh_mesh_size", "zzzh_count_suffix", "zzzh_part_create", "zzzh_part_create_native", "zzzh_part_create_spoke", "zzzh_part_create_spoke_part", "zzzh_spoke_size", "zzzh_part_destroy", "zzzh_part_global_verts",
    "zzzh_last_error", "zzzh_part_sizes", "zzzh_part_x", "zzzh_part_cells", "zzzh_part_cell_dofs",
    "zzzh_part_facets", "zzzh_part_bc_dofs", "zzzh_part_dof_x", "zzzh_part_global_dofs", "zzzh_part_coeff",
    "zzzh_part_neigh", "zzzh_part_send_off", "zzzh_part_send_idx", "zzzh_part_recv_cnt",
]

(NVERTS, NCELLS, NOWNED, NGHOST, ND, BS, NFACETS, NBC, NNEIGH, NSEND, GLOBAL_DOFS, GLOBAL_CELLS, OWNED_CELLS,
 OWN_OFFSET, GLOBAL_NBC, BC_MODE, NSIZES) = range(17)


class SolverOpts(C.Structure):
    _fields_ = [("variant", C.c_int32), ("pc", C.c_int32), ("norm", C.c_int32), ("op", C.c_int32),
                ("max_it", C.c_int32), ("profile", C.c_int32), ("single_reduction", C.c_int32), ("error_if_not_converged", C.c_int32),
                ("rtol", C.c_double), ("atol", C.c_double), ("dtol", C.c_double),
                ("pc_degree", C.c_int32), ("pc_esteig_its", C.c_int32), ("pc_ratio", C.c_double)]


class ZzzError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libzzz_hip error {code}: {msg}")
        self.code = code


def build(verbose=False):
    """make -C performance-test_amd (hipcc --offload-arch=gfx950; cross-compiles without a GPU)"""
    subprocess.check_call(["make", "-C", PKG, "-j4"], stdout=None if verbose else subprocess.DEVNULL)


def hip_lib_path():
    # ZZZ_HIP_LIB: another build of the same library, for the A/B tools (tools/ab_lib.py); never a different backend
    return os.environ.get("ZZZ_HIP_LIB") or os.path.join(PKG, "libzzz_hip.so")


def host_lib_path():
    return os.path.join(PKG, "libzzz_host.so")


_HIP = None
_HOST = None

_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")


def hip():
    """Loads libzzz_hip.so; raises (never falls back) when it is missing."""
    global _HIP
    if _HIP is None:
        path = hip_lib_path()
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run `make -C performance-test_amd` (no CPU fallback exists)")
        L = C.CDLL(path)
        L.zzz_last_error.restype = C.c_char_p
        L.zzz_last_error.argtypes = [C.c_void_p]
        L.zzz_ctx_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        L.zzz_ctx_destroy.argtypes = [C.c_void_p]
        L.zzz_ctx_destroy.restype = None
        L.zzz_sync.argtypes = [C.c_void_p]
        L.zzz_mesh_upload.argtypes = [C.c_void_p, C.c_int64, _f64p, C.c_int64, _i32p]
        L.zzz_dofmap_upload.argtypes = [C.c_void_p, C.c_int, C.c_int, _i32p, C.c_int64, C.c_int64]
        L.zzz_bc_upload.argtypes = [C.c_void_p, C.c_int64, _i32p]
        L.zzz_facets_upload.argtypes = [C.c_void_p, C.c_int64, _i32p]
        L.zzz_coeff_upload.argtypes = [C.c_void_p, C.c_int, _f64p]
        L.zzz_cube_generate.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_int,
                                        _i64p]
        L.zzz_csr_pattern_build.argtypes = [C.c_void_p]
        L.zzz_csr_sizes.argtypes = [C.c_void_p] + [C.POINTER(C.c_int64)] * 3
        L.zzz_csr_download.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.zzz_csr_upload_values.argtypes = [C.c_void_p, _f64p]
        L.zzz_assemble_matrix.argtypes = [C.c_void_p, C.c_int]
        L.zzz_assemble_vector.argtypes = [C.c_void_p, C.c_int]
        L.zzz_vec_download.argtypes = [C.c_void_p, C.c_int, _f64p]
        L.zzz_vec_upload.argtypes = [C.c_void_p, C.c_int, _f64p]
        L.zzz_vec_norm.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double)]
        L.zzz_spmv.argtypes = [C.c_void_p, _f64p, _f64p]
        L.zzz_spmv_time.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double)]
        L.zzz_action.argtypes = [C.c_void_p, _f64p, _f64p]
        L.zzz_matfree_setup.argtypes = [C.c_void_p]
        L.zzz_matfree_diagonal.argtypes = [C.c_void_p, _f64p]
        L.zzz_matfree_info.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
        L.zzz_action_time.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double)]
        L.zzz_near_nullspace_build.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        L.zzz_near_nullspace_download.argtypes = [C.c_void_p, C.c_int, _f64p]
        L.zzz_cg_solve.argtypes = [C.c_void_p, C.POINTER(SolverOpts), C.POINTER(C.c_int), C.POINTER(C.c_double)]
        L.zzz_cg_history.argtypes = [C.c_void_p, C.c_int, _f64p]
        L.zzz_profile_get.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
        L.zzz_spmv_info.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
        # (tools/ab_product.sh loads an older build of the library beside the current one: ZZZ_AB_OLD tolerates the entry
        # points that build does not have yet; without it a missing symbol is an error, as everywhere)
        if hasattr(L, "zzz_spmv_values_info") or not os.environ.get("ZZZ_AB_OLD"):
            L.zzz_spmv_values_info.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
        if hasattr(L, "zzz_spmv_values_info2") or not os.environ.get("ZZZ_AB_OLD"):
            L.zzz_spmv_values_info2.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int64)]
        L.zzz_internal_order_download.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int32)]
        L.zzz_comm_unique_id.argtypes = [C.c_void_p]
        L.zzz_comm_init.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.zzz_local_group_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        L.zzz_local_group_destroy.argtypes = [C.c_void_p]
        L.zzz_local_group_destroy.restype = None
        L.zzz_comm_init_local.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.zzz_comm_init_peer_only.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.zzz_comm_p2p_export.argtypes = [C.c_void_p, C.c_void_p]
        L.zzz_comm_p2p_attach.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
        L.zzz_comm_p2p_disable.argtypes = [C.c_void_p]
        L.zzz_comm_p2p_enable.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        L.zzz_halo_upload.argtypes = [C.c_void_p, C.c_int, _i32p, _i64p, _i32p, _i64p]
        _HIP = L
    return _HIP


def host():
    global _HOST
    if _HOST is None:
        path = host_lib_path()
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run `make -C performance-test_amd`")
        L = C.CDLL(path)
        L.zzzh_num_pdofs.restype = C.c_int64
        L.zzzh_num_pdofs.argtypes = [C.c_int64] * 3 + [C.c_int, C.c_int]
        L.zzzh_num_entities.argtypes = [C.c_int64] * 3 + [C.c_int, _i64p]
        L.zzzh_mesh_size.argtypes = [C.c_int64, C.c_int, C.c_int64, C.c_int64, C.c_int, _i64p]
        L.zzzh_part_create.restype = C.c_void_p
        L.zzzh_part_create.argtypes = [C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_int]
        L.zzzh_part_create_native.restype = C.c_void_p
        L.zzzh_part_create_spoke.restype = C.c_void_p
        L.zzzh_part_create_spoke.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int]
        L.zzzh_part_create_spoke_part.restype = C.c_void_p
        L.zzzh_part_create_spoke_part.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.zzzh_spoke_size.restype = C.c_int
        L.zzzh_spoke_size.argtypes = [C.c_int64, C.c_int]
        L.zzzh_part_create_native.argtypes = L.zzzh_part_create.argtypes
        L.zzzh_part_destroy.argtypes = [C.c_void_p]
        L.zzzh_part_destroy.restype = None
        L.zzzh_last_error.restype = C.c_char_p
        L.zzzh_part_sizes.argtypes = [C.c_void_p, _i64p]
        for name, ty in (("x", C.c_double), ("cells", C.c_int32), ("cell_dofs", C.c_int32), ("facets", C.c_int32),
                         ("bc_dofs", C.c_int32), ("dof_x", C.c_double), ("global_dofs", C.c_int64), ("global_verts", C.c_int64),
                         ("neigh", C.c_int32), ("send_off", C.c_int64), ("send_idx", C.c_int32),
                         ("recv_cnt", C.c_int64)):
            f = getattr(L, "zzzh_part_" + name)
            f.restype = C.POINTER(ty)
            f.argtypes = [C.c_void_p]
        L.zzzh_part_coeff.restype = C.POINTER(C.c_double)
        L.zzzh_part_coeff.argtypes = [C.c_void_p, C.c_int]
        _HOST = L
    return _HOST


def device_count():
    return int(hip().zzz_device_count())


def device_memory(device=0):
    """(free, total) bytes of HBM"""
    f, t = C.c_size_t(), C.c_size_t()
    rc = hip().zzz_device_memory(int(device), C.byref(f), C.byref(t))
    if rc:
        raise ZzzError(rc, hip().zzz_last_error(None).decode())
    return int(f.value), int(t.value)


# ------------------------------------------------------------------------------------------------
def count_suffix(n):
    buf = C.create_string_buffer(64)
    if host().zzzh_count_suffix(C.c_int64(int(n)), buf, 64) < 0:
        raise ValueError("number too big")
    return buf.value.decode()


def mesh_size(ndofs, strong, nproc, dofs_per_node, order):
    out = np.zeros(4, np.int64)
    host().zzzh_mesh_size(int(ndofs), 1 if strong else 0, int(nproc), int(dofs_per_node), int(order), out)
    return tuple(int(v) for v in out)


def _arr(ptr, n, dtype, shape=None):
    if n == 0:
        a = np.zeros(0, dtype)
    else:
        a = np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True)
    return a.reshape(shape) if shape is not None else a


class Part:
    """One z-slab partition of the cube problem (host/mesh_part.cpp)."""

    def __init__(self, problem, order, nx, ny, nz, nparts=1, part=0, native=False, spoke=None):
        H = host()
        pid = FORM_ELASTICITY if problem == "elasticity" else FORM_POISSON
        if spoke is not None:
            # the unstructured ring-with-spurs mesh (host/spoke_mesh.cpp): nx = ny = nz = m sub-blocks per block edge
            h = H.zzzh_part_create_spoke_part(pid, order, nx, int(spoke), nparts, part)
        else:
            h = (H.zzzh_part_create_native if native else H.zzzh_part_create)(pid, order, nx, ny, nz, nparts, part)
        if not h:
            raise ValueError(H.zzzh_last_error().decode())
        try:
            s = np.zeros(NSIZES, np.int64)
            H.zzzh_part_sizes(h, s)
            self.sizes = s
            self.problem, self.order, self.form = problem, order, pid
            self.dims = (nx, ny, nz)
            self.nparts, self.part = nparts, part
            self.nverts, self.ncells = int(s[NVERTS]), int(s[NCELLS])
            self.n_owned, self.n_ghost = int(s[NOWNED]), int(s[NGHOST])
            self.nd, self.bs = int(s[ND]), int(s[BS])
            self.nloc = self.n_owned + self.n_ghost
            self.global_dofs_total, self.global_cells = int(s[GLOBAL_DOFS]), int(s[GLOBAL_CELLS])
            self.owned_cells, self.own_offset = int(s[OWNED_CELLS]), int(s[OWN_OFFSET])
            self.global_nbc, self.bc_mode = int(s[GLOBAL_NBC]), int(s[BC_MODE])
            self.x = _arr(H.zzzh_part_x(h), 3 * self.nverts, np.float64, (-1, 3))
            self.cells = _arr(H.zzzh_part_cells(h), 4 * self.ncells, np.int32, (-1, 4))
            self.cell_dofs = _arr(H.zzzh_part_cell_dofs(h), self.nd * self.ncells, np.int32, (-1, self.nd))
            self.facets = _arr(H.zzzh_part_facets(h), 2 * int(s[NFACETS]), np.int32, (-1, 2))
            self.bc_dofs = _arr(H.zzzh_part_bc_dofs(h), int(s[NBC]), np.int32)
            self.dof_x = _arr(H.zzzh_part_dof_x(h), 3 * self.nloc, np.float64, (-1, 3))
            self.global_dofs = _arr(H.zzzh_part_global_dofs(h), self.nloc, np.int64)
            self.global_verts = _arr(H.zzzh_part_global_verts(h), self.nverts, np.int64)
            self.f = _arr(H.zzzh_part_coeff(h, 0), self.nloc * self.bs, np.float64)
            self.g = _arr(H.zzzh_part_coeff(h, 1), self.nloc, np.float64) if pid == FORM_POISSON else None
            nn = int(s[NNEIGH])
            self.neigh = _arr(H.zzzh_part_neigh(h), nn, np.int32)
            self.send_off = _arr(H.zzzh_part_send_off(h), nn + 1, np.int64)
            self.send_idx = _arr(H.zzzh_part_send_idx(h), int(s[NSEND]), np.int32)
            self.recv_cnt = _arr(H.zzzh_part_recv_cnt(h), nn, np.int64)
        finally:
            H.zzzh_part_destroy(h)

    @classmethod
    def spoke(cls, problem, order, m, bc_mode=1, nparts=1, part=0):
        """`--mesh_type unstructured` (src/mesh.cpp:209-453) with m sub-blocks per block edge; bc_mode 1: the whole exterior
        boundary is constrained (0: the reference's markers, possibly an empty set on this geometry); nparts > 1: the
        partition `part` of the cut by polar angle (owner-major global numbering)"""
        return cls(problem, order, m, m, m, nparts, part, spoke=bc_mode)

    def bc_marker(self):
        m = np.zeros(self.nloc * self.bs, np.uint8)
        m[self.bc_dofs] = 1
        return m

    def renumbered(self, kind, seed=0, pattern=None):
        """The same partition as a caller with ANOTHER numbering would feed it (what DOLFINx's partitioner and graph
        reordering leave, src/mesh.cpp:153-162,182-186): owned block dofs, mesh vertices and cells renumbered.
          "random": three independent random permutations (the worst case);
          "rcm":    dofs in reverse Cuthill-McKee order of the dof graph (pattern = (rowptr, cols) of the BLOCK graph,
                    or None to build it from the connectivity), vertices in the order of their P1 dofs' new numbers
                    where that applies, cells sorted by their lowest new dof (what a graph reordering of both gives);
          "reverse": every numbering reversed (a cheap deterministic case).
        Ghost dofs keep their places (the forward scatter defines them).  Q.dof_new_of_old maps old -> new owned dofs."""
        import copy

        rng = np.random.default_rng(seed)
        n, nv, nc = self.n_owned, self.nverts, self.ncells
        if kind == "random":
            new_of_old, v_new_of_old, c_order = rng.permutation(n), rng.permutation(nv), rng.permutation(nc)
        elif kind == "reverse":
            new_of_old, v_new_of_old, c_order = np.arange(n)[::-1].copy(), np.arange(nv)[::-1].copy(), np.arange(nc)[::-1].copy()
        elif kind == "rcm":
            import scipy.sparse as sp
            from scipy.sparse.csgraph import reverse_cuthill_mckee

            if pattern is None:
                cd = self.cell_dofs
                own = cd < n
                rows = np.repeat(cd, self.nd, axis=1).reshape(-1)
                cols = np.tile(cd, (1, self.nd)).reshape(-1)
                keep = (rows < n) & (cols < n)
                G = sp.csr_matrix((np.ones(int(keep.sum()), np.int8), (rows[keep], cols[keep])), shape=(n, n))
                del own
            else:
                rp, cl = pattern
                keep = cl < n
                rows = np.repeat(np.arange(n), np.diff(rp))
                G = sp.csr_matrix((np.ones(int(keep.sum()), np.int8), (rows[keep], cl[keep])), shape=(n, n))
            order = reverse_cuthill_mckee(G, symmetric_mode=True).astype(np.int64)  # order[new] = old
            new_of_old = np.empty(n, np.int64)
            new_of_old[order] = np.arange(n)
            # vertices: by the lowest new dof of the cells' vertex dofs (P1: the vertex's own dof); cells: by lowest new dof
            full0 = np.concatenate([new_of_old, np.arange(n, self.nloc)])
            vkey = np.full(nv, np.iinfo(np.int64).max)
            np.minimum.at(vkey, self.cells.reshape(-1), full0[self.cell_dofs[:, :4]].reshape(-1))
            v_order = np.argsort(vkey, kind="stable")
            v_new_of_old = np.empty(nv, np.int64)
            v_new_of_old[v_order] = np.arange(nv)
            c_order = np.argsort(full0[self.cell_dofs].min(axis=1), kind="stable")
        else:
            raise ValueError(kind)
        full = np.concatenate([new_of_old, np.arange(n, self.nloc)]).astype(np.int64)
        Q = copy.copy(self)
        Q.dof_new_of_old = new_of_old.astype(np.int64)
        bs = self.bs

        def blocks(v, width):
            out = np.empty_like(v)
            out.reshape(self.nloc, width)[full] = v.reshape(self.nloc, width)
            return out

        Q.cell_dofs = full[self.cell_dofs][c_order].astype(np.int32)
        Q.cells = v_new_of_old[self.cells][c_order].astype(np.int32)
        Q.x = np.empty_like(self.x)
        Q.x[v_new_of_old] = self.x
        Q.global_verts = np.empty_like(self.global_verts)
        Q.global_verts[v_new_of_old] = self.global_verts
        c_new_of_old = np.empty(nc, np.int64)
        c_new_of_old[c_order] = np.arange(nc)
        Q.facets = self.facets.copy()
        if Q.facets.size:
            Q.facets[:, 0] = c_new_of_old[self.facets[:, 0]]
        Q.bc_dofs = np.sort(full[self.bc_dofs // bs] * bs + self.bc_dofs % bs).astype(np.int32)
        Q.f = blocks(self.f, bs)
        Q.g = blocks(self.g, 1) if self.g is not None else None
        Q.dof_x = blocks(self.dof_x.reshape(-1), 3).reshape(-1, 3)
        Q.global_dofs = blocks(self.global_dofs, 1)
        Q.send_idx = full[self.send_idx].astype(np.int32)
        return Q


# ------------------------------------------------------------------------------------------------
class Context:
    """RAII wrapper of zzz_ctx (one GPU)."""

    def __init__(self, device=0):
        self.L = hip()
        h = C.c_void_p()
        rc = self.L.zzz_ctx_create(int(device), C.byref(h))
        if rc:
            raise ZzzError(rc, self.L.zzz_last_error(None).decode())
        self.h = h
        self.bs = 1
        self.n_owned = 0

    def close(self):
        if getattr(self, "h", None):
            self.L.zzz_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _ck(self, rc):
        if rc:
            raise ZzzError(rc, self.L.zzz_last_error(self.h).decode())

    def sync(self):
        self._ck(self.L.zzz_sync(self.h))

    def upload_mesh(self, x, cells):
        x = np.ascontiguousarray(x, np.float64)
        cells = np.ascontiguousarray(cells, np.int32)
        self._ck(self.L.zzz_mesh_upload(self.h, x.shape[0], x, cells.shape[0], cells))

    def upload_dofmap(self, order, bs, cell_dofs, n_owned, n_ghost=0):
        cd = np.ascontiguousarray(cell_dofs, np.int32)
        self._ck(self.L.zzz_dofmap_upload(self.h, order, bs, cd, n_owned, n_ghost))
        self.bs, self.n_owned, self.n_ghost = bs, n_owned, n_ghost

    def upload_bc(self, bc_dofs):
        b = np.ascontiguousarray(bc_dofs, np.int32)
        self._ck(self.L.zzz_bc_upload(self.h, b.shape[0], b if b.size else np.zeros(1, np.int32)))

    def upload_facets(self, facets):
        f = np.ascontiguousarray(facets, np.int32).reshape(-1)
        self._ck(self.L.zzz_facets_upload(self.h, f.shape[0] // 2, f if f.size else np.zeros(2, np.int32)))

    def upload_coeff(self, which, values):
        self._ck(self.L.zzz_coeff_upload(self.h, which, np.ascontiguousarray(values, np.float64)))

    def upload_global_ids(self, dof_global, vert_global):
        d = np.ascontiguousarray(dof_global, np.int64)
        v = np.ascontiguousarray(vert_global, np.int64)
        self._ck(self.L.zzz_global_ids_upload(self.h, d.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p)))

    def ghost_layer_build(self):
        """collective: native (GhostMode::none) feed -> ghost-cell layer; returns the new local sizes"""
        self._ck(self.L.zzz_ghost_layer_build(self.h))
        s = self.local_sizes()
        self.n_ghost = s[3]
        return s

    def local_sizes(self):
        s = (C.c_int64 * 6)()
        self._ck(self.L.zzz_local_sizes(self.h, s))
        return [int(v) for v in s]

    def global_ids(self):
        s = self.local_sizes()
        out = np.zeros(s[2] + s[3], np.int64)
        self._ck(self.L.zzz_global_ids_download(self.h, out.ctypes.data_as(C.c_void_p)))
        return out

    def upload_part(self, P):
        """everything problem() sets up before `ZZZ Assemble matrix` (src/poisson_problem.cpp:33-123)"""
        self.upload_mesh(P.x, P.cells)
        self.upload_dofmap(P.order, P.bs, P.cell_dofs, P.n_owned, P.n_ghost)
        self.upload_bc(P.bc_dofs)
        self.upload_facets(P.facets)
        self.upload_coeff(COEFF_F, P.f)
        if P.g is not None:
            self.upload_coeff(COEFF_G, P.g)

    def cube_generate(self, problem, order, nx, ny, nz, nparts=1, part=0):
        """device-side feed (zzz_cube_generate); returns the info array"""
        info = np.zeros(6, np.int64)
        pid = FORM_ELASTICITY if problem == "elasticity" else FORM_POISSON
        self._ck(self.L.zzz_cube_generate(self.h, pid, order, nx, ny, nz, nparts, part, info))
        self.bs = 3 if pid == FORM_ELASTICITY else 1
        self.n_owned, self.n_ghost = int(info[2]), int(info[3])
        return info

    def pattern_build(self):
        self._ck(self.L.zzz_csr_pattern_build(self.h))

    def csr_sizes(self):
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
        self._ck(self.L.zzz_csr_sizes(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def csr_download(self, values=True):
        nrows, _, nnz = self.csr_sizes()
        rowptr = np.zeros(nrows + 1, np.int32)
        cols = np.zeros(nnz, np.int32)
        vals = np.zeros(nnz) if values else None
        self._ck(self.L.zzz_csr_download(self.h, rowptr.ctypes.data, cols.ctypes.data,
                                         vals.ctypes.data if values else None))
        return rowptr, cols, vals

    def csr_rowptr64(self):
        nrows, _, _ = self.csr_sizes()
        out = np.zeros(nrows + 1, np.int64)
        self._ck(self.L.zzz_csr_rowptr64_download(self.h, out.ctypes.data_as(C.c_void_p)))
        return out

    def csr_upload_values(self, vals):
        self._ck(self.L.zzz_csr_upload_values(self.h, np.ascontiguousarray(vals, np.float64)))

    def assemble_matrix(self, form):
        self._ck(self.L.zzz_assemble_matrix(self.h, form))

    def assemble_vector(self, form):
        self._ck(self.L.zzz_assemble_vector(self.h, form))

    def vec_download(self, which):
        out = np.zeros(self.n_owned * self.bs)
        self._ck(self.L.zzz_vec_download(self.h, which, out))
        return out

    def vec_upload(self, which, v):
        v = np.ascontiguousarray(v, np.float64)
        assert v.shape[0] == self.n_owned * self.bs
        self._ck(self.L.zzz_vec_upload(self.h, which, v))

    def vec_norm(self, which):
        out = C.c_double()
        self._ck(self.L.zzz_vec_norm(self.h, which, C.byref(out)))
        return out.value

    def spmv(self, x):
        x = np.ascontiguousarray(x, np.float64)
        y = np.zeros_like(x)
        self._ck(self.L.zzz_spmv(self.h, x, y))
        return y

    def spmv_time(self, reps=20, variant=-1):
        ms = C.c_double()
        self._ck(self.L.zzz_spmv_time(self.h, reps, variant, C.byref(ms)))
        return ms.value

    def action(self, x):
        x = np.ascontiguousarray(x, np.float64)
        y = np.zeros_like(x)
        self._ck(self.L.zzz_action(self.h, x, y))
        return y

    def near_nullspace(self):
        """build_near_nullspace (src/elasticity_problem.cpp:36-94): returns (basis [6][3 n_owned], largest deviation from
        orthonormality)"""
        dev = C.c_double()
        self._ck(self.L.zzz_near_nullspace_build(self.h, C.byref(dev)))
        B = np.zeros((6, 3 * self.n_owned))
        for k in range(6):
            self._ck(self.L.zzz_near_nullspace_download(self.h, k, B[k]))
        return B, dev.value

    def matfree_setup(self):
        self._ck(self.L.zzz_matfree_setup(self.h))

    def matfree_diagonal(self):
        """diag(A) of the matrix-free operator (1.0 on constrained rows), nothing assembled"""
        d = np.zeros(self.n_owned * self.bs)
        self._ck(self.L.zzz_matfree_diagonal(self.h, d))
        return d

    def matfree_info(self):
        a = (C.c_int64 * 8)()
        self._ck(self.L.zzz_matfree_info(self.h, a))
        return dict(zip(("valid", "blocks", "cells_per_block", "threads", "nloc_max", "shared_dofs", "partial_slots",
                         "bytes_per_action"), [int(v) for v in a]))

    def action_time(self, reps=20):
        ms = C.c_double()
        self._ck(self.L.zzz_action_time(self.h, reps, C.byref(ms)))
        return ms.value

    def cg_solve(self, variant=CG_PETSC, pc=PC_JACOBI, norm=NORM_PRECONDITIONED, op=OP_CSR, rtol=1e-8, atol=1e-50,
                 max_it=10000, profile=False, single_reduction=False, dtol=0.0, error_if_not_converged=False,
                 pc_degree=0, pc_ratio=0.0, pc_esteig_its=0):
        o = SolverOpts(variant, pc, norm, op, max_it, 1 if profile else 0, 1 if single_reduction else 0,
                       1 if error_if_not_converged else 0, rtol, atol, dtol, pc_degree, pc_esteig_its, pc_ratio)
        it = C.c_int()
        rn = (C.c_double * 2)()
        self._ck(self.L.zzz_cg_solve(self.h, C.byref(o), C.byref(it), rn))
        return it.value, rn[0], rn[1]

    def cg_history(self, n):
        out = np.zeros(n)
        self._ck(self.L.zzz_cg_history(self.h, n, out))
        return out

    def spmv_info(self):
        """(packed 16-bit columns in use, offset bits, tiles on int32 columns, tiles)"""
        info = (C.c_int64 * 8)()
        self._ck(self.L.zzz_spmv_info(self.h, info))
        return tuple(int(v) for v in info[:4])

    def spmv_operator_form(self):
        """0: CSR tile kernel, 1: sliced-ELL operator stream in natural row order, 2: ... rows sorted by length"""
        return self.spmv_info_raw()[5]

    def spmv_info_raw(self):
        info = (C.c_int64 * 8)()
        self._ck(self.L.zzz_spmv_info(self.h, info))
        return [int(v) for v in info]

    def spmv_values_info(self):
        """how the operator stream holds its values: dict(form = 'doubles' | 'dictionary in memory' | 'dictionary in LDS',
        distinct values, bytes per product in that form, bytes per product as doubles)"""
        info = (C.c_int64 * 10)()
        if os.environ.get("ZZZ_AB_OLD") and not hasattr(self.L, "zzz_spmv_values_info2"):
            return dict(form="doubles", distinct_values=0, bytes_per_product=0, bytes_per_product_as_doubles=0,
                        one_chunk_kernel=False, workgroups_per_cu=8, block_rows=False, row_windows=False, special_form="", block_table_entries=0, block_chunks=0, block_form=0)
        self._ck(self.L.zzz_spmv_values_info2(self.h, 10, info))
        return dict(form=("doubles", "dictionary in memory", "dictionary in LDS", "slice dictionaries")[int(info[0])], distinct_values=int(info[1]),
                    bytes_per_product=int(info[2]), bytes_per_product_as_doubles=int(info[3]),
                    one_chunk_kernel=bool(info[4]), workgroups_per_cu=int(info[5]), block_rows=int(info[6]) == 1, row_windows=int(info[6]) == 2,
                    special_form=("", "block rows", "block windows")[int(info[6])],
                    block_table_entries=int(info[7]), block_chunks=int(info[8]), block_form=int(info[9]))

    def spmv_x_windows(self):
        """(LDS doubles per workgroup, bytes of x loaded into LDS per product) when the operator stream carries x windows,
        else (0, 0)"""
        info = self.spmv_info_raw()
        return (int(-info[3]), int(info[2])) if info[5] and info[3] < 0 else (0, 0)

    def spmv_lanes_per_row(self):
        info = (C.c_int64 * 8)()
        self._ck(self.L.zzz_spmv_info(self.h, info))
        return int(info[4])

    def cg_fused(self):
        """did the last solve run the fused product + direction kernel (two kernels per iteration)?"""
        info = (C.c_int64 * 4)()
        self._ck(self.L.zzz_cg_info(self.h, info))
        return bool(info[0])

    def internal_order(self):
        """(perm, kind): perm[i] = caller index of the library's internal owned block dof i (identity when the caller's
        order was kept); kind 0 kept, 1 lattice order, 2 coordinate bins"""
        perm = np.zeros(self.n_owned, np.int32)
        kind = C.c_int32()
        self._ck(self.L.zzz_internal_order_download(self.h, perm.ctypes.data_as(C.c_void_p), C.byref(kind)))
        return perm, int(kind.value) & 15

    def cells_renumbered(self):
        kind = C.c_int32()
        self._ck(self.L.zzz_internal_order_download(self.h, None, C.byref(kind)))
        return bool(int(kind.value) & 16)

    def cg_reason(self):
        """KSPConvergedReason of the last solve: 2 rtol, 3 atol, -3 max_it, -4 KSP_DIVERGED_DTOL, -9 NaN/Inf"""
        info = (C.c_int64 * 4)()
        self._ck(self.L.zzz_cg_info(self.h, info))
        return int(info[2])

    def cg_info(self):
        info = (C.c_int64 * 4)()
        self._ck(self.L.zzz_cg_info(self.h, info))
        return {"fused": bool(info[0] & 1), "dinv_codes": int(info[0] >> 8) if info[0] & 2 else 0, "reason": int(info[2]),
                "pc_spectrum_bound": info[3] * 1.0e-6}

    def profile(self):
        ms, n = C.c_double(), C.c_int64()
        self._ck(self.L.zzz_profile_get(self.h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def comm_info(self):
        info = (C.c_int64 * 12)()
        self._ck(self.L.zzz_comm_info(self.h, info))
        keys = ("ranks", "rank", "neighbours", "halo_bytes_sent", "halo_bytes_received", "halo_own_communicator",
                "peer_memory_allreduce", "halo_overlapped", "interior_items", "boundary_items", "local_backend",
                "halo_wait_ns_per_product")
        return {k: int(v) for k, v in zip(keys, info)}

    def comm_init(self, nranks, rank, uid_bytes):
        buf = C.create_string_buffer(bytes(uid_bytes), 128)
        self._ck(self.L.zzz_comm_init(self.h, nranks, rank, buf))

    def comm_init_local(self, group, rank):
        self._ck(self.L.zzz_comm_init_local(self.h, group, rank))

    def comm_init_peer_only(self, nranks, rank):
        self._ck(self.L.zzz_comm_init_peer_only(self.h, nranks, rank))

    def comm_p2p_export(self):
        """this rank's mailbox handle (P2P_HANDLE_BYTES) for the peer-memory all-reduce"""
        buf = C.create_string_buffer(P2P_HANDLE_BYTES)
        self._ck(self.L.zzz_comm_p2p_export(self.h, buf))
        return bytes(buf.raw)

    def comm_p2p_attach(self, handles):
        """handles: the nranks exported handles concatenated in rank order; True if the peer-memory
        all-reduce is now in use on every rank"""
        buf = C.create_string_buffer(bytes(handles), len(handles))
        en = C.c_int(0)
        self._ck(self.L.zzz_comm_p2p_attach(self.h, buf, C.byref(en)))
        return bool(en.value)

    def comm_p2p_disable(self):
        self._ck(self.L.zzz_comm_p2p_disable(self.h))

    def comm_p2p_enable(self):
        en = C.c_int(0)
        self._ck(self.L.zzz_comm_p2p_enable(self.h, C.byref(en)))
        return bool(en.value)

    def comm_p2p_halo(self, on=True):
        """halo through the peer-memory window (True) or the communicator's send / recv (False); returns whether the
        next exchange goes through the window"""
        used = C.c_int(0)
        self._ck(self.L.zzz_comm_p2p_halo(self.h, 1 if on else 0, C.byref(used)))
        return bool(used.value)

    def upload_halo(self, P):
        nn = len(P.neigh)
        z32, z64 = np.zeros(1, np.int32), np.zeros(1, np.int64)
        self._ck(self.L.zzz_halo_upload(self.h, nn, P.neigh if nn else z32, P.send_off if nn else z64,
                                        P.send_idx if P.send_idx.size else z32, P.recv_cnt if nn else z64))


P2P_HANDLE_BYTES = 128


def comm_load():
    """bind /opt/rocm's librccl now (before anything else in the process brings its own copy)"""
    rc = hip().zzz_comm_load()
    if rc:
        raise ZzzError(rc, hip().zzz_last_error(None).decode())


def comm_library_path():
    f = hip().zzz_comm_library_path
    f.restype = C.c_char_p
    return f().decode()


def comm_unique_id():
    buf = C.create_string_buffer(128)
    rc = hip().zzz_comm_unique_id(buf)
    if rc:
        raise ZzzError(rc, hip().zzz_last_error(None).decode())
    return buf.raw


class LocalGroup:
    """zzz_local_group_create: host-mediated communicator for N contexts driven by N threads."""

    def __init__(self, nranks):
        self.h = C.c_void_p()
        rc = hip().zzz_local_group_create(int(nranks), C.byref(self.h))
        if rc:
            raise ZzzError(rc, hip().zzz_last_error(None).decode())
        self.n = nranks

    def abort(self):
        if self.h:
            hip().zzz_local_group_abort(self.h)

    def close(self):
        if self.h:
            hip().zzz_local_group_destroy(self.h)
            self.h = None

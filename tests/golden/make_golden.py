#!/usr/bin/env python3
"""Generates tests/golden/*.npz -- input/output vectors for the hot path.

The reference holds no golden vectors (CI checks the exit code only, .github/workflows/ccpp.yml:56-197)
and cannot run in this image (DOLFINx/FFCx/Basix/PETSc absent), so these vectors come from a THIRD,
independent restatement written in numpy/scipy.  It deliberately shares no code and few choices with
oracle/zzz_oracle.c or with the HIP kernels:

  * basis: Vandermonde in the homogeneous barycentric (Bernstein-monomial) basis, float64 numpy
    (oracle: Cartesian monomials in long double; HIP: exact pre-contracted reference tensors)
  * quadrature: collapsed Gauss-Jacobi from scipy.special.roots_jacobi
    (oracle: Gauss-Legendre with the Duffy Jacobian as part of the integrand)
  * mesh: Kuhn simplices generated from axis permutations, vertex-sorted cells, own cell order
  * dof numbering: first-touch order through python dicts (oracle: sorted keys)
  * elasticity: closed-form mu/lambda gradient formula (oracle: literal sigma(u):eps(v) tensors)
  * assembly: scipy.sparse COO duplicate summation; solvers: numpy loops

Each .npz is self-contained: inputs (x, cells, cell_dofs, bc, f, g, facets) and expected outputs
(rowptr, cols, vals, b, u_pcg, it_pcg, u_cg, it_cg, ...).  What the inputs mean is the boundary
documented in include/zzz_abi.h.  Spec followed: src/Poisson.py:15-39, src/Elasticity.py:11-43,
src/poisson_problem.cpp:53-157, src/elasticity_problem.cpp:119-231, src/cg.h:38-86.

Run:  python tests/golden/make_golden.py      (deterministic; rewrites the .npz files)
"""
import itertools
import os

import numpy as np
import scipy.sparse as sp
from scipy.special import roots_jacobi

HERE = os.path.dirname(os.path.abspath(__file__))

EDGE_V = [(2, 3), (1, 3), (1, 2), (0, 3), (0, 2), (0, 1)]
FACE_V = [(1, 2, 3), (0, 2, 3), (0, 1, 3), (0, 1, 2)]


# ---------------------------------------------------------------- reference element
def ref_nodes_bary(order):
    """barycentric coordinates (l0..l3) of the gll_warped Lagrange nodes"""
    E = np.eye(4)
    nodes = [E[v] for v in range(4)]
    if order == 2:
        ts = [0.5]
    elif order == 3:
        ts = [(1 - 1 / np.sqrt(5)) / 2, (1 + 1 / np.sqrt(5)) / 2]
    else:
        ts = []
    for (a, b) in EDGE_V:
        for t in ts:
            nodes.append((1 - t) * E[a] + t * E[b])
    if order == 3:
        for (a, b, c) in FACE_V:
            nodes.append((E[a] + E[b] + E[c]) / 3)
    return np.array(nodes)


def exps(order):
    return [e for e in itertools.product(range(order + 1), repeat=4) if sum(e) == order]


def basis_coeffs(order):
    P = ref_nodes_bary(order)
    ex = exps(order)
    V = np.array([[np.prod(p ** np.array(e)) for e in ex] for p in P])
    return np.linalg.inv(V).T, ex  # phi_i = sum_m C[i,m] * lambda^e_m


def eval_basis(order, lam):
    """lam: (nq,4) barycentric points -> phi (nq,nd), dphi/dlambda (nq,nd,4)"""
    Cf, ex = basis_coeffs(order)
    nq = lam.shape[0]
    mono = np.zeros((nq, len(ex)))
    dmono = np.zeros((nq, len(ex), 4))
    for m, e in enumerate(ex):
        e = np.array(e)
        mono[:, m] = np.prod(lam ** e, axis=1)
        for k in range(4):
            if e[k] > 0:
                ek = e.copy()
                ek[k] -= 1
                dmono[:, m, k] = e[k] * np.prod(lam ** ek, axis=1)
    phi = mono @ Cf.T
    dphi = np.einsum("qmk,im->qik", dmono, Cf)
    return phi, dphi


def tet_rule(deg):
    n = deg // 2 + 1
    xu, wu = roots_jacobi(n, 2, 0)
    xv, wv = roots_jacobi(n, 1, 0)
    xw, ww = roots_jacobi(n, 0, 0)
    pts, wts = [], []
    for a in range(n):
        for b in range(n):
            for c in range(n):
                u, v, w = (1 + xu[a]) / 2, (1 + xv[b]) / 2, (1 + xw[c]) / 2
                X = np.array([u, v * (1 - u), w * (1 - u) * (1 - v)])
                pts.append([1 - X.sum(), X[0], X[1], X[2]])
                wts.append(wu[a] / 8 * wv[b] / 4 * ww[c] / 2)
    return np.array(pts), np.array(wts)


def tri_rule(deg):
    n = deg // 2 + 1
    xu, wu = roots_jacobi(n, 1, 0)
    xv, wv = roots_jacobi(n, 0, 0)
    pts, wts = [], []
    for a in range(n):
        for b in range(n):
            u, v = (1 + xu[a]) / 2, (1 + xv[b]) / 2
            pts.append([1 - u - v * (1 - u), u, v * (1 - u)])
            wts.append(wu[a] / 4 * wv[b] / 2)
    return np.array(pts), np.array(wts)  # weights sum to 1/2


class Element:
    def __init__(self, order):
        self.order = order
        self.nd = len(exps(order))
        lam, self.ws = tet_rule(2 * (order - 1))
        _, self.dphi_s = eval_basis(order, lam)
        lam, self.wm = tet_rule(2 * order)
        self.phi_m, _ = eval_basis(order, lam)
        self.facet = []
        tl, tw = tri_rule(2 * order)
        for fv in FACE_V:
            lam = np.zeros((tl.shape[0], 4))
            for k in range(3):
                lam[:, fv[k]] = tl[:, k]
            phi, _ = eval_basis(order, lam)
            self.facet.append((phi, tw))

    def phys_grads(self, xc):
        """xc (4,3) -> |det J|, g (nq, nd, 3)"""
        # d lambda_k / dx: rows of inverse of [1 x y z] matrix
        M = np.hstack([np.ones((4, 1)), xc])
        Minv = np.linalg.inv(M)  # lambda_k(x) = Minv[0,k] + Minv[1:,k].x
        dl = Minv[1:, :].T  # (4,3)
        adet = abs(np.linalg.det(xc[1:] - xc[0]))
        return adet, self.dphi_s @ dl

    def poisson_a(self, xc):
        adet, g = self.phys_grads(xc)
        return adet * np.einsum("q,qia,qja->ij", self.ws, g, g)

    def elasticity_a(self, xc):
        Ey, nu = 1.0e6, 0.3
        mu = Ey / (2.0 * (1.0 + nu))
        lm = Ey * nu / ((1.0 + nu) * (1.0 - 2.0 * nu))
        adet, g = self.phys_grads(xc)
        gg = np.einsum("q,qia,qja->ij", self.ws, g, g)
        G = np.einsum("q,qic,qjd->icjd", self.ws, g, g)  # int d_c phi_i d_d phi_j
        A = mu * (np.einsum("ij,cd->icjd", gg, np.eye(3)) + np.einsum("idjc->icjd", G)) + lm * G
        n = 3 * self.nd
        return adet * A.reshape(n, n)

    def mass(self, xc):
        adet = abs(np.linalg.det(xc[1:] - xc[0]))
        return adet * np.einsum("q,qi,qj->ij", self.wm, self.phi_m, self.phi_m)

    def facet_mass(self, xc, lf):
        fv = FACE_V[lf]
        scale = np.linalg.norm(np.cross(xc[fv[1]] - xc[fv[0]], xc[fv[2]] - xc[fv[0]]))
        phi, w = self.facet[lf]
        return scale * np.einsum("q,qi,qj->ij", w, phi, phi)


# ---------------------------------------------------------------- mesh + dofmap
def kuhn_mesh(nx, ny, nz):
    px, py = nx + 1, ny + 1
    vid = lambda ix, iy, iz: (iz * py + iy) * px + ix
    x = np.zeros(((nx + 1) * (ny + 1) * (nz + 1), 3))
    for iz in range(nz + 1):
        for iy in range(ny + 1):
            for ix in range(nx + 1):
                x[vid(ix, iy, iz)] = (ix / nx, iy / ny, iz / nz)
    cells = []
    # different traversal order from the oracle on purpose: x slowest
    for ix in range(nx):
        for iy in range(ny):
            for iz in range(nz):
                for perm in itertools.permutations(range(3)):
                    p = [ix, iy, iz]
                    tet = [vid(*p)]
                    for a in perm:
                        p[a] += 1
                        tet.append(vid(*p))
                    cells.append(tet)  # ascending vertex ids along the Kuhn path
    return x, np.array(cells, np.int32)


def build_dofmap(order, x, cells):
    nv = x.shape[0]
    nd = len(exps(order))
    npe = order - 1
    dof_of = {}
    coords = [tuple(p) for p in x]

    def get(key, pos):
        if key not in dof_of:
            dof_of[key] = len(coords)
            coords.append(tuple(pos))
        return dof_of[key]

    ts = {2: [0.5], 3: [(1 - 1 / np.sqrt(5)) / 2, (1 + 1 / np.sqrt(5)) / 2]}.get(order, [])
    cd = np.zeros((cells.shape[0], nd), np.int32)
    for c, cv in enumerate(cells):
        row = list(cv)
        if order >= 2:
            for (a, b) in EDGE_V:
                ga, gb = int(cv[a]), int(cv[b])
                lo, hi = min(ga, gb), max(ga, gb)
                for s in range(npe):
                    gs = s if ga < gb else npe - 1 - s  # global sub-dof index counted from the low vertex
                    pos = (1 - ts[gs]) * x[lo] + ts[gs] * x[hi]
                    row.append(get(("e", lo, hi, gs), pos))
        if order == 3:
            for fv in FACE_V:
                tri = tuple(sorted(int(cv[k]) for k in fv))
                row.append(get(("f",) + tri, x[list(tri)].mean(axis=0)))
        cd[c] = row
    return len(coords), cd, np.array(coords)


def exterior_facets(cells):
    seen = {}
    for c, cv in enumerate(cells):
        for lf, fv in enumerate(FACE_V):
            key = tuple(sorted(int(cv[k]) for k in fv))
            seen.setdefault(key, []).append((c, lf))
    out = sorted(v[0] for v in seen.values() if len(v) == 1)
    return np.array(out, np.int32).reshape(-1, 2)


# ---------------------------------------------------------------- problems
def make_case(problem, order, dims):
    el = Element(order)
    nd = el.nd
    x, cells = kuhn_mesh(*dims)
    nblock, cell_dofs, dof_x = build_dofmap(order, x, cells)
    bs = 3 if problem == "elasticity" else 1
    n = nblock * bs
    eps = 1e-8
    if problem == "poisson":
        bcb = (np.abs(dof_x[:, 0]) < eps) | (np.abs(dof_x[:, 0] - 1) < eps)
        f = 10 * np.exp(-((dof_x[:, 0] - 0.5) ** 2 + (dof_x[:, 1] - 0.5) ** 2) / 0.02)
        g = np.sin(5 * dof_x[:, 0])
        facets = exterior_facets(cells)
    else:
        bcb = np.abs(dof_x[:, 1]) < eps
        dx, dz = dof_x[:, 0] - 0.5, dof_x[:, 2] - 0.5
        r = np.sqrt(dx * dx + dz * dz)
        f = np.stack([-dz * r * dof_x[:, 1], np.ones(nblock), dx * r * dof_x[:, 1]], axis=1).reshape(-1)
        g = np.zeros(0)
        facets = np.zeros((0, 2), np.int32)
    bc = np.repeat(bcb, bs)

    rows, cols_, vals_ = [], [], []
    b = np.zeros(n)
    for c in range(cells.shape[0]):
        xc = x[cells[c]]
        dofs = (cell_dofs[c][:, None] * bs + np.arange(bs)[None, :]).reshape(-1)
        Ae = el.poisson_a(xc) if problem == "poisson" else el.elasticity_a(xc)
        m = bc[dofs]
        Ae[m, :] = 0.0
        Ae[:, m] = 0.0
        rows.append(np.repeat(dofs, dofs.size))
        cols_.append(np.tile(dofs, dofs.size))
        vals_.append(Ae.reshape(-1))
        Mc = el.mass(xc)
        if problem == "poisson":
            b[dofs] += Mc @ f[cell_dofs[c]]
        else:
            b[dofs] += (Mc @ f.reshape(-1, 3)[cell_dofs[c]]).reshape(-1)
    for (c, lf) in facets:
        xc = x[cells[c]]
        b[cell_dofs[c]] += el.facet_mass(xc, lf) @ g[cell_dofs[c]]
    b[bc] = 0.0
    A = sp.coo_matrix((np.concatenate(vals_), (np.concatenate(rows), np.concatenate(cols_))), shape=(n, n)).tocsr()
    A.sort_indices()
    A = A.tolil(copy=True).tocsr() if False else A
    # set_diagonal: 1.0 on constrained diagonals (entries exist in the pattern, currently 0)
    A = A.copy()
    for r in np.nonzero(bc)[0]:
        lo, hi = A.indptr[r], A.indptr[r + 1]
        k = lo + np.searchsorted(A.indices[lo:hi], r)
        assert A.indices[k] == r
        A.data[k] = 1.0

    # ---- solvers
    def spmv(v):
        return A @ v

    # cg.h
    def cg(kmax, rtol):
        xk = np.zeros(n)
        y = spmv(xk)
        r = -1.0 * y + b
        p = r.copy()
        rn0 = float(r @ r)
        rn = rn0
        k = 0
        while k < kmax:
            k += 1
            y = spmv(p)
            alpha = rn / float(p @ y)
            xk = alpha * p + xk
            r = -alpha * y + r
            rn_new = float(r @ r)
            beta = rn_new / rn
            rn = rn_new
            if rn / rn0 < rtol * rtol:
                break
            p = beta * p + r
        return k, xk

    # PETSc-style Jacobi PCG, preconditioned norm
    def pcg(rtol, max_it):
        dinv = 1.0 / A.diagonal()
        xk = np.zeros(n)
        r = b.copy()
        z = dinv * r
        beta = float(r @ z)
        dp0 = dp = float(np.sqrt(z @ z))
        ttol = max(rtol * dp0, 1e-50)
        it = 0
        p = None
        betaold = 1.0
        if dp <= ttol:
            return 0, xk
        while it < max_it:
            p = z.copy() if it == 0 else (beta / betaold) * p + z
            w = spmv(p)
            a = beta / float(p @ w)
            xk = a * p + xk
            r = -a * w + r
            z = dinv * r
            betaold = beta
            beta = float(r @ z)
            dp = float(np.sqrt(z @ z))
            it += 1
            if dp <= ttol:
                break
        return it, xk

    it_cg, u_cg = cg(2000, 1e-8)
    it_cg6, u_cg6 = cg(100, 1e-6)  # the reference's only call: src/cgpoisson_problem.cpp:233
    it_pcg, u_pcg = pcg(1e-8, 10000)
    return dict(
        problem=problem, order=order, dims=np.array(dims), bs=bs, nblock=nblock,
        x=x, cells=cells, cell_dofs=cell_dofs, dof_x=dof_x, bc=bc.astype(np.uint8), f=f, g=g, facets=facets,
        rowptr=A.indptr.astype(np.int64), cols=A.indices.astype(np.int32), vals=A.data, b=b,
        it_cg=it_cg, u_cg=u_cg, it_cg6=it_cg6, u_cg6=u_cg6, it_pcg=it_pcg, u_pcg=u_pcg,
    )


CASES = [
    ("poisson", 1, (3, 2, 4)),
    ("poisson", 2, (3, 2, 2)),
    ("poisson", 3, (2, 2, 3)),
    ("elasticity", 1, (3, 2, 4)),
    ("elasticity", 2, (2, 2, 2)),
    ("elasticity", 3, (1, 2, 2)),
]


def main():
    for problem, order, dims in CASES:
        d = make_case(problem, order, dims)
        name = f"{problem}_p{order}_{dims[0]}x{dims[1]}x{dims[2]}.npz"
        np.savez_compressed(os.path.join(HERE, name), **d)
        print(name, "n =", d["nblock"] * d["bs"], "nnz =", d["cols"].shape[0], "it_pcg =", d["it_pcg"], "it_cg =",
              d["it_cg"], "|u| =", np.linalg.norm(d["u_pcg"]))

    # element-level known answers (analytic, sympy-checkable): reference tetrahedron
    ref = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]], float)
    rng = np.random.default_rng(20261003)
    tet = rng.random((4, 3))
    out = {"ref": ref, "tet": tet}
    for order in (1, 2, 3):
        el = Element(order)
        for nm, xc in (("ref", ref), ("tet", tet)):
            out[f"poisson_a_p{order}_{nm}"] = el.poisson_a(xc)
            out[f"elasticity_a_p{order}_{nm}"] = el.elasticity_a(xc)
            out[f"mass_p{order}_{nm}"] = el.mass(xc)
            for lf in range(4):
                out[f"facet_mass{lf}_p{order}_{nm}"] = el.facet_mass(xc, lf)
    np.savez_compressed(os.path.join(HERE, "element_tensors.npz"), **out)
    print("element_tensors.npz")


if __name__ == "__main__":
    main()

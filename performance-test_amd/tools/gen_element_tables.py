#!/usr/bin/env python3
"""Generates performance-test_amd/csrc/element_tables.inc: the reference tensors the HIP assembly
kernels contract with per-cell geometry (what FFCx bakes into tabulate_tensor for affine cells).

  S[a][b][i][j] = int_K^ d_a phi_i d_b phi_j dX          (stiffness pieces, a,b in 0..2)
  M[i][j]       = int_K^ phi_i phi_j dX                  (mass)
  F[f][i][j]    = int_T^ phi_i(X_f(s,t)) phi_j(X_f(s,t)) ds dt   (facet mass on the unit triangle)

for the Lagrange P1..P3 gll_warped element on the reference tetrahedron
(src/poisson_problem.cpp:35-38, src/Poisson.py:16, src/Elasticity.py:23) in Basix's local dof order
(see include/zzz_abi.h).  Derivation: Vandermonde inverse in Cartesian monomials and EXACT monomial
integration (a!b!c!/(a+b+c+3)!), all in 50-digit mpmath arithmetic, printed with 17 significant
digits.  This is independent of oracle/zzz_oracle.c (Gauss quadrature, long double) and of
tests/golden/make_golden.py (barycentric basis, Gauss-Jacobi): the parity tests compare all three.

Run: python performance-test_amd/tools/gen_element_tables.py   (rewrites the .inc; deterministic)
"""
import itertools
import os
from math import factorial

import mpmath as mp

mp.mp.dps = 50

EDGE_V = [(2, 3), (1, 3), (1, 2), (0, 3), (0, 2), (0, 1)]
FACE_V = [(1, 2, 3), (0, 2, 3), (0, 1, 3), (0, 1, 2)]
REFV = [(0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1)]


def nodes(order):
    V = [tuple(mp.mpf(c) for c in v) for v in REFV]
    pts = list(V)
    if order == 2:
        ts = [mp.mpf(1) / 2]
    elif order == 3:
        ts = [(1 - 1 / mp.sqrt(5)) / 2, (1 + 1 / mp.sqrt(5)) / 2]
    else:
        ts = []
    for a, b in EDGE_V:
        for t in ts:
            pts.append(tuple((1 - t) * V[a][k] + t * V[b][k] for k in range(3)))
    if order == 3:
        for a, b, c in FACE_V:
            pts.append(tuple((V[a][k] + V[b][k] + V[c][k]) / 3 for k in range(3)))
    return pts


def monos(order):
    return [e for d in range(order + 1) for e in itertools.product(range(d + 1), repeat=3) if sum(e) == d]


def poly_mul(p, q):
    r = {}
    for e1, c1 in p.items():
        for e2, c2 in q.items():
            e = tuple(a + b for a, b in zip(e1, e2))
            r[e] = r.get(e, 0) + c1 * c2
    return r


def poly_diff(p, k):
    r = {}
    for e, c in p.items():
        if e[k] > 0:
            e2 = list(e)
            e2[k] -= 1
            r[tuple(e2)] = r.get(tuple(e2), 0) + c * e[k]
    return r


def int_tet(p):
    return sum(c * mp.mpf(factorial(e[0]) * factorial(e[1]) * factorial(e[2])) / factorial(sum(e) + 3) for e, c in p.items())


def int_tri(p):  # polynomial in (s, t): exponents (p, q)
    return sum(c * mp.mpf(factorial(e[0]) * factorial(e[1])) / factorial(sum(e) + 2) for e, c in p.items())


def poly_pow(p, n):
    r = {tuple([0] * len(next(iter(p)))): mp.mpf(1)}
    for _ in range(n):
        r = poly_mul(r, p)
    return r


def restrict_to_facet(p, lf):
    """substitute X = q0 + s (q1-q0) + t (q2-q0) into a polynomial in (x,y,z) -> polynomial in (s,t)"""
    q0, q1, q2 = (REFV[v] for v in FACE_V[lf])
    coord = []
    for k in range(3):
        c = {}
        if q0[k] != 0:
            c[(0, 0)] = mp.mpf(q0[k])
        if q1[k] - q0[k] != 0:
            c[(1, 0)] = mp.mpf(q1[k] - q0[k])
        if q2[k] - q0[k] != 0:
            c[(0, 1)] = mp.mpf(q2[k] - q0[k])
        if not c:
            c[(0, 0)] = mp.mpf(0)
        coord.append(c)
    out = {}
    for e, cf in p.items():
        term = {(0, 0): cf}
        for k in range(3):
            if e[k]:
                term = poly_mul(term, poly_pow(coord[k], e[k]))
        for ee, cc in term.items():
            out[ee] = out.get(ee, 0) + cc
    return out


def tables(order):
    P = nodes(order)
    ex = monos(order)
    nd = len(P)
    assert nd == len(ex)
    V = mp.matrix(nd, nd)
    for j, p in enumerate(P):
        for m, e in enumerate(ex):
            V[j, m] = p[0] ** e[0] * p[1] ** e[1] * p[2] ** e[2]
    Vi = V ** -1  # Vi[m, i] = coefficient of monomial m in phi_i
    phi = [{ex[m]: Vi[m, i] for m in range(nd)} for i in range(nd)]
    dphi = [[poly_diff(phi[i], a) for a in range(3)] for i in range(nd)]
    S = [[[[int_tet(poly_mul(dphi[i][a], dphi[j][b])) for j in range(nd)] for i in range(nd)] for b in range(3)]
         for a in range(3)]
    M = [[int_tet(poly_mul(phi[i], phi[j])) for j in range(nd)] for i in range(nd)]
    F = []
    for lf in range(4):
        r = [restrict_to_facet(phi[i], lf) for i in range(nd)]
        F.append([[int_tri(poly_mul(r[i], r[j])) for j in range(nd)] for i in range(nd)])
    return nd, S, M, F, dtables(order, nd, dphi, S)


def dtables(order, nd, dphi, S):
    """Factorised stiffness for the matrix-free action (csrc/zzz_matfree.hip): d_a phi_j lies in P_(k-1); with an
    L2-orthonormal basis psi_q of P_(k-1) on the reference tetrahedron (Gram-Schmidt on the monomials, exact integrals)
        D[a][q][j] = int_K^ psi_q d_a phi_j dX      =>      S[a][b][i][j] = sum_q D[a][q][i] D[b][q][j],
    so Ae u = |detJ| sum_a D_a^T ( sum_b (K K^T)_ab D_b u ): 2 x 3 nq nd + 9 nq multiply-adds per cell instead of 6 nd^2."""
    ex = monos(order - 1)
    nq = len(ex)
    psi = []
    for e in ex:
        p = {e: mp.mpf(1)}
        for q in psi:
            c = int_tet(poly_mul(p, q))
            for ee, cc in q.items():
                p[ee] = p.get(ee, 0) - c * cc
        nrm = mp.sqrt(int_tet(poly_mul(p, p)))
        psi.append({ee: cc / nrm for ee, cc in p.items()})
    D = [[[int_tet(poly_mul(psi[q], dphi[j][a])) if dphi[j][a] else mp.mpf(0) for j in range(nd)] for q in range(nq)]
         for a in range(3)]
    for a in range(3):
        for b in range(3):
            for i in range(nd):
                for j in range(nd):
                    assert abs(sum(D[a][q][i] * D[b][q][j] for q in range(nq)) - S[a][b][i][j]) < mp.mpf(10) ** -40
    return nq, D


def fmt(v):
    if abs(v) < mp.mpf(10) ** -30:
        return "0.0"
    return mp.nstr(v, 17, min_fixed=0, max_fixed=0)


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    out = os.path.join(os.path.dirname(here), "csrc", "element_tables.inc")
    L = ["// GENERATED by performance-test_amd/tools/gen_element_tables.py -- do not edit.",
         "// Reference tensors of the Lagrange P1..P3 gll_warped tetrahedron (exact integration, 17 digits).",
         "// Layout per order k: S[9][nd][nd] (a*3+b major) | M[nd][nd] | F[4][nd][nd], doubles.", ""]
    for order in (1, 2, 3):
        nd, S, M, F, (nq, D) = tables(order)
        # sanity: P1 closed forms
        if order == 1:
            assert abs(M[0][0] - mp.mpf(1) / 60) < 1e-40 and abs(S[0][0][1][1] - mp.mpf(1) / 6) < 1e-40
            assert abs(F[3][0][0] - mp.mpf(1) / 12) < 1e-40 and abs(F[3][3][3]) < 1e-40
        flat = []
        for a in range(3):
            for b in range(3):
                for i in range(nd):
                    flat += S[a][b][i]
        for i in range(nd):
            flat += M[i]
        for lf in range(4):
            for i in range(nd):
                flat += F[lf][i]
        L.append(f"static const double ZZZ_TAB_P{order}[{len(flat)}] = {{")
        for k in range(0, len(flat), 6):
            L.append("    " + ", ".join(fmt(v) for v in flat[k:k + 6]) + ",")
        L.append("};")
        L.append("")
        dflat = [D[a][q][j] for a in range(3) for q in range(nq) for j in range(nd)]
        L.append(f"// D[3][{nq}][{nd}]: factorised stiffness, S[a][b][i][j] = sum_q D[a][q][i] D[b][q][j] (see dtables())")
        L.append(f"static constexpr double ZZZ_DTAB_P{order}[{len(dflat)}] = {{")
        for k in range(0, len(dflat), 6):
            L.append("    " + ", ".join(fmt(v) for v in dflat[k:k + 6]) + ",")
        L.append("};")
        L.append("")
    open(out, "w").write("\n".join(L))
    print("wrote", out)


if __name__ == "__main__":
    main()

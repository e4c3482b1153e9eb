/*
 * oracle/zzz_oracle.c -- CPU restatement of the FEniCS/performance-test hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (performance-test_amd/, include/,
 * the dolfinx-scaling-test driver) may include, link or call this file.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker.
 *
 * PARITY UNPINNED: the reference ships no golden vectors, no known-answer tests and no
 * fixtures for this path (CI checks the exit status only, .github/workflows/ccpp.yml:56-197)
 * and its arithmetic lives in un-vendored, unpinned dependencies (DOLFINx main, FFCx main,
 * Basix main, PETSc) that are absent from this image, so the reference cannot be run here.
 * This file restates the published algorithms of those dependencies at the reference's own
 * call sites; it is pinned only by mathematics (analytic element matrices, patch tests,
 * manufactured-solution convergence rates, entity-count formulas of src/mesh.cpp:44-74) and
 * by an independent numpy/scipy cross-implementation (tests/golden/make_golden.py).
 *
 * What each part follows (paths relative to /root/reference):
 *   mesh size search ............ src/mesh.cpp:44-151
 *   unit-cube tetrahedral mesh .. src/mesh.cpp:184-186 (dolfinx::mesh::create_box, tetrahedron) [EXT]
 *   Lagrange gll_warped P1..P3 .. src/poisson_problem.cpp:35-38, src/elasticity_problem.cpp:103-109 (Basix) [EXT]
 *   Poisson forms a, L, M ....... src/Poisson.py:15-39 (what FFCx tabulate_tensor evaluates) [EXT]
 *   Elasticity forms a, L ....... src/Elasticity.py:11-43
 *   matrix assembly + BCs ....... src/poisson_problem.cpp:125-139, src/elasticity_problem.cpp:199-213
 *   vector assembly + BCs ....... src/poisson_problem.cpp:146-157, src/elasticity_problem.cpp:220-231
 *   boundary conditions ......... src/poisson_problem.cpp:53-77, src/elasticity_problem.cpp:119-145
 *   coefficients f, g ........... src/poisson_problem.cpp:83-106, src/elasticity_problem.cpp:153-176
 *   CG .......................... src/cg.h:18-25 (axpy), src/cg.h:38-86 (cg)
 *   Jacobi-preconditioned CG .... PETSc KSPCG + PCJACOBI as selected at src/poisson_problem.cpp:168-177 [EXT]
 *
 * All arithmetic is IEEE double (T = PetscScalar = double, src/poisson_problem.cpp:27).
 * Element integrals are evaluated literally by Gauss quadrature of sufficient degree, the way
 * a form compiler would, not by the pre-contracted reference tensors the HIP kernels use.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef int32_t i32;
typedef int64_t i64;

#define ZO_MAXND 20
#define ZO_MAXQ 256

/* ------------------------------------------------------------------------------------------
 * 1. Mesh size search -- src/mesh.cpp:44-151
 * ---------------------------------------------------------------------------------------- */

/* src/mesh.cpp:44-54 */
static void num_entities(i64 i, i64 j, i64 k, int nrefine, i64 out[4])
{
  i <<= nrefine;
  j <<= nrefine;
  k <<= nrefine;
  out[0] = (i + 1) * (j + 1) * (k + 1);
  out[1] = 7 * i * j * k + 3 * (i * j + i * k + j * k) + (i + j + k);
  out[2] = 12 * i * j * k + 2 * (i * j + i * k + j * k);
  out[3] = 6 * (i * j * k);
}

void zo_num_entities(i64 i, i64 j, i64 k, int nrefine, i64 out[4]) { num_entities(i, j, k, nrefine, out); }

/* src/mesh.cpp:56-74; returns -1 where the reference throws "Order not supported" */
i64 zo_num_pdofs(i64 i, i64 j, i64 k, int nrefine, int order)
{
  i64 e[4];
  num_entities(i, j, k, nrefine, e);
  switch (order)
  {
  case 1: return e[0];
  case 2: return e[0] + e[1];
  case 3: return e[0] + 2 * e[1] + e[2];
  case 4: return e[0] + 3 * e[1] + 3 * e[2] + e[3];
  default: return -1;
  }
}

/* src/mesh.cpp:78-151: out = {Nx, Ny, Nz, r}.  target_total != 0 <=> strong scaling. */
void zo_mesh_size(i64 target_dofs, int target_total, i64 num_processes, i64 dofs_per_node, int order,
                  i64 out[4])
{
  i64 N = target_total ? target_dofs / dofs_per_node : target_dofs * num_processes / dofs_per_node;
  i64 Nx, Ny, Nz;
  int r = 0;
  const i64 Nx_max = 200;
  Nx = 1;
  i64 ndofs = 0;
  while (ndofs < N)
  {
    ++Nx;
    if (Nx > Nx_max)
    {
      while (ndofs < N)
      {
        ++r;
        ndofs = zo_num_pdofs(Nx, Nx, Nx, r, order);
      }
      while (ndofs > N)
      {
        --Nx;
        ndofs = zo_num_pdofs(Nx, Nx, Nx, r, order);
      }
    }
    ndofs = zo_num_pdofs(Nx, Nx, Nx, r, order);
  }
  Ny = Nx;
  Nz = Nx;
  uint64_t mindiff = 1000000;
  /* src/mesh.cpp:135 literally: the start Nx - 10 is evaluated once, the bound `i < Nx + 10` on the LIVE Nx that
   * the body overwrites -- the search ends 10 past the best i found so far, not 10 past the cubic guess */
  for (i64 i = Nx - 10; i < Nx + 10; ++i)
    for (i64 j = i - 5; j < i + 5; ++j)
      for (i64 k = i - 5; k < i + 5; ++k)
      {
        i64 d = zo_num_pdofs(i, j, k, r, order) - N;
        uint64_t diff = (uint64_t)(d < 0 ? -d : d);
        if (diff < mindiff)
        {
          mindiff = diff;
          Nx = i;
          Ny = j;
          Nz = k;
        }
      }
  out[0] = Nx;
  out[1] = Ny;
  out[2] = Nz;
  out[3] = r;
}

/* ------------------------------------------------------------------------------------------
 * 2. Unit-cube mesh: nx*ny*nz sub-cubes, 6 tetrahedra each (create_box, src/mesh.cpp:184-186)
 *    vertex id = iz*(ny+1)*(nx+1) + iy*(nx+1) + ix, x = ix/nx ...; cells are NOT vertex-sorted.
 * ---------------------------------------------------------------------------------------- */
void zo_box_mesh(i64 nx, i64 ny, i64 nz, double* x, i32* cells)
{
  const i64 px = nx + 1, py = ny + 1;
  for (i64 iz = 0; iz <= nz; ++iz)
    for (i64 iy = 0; iy <= ny; ++iy)
      for (i64 ix = 0; ix <= nx; ++ix)
      {
        i64 v = (iz * py + iy) * px + ix;
        x[3 * v + 0] = (double)ix / (double)nx;
        x[3 * v + 1] = (double)iy / (double)ny;
        x[3 * v + 2] = (double)iz / (double)nz;
      }
  i64 c = 0;
  for (i64 iz = 0; iz < nz; ++iz)
    for (i64 iy = 0; iy < ny; ++iy)
      for (i64 ix = 0; ix < nx; ++ix)
      {
        i32 v0 = (i32)((iz * py + iy) * px + ix);
        i32 v1 = v0 + 1, v2 = v0 + (i32)px, v3 = v1 + (i32)px;
        i32 v4 = v0 + (i32)(px * py), v5 = v1 + (i32)(px * py), v6 = v2 + (i32)(px * py),
            v7 = v3 + (i32)(px * py);
        const i32 t[6][4] = {{v0, v1, v3, v7}, {v0, v1, v7, v5}, {v0, v5, v7, v4},
                             {v0, v3, v2, v7}, {v0, v6, v4, v7}, {v0, v2, v6, v7}};
        for (int k = 0; k < 6; ++k, ++c)
          for (int a = 0; a < 4; ++a)
            cells[4 * c + a] = t[k][a];
      }
}

/* ------------------------------------------------------------------------------------------
 * 3. Reference element: Lagrange P1..P3 on the tetrahedron, gll_warped variant.
 *    Reference vertices (0,0,0),(1,0,0),(0,1,0),(0,0,1).  Local dof order (Basix) [EXT]:
 *    vertices 0-3; edges e0=(2,3) e1=(1,3) e2=(1,2) e3=(0,3) e4=(0,2) e5=(0,1);
 *    faces f0=(1,2,3) f1=(0,2,3) f2=(0,1,3) f3=(0,1,2).
 *    P3 edge points sit at the interior Gauss-Lobatto-Legendre abscissae (1 -+ 1/sqrt5)/2,
 *    ordered from the edge's first to its second local vertex; the face point is the centroid.
 * ---------------------------------------------------------------------------------------- */
static const int EDGE_V[6][2] = {{2, 3}, {1, 3}, {1, 2}, {0, 3}, {0, 2}, {0, 1}};
static const int FACE_V[4][3] = {{1, 2, 3}, {0, 2, 3}, {0, 1, 3}, {0, 1, 2}};
static const double REFV[4][3] = {{0, 0, 0}, {1, 0, 0}, {0, 1, 0}, {0, 0, 1}};

int zo_ndofs_cell(int order) { return order == 1 ? 4 : order == 2 ? 10 : order == 3 ? 20 : -1; }

static void ref_nodes(int order, double X[ZO_MAXND][3])
{
  int n = 0;
  for (int v = 0; v < 4; ++v, ++n)
    for (int a = 0; a < 3; ++a)
      X[n][a] = REFV[v][a];
  if (order == 1)
    return;
  const int npe = order - 1;
  double t[2];
  if (order == 2)
    t[0] = 0.5;
  else
  {
    t[0] = 0.5 * (1.0 - 1.0 / sqrt(5.0));
    t[1] = 0.5 * (1.0 + 1.0 / sqrt(5.0));
  }
  for (int e = 0; e < 6; ++e)
    for (int s = 0; s < npe; ++s, ++n)
      for (int a = 0; a < 3; ++a)
        X[n][a] = (1.0 - t[s]) * REFV[EDGE_V[e][0]][a] + t[s] * REFV[EDGE_V[e][1]][a];
  if (order == 3)
    for (int f = 0; f < 4; ++f, ++n)
      for (int a = 0; a < 3; ++a)
        X[n][a] = (REFV[FACE_V[f][0]][a] + REFV[FACE_V[f][1]][a] + REFV[FACE_V[f][2]][a]) / 3.0;
}

typedef struct
{
  int order, nd, nmono;
  int mono[ZO_MAXND][3];
  long double C[ZO_MAXND][ZO_MAXND]; /* phi_i = sum_m C[i][m] * mono_m */
} basis_t;

static long double ipowl(long double x, int p)
{
  long double r = 1.0L;
  for (int i = 0; i < p; ++i)
    r *= x;
  return r;
}

static void basis_init(basis_t* B, int order)
{
  B->order = order;
  B->nd = zo_ndofs_cell(order);
  int m = 0;
  for (int d = 0; d <= order; ++d)
    for (int a = d; a >= 0; --a)
      for (int b = d - a; b >= 0; --b, ++m)
      {
        B->mono[m][0] = a;
        B->mono[m][1] = b;
        B->mono[m][2] = d - a - b;
      }
  B->nmono = m; /* == nd */
  double X[ZO_MAXND][3];
  ref_nodes(order, X);
  /* recompute the P3 nodes in long double so the Vandermonde inverse is accurate to ~1e-18 */
  long double XL[ZO_MAXND][3];
  for (int i = 0; i < B->nd; ++i)
    for (int a = 0; a < 3; ++a)
      XL[i][a] = X[i][a];
  if (order == 3)
  {
    long double t[2] = {0.5L * (1.0L - 1.0L / sqrtl(5.0L)), 0.5L * (1.0L + 1.0L / sqrtl(5.0L))};
    int n = 4;
    for (int e = 0; e < 6; ++e)
      for (int s = 0; s < 2; ++s, ++n)
        for (int a = 0; a < 3; ++a)
          XL[n][a] = (1.0L - t[s]) * REFV[EDGE_V[e][0]][a] + t[s] * REFV[EDGE_V[e][1]][a];
    for (int f = 0; f < 4; ++f, ++n)
      for (int a = 0; a < 3; ++a)
        XL[n][a] = ((long double)REFV[FACE_V[f][0]][a] + REFV[FACE_V[f][1]][a] + REFV[FACE_V[f][2]][a]) / 3.0L;
  }
  const int n = B->nd;
  long double V[ZO_MAXND][2 * ZO_MAXND];
  for (int j = 0; j < n; ++j)
    for (int k = 0; k < n; ++k)
    {
      V[j][k] = ipowl(XL[j][0], B->mono[k][0]) * ipowl(XL[j][1], B->mono[k][1]) * ipowl(XL[j][2], B->mono[k][2]);
      V[j][n + k] = (j == k) ? 1.0L : 0.0L;
    }
  /* Gauss-Jordan with partial pivoting: [V | I] -> [I | V^-1] */
  for (int c = 0; c < n; ++c)
  {
    int p = c;
    for (int r = c + 1; r < n; ++r)
      if (fabsl(V[r][c]) > fabsl(V[p][c]))
        p = r;
    if (p != c)
      for (int k = 0; k < 2 * n; ++k)
      {
        long double tmp = V[c][k];
        V[c][k] = V[p][k];
        V[p][k] = tmp;
      }
    long double inv = 1.0L / V[c][c];
    for (int k = 0; k < 2 * n; ++k)
      V[c][k] *= inv;
    for (int r = 0; r < n; ++r)
      if (r != c)
      {
        long double f = V[r][c];
        if (f != 0.0L)
          for (int k = 0; k < 2 * n; ++k)
            V[r][k] -= f * V[c][k];
      }
  }
  /* V^-1[m][i] : coefficient of monomial m in phi_i */
  for (int i = 0; i < n; ++i)
    for (int k = 0; k < n; ++k)
      B->C[i][k] = V[k][n + i];
}

/* phi[i], dphi[i][a] at reference point X */
static void basis_eval(const basis_t* B, const long double X[3], double* phi, double (*dphi)[3])
{
  for (int i = 0; i < B->nd; ++i)
  {
    long double v = 0, g0 = 0, g1 = 0, g2 = 0;
    for (int m = 0; m < B->nmono; ++m)
    {
      const int a = B->mono[m][0], b = B->mono[m][1], c = B->mono[m][2];
      const long double cf = B->C[i][m];
      v += cf * ipowl(X[0], a) * ipowl(X[1], b) * ipowl(X[2], c);
      if (a > 0)
        g0 += cf * a * ipowl(X[0], a - 1) * ipowl(X[1], b) * ipowl(X[2], c);
      if (b > 0)
        g1 += cf * b * ipowl(X[0], a) * ipowl(X[1], b - 1) * ipowl(X[2], c);
      if (c > 0)
        g2 += cf * c * ipowl(X[0], a) * ipowl(X[1], b) * ipowl(X[2], c - 1);
    }
    phi[i] = (double)v;
    if (dphi)
    {
      dphi[i][0] = (double)g0;
      dphi[i][1] = (double)g1;
      dphi[i][2] = (double)g2;
    }
  }
}

/* Gauss-Legendre on [0,1], n points (Newton on P_n) */
static void gauss_legendre01(int n, long double* x, long double* w)
{
  const long double PI = 3.14159265358979323846264338327950288L;
  for (int i = 0; i < n; ++i)
  {
    long double z = cosl(PI * (i + 0.75L) / (n + 0.5L));
    long double pp = 1;
    for (int it = 0; it < 100; ++it)
    {
      long double p1 = 1.0L, p2 = 0.0L;
      for (int j = 1; j <= n; ++j)
      {
        long double p3 = p2;
        p2 = p1;
        p1 = ((2.0L * j - 1.0L) * z * p2 - (j - 1.0L) * p3) / j;
      }
      pp = n * (z * p1 - p2) / (z * z - 1.0L);
      long double dz = p1 / pp;
      z -= dz;
      if (fabsl(dz) < 1e-19L)
        break;
    }
    x[i] = 0.5L * (1.0L - z);
    w[i] = 1.0L / ((1.0L - z * z) * pp * pp); /* = 0.5 * 2/((1-z^2) pp^2) */
  }
}

typedef struct
{
  int nq;
  double w[ZO_MAXQ];
  double phi[ZO_MAXQ][ZO_MAXND];
  double dphi[ZO_MAXQ][ZO_MAXND][3];
} qtab_t;

/* quadrature exact to total degree deg on the reference tetrahedron (collapsed Gauss-Legendre) */
static void tet_table(const basis_t* B, int deg, qtab_t* T)
{
  int nu = (deg + 4) / 2, nv = (deg + 3) / 2, nw = (deg + 2) / 2;
  long double xu[16], wu[16], xv[16], wv[16], xw[16], ww[16];
  gauss_legendre01(nu, xu, wu);
  gauss_legendre01(nv, xv, wv);
  gauss_legendre01(nw, xw, ww);
  int q = 0;
  for (int a = 0; a < nu; ++a)
    for (int b = 0; b < nv; ++b)
      for (int c = 0; c < nw; ++c, ++q)
      {
        long double u = xu[a], v = xv[b], w = xw[c];
        long double X[3] = {u, v * (1 - u), w * (1 - u) * (1 - v)};
        T->w[q] = (double)(wu[a] * wv[b] * ww[c] * (1 - u) * (1 - u) * (1 - v));
        basis_eval(B, X, T->phi[q], T->dphi[q]);
      }
  T->nq = q;
}

/* quadrature exact to degree deg on reference facet lf, parametrised over the unit triangle
 * (weights sum to 1/2); the physical scale factor is |(p1-p0) x (p2-p0)| = 2*area. */
static void facet_table(const basis_t* B, int deg, int lf, qtab_t* T)
{
  int nu = (deg + 3) / 2, nv = (deg + 2) / 2;
  long double xu[16], wu[16], xv[16], wv[16];
  gauss_legendre01(nu, xu, wu);
  gauss_legendre01(nv, xv, wv);
  const double* q0 = REFV[FACE_V[lf][0]];
  const double* q1 = REFV[FACE_V[lf][1]];
  const double* q2 = REFV[FACE_V[lf][2]];
  int q = 0;
  for (int a = 0; a < nu; ++a)
    for (int b = 0; b < nv; ++b, ++q)
    {
      long double s = xu[a], t = xv[b] * (1 - xu[a]);
      long double X[3];
      for (int d = 0; d < 3; ++d)
        X[d] = q0[d] + s * (q1[d] - q0[d]) + t * (q2[d] - q0[d]);
      T->w[q] = (double)(wu[a] * wv[b] * (1 - xu[a]));
      basis_eval(B, X, T->phi[q], NULL);
    }
  T->nq = q;
}

typedef struct
{
  int ready;
  basis_t B;
  qtab_t stiff;    /* degree 2(k-1): a forms */
  qtab_t mass;     /* degree 2k: L cell terms */
  qtab_t facet[4]; /* degree 2k: L exterior-facet terms */
} elem_t;

static elem_t ELEM[4];

static const elem_t* elem_get(int order)
{
  elem_t* E = &ELEM[order];
  int ready;
#pragma omp atomic read
  ready = E->ready;
  if (ready)
    return E;
#pragma omp critical(zo_elem_init)
  {
    if (!E->ready)
    {
      basis_init(&E->B, order);
      tet_table(&E->B, 2 * (order - 1), &E->stiff);
      tet_table(&E->B, 2 * order, &E->mass);
      for (int f = 0; f < 4; ++f)
        facet_table(&E->B, 2 * order, f, &E->facet[f]);
#pragma omp atomic write
      E->ready = 1;
    }
  }
  return E;
}

void zo_ref_nodes(int order, double* X /* nd*3 */)
{
  double N[ZO_MAXND][3];
  ref_nodes(order, N);
  memcpy(X, N, sizeof(double) * 3 * (size_t)zo_ndofs_cell(order));
}

/* ------------------------------------------------------------------------------------------
 * 4. Element kernels (what FFCx's tabulate_tensor_float64 evaluates; A/b are ACCUMULATED into,
 *    caller zeroes -- the ufcx convention).  coordinate_dofs = 4x3 row-major.
 * ---------------------------------------------------------------------------------------- */
static double geom(const double* cd, double K[3][3])
{
  double J[3][3];
  for (int a = 0; a < 3; ++a)
    for (int al = 0; al < 3; ++al)
      J[a][al] = cd[3 * (al + 1) + a] - cd[a];
  double det = J[0][0] * (J[1][1] * J[2][2] - J[1][2] * J[2][1]) - J[0][1] * (J[1][0] * J[2][2] - J[1][2] * J[2][0])
               + J[0][2] * (J[1][0] * J[2][1] - J[1][1] * J[2][0]);
  double id = 1.0 / det;
  K[0][0] = (J[1][1] * J[2][2] - J[1][2] * J[2][1]) * id;
  K[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) * id;
  K[0][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) * id;
  K[1][0] = (J[1][2] * J[2][0] - J[1][0] * J[2][2]) * id;
  K[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) * id;
  K[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) * id;
  K[2][0] = (J[1][0] * J[2][1] - J[1][1] * J[2][0]) * id;
  K[2][1] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) * id;
  K[2][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) * id;
  return det; /* K = J^-1: K[al][a] = dX_al/dx_a */
}

/* src/Poisson.py:31  a = inner(grad(u), grad(v))*dx */
void zo_tabulate_poisson_a(int order, const double* cd, double* A)
{
  const elem_t* E = elem_get(order);
  const int nd = E->B.nd;
  double K[3][3];
  const double adet = fabs(geom(cd, K));
  for (int q = 0; q < E->stiff.nq; ++q)
  {
    double g[ZO_MAXND][3];
    for (int i = 0; i < nd; ++i)
      for (int a = 0; a < 3; ++a)
        g[i][a] = K[0][a] * E->stiff.dphi[q][i][0] + K[1][a] * E->stiff.dphi[q][i][1] + K[2][a] * E->stiff.dphi[q][i][2];
    const double w = E->stiff.w[q] * adet;
    for (int i = 0; i < nd; ++i)
      for (int j = 0; j < nd; ++j)
        A[i * nd + j] += w * (g[i][0] * g[j][0] + g[i][1] * g[j][1] + g[i][2] * g[j][2]);
  }
}

/* src/Poisson.py:32  L = f*v*dx (cell part);  w = [f on cell dofs | g on cell dofs] */
void zo_tabulate_poisson_L_cell(int order, const double* cd, const double* w, double* b)
{
  const elem_t* E = elem_get(order);
  const int nd = E->B.nd;
  double K[3][3];
  const double adet = fabs(geom(cd, K));
  for (int q = 0; q < E->mass.nq; ++q)
  {
    double fq = 0;
    for (int j = 0; j < nd; ++j)
      fq += w[j] * E->mass.phi[q][j];
    const double s = E->mass.w[q] * adet * fq;
    for (int i = 0; i < nd; ++i)
      b[i] += s * E->mass.phi[q][i];
  }
}

/* src/Poisson.py:32  L = ... + g*v*ds (exterior-facet part), local facet lf */
void zo_tabulate_poisson_L_facet(int order, const double* cd, const double* w, int lf, double* b)
{
  const elem_t* E = elem_get(order);
  const int nd = E->B.nd;
  const double* p0 = cd + 3 * FACE_V[lf][0];
  const double* p1 = cd + 3 * FACE_V[lf][1];
  const double* p2 = cd + 3 * FACE_V[lf][2];
  double e1[3], e2[3];
  for (int a = 0; a < 3; ++a)
  {
    e1[a] = p1[a] - p0[a];
    e2[a] = p2[a] - p0[a];
  }
  const double cx = e1[1] * e2[2] - e1[2] * e2[1], cy = e1[2] * e2[0] - e1[0] * e2[2], cz = e1[0] * e2[1] - e1[1] * e2[0];
  const double scale = sqrt(cx * cx + cy * cy + cz * cz);
  const qtab_t* T = &E->facet[lf];
  for (int q = 0; q < T->nq; ++q)
  {
    double gq = 0;
    for (int j = 0; j < nd; ++j)
      gq += w[nd + j] * T->phi[q][j];
    const double s = T->w[q] * scale * gq;
    for (int i = 0; i < nd; ++i)
      b[i] += s * T->phi[q][i];
  }
}

/* src/Poisson.py:33  M = action(a, un); w = un on cell dofs */
void zo_tabulate_poisson_M(int order, const double* cd, const double* w, double* y)
{
  const int nd = zo_ndofs_cell(order);
  double A[ZO_MAXND * ZO_MAXND];
  memset(A, 0, sizeof(double) * (size_t)(nd * nd));
  zo_tabulate_poisson_a(order, cd, A);
  for (int i = 0; i < nd; ++i)
  {
    double s = 0;
    for (int j = 0; j < nd; ++j)
      s += A[i * nd + j] * w[j];
    y[i] += s;
  }
}

/* src/Elasticity.py:12-15,30-39: a = inner(sigma(u), eps(v))*dx, dofs interleaved 3*node+comp */
void zo_tabulate_elasticity_a(int order, const double* cd, double* A)
{
  const double Ey = 1.0e6, nu = 0.3;
  const double mu = Ey / (2.0 * (1.0 + nu));
  const double lmbda = Ey * nu / ((1.0 + nu) * (1.0 - 2.0 * nu));
  const elem_t* E = elem_get(order);
  const int nd = E->B.nd, n3 = 3 * nd;
  double K[3][3];
  const double adet = fabs(geom(cd, K));
  for (int q = 0; q < E->stiff.nq; ++q)
  {
    double g[ZO_MAXND][3];
    for (int i = 0; i < nd; ++i)
      for (int a = 0; a < 3; ++a)
        g[i][a] = K[0][a] * E->stiff.dphi[q][i][0] + K[1][a] * E->stiff.dphi[q][i][1] + K[2][a] * E->stiff.dphi[q][i][2];
    const double w = E->stiff.w[q] * adet;
    for (int j = 0; j < nd; ++j)
      for (int d = 0; d < 3; ++d)
      {
        /* trial function u = phi_j e_d: grad(u)[a][b] = delta_ad * g_j[b] */
        double eu[3][3], sig[3][3];
        for (int a = 0; a < 3; ++a)
          for (int b = 0; b < 3; ++b)
            eu[a][b] = 0.5 * ((a == d ? g[j][b] : 0.0) + (b == d ? g[j][a] : 0.0));
        const double tr = eu[0][0] + eu[1][1] + eu[2][2];
        for (int a = 0; a < 3; ++a)
          for (int b = 0; b < 3; ++b)
            sig[a][b] = 2.0 * mu * eu[a][b] + (a == b ? lmbda * tr : 0.0);
        for (int i = 0; i < nd; ++i)
          for (int c = 0; c < 3; ++c)
          {
            double s = 0;
            for (int a = 0; a < 3; ++a)
              for (int b = 0; b < 3; ++b)
                s += sig[a][b] * 0.5 * ((a == c ? g[i][b] : 0.0) + (b == c ? g[i][a] : 0.0));
            A[(3 * i + c) * n3 + (3 * j + d)] += w * s;
          }
      }
  }
}

/* src/Elasticity.py:40  L = inner(f, v)*dx;  w = f interleaved (3*nd) */
void zo_tabulate_elasticity_L(int order, const double* cd, const double* w, double* b)
{
  const elem_t* E = elem_get(order);
  const int nd = E->B.nd;
  double K[3][3];
  const double adet = fabs(geom(cd, K));
  for (int q = 0; q < E->mass.nq; ++q)
  {
    double fq[3] = {0, 0, 0};
    for (int j = 0; j < nd; ++j)
      for (int c = 0; c < 3; ++c)
        fq[c] += w[3 * j + c] * E->mass.phi[q][j];
    const double s = E->mass.w[q] * adet;
    for (int i = 0; i < nd; ++i)
      for (int c = 0; c < 3; ++c)
        b[3 * i + c] += s * fq[c] * E->mass.phi[q][i];
  }
}

/* ------------------------------------------------------------------------------------------
 * 5. Topology -> dofmap (what fem::create_functionspace produces, up to numbering) [EXT]
 *    Numbering: vertices, then edges (sorted by (lo,hi) vertex), then faces (sorted triple).
 *    Edge orientation is resolved in the dofmap (reference direction vs global low->high),
 *    so element kernels need no dof transformation (SURVEY 3.2 note).
 * ---------------------------------------------------------------------------------------- */
typedef struct
{
  i64 k0, k1, k2;
} key3;

static int key3_cmp(const void* a, const void* b)
{
  const key3* x = a;
  const key3* y = b;
  if (x->k0 != y->k0)
    return x->k0 < y->k0 ? -1 : 1;
  if (x->k1 != y->k1)
    return x->k1 < y->k1 ? -1 : 1;
  if (x->k2 != y->k2)
    return x->k2 < y->k2 ? -1 : 1;
  return 0;
}

static void sort3(i64 v[3])
{
  i64 t;
  if (v[0] > v[1]) { t = v[0]; v[0] = v[1]; v[1] = t; }
  if (v[1] > v[2]) { t = v[1]; v[1] = v[2]; v[2] = t; }
  if (v[0] > v[1]) { t = v[0]; v[0] = v[1]; v[1] = t; }
}

/* unique sorted keys; returns count; ids[k] for each of the n input keys */
static i64 unique_ids(key3* keys, i64 n, i64* ids)
{
  /* sort an index permutation */
  typedef struct { key3 k; i64 idx; } rec;
  rec* r = malloc(sizeof(rec) * (size_t)n);
  for (i64 i = 0; i < n; ++i)
  {
    r[i].k = keys[i];
    r[i].idx = i;
  }
  qsort(r, (size_t)n, sizeof(rec), key3_cmp);
  i64 u = -1;
  for (i64 i = 0; i < n; ++i)
  {
    if (i == 0 || key3_cmp(&r[i].k, &r[i - 1].k) != 0)
      ++u;
    ids[r[i].idx] = u;
  }
  free(r);
  return u + 1;
}

/* returns number of (block) dofs; fills cell_dofs[nc*nd] and dof coordinates dof_x[3*ndofs]
 * (pass NULL to skip).  out_counts = {nverts_used, nedges, nfaces}. */
i64 zo_build_dofmap(int order, i64 nv, i64 nc, const i32* cells, const double* x, i32* cell_dofs, double* dof_x,
                    i64* out_counts)
{
  const int nd = zo_ndofs_cell(order);
  i64 ne = 0, nf = 0;
  i64* eid = NULL;
  i64* fid = NULL;
  if (order >= 2)
  {
    key3* k = malloc(sizeof(key3) * (size_t)(6 * nc));
    eid = malloc(sizeof(i64) * (size_t)(6 * nc));
    for (i64 c = 0; c < nc; ++c)
      for (int e = 0; e < 6; ++e)
      {
        i64 a = cells[4 * c + EDGE_V[e][0]], b = cells[4 * c + EDGE_V[e][1]];
        k[6 * c + e].k0 = a < b ? a : b;
        k[6 * c + e].k1 = a < b ? b : a;
        k[6 * c + e].k2 = 0;
      }
    ne = unique_ids(k, 6 * nc, eid);
    free(k);
  }
  if (order >= 3)
  {
    key3* k = malloc(sizeof(key3) * (size_t)(4 * nc));
    fid = malloc(sizeof(i64) * (size_t)(4 * nc));
    for (i64 c = 0; c < nc; ++c)
      for (int f = 0; f < 4; ++f)
      {
        i64 v[3] = {cells[4 * c + FACE_V[f][0]], cells[4 * c + FACE_V[f][1]], cells[4 * c + FACE_V[f][2]]};
        sort3(v);
        k[4 * c + f].k0 = v[0];
        k[4 * c + f].k1 = v[1];
        k[4 * c + f].k2 = v[2];
      }
    nf = unique_ids(k, 4 * nc, fid);
    free(k);
  }
  const int npe = order - 1;
  const i64 ndofs = nv + npe * ne + (order == 3 ? nf : 0);
  for (i64 c = 0; c < nc; ++c)
  {
    i32* d = cell_dofs + (size_t)c * nd;
    int n = 0;
    for (int v = 0; v < 4; ++v)
      d[n++] = cells[4 * c + v];
    if (order >= 2)
      for (int e = 0; e < 6; ++e)
      {
        const i64 a = cells[4 * c + EDGE_V[e][0]], b = cells[4 * c + EDGE_V[e][1]];
        for (int s = 0; s < npe; ++s)
        {
          /* global sub-dof 0 is the one nearest the lower global vertex */
          const int gs = (a < b) ? s : npe - 1 - s;
          d[n++] = (i32)(nv + npe * eid[6 * c + e] + gs);
        }
      }
    if (order == 3)
      for (int f = 0; f < 4; ++f)
        d[n++] = (i32)(nv + npe * ne + fid[4 * c + f]);
  }
  if (dof_x)
  {
    double X[ZO_MAXND][3];
    ref_nodes(order, X);
    for (i64 c = 0; c < nc; ++c)
      for (int i = 0; i < nd; ++i)
      {
        const i64 dof = cell_dofs[(size_t)c * nd + i];
        const double l0 = 1.0 - X[i][0] - X[i][1] - X[i][2];
        for (int a = 0; a < 3; ++a)
          dof_x[3 * dof + a] = l0 * x[3 * (i64)cells[4 * c + 0] + a] + X[i][0] * x[3 * (i64)cells[4 * c + 1] + a]
                               + X[i][1] * x[3 * (i64)cells[4 * c + 2] + a] + X[i][2] * x[3 * (i64)cells[4 * c + 3] + a];
      }
  }
  if (out_counts)
  {
    out_counts[0] = nv;
    out_counts[1] = ne;
    out_counts[2] = nf;
  }
  free(eid);
  free(fid);
  return ndofs;
}

/* Exterior facets = faces belonging to exactly one cell (what the `ds` measure integrates over,
 * src/Poisson.py:32).  out = (cell, local_facet) pairs; pass NULL to count. */
i64 zo_exterior_facets(i64 nc, const i32* cells, i32* out)
{
  typedef struct { key3 k; i64 idx; } rec;
  rec* r = malloc(sizeof(rec) * (size_t)(4 * nc));
  for (i64 c = 0; c < nc; ++c)
    for (int f = 0; f < 4; ++f)
    {
      i64 v[3] = {cells[4 * c + FACE_V[f][0]], cells[4 * c + FACE_V[f][1]], cells[4 * c + FACE_V[f][2]]};
      sort3(v);
      r[4 * c + f].k.k0 = v[0];
      r[4 * c + f].k.k1 = v[1];
      r[4 * c + f].k.k2 = v[2];
      r[4 * c + f].idx = 4 * c + f;
    }
  qsort(r, (size_t)(4 * nc), sizeof(rec), key3_cmp);
  /* collect singletons, then order by (cell, facet) */
  i64 n = 0;
  i64* sel = malloc(sizeof(i64) * (size_t)(4 * nc));
  for (i64 i = 0; i < 4 * nc;)
  {
    i64 j = i + 1;
    while (j < 4 * nc && key3_cmp(&r[j].k, &r[i].k) == 0)
      ++j;
    if (j - i == 1)
      sel[n++] = r[i].idx;
    i = j;
  }
  if (out)
  {
    /* insertion into cell order via counting on idx */
    char* mark = calloc((size_t)(4 * nc), 1);
    for (i64 i = 0; i < n; ++i)
      mark[sel[i]] = 1;
    i64 m = 0;
    for (i64 i = 0; i < 4 * nc; ++i)
      if (mark[i])
      {
        out[2 * m] = (i32)(i / 4);
        out[2 * m + 1] = (i32)(i % 4);
        ++m;
      }
    free(mark);
  }
  free(sel);
  free(r);
  return n;
}

/* Dirichlet dofs, topologically, as locate_entities + locate_dofs_topological do
 * (src/poisson_problem.cpp:58-75, src/elasticity_problem.cpp:125-142) [EXT]: facets ALL of whose
 * vertices are marked; then every dof on the closure of those facets.
 * kind 0: |x0| < 1e-8 or |x0 - 1| < 1e-8 (Poisson);  kind 1: |x1| < 1e-8 (elasticity).
 * marker has one byte per BLOCK dof (all bs components are constrained). */
void zo_locate_bc(int kind, int order, i64 nv, i64 nc, const i32* cells, const double* x, const i32* cell_dofs,
                  i64 ndofs, uint8_t* marker)
{
  const double eps = 1.0e-8;
  const int nd = zo_ndofs_cell(order);
  const int npe = order - 1;
  uint8_t* vm = calloc((size_t)nv, 1);
  for (i64 v = 0; v < nv; ++v)
  {
    if (kind == 0)
      vm[v] = (fabs(x[3 * v]) < eps || fabs(x[3 * v] - 1) < eps);
    else
      vm[v] = (fabs(x[3 * v + 1]) < eps);
  }
  memset(marker, 0, (size_t)ndofs);
  for (i64 c = 0; c < nc; ++c)
    for (int f = 0; f < 4; ++f)
    {
      const int* fv = FACE_V[f];
      if (!(vm[cells[4 * c + fv[0]]] && vm[cells[4 * c + fv[1]]] && vm[cells[4 * c + fv[2]]]))
        continue;
      const i32* d = cell_dofs + (size_t)c * nd;
      for (int k = 0; k < 3; ++k)
        marker[d[fv[k]]] = 1;
      if (order >= 2)
        for (int e = 0; e < 6; ++e)
          if (EDGE_V[e][0] != f && EDGE_V[e][1] != f) /* edge lies in facet f (opposite vertex f) */
            for (int s = 0; s < npe; ++s)
              marker[d[4 + npe * e + s]] = 1;
      if (order == 3)
        marker[d[4 + 12 + f]] = 1;
    }
  free(vm);
}

/* Coefficients by nodal interpolation (Function::interpolate for Lagrange = point evaluation) [EXT]
 * which 0: Poisson f (src/poisson_problem.cpp:85-98), 1: Poisson g (:99-106),
 *       2: elasticity f, 3 interleaved components (src/elasticity_problem.cpp:154-176) */
void zo_interpolate(int which, i64 ndofs, const double* dof_x, double* out)
{
  for (i64 p = 0; p < ndofs; ++p)
  {
    const double* X = dof_x + 3 * p;
    if (which == 0)
    {
      double dx = X[0] - 0.5, dy = X[1] - 0.5;
      double dr = dx * dx + dy * dy;
      out[p] = 10 * exp(-dr / 0.02);
    }
    else if (which == 1)
      out[p] = sin(5 * X[0]);
    else
    {
      double dx = X[0] - 0.5, dz = X[2] - 0.5;
      double r = sqrt(dx * dx + dz * dz);
      out[3 * p + 0] = -dz * r * X[1];
      out[3 * p + 1] = 1.0;
      out[3 * p + 2] = dx * r * X[1];
    }
  }
}

/* ------------------------------------------------------------------------------------------
 * 6. Sparsity pattern (fem::petsc::create_matrix, src/poisson_problem.cpp:122-123) [EXT]:
 *    scalar CSR, every (row dof, col dof) pair sharing a cell, columns ascending, bs expanded.
 * ---------------------------------------------------------------------------------------- */
static int i32_cmp(const void* a, const void* b)
{
  i32 x = *(const i32*)a, y = *(const i32*)b;
  return x < y ? -1 : x > y;
}

/* two calls: cols == NULL -> fills rowptr (n+1 entries, i64) and returns nnz; else fills cols */
i64 zo_pattern(i64 nblock, i64 nc, int nd, int bs, const i32* cell_dofs, i64* rowptr, i32* cols)
{
  /* dof -> cells adjacency */
  i64* off = calloc((size_t)(nblock + 1), sizeof(i64));
  for (i64 c = 0; c < nc; ++c)
    for (int i = 0; i < nd; ++i)
      off[cell_dofs[(size_t)c * nd + i] + 1]++;
  for (i64 i = 0; i < nblock; ++i)
    off[i + 1] += off[i];
  i32* adj = malloc(sizeof(i32) * (size_t)off[nblock]);
  i64* pos = malloc(sizeof(i64) * (size_t)nblock);
  memcpy(pos, off, sizeof(i64) * (size_t)nblock);
  for (i64 c = 0; c < nc; ++c)
    for (int i = 0; i < nd; ++i)
      adj[pos[cell_dofs[(size_t)c * nd + i]]++] = (i32)c;
  free(pos);
  if (!cols)
    rowptr[0] = 0;
  i64 maxc = 0;
  for (i64 i = 0; i < nblock; ++i)
    if (off[i + 1] - off[i] > maxc)
      maxc = off[i + 1] - off[i];
  i32* tmp = malloc(sizeof(i32) * (size_t)(maxc * nd + 1));
  for (i64 r = 0; r < nblock; ++r)
  {
    i64 m = 0;
    for (i64 a = off[r]; a < off[r + 1]; ++a)
      for (int j = 0; j < nd; ++j)
        tmp[m++] = cell_dofs[(size_t)adj[a] * nd + j];
    qsort(tmp, (size_t)m, sizeof(i32), i32_cmp);
    i64 u = 0;
    for (i64 k = 0; k < m; ++k)
      if (k == 0 || tmp[k] != tmp[k - 1])
        tmp[u++] = tmp[k];
    for (int c = 0; c < bs; ++c)
    {
      const i64 row = r * bs + c;
      if (!cols)
        rowptr[row + 1] = u * bs; /* counts; prefix-summed below */
      else
        for (i64 k = 0; k < u; ++k)
          for (int d = 0; d < bs; ++d)
            cols[rowptr[row] + k * bs + d] = tmp[k] * bs + d;
    }
  }
  if (!cols)
    for (i64 r = 0; r < nblock * bs; ++r)
      rowptr[r + 1] += rowptr[r];
  free(tmp);
  free(adj);
  free(off);
  return rowptr[nblock * bs];
}

/* ------------------------------------------------------------------------------------------
 * 7. Assembly
 * ---------------------------------------------------------------------------------------- */
static inline i64 find_col(const i64* rowptr, const i32* cols, i64 row, i32 col)
{
  i64 lo = rowptr[row], hi = rowptr[row + 1] - 1;
  while (lo <= hi)
  {
    i64 mid = (lo + hi) >> 1;
    if (cols[mid] == col)
      return mid;
    if (cols[mid] < col)
      lo = mid + 1;
    else
      hi = mid - 1;
  }
  return -1;
}

/* form 0: Poisson a (bs = 1), form 1: elasticity a (bs = 3).
 * Restates src/poisson_problem.cpp:125-139: fem::assemble_matrix(ADD_VALUES, {bc}) -- per cell
 * gather x_c, tabulate, zero the rows AND columns of Ae whose dof is constrained, add --
 * then fem::set_diagonal(INSERT_VALUES) = 1.0 on constrained diagonals.
 * bc_marker: one byte per SCALAR dof.  vals must be zeroed by the caller (MatZeroEntries state
 * of a fresh matrix).  Returns 0, or -1 if a (row,col) is missing from the pattern. */
int zo_assemble_matrix(int form, int order, const double* x, i64 nc, const i32* cells, const i32* cell_dofs,
                       const uint8_t* bc_marker, i64 nrows, const i64* rowptr, const i32* cols, double* vals)
{
  const int nd = zo_ndofs_cell(order);
  const int bs = form == 1 ? 3 : 1;
  const int n = nd * bs;
  int err = 0;
  (void)elem_get(order);
  /* with one thread (the setting every parity test uses) the cell order, and so every sum, is
   * the serial one; more threads (cpu_baseline timing only) add through atomics */
#pragma omp parallel for schedule(static)
  for (i64 c = 0; c < nc; ++c)
  {
    double cd[12];
    double Ae[3 * ZO_MAXND * 3 * ZO_MAXND];
    i32 dofs[3 * ZO_MAXND];
    for (int v = 0; v < 4; ++v)
      for (int a = 0; a < 3; ++a)
        cd[3 * v + a] = x[3 * (i64)cells[4 * c + v] + a];
    memset(Ae, 0, sizeof(double) * (size_t)(n * n));
    if (form == 0)
      zo_tabulate_poisson_a(order, cd, Ae);
    else
      zo_tabulate_elasticity_a(order, cd, Ae);
    for (int i = 0; i < nd; ++i)
      for (int k = 0; k < bs; ++k)
        dofs[i * bs + k] = cell_dofs[(size_t)c * nd + i] * bs + k;
    for (int i = 0; i < n; ++i)
      if (bc_marker[dofs[i]])
        for (int j = 0; j < n; ++j)
        {
          Ae[i * n + j] = 0.0;
          Ae[j * n + i] = 0.0;
        }
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j)
      {
        /* a partition's feed (owned rows, then ghosts): rows of ghost dofs belong to their owners, who hold every cell
         * that touches them (ghost-cell layer) -- what MatSetValuesLocal + MatAssembly leave on this rank is the owned rows */
        if (dofs[i] >= nrows)
          continue;
        i64 p = find_col(rowptr, cols, dofs[i], dofs[j]);
        if (p < 0)
        {
#pragma omp atomic write
          err = -1;
        }
        else
        {
#pragma omp atomic
          vals[p] += Ae[i * n + j];
        }
      }
  }
  for (i64 r = 0; r < nrows; ++r)
    if (bc_marker[r])
    {
      i64 p = find_col(rowptr, cols, r, (i32)r);
      if (p < 0)
        err = -1;
      else
        vals[p] = 1.0;
    }
  return err;
}

/* form 0: Poisson L (cells + exterior facets; coefficients f, g), form 1: elasticity L (f, bs 3).
 * Restates src/poisson_problem.cpp:146-157: assemble_vector (cell integrals, then exterior-facet
 * integrals), apply_lifting (adds exactly zero: the BC value u0 is filled with 0.0 at :53-54, so
 * it is not evaluated), scatter_rev (no-op on one rank), bc->set => b[bc] = 0. */
void zo_assemble_vector(int form, int order, const double* x, i64 nc, const i32* cells, const i32* cell_dofs,
                        const double* f, const double* g, i64 nfacets, const i32* facets, const uint8_t* bc_marker,
                        i64 nrows, double* b)
{
  const int nd = zo_ndofs_cell(order);
  const int bs = form == 1 ? 3 : 1;
  memset(b, 0, sizeof(double) * (size_t)nrows);
  (void)elem_get(order);
#pragma omp parallel for schedule(static)
  for (i64 c = 0; c < nc; ++c)
  {
    double cd[12], w[3 * ZO_MAXND], be[3 * ZO_MAXND];
    for (int v = 0; v < 4; ++v)
      for (int a = 0; a < 3; ++a)
        cd[3 * v + a] = x[3 * (i64)cells[4 * c + v] + a];
    memset(be, 0, sizeof(be));
    const i32* d = cell_dofs + (size_t)c * nd;
    if (form == 0)
    {
      for (int j = 0; j < nd; ++j)
      {
        w[j] = f[d[j]];
        w[nd + j] = g[d[j]];
      }
      zo_tabulate_poisson_L_cell(order, cd, w, be);
    }
    else
    {
      for (int j = 0; j < nd; ++j)
        for (int k = 0; k < 3; ++k)
          w[3 * j + k] = f[3 * (i64)d[j] + k];
      zo_tabulate_elasticity_L(order, cd, w, be);
    }
    for (int i = 0; i < nd; ++i)
      for (int k = 0; k < bs; ++k)
      {
#pragma omp atomic
        b[(i64)d[i] * bs + k] += be[i * bs + k];
      }
  }
  if (form == 0)
    for (i64 fi = 0; fi < nfacets; ++fi)
    {
      const i64 c = facets[2 * fi];
      const int lf = facets[2 * fi + 1];
      double cd[12], w[2 * ZO_MAXND], be[ZO_MAXND];
      for (int v = 0; v < 4; ++v)
        for (int a = 0; a < 3; ++a)
          cd[3 * v + a] = x[3 * (i64)cells[4 * c + v] + a];
      memset(be, 0, sizeof(be));
      const i32* d = cell_dofs + (size_t)c * nd;
      for (int j = 0; j < nd; ++j)
      {
        w[j] = f[d[j]];
        w[nd + j] = g[d[j]];
      }
      zo_tabulate_poisson_L_facet(order, cd, w, lf, be);
      for (int i = 0; i < nd; ++i)
        b[d[i]] += be[i];
    }
  for (i64 r = 0; r < nrows; ++r)
    if (bc_marker[r])
      b[r] = 0.0;
}

/* Matrix-free operator of cgpoisson (src/cgpoisson_problem.cpp:193-230): y = 0; assemble M with
 * un = x; y[bc] = 0 (scatter steps are no-ops on one rank). */
void zo_action_poisson(int order, const double* x, i64 nc, const i32* cells, const i32* cell_dofs,
                       const uint8_t* bc_marker, i64 n, const double* u, double* y)
{
  const int nd = zo_ndofs_cell(order);
  memset(y, 0, sizeof(double) * (size_t)n);
  (void)elem_get(order);
  for (i64 c = 0; c < nc; ++c)
  {
    double cd[12], w[ZO_MAXND], ye[ZO_MAXND];
    for (int v = 0; v < 4; ++v)
      for (int a = 0; a < 3; ++a)
        cd[3 * v + a] = x[3 * (i64)cells[4 * c + v] + a];
    const i32* d = cell_dofs + (size_t)c * nd;
    for (int j = 0; j < nd; ++j)
      w[j] = u[d[j]];
    memset(ye, 0, sizeof(ye));
    zo_tabulate_poisson_M(order, cd, w, ye);
    for (int i = 0; i < nd; ++i)
      y[d[i]] += ye[i];
  }
  for (i64 r = 0; r < n; ++r)
    if (bc_marker[r])
      y[r] = 0.0;
}

/* ------------------------------------------------------------------------------------------
 * 8. Linear algebra: CSR SpMV, cg.h CG, PETSc-style Jacobi PCG
 * ---------------------------------------------------------------------------------------- */
void zo_spmv(i64 n, const i64* rowptr, const i32* cols, const double* vals, const double* x, double* y)
{
#pragma omp parallel for schedule(static)
  for (i64 r = 0; r < n; ++r)
  {
    double s = 0;
    for (i64 k = rowptr[r]; k < rowptr[r + 1]; ++k)
      s += vals[k] * x[cols[k]];
    y[r] = s;
  }
}

static double dot(i64 n, const double* a, const double* b)
{
  double s = 0;
#pragma omp parallel for schedule(static) reduction(+ : s)
  for (i64 i = 0; i < n; ++i)
    s += a[i] * b[i];
  return s;
}

/* src/cg.h:18-25  r = alpha*x + y (aliasing allowed) */
static void axpy(i64 n, double* r, double alpha, const double* x, const double* y)
{
#pragma omp parallel for schedule(static)
  for (i64 i = 0; i < n; ++i)
    r[i] = alpha * x[i] + y[i];
}

/* src/cg.h:38-86 with action(x, y) = CSR SpMV.  Returns the iteration count k (src/cg.h:85).
 * rnorm_out (optional) = final <r,r>/<r0,r0>. */
int zo_cg(i64 n, const i64* rowptr, const i32* cols, const double* vals, const double* b, double* x, int kmax,
          double rtol, double* rnorm_out)
{
  double* r = malloc(sizeof(double) * (size_t)n);
  double* y = malloc(sizeof(double) * (size_t)n);
  double* p = malloc(sizeof(double) * (size_t)n);
  zo_spmv(n, rowptr, cols, vals, x, y); /* :46 */
  axpy(n, r, -1.0, y, b);               /* :47 */
  memcpy(p, r, sizeof(double) * (size_t)n);
  const double rnorm0 = dot(n, r, r); /* :53 */
  const double rtol2 = rtol * rtol;
  double rnorm = rnorm0;
  int k = 0;
  while (k < kmax)
  {
    ++k;
    zo_spmv(n, rowptr, cols, vals, p, y);       /* :62 */
    const double alpha = rnorm / dot(n, p, y);  /* :65 */
    axpy(n, x, alpha, p, x);                    /* :68 */
    axpy(n, r, -alpha, y, r);                   /* :71 */
    const double rnorm_new = dot(n, r, r);      /* :74 */
    const double beta = rnorm_new / rnorm;      /* :75 */
    rnorm = rnorm_new;
    if (rnorm / rnorm0 < rtol2) /* :78 */
      break;
    axpy(n, p, beta, p, r); /* :82 */
  }
  if (rnorm_out)
    *rnorm_out = rnorm / rnorm0;
  free(r);
  free(y);
  free(p);
  return k;
}

/* src/cg.h:38-86 with the matrix-free operator of cgpoisson (src/cgpoisson_problem.cpp:233) */
int zo_cg_matfree_poisson(int order, const double* xg, i64 nc, const i32* cells, const i32* cell_dofs,
                          const uint8_t* bc_marker, i64 n, const double* b, double* x, int kmax, double rtol)
{
  double* r = malloc(sizeof(double) * (size_t)n);
  double* y = malloc(sizeof(double) * (size_t)n);
  double* p = malloc(sizeof(double) * (size_t)n);
  zo_action_poisson(order, xg, nc, cells, cell_dofs, bc_marker, n, x, y);
  axpy(n, r, -1.0, y, b);
  memcpy(p, r, sizeof(double) * (size_t)n);
  const double rnorm0 = dot(n, r, r);
  const double rtol2 = rtol * rtol;
  double rnorm = rnorm0;
  int k = 0;
  while (k < kmax)
  {
    ++k;
    zo_action_poisson(order, xg, nc, cells, cell_dofs, bc_marker, n, p, y);
    const double alpha = rnorm / dot(n, p, y);
    axpy(n, x, alpha, p, x);
    axpy(n, r, -alpha, y, r);
    const double rnorm_new = dot(n, r, r);
    const double beta = rnorm_new / rnorm;
    rnorm = rnorm_new;
    if (rnorm / rnorm0 < rtol2)
      break;
    axpy(n, p, beta, p, r);
  }
  free(r);
  free(y);
  free(p);
  return k;
}

/* PETSc KSPCG (src/ksp/ksp/impls/cg/cg.c, symmetric variant) with PCJACOBI or PCNONE, zero
 * initial guess, as the reference selects it through "-ksp_type cg -pc_type jacobi -ksp_rtol R"
 * (src/poisson_problem.cpp:168-177, README.md:66-82) [EXT, restated from the published algorithm]:
 *   r = b; z = M^-1 r; dp = norm; converged if dp <= max(rtol*dp0, atol) (KSPConvergedDefault);
 *   loop: beta = (r,z); p = z + (beta/betaold) p; w = A p; a = beta/(p,w); x += a p; r -= a w;
 *         z = M^-1 r; dp = norm; ++it; test.
 * norm_type 0: preconditioned ||z|| (PETSc default for CG), 1: unpreconditioned ||r||,
 *           2: natural sqrt((r,z)).
 * pc 0: none, 1: Jacobi (PCJACOBI default: z = r / diag(A), diagonal entries of 0 replaced by 1).
 * Returns iteration count; x is overwritten (zero initial guess: KSP default).
 * rnorm_out[0] = final norm, rnorm_out[1] = initial norm. */
static double now_s(void)
{
#ifdef _OPENMP
  return omp_get_wtime();
#else
  return 0.0;
#endif
}

/* parallel first-touch copy: each thread touches the pages of the rows it will stream in zo_spmv
 * (static schedule), so that on a multi-socket host the matrix is spread over the NUMA nodes the way
 * an MPI run of the reference would have it */
static void* numa_copy(const void* src, size_t elem, i64 count, i64 nrows, const i64* rowptr_for_split)
{
  char* dst = malloc(elem * (size_t)(count > 0 ? count : 1));
  if (!rowptr_for_split)
  {
#pragma omp parallel for schedule(static)
    for (i64 i = 0; i < count; ++i)
      memcpy(dst + elem * (size_t)i, (const char*)src + elem * (size_t)i, elem);
  }
  else
  {
#pragma omp parallel for schedule(static)
    for (i64 r = 0; r < nrows; ++r)
      memcpy(dst + elem * (size_t)rowptr_for_split[r], (const char*)src + elem * (size_t)rowptr_for_split[r],
             elem * (size_t)(rowptr_for_split[r + 1] - rowptr_for_split[r]));
  }
  return dst;
}

int zo_pcg(i64 n, const i64* rowptr_in, const i32* cols_in, const double* vals_in, const double* b_in, double* x, int pc,
           int norm_type, double rtol, double atol, int max_it, double* rnorm_out)
{
  /* working copies with NUMA-friendly placement (not timed: rnorm_out[2] reports the loop alone) */
  i64* rowptr = numa_copy(rowptr_in, sizeof(i64), n + 1, 0, NULL);
  i32* cols = numa_copy(cols_in, sizeof(i32), rowptr_in[n], n, rowptr_in);
  double* vals = numa_copy(vals_in, sizeof(double), rowptr_in[n], n, rowptr_in);
  double* b = numa_copy(b_in, sizeof(double), n, 0, NULL);
  double* r = malloc(sizeof(double) * (size_t)n);
  double* z = malloc(sizeof(double) * (size_t)n);
  double* p = malloc(sizeof(double) * (size_t)n);
  double* w = malloc(sizeof(double) * (size_t)n);
  double* dinv = malloc(sizeof(double) * (size_t)n);
  double* xx = malloc(sizeof(double) * (size_t)n);
#pragma omp parallel for schedule(static)
  for (i64 i = 0; i < n; ++i)
  {
    double d = 1.0;
    if (pc == 1)
    {
      i64 q = find_col(rowptr, cols, i, (i32)i);
      d = q >= 0 ? vals[q] : 0.0;
      if (d == 0.0)
        d = 1.0;
    }
    dinv[i] = 1.0 / d;
    xx[i] = 0.0;      /* KSP zero initial guess */
    r[i] = b[i];
    z[i] = dinv[i] * r[i];
    p[i] = 0.0;
    w[i] = 0.0;
  }
  const double t_begin = now_s();
  double beta = dot(n, r, z), betaold = 1.0;
  double dp = norm_type == 0 ? sqrt(dot(n, z, z)) : norm_type == 1 ? sqrt(dot(n, r, r)) : sqrt(fabs(beta));
  const double dp0 = dp;
  const double ttol = fmax(rtol * dp0, atol);
  int it = 0;
  if (!(dp <= ttol))
  {
    while (it < max_it)
    {
      if (it == 0)
        axpy(n, p, 0.0, p, z); /* p = z */
      else
        axpy(n, p, beta / betaold, p, z);
      zo_spmv(n, rowptr, cols, vals, p, w);
      const double dpi = dot(n, p, w);
      const double a = beta / dpi;
      axpy(n, xx, a, p, xx);
      axpy(n, r, -a, w, r);
#pragma omp parallel for schedule(static)
      for (i64 i = 0; i < n; ++i)
        z[i] = dinv[i] * r[i];
      betaold = beta;
      beta = dot(n, r, z);
      dp = norm_type == 0 ? sqrt(dot(n, z, z)) : norm_type == 1 ? sqrt(dot(n, r, r)) : sqrt(fabs(beta));
      ++it;
      if (dp <= ttol)
        break;
    }
  }
  const double t_loop = now_s() - t_begin;
  memcpy(x, xx, sizeof(double) * (size_t)n);
  if (rnorm_out)
  {
    rnorm_out[0] = dp;
    rnorm_out[1] = dp0;
    rnorm_out[2] = t_loop;
  }
  free(rowptr);
  free(cols);
  free(vals);
  free(b);
  free(r);
  free(z);
  free(p);
  free(w);
  free(dinv);
  free(xx);
  return it;
}

/* KSPCG with a POLYNOMIAL preconditioner: z = p_k(D^-1 A) D^-1 r, k steps of the Chebyshev iteration for D^-1 A
 * started from zero -- what PETSc selects with "-pc_type ksp -ksp_ksp_type chebyshev -ksp_ksp_max_it k -ksp_pc_type
 * jacobi" [EXT]; the README's answer to Jacobi's iteration counts is a stronger preconditioner (README.md:61-62,108-110;
 * BoomerAMG / GAMG there, which are out of scope) and this is the one that needs no more than the product and no
 * reduction inside its application.  Spectrum bounds as KSPChebyshevEstEigSet's defaults [EXT]: [0.1, 1.1] x an estimate
 * of the largest eigenvalue of D^-1 A, here from `power_its` steps of the power method started from the vector of ones
 * (deterministic and independent of the numbering).  The Chebyshev recurrence is Saad, Iterative Methods, Alg. 12.1:
 *   theta = (hi + lo)/2, delta = (hi - lo)/2, sigma = theta/delta, rho_0 = 1/sigma, d_0 = g/theta (g = D^-1 r), z_0 = 0
 *   z_{i+1} = z_i + d_i;  g_{i+1} = g_i - D^-1 A d_i;  rho_{i+1} = 1/(2 sigma - rho_i);
 *   d_{i+1} = rho_{i+1} rho_i d_i + (2 rho_{i+1}/delta) g_{i+1}
 * Preconditioned-norm test as KSPCG's default.  rnorm_out = {final norm, initial norm, eigenvalue estimate}. */
/* The noise vector of the spectrum estimate: a fixed hash of the (global) row number, in [-1/2, 1/2). */
double zo_noise(i64 i)
{
  unsigned int h = (unsigned int)((unsigned long long)i * 2654435761ull + 12345ull);
  h ^= h >> 16;
  h *= 0x45d9f3bu;
  h ^= h >> 16;
  return (double)h / 4294967296.0 - 0.5;
}

/* Largest eigenvalue of the k x k symmetric tridiagonal (diag t, off-diagonal e[0..k-2]): bisection on the Sturm count
 * between Gershgorin's limits, to full precision. */
double zo_tridiag_lmax(int k, const double* t, const double* e)
{
  double lo = t[0], hi = t[0];
  for (int j = 0; j < k; ++j)
  {
    const double rad = (j > 0 ? fabs(e[j - 1]) : 0.0) + (j + 1 < k ? fabs(e[j]) : 0.0);
    if (t[j] - rad < lo)
      lo = t[j] - rad;
    if (t[j] + rad > hi)
      hi = t[j] + rad;
  }
  for (int itb = 0; itb < 200 && hi - lo > 4.0e-16 * fabs(hi); ++itb)
  {
    const double mid = 0.5 * (lo + hi);
    /* eigenvalues above mid = positive pivots of the LDL^T of T - mid I */
    int above = 0;
    double q = 1.0;
    for (int j = 0; j < k; ++j)
    {
      const double off2 = j > 0 ? e[j - 1] * e[j - 1] : 0.0;
      q = (t[j] - mid) - (j > 0 ? off2 / q : 0.0);
      if (q == 0.0)
        q = 1.0e-300;
      if (q > 0.0)
        ++above;
    }
    if (above > 0)
      lo = mid;
    else
      hi = mid;
  }
  return 0.5 * (lo + hi);
}

/* Largest eigenvalue of D^-1 A estimated as PETSc's KSPChebyshev does by default (-ksp_chebyshev_esteig, noisy
 * right-hand side): est_its iterations of Jacobi-PCG on the noise vector; the Lanczos tridiagonal of its coefficients
 * (diag 1/a_j + b_(j-1)/a_(j-1), off-diagonal sqrt(b_j)/a_j) has the Ritz values; the largest, from below.  offset: global
 * row number of row 0 (the noise is a function of the global row).  Returns 0 when fewer than two iterations ran. */
double zo_esteig(i64 n, const i64* rowptr, const i32* cols, const double* vals, int est_its, i64 offset)
{
  if (est_its > 64)
    est_its = 64;
  double* v = malloc(sizeof(double) * (size_t)n);
  double* xx = malloc(sizeof(double) * (size_t)n);
  for (i64 i = 0; i < n; ++i)
    v[i] = zo_noise(offset + i);
  double alpha[64], rho[65];
  /* KSPCG + PCJACOBI, recording a_j and rho_j = <r_j, z_j> */
  double* r = malloc(sizeof(double) * (size_t)n);
  double* z = malloc(sizeof(double) * (size_t)n);
  double* p = malloc(sizeof(double) * (size_t)n);
  double* w = malloc(sizeof(double) * (size_t)n);
  double* dinv = malloc(sizeof(double) * (size_t)n);
  for (i64 i = 0; i < n; ++i)
  {
    i64 q = find_col(rowptr, cols, i, (i32)i);
    double dd = q >= 0 ? vals[q] : 0.0;
    if (dd == 0.0)
      dd = 1.0;
    dinv[i] = 1.0 / dd;
    xx[i] = 0.0;
    r[i] = v[i];
    z[i] = dinv[i] * r[i];
    p[i] = 0.0;
  }
  int k = 0;
  double beta = dot(n, r, z), betaold = 1.0;
  for (; k < est_its; ++k)
  {
    rho[k] = beta;
    if (!(beta > 0.0) || !isfinite(beta))
      break;
    if (k == 0)
      axpy(n, p, 0.0, p, z);
    else
      axpy(n, p, beta / betaold, p, z);
    zo_spmv(n, rowptr, cols, vals, p, w);
    const double a = beta / dot(n, p, w);
    if (!isfinite(a) || !(a > 0.0))
      break;
    alpha[k] = a;
    axpy(n, xx, a, p, xx);
    axpy(n, r, -a, w, r);
    for (i64 i = 0; i < n; ++i)
      z[i] = dinv[i] * r[i];
    betaold = beta;
    beta = dot(n, r, z);
  }
  double est = 0.0;
  if (k >= 2)
  {
    double t[64], e[64];
    for (int j = 0; j < k; ++j)
    {
      t[j] = 1.0 / alpha[j] + (j > 0 ? (rho[j] / rho[j - 1]) / alpha[j - 1] : 0.0);
      if (j + 1 < k)
        e[j] = sqrt(rho[j + 1] / rho[j]) / alpha[j];
    }
    est = zo_tridiag_lmax(k, t, e);
  }
  free(v), free(xx), free(r), free(z), free(p), free(w), free(dinv);
  return est;
}

int zo_pcg_cheb(i64 n, const i64* rowptr, const i32* cols, const double* vals, const double* b, double* x, int degree,
                int est_its, double ratio, double rtol, double atol, int max_it, double* rnorm_out)
{
  double* r = malloc(sizeof(double) * (size_t)n);
  double* z = malloc(sizeof(double) * (size_t)n);
  double* p = malloc(sizeof(double) * (size_t)n);
  double* w = malloc(sizeof(double) * (size_t)n);
  double* g = malloc(sizeof(double) * (size_t)n);
  double* d = malloc(sizeof(double) * (size_t)n);
  double* dinv = malloc(sizeof(double) * (size_t)n);
  for (i64 i = 0; i < n; ++i)
  {
    i64 q = find_col(rowptr, cols, i, (i32)i);
    double dd = q >= 0 ? vals[q] : 0.0;
    if (dd == 0.0)
      dd = 1.0;
    dinv[i] = 1.0 / dd;
    x[i] = 0.0;
    r[i] = b[i];
    p[i] = 0.0;
  }
  /* upper bound of the spectrum of D^-1 A: Gershgorin's, the largest row sum of |D^-1 A| (rigorous, so the polynomial is
   * positive on the whole spectrum and the preconditioner stays SPD; independent of numbering and partition) */
  double est = 0.0;
  for (i64 i = 0; i < n; ++i)
  {
    double sum = 0.0;
    for (i64 k = rowptr[i]; k < rowptr[i + 1]; ++k)
      sum += fabs(vals[k]);
    sum *= fabs(dinv[i]);
    if (sum > est)
      est = sum;
  }
  /* ... tightened by the Lanczos estimate where that is lower: Gershgorin's bound is exact for P1 Laplacians (2) and up
   * to 2.7 x too high for P2 / P3 / elasticity; the estimate comes from below (0.97-0.98 of the largest eigenvalue after
   * 10 iterations from noise), hence PETSc's safety factor 1.1 */
  if (est_its > 0)
  {
    const double ritz = zo_esteig(n, rowptr, cols, vals, est_its, 0);
    if (ritz > 0.0 && isfinite(ritz) && 1.1 * ritz < est)
      est = 1.1 * ritz;
  }
  const double hi = est, lo = est / ratio;
  const double theta = 0.5 * (hi + lo), delta = 0.5 * (hi - lo), sigma = theta / delta;
#define ZO_CHEB_APPLY()                                                                        \
  do                                                                                           \
  {                                                                                            \
    double rho = 1.0 / sigma;                                                                  \
    for (i64 i = 0; i < n; ++i)                                                                \
    {                                                                                          \
      g[i] = dinv[i] * r[i];                                                                   \
      d[i] = g[i] / theta;                                                                     \
      z[i] = 0.0;                                                                              \
    }                                                                                          \
    for (int s = 0; s < degree; ++s)                                                           \
    {                                                                                          \
      zo_spmv(n, rowptr, cols, vals, d, w);                                                    \
      const double rhon = 1.0 / (2.0 * sigma - rho);                                           \
      const double c1 = rhon * rho, c2 = 2.0 * rhon / delta;                                   \
      for (i64 i = 0; i < n; ++i)                                                              \
      {                                                                                        \
        z[i] = z[i] + d[i];                                                                    \
        g[i] = -1.0 * (dinv[i] * w[i]) + g[i];                                                 \
        d[i] = c1 * d[i] + c2 * g[i];                                                          \
      }                                                                                        \
      rho = rhon;                                                                              \
    }                                                                                          \
    if (degree == 0)                                                                           \
      for (i64 i = 0; i < n; ++i)                                                              \
        z[i] = g[i];                                                                           \
  } while (0)
  ZO_CHEB_APPLY();
  double beta = dot(n, r, z), betaold = 1.0;
  double dp = sqrt(dot(n, z, z));
  const double dp0 = dp;
  const double ttol = fmax(rtol * dp0, atol);
  int it = 0;
  if (!(dp <= ttol))
  {
    while (it < max_it)
    {
      if (it == 0)
        axpy(n, p, 0.0, p, z);
      else
        axpy(n, p, beta / betaold, p, z);
      zo_spmv(n, rowptr, cols, vals, p, w);
      const double dpi = dot(n, p, w);
      const double a = beta / dpi;
      axpy(n, x, a, p, x);
      axpy(n, r, -a, w, r);
      ZO_CHEB_APPLY();
      betaold = beta;
      beta = dot(n, r, z);
      dp = sqrt(dot(n, z, z));
      ++it;
      if (dp <= ttol)
        break;
    }
  }
#undef ZO_CHEB_APPLY
  if (rnorm_out)
  {
    rnorm_out[0] = dp;
    rnorm_out[1] = dp0;
    rnorm_out[2] = est;
  }
  free(r);
  free(z);
  free(p);
  free(w);
  free(g);
  free(d);
  free(dinv);
  return it;
}

/* y = A x with the summation order of the GPU row phase for long rows (csrc/zzz_spmv.hip, lpr_shift > 0):
 * `lanes` (a power of two) partial sums over contiguous chunks of ceil(len/lanes) products, each in column
 * order, combined by a butterfly: ((c0+c1)+(c2+c3))+((c4+c5)+(c6+c7)).  lanes == 1 is zo_spmv.  No
 * reference counterpart (MatMult adds serially): it exists so that the parity test of that kernel mode can
 * stay bit-exact; the results differ from zo_spmv by round-off only. */
void zo_spmv_chunked(i64 n, const i64* rowptr, const i32* cols, const double* vals, const double* x, double* y, int lanes)
{
  for (i64 r = 0; r < n; ++r)
  {
    const i64 a = rowptr[r], b = rowptr[r + 1];
    const i64 chunk = (b - a + lanes - 1) / lanes;
    double c[64];
    for (int j = 0; j < lanes; ++j)
    {
      i64 ka = a + j * chunk, kb = ka + chunk < b ? ka + chunk : b;
      double s = 0.0;
      for (i64 k = ka; k < kb; ++k)
        s += vals[k] * x[cols[k]];
      c[j] = s;
    }
    for (int o = 1; o < lanes; o <<= 1)
      for (int j = 0; j < lanes; j += 2 * o)
        c[j] = c[j] + c[j + o];
    y[r] = c[0];
  }
}

/* KSPCG with -ksp_cg_single_reduction (KSPCGUseSingleReduction; PETSc cg.c, the branches guarded by
 * cg->singlereduction) [EXT, restated from the published algorithm]: the same iteration with
 *   s = A z kept beside z,  w = s + b w (= A p by recurrence),  delta = (z,s),
 *   (p,w) = delta - beta^2 (p,w)_old / betaold^2,
 * so that beta, delta and the norm are reduced together once per iteration.  Same arguments and
 * return values as zo_pcg.  In exact arithmetic the iterates equal zo_pcg's. */
int zo_pcg_sr(i64 n, const i64* rowptr, const i32* cols, const double* vals, const double* b, double* x, int pc,
              int norm_type, double rtol, double atol, int max_it, double* rnorm_out)
{
  double* r = malloc(sizeof(double) * (size_t)n);
  double* z = malloc(sizeof(double) * (size_t)n);
  double* s = malloc(sizeof(double) * (size_t)n);
  double* p = calloc((size_t)n, sizeof(double));
  double* w = calloc((size_t)n, sizeof(double));
  double* dinv = malloc(sizeof(double) * (size_t)n);
  for (i64 i = 0; i < n; ++i)
  {
    double d = 1.0;
    if (pc == 1)
    {
      i64 q = find_col(rowptr, cols, i, (i32)i);
      d = q >= 0 ? vals[q] : 0.0;
      if (d == 0.0)
        d = 1.0;
    }
    dinv[i] = 1.0 / d;
    x[i] = 0.0;
    r[i] = b[i];
    z[i] = dinv[i] * r[i];
  }
  zo_spmv(n, rowptr, cols, vals, z, s);
  double delta = dot(n, z, s);
  double beta = dot(n, r, z), betaold = 1.0, dpi = 0.0, dpiold;
  double dp = norm_type == 0 ? sqrt(dot(n, z, z)) : norm_type == 1 ? sqrt(dot(n, r, r)) : sqrt(fabs(beta));
  const double dp0 = dp;
  const double ttol = fmax(rtol * dp0, atol);
  int it = 0;
  if (!(dp <= ttol))
  {
    while (it < max_it)
    {
      const double bb = it == 0 ? 0.0 : beta / betaold;
      if (it == 0)
      {
        memcpy(p, z, sizeof(double) * (size_t)n);
        memcpy(w, s, sizeof(double) * (size_t)n);
      }
      else
      {
        axpy(n, p, bb, p, z);
        axpy(n, w, bb, w, s);
      }
      dpiold = dpi;
      dpi = it == 0 ? delta : delta - beta * beta * dpiold / (betaold * betaold);
      betaold = beta;
      const double a = beta / dpi;
      axpy(n, x, a, p, x);
      axpy(n, r, -a, w, r);
      for (i64 i = 0; i < n; ++i)
        z[i] = dinv[i] * r[i];
      zo_spmv(n, rowptr, cols, vals, z, s);
      delta = dot(n, z, s);
      beta = dot(n, r, z);
      dp = norm_type == 0 ? sqrt(dot(n, z, z)) : norm_type == 1 ? sqrt(dot(n, r, r)) : sqrt(fabs(beta));
      ++it;
      if (dp <= ttol)
        break;
    }
  }
  if (rnorm_out)
  {
    rnorm_out[0] = dp;
    rnorm_out[1] = dp0;
  }
  free(r);
  free(z);
  free(s);
  free(p);
  free(w);
  free(dinv);
  return it;
}

double zo_norm2(i64 n, const double* x) { return sqrt(dot(n, x, x)); } /* la::norm, src/main.cpp:229 */

int zo_num_threads(void)
{
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

void zo_set_num_threads(int n)
{
#ifdef _OPENMP
  omp_set_num_threads(n);
#else
  (void)n;
#endif
}

/* build_near_nullspace, src/elasticity_problem.cpp:36-94: basis[k] (k < 3) = 1 in component k; rotations
 * x3 = (-x1, x0, 0), x4 = (x2, 0, -x0), x5 = (0, -x2, x1) at the dof coordinates (:56-71); la::orthonormalize
 * [EXT: dolfinx/la/utils.h]: for i: for k < i: x_i -= <x_i, x_k> x_k; x_i /= |x_i| (:74-75); la::is_orthonormal (:76-81):
 * returns the largest |<x_i, x_j> - delta_ij|.  B: [6][3 n], row-major.  One rank: all entries are owned. */
double zo_near_nullspace(i64 n, const double* dof_x, double* B)
{
  const i64 ld = 3 * n;
  memset(B, 0, sizeof(double) * (size_t)(6 * ld));
  for (i64 i = 0; i < n; ++i)
  {
    const double x0 = dof_x[3 * i], x1 = dof_x[3 * i + 1], x2 = dof_x[3 * i + 2];
    for (int k = 0; k < 3; ++k)
      B[k * ld + 3 * i + k] = 1.0;
    B[3 * ld + 3 * i + 0] = -x1;
    B[3 * ld + 3 * i + 1] = x0;
    B[4 * ld + 3 * i + 0] = x2;
    B[4 * ld + 3 * i + 2] = -x0;
    B[5 * ld + 3 * i + 2] = x1;
    B[5 * ld + 3 * i + 1] = -x2;
  }
  for (int i = 0; i < 6; ++i)
  {
    for (int k = 0; k < i; ++k)
    {
      double d = 0.0;
      for (i64 j = 0; j < ld; ++j)
        d += B[i * ld + j] * B[k * ld + j];
      for (i64 j = 0; j < ld; ++j)
        B[i * ld + j] -= d * B[k * ld + j];
    }
    double nn = 0.0;
    for (i64 j = 0; j < ld; ++j)
      nn += B[i * ld + j] * B[i * ld + j];
    nn = sqrt(nn);
    for (i64 j = 0; j < ld; ++j)
      B[i * ld + j] /= nn;
  }
  double dev = 0.0;
  for (int i = 0; i < 6; ++i)
    for (int k = 0; k <= i; ++k)
    {
      double d = 0.0;
      for (i64 j = 0; j < ld; ++j)
        d += B[i * ld + j] * B[k * ld + j];
      d = fabs(d - (i == k ? 1.0 : 0.0));
      if (d > dev)
        dev = d;
    }
  return dev;
}

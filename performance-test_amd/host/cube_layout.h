// Closed-form description of the structured problem feed, shared by the host generator
// (mesh_part.cpp, g++) and the device generator (csrc/zzz_cubegen.hip, hipcc), so that both produce
// the same integers and the same coordinates bit for bit.
//
//  * unit cube cut into nx*ny*nz sub-cubes, 6 Kuhn simplices each (create_box with tetrahedra,
//    src/mesh.cpp:184-186): simplex q of a sub-cube follows the lattice path origin -> +e_PERM[q][0]
//    -> +e_PERM[q][1] -> +e_PERM[q][2]; its vertices are therefore in ascending global order;
//  * P1..P3 Lagrange dofs (src/poisson_problem.cpp:35-38) numbered in "level units": for k = 0..nz
//    the dofs of plane z = k (vertices, in-plane edges, in-plane faces) then those of layer k, inside
//    each block entity type by entity type (struct Layout);
//  * z-slab partition: part p owns sub-cube layers [zs, ze) and one contiguous global dof range;
//    ghosts = plane zs (from p-1) and layer ze + plane ze+1 (from p+1), in that order.
#pragma once

#include <cmath>
#include <cstdint>

#if defined(__HIPCC__)
#define ZZZ_HD __host__ __device__
#else
#define ZZZ_HD
#endif

namespace zzzcube
{
struct Layout
{
  int64_t nx, ny, nz, PX, PY;
  int order, npe, nfd;
  // Within the plane block and within the layer block of a level the dofs are numbered ENTITY TYPE by entity type
  // (vertices; x-, y-, xy-edges; in-plane faces; resp. z-, xz-, yz-, xyz-edges; xz-faces, yz-faces, the six faces
  // inside a sub-cube), inside a type lattice point by lattice point (iy major, ix minor), the sub-dofs of one
  // entity adjacent.  Entities that would stick out of the cube at ix == nx / iy == ny do not exist.
  // Consecutive rows of the matrix then belong to consecutive entities of ONE type: equal row lengths and, entry
  // for entry, consecutive columns -- the layout the sliced-ELL operator stream (csrc/zzz_sellp.hip) wants: its
  // 64-row slices are not padded by a long vertex row next to short face rows, and the x gather of a wave
  // instruction is dense.  (A point-by-point numbering, all entities of a lattice point adjacent, suited the CSR
  // tile kernel better -- a P3 row touched ~15 column clusters instead of ~90 -- but pads the slices 2x.)
  int64_t oEx, oEy, oExy, oFxy; // plane block: offsets of the sub-blocks behind the vertices
  int64_t oExz, oEyz, oExyz, oFxz, oFyz, oFin; // layer block (z-edges first)
  int64_t NP, NL;

  ZZZ_HD Layout(int64_t nx_, int64_t ny_, int64_t nz_, int order_)
      : nx(nx_), ny(ny_), nz(nz_), PX(nx_ + 1), PY(ny_ + 1), order(order_), npe(order_ - 1), nfd(order_ == 3 ? 1 : 0)
  {
    oEx = PX * PY;
    oEy = oEx + nx * PY * npe;
    oExy = oEy + PX * ny * npe;
    oFxy = oExy + nx * ny * npe;
    NP = oFxy + nx * ny * 2 * nfd;
    oExz = PX * PY * npe;
    oEyz = oExz + nx * PY * npe;
    oExyz = oEyz + PX * ny * npe;
    oFxz = oExyz + nx * ny * npe;
    oFyz = oFxz + nx * PY * 2 * nfd;
    oFin = oFyz + PX * ny * 2 * nfd;
    NL = oFin + nx * ny * 6 * nfd;
  }
  ZZZ_HD int64_t level_base(int64_t k) const { return k * (NP + NL); }
  ZZZ_HD int64_t total() const { return (nz + 1) * NP + nz * NL; }

  ZZZ_HD int64_t vertex(const int64_t a[3]) const { return level_base(a[2]) + a[1] * PX + a[0]; }
  // edge anchored at lattice point a (its lowest vertex) with axis mask m (x=1, y=2, z=4), sub-dof s
  ZZZ_HD int64_t edge(const int64_t a[3], int m, int s) const
  {
    const int64_t lb = level_base(a[2]);
    const int64_t full = a[1] * PX + a[0], cut = a[1] * nx + a[0]; // point index where entities exist up to ix <= nx / ix < nx
    switch (m)
    {
    case 1:
      return lb + oEx + cut * npe + s;
    case 2:
      return lb + oEy + full * npe + s;
    case 3:
      return lb + oExy + cut * npe + s;
    case 4:
      return lb + NP + full * npe + s;
    case 5:
      return lb + NP + oExz + cut * npe + s;
    case 6:
      return lb + NP + oEyz + full * npe + s;
    default:
      return lb + NP + oExyz + cut * npe + s;
    }
  }
  // face with vertices a, a+S1, a+S1+S2 (axis masks)
  ZZZ_HD int64_t face(const int64_t a[3], int S1, int S2) const
  {
    const int u = S1 | S2;
    const int64_t lb = level_base(a[2]);
    const int64_t full = a[1] * PX + a[0], cut = a[1] * nx + a[0];
    if (u == 3)
      return lb + oFxy + cut * 2 + (S1 == 1 ? 0 : 1);
    if (u == 5)
      return lb + NP + oFxz + cut * 2 + (S1 == 1 ? 0 : 1);
    if (u == 6)
      return lb + NP + oFyz + full * 2 + (S1 == 2 ? 0 : 1);
    int t;
    if (S1 == 1 || S1 == 2 || S1 == 4)
      t = S1 == 1 ? 0 : (S1 == 2 ? 1 : 2);
    else
      t = 3 + (S2 == 1 ? 0 : (S2 == 2 ? 1 : 2));
    return lb + NP + oFin + cut * 6 + t;
  }
};

// one z-slab of the cube
struct Slab
{
  Layout L;
  int nparts, part, bs, nd;
  int64_t zs, ze, zl_end; // own layers [zs, ze), local layers [zs, zl_end)
  bool lower, upper;
  int64_t own_lo, own_hi, n_owned, n_lower, n_upper, up_lo, nloc, nverts, ncubes, ncells;

  // native: the partition as the reference's cell partitioner leaves it (GhostMode::none, src/mesh.cpp:182-183):
  // own cells only, ghosts = the dofs of the own cells that the lower neighbour owns (plane zs); the rows of the
  // top plane are then incomplete until the cells above arrive (zzz_ghost_layer_build).
  ZZZ_HD Slab(int64_t nx, int64_t ny, int64_t nz, int order, int bs_, int nparts_, int part_, bool native = false)
      : L(nx, ny, nz, order), nparts(nparts_), part(part_), bs(bs_), nd(order == 1 ? 4 : (order == 2 ? 10 : 20))
  {
    zs = nz * part / nparts;
    ze = nz * (part + 1) / nparts;
    lower = part > 0;
    upper = part < nparts - 1;
    const bool upper_layer = upper && !native;
    zl_end = upper_layer ? ze + 1 : ze;
    own_lo = L.level_base(zs) + (lower ? L.NP : 0);
    own_hi = L.level_base(ze) + L.NP;
    n_owned = own_hi - own_lo;
    n_lower = lower ? L.NP : 0;
    n_upper = upper_layer ? L.NL + L.NP : 0;
    up_lo = L.level_base(ze) + L.NP;
    nloc = n_owned + n_lower + n_upper;
    nverts = (zl_end - zs + 1) * L.PX * L.PY;
    ncubes = nx * ny * (zl_end - zs);
    ncells = 6 * ncubes;
  }
  ZZZ_HD int32_t to_local(int64_t g) const
  {
    if (g >= own_lo && g < own_hi)
      return (int32_t)(g - own_lo);
    if (lower && g >= L.level_base(zs) && g < own_lo)
      return (int32_t)(n_owned + (g - L.level_base(zs)));
    return (int32_t)(n_owned + n_lower + (g - up_lo)); // upper ghost
  }
  ZZZ_HD int32_t local_vertex(const int64_t a[3]) const { return (int32_t)(((a[2] - zs) * L.PY + a[1]) * L.PX + a[0]); }
};

// Everything about one cell.  Cell numbering is simplex-type major: c = q * ncubes + cube index,
// so "the a-th cell of my dof" of 64 neighbouring dofs is 64 consecutive cells (dense reads).
struct Cell
{
  int32_t verts[4];
  int32_t dofs[20];
  int64_t gdofs[20];
  double dof_x[20][3];
  unsigned facet_mask; // bit f: local facet f (opposite local vertex f) lies on the cube boundary
};

ZZZ_HD inline void make_cell(const Slab& S, int64_t ix, int64_t iy, int64_t iz, int q, Cell& C)
{
  // Basix local entity ordering of the tetrahedron (src/poisson_problem.cpp:35-38) [EXT]
  const int EDGE_V[6][2] = {{2, 3}, {1, 3}, {1, 2}, {0, 3}, {0, 2}, {0, 1}};
  const int FACE_V[4][3] = {{1, 2, 3}, {0, 2, 3}, {0, 1, 3}, {0, 1, 2}};
  const int PERM[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};
  const Layout& L = S.L;
  const int order = L.order, npe = L.npe;
  const double sq5 = 2.23606797749978969640917366873128; // sqrt(5): GLL points (1 -+ 1/sqrt5)/2
  const double tt[2] = {order == 2 ? 0.5 : 0.5 * (1.0 - 1.0 / sq5), 0.5 * (1.0 + 1.0 / sq5)};
  const double nn[3] = {(double)L.nx, (double)L.ny, (double)L.nz};
  int64_t p[4][3] = {{ix, iy, iz}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  int step[3];
  for (int k = 0; k < 3; ++k)
  {
    step[k] = 1 << PERM[q][k];
    for (int a = 0; a < 3; ++a)
      p[k + 1][a] = p[k][a] + (a == PERM[q][k] ? 1 : 0);
  }
  int n = 0;
  for (int v = 0; v < 4; ++v)
  {
    C.verts[v] = S.local_vertex(p[v]);
    C.gdofs[n] = L.vertex(p[v]);
    for (int a = 0; a < 3; ++a)
      C.dof_x[n][a] = (double)p[v][a] / nn[a];
    ++n;
  }
  if (order >= 2)
    for (int e = 0; e < 6; ++e)
    {
      const int a = EDGE_V[e][0], b = EDGE_V[e][1];
      int m = 0;
      for (int k = a; k < b; ++k)
        m |= step[k];
      for (int s = 0; s < npe; ++s)
      {
        C.gdofs[n] = L.edge(p[a], m, s);
        for (int d = 0; d < 3; ++d)
          C.dof_x[n][d] = ((double)p[a][d] + tt[s] * (double)(p[b][d] - p[a][d])) / nn[d];
        ++n;
      }
    }
  if (order == 3)
    for (int f = 0; f < 4; ++f)
    {
      const int a = FACE_V[f][0], b = FACE_V[f][1], c = FACE_V[f][2];
      int m1 = 0, m2 = 0;
      for (int k = a; k < b; ++k)
        m1 |= step[k];
      for (int k = b; k < c; ++k)
        m2 |= step[k];
      C.gdofs[n] = L.face(p[a], m1, m2);
      for (int d = 0; d < 3; ++d)
        C.dof_x[n][d] = ((double)(p[a][d] + p[b][d] + p[c][d]) / 3.0) / nn[d];
      ++n;
    }
  for (int i = 0; i < n; ++i)
    C.dofs[i] = S.to_local(C.gdofs[i]);
  C.facet_mask = 0;
  for (int f = 0; f < 4; ++f)
    for (int d = 0; d < 3; ++d)
    {
      const int64_t lim = d == 0 ? L.nx : (d == 1 ? L.ny : L.nz);
      const int64_t v0 = p[FACE_V[f][0]][d];
      if ((v0 == 0 || v0 == lim) && p[FACE_V[f][1]][d] == v0 && p[FACE_V[f][2]][d] == v0)
        C.facet_mask |= 1u << f;
    }
}

// Dirichlet marker lambdas (src/poisson_problem.cpp:60-71, src/elasticity_problem.cpp:127-138) on a
// dof coordinate; problem 0 = Poisson (x = 0 or 1), 1 = elasticity (y = 0)
ZZZ_HD inline bool is_dirichlet(int problem, const double X[3])
{
  const double eps = 1.0e-8;
  return problem == 0 ? (fabs(X[0]) < eps || fabs(X[0] - 1) < eps) : (fabs(X[1]) < eps);
}

// coefficient expressions (src/poisson_problem.cpp:85-106, src/elasticity_problem.cpp:154-176)
ZZZ_HD inline double poisson_f(const double X[3])
{
  const double dx = X[0] - 0.5, dy = X[1] - 0.5;
  const double dr = dx * dx + dy * dy;
  return 10 * exp(-dr / 0.02);
}
ZZZ_HD inline double poisson_g(const double X[3]) { return sin(5 * X[0]); }
ZZZ_HD inline void elasticity_f(const double X[3], double out[3])
{
  const double dx = X[0] - 0.5, dz = X[2] - 0.5;
  const double r = sqrt(dx * dx + dz * dz);
  out[0] = -dz * r * X[1];
  out[1] = 1.0;
  out[2] = dx * r * X[1];
}
} // namespace zzzcube

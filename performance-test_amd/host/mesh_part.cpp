// Host-side feed of the hot path: see include/zzz_host.h for what each piece replaces in the
// reference.  Pure C++; no GPU, no oracle.  Everything is closed-form on the structured Kuhn
// triangulation, so a rank generates only its own z-slab (O(local) time and memory).
//
// Global dof numbering ("level units"): for k = 0..nz the dofs of plane z = k (vertices, in-plane
// edges, in-plane faces) are followed by those of layer k (entities spanning z = k..k+1).  A z-slab
// partition then owns one contiguous global range, ghost blocks arrive already in ghost order
// (no unpack kernel), and rows that are neighbours in the mesh are neighbours in memory.
#include "../../include/zzz_host.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <utility>
#include <vector>

namespace
{
thread_local std::string g_err;

void set_err(const char* fmt, ...)
{
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
}

// src/mesh.cpp:44-54
void num_entities(int64_t i, int64_t j, int64_t k, int nrefine, int64_t out[4])
{
  i <<= nrefine;
  j <<= nrefine;
  k <<= nrefine;
  out[0] = (i + 1) * (j + 1) * (k + 1);
  out[1] = 7 * i * j * k + 3 * (i * j + i * k + j * k) + (i + j + k);
  out[2] = 12 * i * j * k + 2 * (i * j + i * k + j * k);
  out[3] = 6 * (i * j * k);
}

// src/mesh.cpp:56-74
int64_t num_pdofs(int64_t i, int64_t j, int64_t k, int nrefine, int order)
{
  int64_t e[4];
  num_entities(i, j, k, nrefine, e);
  switch (order)
  {
  case 1:
    return e[0];
  case 2:
    return e[0] + e[1];
  case 3:
    return e[0] + 2 * e[1] + e[2];
  case 4:
    return e[0] + 3 * e[1] + 3 * e[2] + e[3];
  default:
    return -1; // the reference throws "Order not supported"
  }
}

// Basix local entity ordering of the tetrahedron (src/poisson_problem.cpp:35-38) [EXT]
const int EDGE_V[6][2] = {{2, 3}, {1, 3}, {1, 2}, {0, 3}, {0, 2}, {0, 1}};
const int FACE_V[4][3] = {{1, 2, 3}, {0, 2, 3}, {0, 1, 3}, {0, 1, 2}};
// the six Kuhn simplices of a sub-cube: axis permutations (0 = x, 1 = y, 2 = z)
const int PERM[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};

struct Layout
{
  int64_t nx, ny, nz, PX, PY;
  int order, npe, nfd;
  int64_t offP[5], NP, offL[7], NL;

  Layout(int64_t nx_, int64_t ny_, int64_t nz_, int order_)
      : nx(nx_), ny(ny_), nz(nz_), PX(nx_ + 1), PY(ny_ + 1), order(order_), npe(order_ - 1), nfd(order_ == 3 ? 1 : 0)
  {
    const int64_t cntP[5] = {PX * PY, nx * PY * npe, PX * ny * npe, nx * ny * npe, 2 * nx * ny * nfd};
    NP = 0;
    for (int t = 0; t < 5; ++t)
    {
      offP[t] = NP;
      NP += cntP[t];
    }
    const int64_t cntL[7] = {PX * PY * npe, nx * PY * npe,      PX * ny * npe,     nx * ny * npe,
                             2 * nx * PY * nfd, 2 * PX * ny * nfd, 6 * nx * ny * nfd};
    NL = 0;
    for (int t = 0; t < 7; ++t)
    {
      offL[t] = NL;
      NL += cntL[t];
    }
  }
  int64_t level_base(int64_t k) const { return k * (NP + NL); }
  int64_t total() const { return (nz + 1) * NP + nz * NL; }

  int64_t vertex(const int64_t a[3]) const { return level_base(a[2]) + offP[0] + a[1] * PX + a[0]; }
  // edge anchored at lattice point a with axis mask m (x=1, y=2, z=4), sub-dof s
  int64_t edge(const int64_t a[3], int m, int s) const
  {
    switch (m)
    {
    case 1:
      return level_base(a[2]) + offP[1] + (a[1] * nx + a[0]) * npe + s;
    case 2:
      return level_base(a[2]) + offP[2] + (a[1] * PX + a[0]) * npe + s;
    case 3:
      return level_base(a[2]) + offP[3] + (a[1] * nx + a[0]) * npe + s;
    case 4:
      return level_base(a[2]) + NP + offL[0] + (a[1] * PX + a[0]) * npe + s;
    case 5:
      return level_base(a[2]) + NP + offL[1] + (a[1] * nx + a[0]) * npe + s;
    case 6:
      return level_base(a[2]) + NP + offL[2] + (a[1] * PX + a[0]) * npe + s;
    default:
      return level_base(a[2]) + NP + offL[3] + (a[1] * nx + a[0]) * npe + s;
    }
  }
  // face with vertices a, a+S1, a+S1+S2 (axis masks)
  int64_t face(const int64_t a[3], int S1, int S2) const
  {
    const int u = S1 | S2;
    if (u == 3)
      return level_base(a[2]) + offP[4] + (a[1] * nx + a[0]) * 2 + (S1 == 1 ? 0 : 1);
    if (u == 5)
      return level_base(a[2]) + NP + offL[4] + (a[1] * nx + a[0]) * 2 + (S1 == 1 ? 0 : 1);
    if (u == 6)
      return level_base(a[2]) + NP + offL[5] + (a[1] * PX + a[0]) * 2 + (S1 == 2 ? 0 : 1);
    int t;
    if (S1 == 1 || S1 == 2 || S1 == 4)
      t = S1 == 1 ? 0 : (S1 == 2 ? 1 : 2);
    else
      t = 3 + (S2 == 1 ? 0 : (S2 == 2 ? 1 : 2));
    return level_base(a[2]) + NP + offL[6] + (a[1] * nx + a[0]) * 6 + t;
  }
};
} // namespace

struct zzzh_part
{
  int problem, order, bs, nd, nparts, part;
  int64_t nx, ny, nz;
  int64_t sizes[ZZZH_NSIZES];
  std::vector<double> x, dof_x, coeff[2];
  std::vector<int32_t> cells, cell_dofs, facets, bc_dofs, neigh, send_idx;
  std::vector<int64_t> global_dofs, send_off, recv_cnt;
};

extern "C" {

int64_t zzzh_num_pdofs(int64_t i, int64_t j, int64_t k, int nrefine, int order) { return num_pdofs(i, j, k, nrefine, order); }
void zzzh_num_entities(int64_t i, int64_t j, int64_t k, int nrefine, int64_t out[4]) { num_entities(i, j, k, nrefine, out); }

// src/mesh.cpp:78-151 (create_cube_mesh up to the choice of Nx, Ny, Nz, r)
void zzzh_mesh_size(int64_t target_dofs, int strong, int64_t num_processes, int64_t dofs_per_node, int order, int64_t out[4])
{
  int64_t N = strong ? target_dofs / dofs_per_node : target_dofs * num_processes / dofs_per_node; // :86-90
  int64_t Nx = 1, Ny, Nz;
  int r = 0;
  const int64_t Nx_max = 200; // :98
  int64_t ndofs = 0;
  while (ndofs < N) // :103-126
  {
    ++Nx;
    if (Nx > Nx_max)
    {
      while (ndofs < N)
      {
        ++r;
        ndofs = num_pdofs(Nx, Nx, Nx, r, order);
      }
      while (ndofs > N)
      {
        --Nx;
        ndofs = num_pdofs(Nx, Nx, Nx, r, order);
      }
    }
    ndofs = num_pdofs(Nx, Nx, Nx, r, order);
  }
  Ny = Nx;
  Nz = Nx;
  uint64_t mindiff = 1000000; // :134
  const int64_t c = Nx;
  for (int64_t i = c - 10; i < c + 10; ++i) // :135-151
    for (int64_t j = i - 5; j < i + 5; ++j)
      for (int64_t k = i - 5; k < i + 5; ++k)
      {
        const int64_t d = num_pdofs(i, j, k, r, order) - N;
        const uint64_t diff = (uint64_t)(d < 0 ? -d : d);
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

const char* zzzh_last_error(void) { return g_err.c_str(); }

zzzh_part* zzzh_part_create(int problem, int order, int64_t nx, int64_t ny, int64_t nz, int nparts, int part)
{
  if (problem != ZZZH_POISSON && problem != ZZZH_ELASTICITY)
  {
    set_err("unknown problem %d", problem);
    return nullptr;
  }
  if (order < 1 || order > 3)
  {
    set_err("order %d not supported (1..3)", order); // form_*.at(order-1), src/poisson_problem.cpp:117
    return nullptr;
  }
  if (nx < 1 || ny < 1 || nz < 1 || nparts < 1 || part < 0 || part >= nparts)
  {
    set_err("bad mesh size %lldx%lldx%lld or partition %d/%d", (long long)nx, (long long)ny, (long long)nz, part, nparts);
    return nullptr;
  }
  if (nz < nparts)
  {
    set_err("z-slab partition needs nz >= number of parts (%lld < %d)", (long long)nz, nparts);
    return nullptr;
  }
  const Layout L(nx, ny, nz, order);
  const int bs = problem == ZZZH_ELASTICITY ? 3 : 1;
  const int nd = order == 1 ? 4 : order == 2 ? 10 : 20;
  const int npe = order - 1;
  const int64_t zs = nz * part / nparts, ze = nz * (part + 1) / nparts; // own layers [zs, ze)
  const bool lower = part > 0, upper = part < nparts - 1;
  const int64_t zl_end = upper ? ze + 1 : ze; // local layers [zs, zl_end)
  const int64_t own_lo = L.level_base(zs) + (lower ? L.NP : 0);
  const int64_t own_hi = L.level_base(ze) + L.NP;
  const int64_t n_owned = own_hi - own_lo;
  const int64_t n_lower = lower ? L.NP : 0;
  const int64_t n_upper = upper ? L.NL + L.NP : 0;
  const int64_t up_lo = L.level_base(ze) + L.NP; // first upper ghost (global)
  const int64_t nloc = n_owned + n_lower + n_upper;
  if (nloc * bs > INT32_MAX - 8)
  {
    set_err("partition has %lld scalar dofs: exceeds int32 local indexing, use more parts", (long long)(nloc * bs));
    return nullptr;
  }
  auto to_local = [&](int64_t g) -> int32_t {
    if (g >= own_lo && g < own_hi)
      return (int32_t)(g - own_lo);
    if (lower && g >= L.level_base(zs) && g < own_lo)
      return (int32_t)(n_owned + (g - L.level_base(zs)));
    return (int32_t)(n_owned + n_lower + (g - up_lo)); // upper ghost
  };

  zzzh_part* P = new zzzh_part();
  P->problem = problem;
  P->order = order;
  P->bs = bs;
  P->nd = nd;
  P->nparts = nparts;
  P->part = part;
  P->nx = nx;
  P->ny = ny;
  P->nz = nz;

  // geometry: vertex planes zs .. zl_end
  const int64_t nplanes = zl_end - zs + 1;
  const int64_t nverts = nplanes * L.PX * L.PY;
  P->x.resize((size_t)(3 * nverts));
  for (int64_t iz = zs; iz <= zl_end; ++iz)
    for (int64_t iy = 0; iy <= ny; ++iy)
      for (int64_t ix = 0; ix <= nx; ++ix)
      {
        const int64_t v = ((iz - zs) * L.PY + iy) * L.PX + ix;
        P->x[3 * v + 0] = (double)ix / (double)nx;
        P->x[3 * v + 1] = (double)iy / (double)ny;
        P->x[3 * v + 2] = (double)iz / (double)nz;
      }

  const int64_t ncells = 6 * nx * ny * (zl_end - zs);
  P->cells.resize((size_t)(4 * ncells));
  P->cell_dofs.resize((size_t)(nd * ncells));
  P->dof_x.assign((size_t)(3 * nloc), 0.0);
  P->global_dofs.assign((size_t)nloc, -1);
  const double tt[2] = {order == 2 ? 0.5 : 0.5 * (1.0 - 1.0 / std::sqrt(5.0)), 0.5 * (1.0 + 1.0 / std::sqrt(5.0))};
  const double nn[3] = {(double)nx, (double)ny, (double)nz};

  // Cell numbering: simplex-type major (all type-0 simplices of the slab in lexicographic sub-cube
  // order, then type 1, ...).  The row-gather kernels walk "the a-th cell of my dof" in lockstep over
  // 64 neighbouring dofs; with this numbering those 64 cells are consecutive in memory (one dense
  // 1-KiB read of the connectivity per wave instruction) instead of 96 B apart.
  const int64_t ncubes = nx * ny * (zl_end - zs);
  for (int64_t iz = zs; iz < zl_end; ++iz)
    for (int64_t iy = 0; iy < ny; ++iy)
      for (int64_t ix = 0; ix < nx; ++ix)
        for (int q = 0; q < 6; ++q)
        {
          const int64_t c = q * ncubes + ((iz - zs) * ny + iy) * nx + ix;
          int64_t p[4][3] = {{ix, iy, iz}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
          int step[3];
          for (int k = 0; k < 3; ++k)
          {
            step[k] = 1 << PERM[q][k];
            for (int a = 0; a < 3; ++a)
              p[k + 1][a] = p[k][a] + (a == PERM[q][k] ? 1 : 0);
          }
          // mask of the steps between path vertices a < b
          auto mask = [&](int a, int b) {
            int m = 0;
            for (int k = a; k < b; ++k)
              m |= step[k];
            return m;
          };
          int32_t* cv = &P->cells[(size_t)(4 * c)];
          int32_t* cd = &P->cell_dofs[(size_t)(nd * c)];
          int n = 0;
          for (int v = 0; v < 4; ++v)
          {
            cv[v] = (int32_t)(((p[v][2] - zs) * L.PY + p[v][1]) * L.PX + p[v][0]);
            const int64_t g = L.vertex(p[v]);
            const int32_t l = to_local(g);
            cd[n++] = l;
            P->global_dofs[l] = g;
            for (int a = 0; a < 3; ++a)
              P->dof_x[3 * (size_t)l + a] = (double)p[v][a] / nn[a];
          }
          if (order >= 2)
            for (int e = 0; e < 6; ++e)
            {
              const int a = EDGE_V[e][0], b = EDGE_V[e][1];
              const int m = mask(a, b);
              for (int s = 0; s < npe; ++s)
              {
                const int64_t g = L.edge(p[a], m, s);
                const int32_t l = to_local(g);
                cd[n++] = l;
                P->global_dofs[l] = g;
                for (int d = 0; d < 3; ++d)
                  P->dof_x[3 * (size_t)l + d] = ((double)p[a][d] + tt[s] * (double)(p[b][d] - p[a][d])) / nn[d];
              }
            }
          if (order == 3)
            for (int f = 0; f < 4; ++f)
            {
              const int a = FACE_V[f][0], b = FACE_V[f][1], cc = FACE_V[f][2];
              const int64_t g = L.face(p[a], mask(a, b), mask(b, cc));
              const int32_t l = to_local(g);
              cd[n++] = l;
              P->global_dofs[l] = g;
              for (int d = 0; d < 3; ++d)
                P->dof_x[3 * (size_t)l + d] = ((double)(p[a][d] + p[b][d] + p[cc][d]) / 3.0) / nn[d];
            }
          // exterior facets: all three vertices on one face of the cube
          for (int f = 0; f < 4; ++f)
          {
            bool ext = false;
            for (int d = 0; d < 3 && !ext; ++d)
            {
              const int64_t lim = d == 0 ? nx : (d == 1 ? ny : nz);
              const int64_t v0 = p[FACE_V[f][0]][d];
              if ((v0 == 0 || v0 == lim) && p[FACE_V[f][1]][d] == v0 && p[FACE_V[f][2]][d] == v0)
                ext = true;
            }
            if (ext)
            {
              P->facets.push_back((int32_t)c);
              P->facets.push_back(f);
            }
          }
        }
  {
    // exterior facets ordered by (cell, local facet)
    std::vector<std::pair<int32_t, int32_t>> fp(P->facets.size() / 2);
    for (size_t k = 0; k < fp.size(); ++k)
      fp[k] = {P->facets[2 * k], P->facets[2 * k + 1]};
    std::sort(fp.begin(), fp.end());
    for (size_t k = 0; k < fp.size(); ++k)
    {
      P->facets[2 * k] = fp[k].first;
      P->facets[2 * k + 1] = fp[k].second;
    }
  }
  for (int64_t l = 0; l < nloc; ++l)
    if (P->global_dofs[l] < 0)
    {
      set_err("internal error: local dof %lld not touched by any local cell", (long long)l);
      delete P;
      return nullptr;
    }

  // Dirichlet dofs (marker lambdas of src/poisson_problem.cpp:60-71, src/elasticity_problem.cpp:127-138
  // evaluated on the dof coordinates; on this mesh the closure of the marked facets is exactly that set)
  const double eps = 1.0e-8;
  for (int64_t l = 0; l < nloc; ++l)
  {
    const double* X = &P->dof_x[3 * (size_t)l];
    const bool m = problem == ZZZH_POISSON ? (std::abs(X[0]) < eps || std::abs(X[0] - 1) < eps) : (std::abs(X[1]) < eps);
    if (m)
      for (int k = 0; k < bs; ++k)
        P->bc_dofs.push_back((int32_t)(l * bs + k));
  }

  // coefficients (src/poisson_problem.cpp:85-106, src/elasticity_problem.cpp:154-176)
  P->coeff[0].resize((size_t)(nloc * bs));
  if (problem == ZZZH_POISSON)
  {
    P->coeff[1].resize((size_t)nloc);
    for (int64_t l = 0; l < nloc; ++l)
    {
      const double* X = &P->dof_x[3 * (size_t)l];
      const double dx = X[0] - 0.5, dy = X[1] - 0.5;
      const double dr = dx * dx + dy * dy;
      P->coeff[0][(size_t)l] = 10 * std::exp(-dr / 0.02);
      P->coeff[1][(size_t)l] = std::sin(5 * X[0]);
    }
  }
  else
    for (int64_t l = 0; l < nloc; ++l)
    {
      const double* X = &P->dof_x[3 * (size_t)l];
      const double dx = X[0] - 0.5, dz = X[2] - 0.5;
      const double r = std::sqrt(dx * dx + dz * dz);
      P->coeff[0][3 * (size_t)l + 0] = -dz * r * X[1];
      P->coeff[0][3 * (size_t)l + 1] = 1.0;
      P->coeff[0][3 * (size_t)l + 2] = dx * r * X[1];
    }

  // forward-scatter plan: neighbours in ghost order (lower, then upper)
  P->send_off.push_back(0);
  if (lower)
  {
    // rank part-1's upper ghosts = my layer zs and plane zs+1 = my first NL+NP owned dofs
    P->neigh.push_back(part - 1);
    for (int64_t i = 0; i < L.NL + L.NP; ++i)
      P->send_idx.push_back((int32_t)i);
    P->send_off.push_back((int64_t)P->send_idx.size());
    P->recv_cnt.push_back(n_lower);
  }
  if (upper)
  {
    // rank part+1's lower ghosts = plane ze = my last NP owned dofs
    P->neigh.push_back(part + 1);
    for (int64_t i = n_owned - L.NP; i < n_owned; ++i)
      P->send_idx.push_back((int32_t)i);
    P->send_off.push_back((int64_t)P->send_idx.size());
    P->recv_cnt.push_back(n_upper);
  }

  int64_t* S = P->sizes;
  S[ZZZH_NVERTS] = nverts;
  S[ZZZH_NCELLS] = ncells;
  S[ZZZH_NOWNED] = n_owned;
  S[ZZZH_NGHOST] = n_lower + n_upper;
  S[ZZZH_ND] = nd;
  S[ZZZH_BS] = bs;
  S[ZZZH_NFACETS] = (int64_t)P->facets.size() / 2;
  S[ZZZH_NBC] = (int64_t)P->bc_dofs.size();
  S[ZZZH_NNEIGH] = (int64_t)P->neigh.size();
  S[ZZZH_NSEND] = (int64_t)P->send_idx.size();
  S[ZZZH_GLOBAL_DOFS] = L.total() * bs;
  S[ZZZH_GLOBAL_CELLS] = 6 * nx * ny * nz;
  S[ZZZH_OWNED_CELLS] = 6 * nx * ny * (ze - zs);
  S[ZZZH_OWN_OFFSET] = own_lo;
  return P;
}

void zzzh_part_destroy(zzzh_part* p) { delete p; }
void zzzh_part_sizes(const zzzh_part* p, int64_t sizes[ZZZH_NSIZES])
{
  for (int i = 0; i < ZZZH_NSIZES; ++i)
    sizes[i] = p->sizes[i];
}
const double* zzzh_part_x(const zzzh_part* p) { return p->x.data(); }
const int32_t* zzzh_part_cells(const zzzh_part* p) { return p->cells.data(); }
const int32_t* zzzh_part_cell_dofs(const zzzh_part* p) { return p->cell_dofs.data(); }
const int32_t* zzzh_part_facets(const zzzh_part* p) { return p->facets.data(); }
const int32_t* zzzh_part_bc_dofs(const zzzh_part* p) { return p->bc_dofs.data(); }
const double* zzzh_part_dof_x(const zzzh_part* p) { return p->dof_x.data(); }
const int64_t* zzzh_part_global_dofs(const zzzh_part* p) { return p->global_dofs.data(); }
const double* zzzh_part_coeff(const zzzh_part* p, int which) { return (which == 0 || which == 1) ? p->coeff[which].data() : nullptr; }
const int32_t* zzzh_part_neigh(const zzzh_part* p) { return p->neigh.data(); }
const int64_t* zzzh_part_send_off(const zzzh_part* p) { return p->send_off.data(); }
const int32_t* zzzh_part_send_idx(const zzzh_part* p) { return p->send_idx.data(); }
const int64_t* zzzh_part_recv_cnt(const zzzh_part* p) { return p->recv_cnt.data(); }

} // extern "C"

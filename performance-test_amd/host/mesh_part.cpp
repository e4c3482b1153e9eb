// Host-side feed of the hot path: see include/zzz_host.h for what each piece replaces in the
// reference.  Pure C++; no GPU, no oracle.  Everything is closed-form on the structured Kuhn
// triangulation, so a rank generates only its own z-slab (O(local) time and memory).
//
// Global dof numbering ("level units"): for k = 0..nz the dofs of plane z = k (vertices, in-plane
// edges, in-plane faces) are followed by those of layer k (entities spanning z = k..k+1).  A z-slab
// partition then owns one contiguous global range, ghost blocks arrive already in ghost order
// (no unpack kernel), and rows that are neighbours in the mesh are neighbours in memory.
#include "../../include/zzz_host.h"
#include "cube_layout.h"
#include "part_struct.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <utility>
#include <vector>

static thread_local std::string g_err;
namespace
{

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

} // namespace

void zzzh_set_error(const char* msg) { g_err = msg; }

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
  // :135-151.  The bound is the reference's: `i < Nx + 10` on the live Nx (the body overwrites it), so the sweep
  // stops 10 past the best i found so far; only the start is the cubic guess minus 10.
  for (int64_t i = Nx - 10; i < Nx + 10; ++i)
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

int zzzh_count_suffix(int64_t n, char* out, int cap)
{
  static const char* const unit[] = {"thousand", "million", "billion", "trillion"};
  if (!out || cap < 1)
    return -1;
  out[0] = 0;
  double scaled = static_cast<double>(n);
  int group = -1; // how many times a factor 1000 was taken out
  for (; scaled > 1000.0; scaled /= 1000.0)
    ++group;
  if (group < 0)
    return 0;
  if (group > 3)
    return -1;
  // %.3g is what an ostream prints with setprecision(3) in its default float format
  const int len = std::snprintf(out, (size_t)cap, " (%.3g %s)", scaled, unit[group]);
  return len < cap ? len : cap - 1;
}

const char* zzzh_last_error(void) { return g_err.c_str(); }

static zzzh_part* part_create(int problem, int order, int64_t nx, int64_t ny, int64_t nz, int nparts, int part, bool native);

zzzh_part* zzzh_part_create(int problem, int order, int64_t nx, int64_t ny, int64_t nz, int nparts, int part)
{
  return part_create(problem, order, nx, ny, nz, nparts, part, false);
}
zzzh_part* zzzh_part_create_native(int problem, int order, int64_t nx, int64_t ny, int64_t nz, int nparts, int part)
{
  return part_create(problem, order, nx, ny, nz, nparts, part, true);
}

static zzzh_part* part_create(int problem, int order, int64_t nx, int64_t ny, int64_t nz, int nparts, int part, bool native)
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
  const int bs = problem == ZZZH_ELASTICITY ? 3 : 1;
  const zzzcube::Slab S(nx, ny, nz, order, bs, nparts, part, native);
  const zzzcube::Layout& L = S.L;
  const int nd = S.nd;
  const int64_t zs = S.zs, ze = S.ze, zl_end = S.zl_end;
  const bool lower = S.lower, upper = S.upper;
  const int64_t n_owned = S.n_owned, n_lower = S.n_lower, n_upper = S.n_upper, nloc = S.nloc, own_lo = S.own_lo;
  if (nloc * bs > INT32_MAX - 8)
  {
    set_err("partition has %lld scalar dofs: exceeds int32 local indexing, use more parts", (long long)(nloc * bs));
    return nullptr;
  }

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
  const int64_t nverts = S.nverts;
  P->x.resize((size_t)(3 * nverts));
  P->global_verts.resize((size_t)nverts);
  for (int64_t iz = zs; iz <= zl_end; ++iz)
    for (int64_t iy = 0; iy <= ny; ++iy)
      for (int64_t ix = 0; ix <= nx; ++ix)
      {
        const int64_t a[3] = {ix, iy, iz};
        const int64_t v = S.local_vertex(a);
        P->x[3 * v + 0] = (double)ix / (double)nx;
        P->x[3 * v + 1] = (double)iy / (double)ny;
        P->x[3 * v + 2] = (double)iz / (double)nz;
        P->global_verts[(size_t)v] = (iz * L.PY + iy) * L.PX + ix;
      }

  const int64_t ncells = S.ncells;
  P->cells.resize((size_t)(4 * ncells));
  P->cell_dofs.resize((size_t)(nd * ncells));
  P->dof_x.assign((size_t)(3 * nloc), 0.0);
  P->global_dofs.assign((size_t)nloc, -1);
  // cells: closed form per (sub-cube, simplex type), see cube_layout.h (shared with the device generator)
  for (int64_t iz = zs; iz < zl_end; ++iz)
    for (int64_t iy = 0; iy < ny; ++iy)
      for (int64_t ix = 0; ix < nx; ++ix)
        for (int q = 0; q < 6; ++q)
        {
          const int64_t c = q * S.ncubes + ((iz - zs) * ny + iy) * nx + ix;
          zzzcube::Cell C;
          zzzcube::make_cell(S, ix, iy, iz, q, C);
          for (int v = 0; v < 4; ++v)
            P->cells[(size_t)(4 * c + v)] = C.verts[v];
          for (int i = 0; i < nd; ++i)
          {
            const int32_t l = C.dofs[i];
            P->cell_dofs[(size_t)(nd * c + i)] = l;
            P->global_dofs[l] = C.gdofs[i];
            for (int a = 0; a < 3; ++a)
              P->dof_x[3 * (size_t)l + a] = C.dof_x[i][a];
          }
          for (int f = 0; f < 4; ++f)
            if ((C.facet_mask >> f) & 1u)
            {
              P->facets.push_back((int32_t)c);
              P->facets.push_back(f);
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

  // Dirichlet dofs: the marker lambdas of src/poisson_problem.cpp:60-71, src/elasticity_problem.cpp:127-138
  // evaluated on the dof coordinates (on this mesh the closure of the marked facets is exactly that set)
  for (int64_t l = 0; l < nloc; ++l)
    if (zzzcube::is_dirichlet(problem, &P->dof_x[3 * (size_t)l]))
      for (int k = 0; k < bs; ++k)
        P->bc_dofs.push_back((int32_t)(l * bs + k));

  // coefficients (src/poisson_problem.cpp:85-106, src/elasticity_problem.cpp:154-176)
  P->coeff[0].resize((size_t)(nloc * bs));
  if (problem == ZZZH_POISSON)
  {
    P->coeff[1].resize((size_t)nloc);
    for (int64_t l = 0; l < nloc; ++l)
    {
      P->coeff[0][(size_t)l] = zzzcube::poisson_f(&P->dof_x[3 * (size_t)l]);
      P->coeff[1][(size_t)l] = zzzcube::poisson_g(&P->dof_x[3 * (size_t)l]);
    }
  }
  else
    for (int64_t l = 0; l < nloc; ++l)
      zzzcube::elasticity_f(&P->dof_x[3 * (size_t)l], &P->coeff[0][3 * (size_t)l]);

  // forward-scatter plan: neighbours in ghost order (lower, then upper)
  P->send_off.push_back(0);
  if (lower)
  {
    // rank part-1's upper ghosts = my layer zs and plane zs+1 = my first NL+NP owned dofs (native: it has none)
    P->neigh.push_back(part - 1);
    for (int64_t i = 0; i < (native ? 0 : L.NL + L.NP); ++i)
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

  int64_t* Sz = P->sizes;
  Sz[ZZZH_NVERTS] = nverts;
  Sz[ZZZH_NCELLS] = ncells;
  Sz[ZZZH_NOWNED] = n_owned;
  Sz[ZZZH_NGHOST] = n_lower + n_upper;
  Sz[ZZZH_ND] = nd;
  Sz[ZZZH_BS] = bs;
  Sz[ZZZH_NFACETS] = (int64_t)P->facets.size() / 2;
  Sz[ZZZH_NBC] = (int64_t)P->bc_dofs.size();
  Sz[ZZZH_NNEIGH] = (int64_t)P->neigh.size();
  Sz[ZZZH_NSEND] = (int64_t)P->send_idx.size();
  Sz[ZZZH_GLOBAL_DOFS] = L.total() * bs;
  Sz[ZZZH_GLOBAL_CELLS] = 6 * nx * ny * nz;
  Sz[ZZZH_OWNED_CELLS] = 6 * nx * ny * (ze - zs);
  Sz[ZZZH_OWN_OFFSET] = own_lo;
  // the marked planes hold every point of the order-k lattice: x = 0 and x = 1 (Poisson), y = 0 (elasticity, 3 components)
  Sz[ZZZH_GLOBAL_NBC] = problem == ZZZH_POISSON ? 2 * (order * ny + 1) * (order * nz + 1) : 3 * (order * nx + 1) * (order * nz + 1);
  Sz[ZZZH_BC_MODE] = 0;
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
const int64_t* zzzh_part_global_verts(const zzzh_part* p) { return p->global_verts.data(); }
const double* zzzh_part_coeff(const zzzh_part* p, int which) { return (which == 0 || which == 1) ? p->coeff[which].data() : nullptr; }
const int32_t* zzzh_part_neigh(const zzzh_part* p) { return p->neigh.data(); }
const int64_t* zzzh_part_send_off(const zzzh_part* p) { return p->send_off.data(); }
const int32_t* zzzh_part_send_idx(const zzzh_part* p) { return p->send_idx.data(); }
const int64_t* zzzh_part_recv_cnt(const zzzh_part* p) { return p->recv_cnt.data(); }

} // extern "C"

// Device-side generation of the structured problem feed: what
// create_cube_mesh (src/mesh.cpp:78-206), fem::create_functionspace (src/poisson_problem.cpp:35-44),
// locate_dofs_topological + DirichletBC (:53-77) and Function::interpolate (:83-106) hand to the
// assembly -- geometry, connectivity, dofmap, exterior-facet mask, Dirichlet marker, nodal
// coefficients -- written straight into HBM by closed-form kernels instead of being built on the
// host and copied over PCIe.  The closed forms live in host/cube_layout.h and are shared with the
// host generator (host/mesh_part.cpp), so both feeds are identical integer for integer and the
// coordinates bit for bit; only exp()/sin() of the coefficients may differ in the last ulp.
#include "zzz_internal.h"

#include "../host/cube_layout.h"

namespace zzz
{
using zzzcube::Slab;

__global__ void k_cube_vertices(Slab S, double* __restrict__ x)
{
  const int64_t PX = S.L.PX, PY = S.L.PY;
  for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < S.nverts; v += (int64_t)gridDim.x * blockDim.x)
  {
    const int64_t ix = v % PX, iy = (v / PX) % PY, iz = v / (PX * PY) + S.zs;
    x[3 * v + 0] = (double)ix / (double)S.L.nx;
    x[3 * v + 1] = (double)iy / (double)S.L.ny;
    x[3 * v + 2] = (double)iz / (double)S.L.nz;
  }
}

__global__ __launch_bounds__(256) void k_cube_cells(Slab S, int32_t* __restrict__ cells, int32_t* __restrict__ cell_dofs,
                                                    uint8_t* __restrict__ facet_mask, double* __restrict__ dof_x)
{
  const int64_t nx = S.L.nx, ny = S.L.ny;
  for (int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; c < S.ncells; c += (int64_t)gridDim.x * blockDim.x)
  {
    const int q = (int)(c / S.ncubes);
    const int64_t cube = c - (int64_t)q * S.ncubes;
    const int64_t ix = cube % nx, iy = (cube / nx) % ny, iz = cube / (nx * ny) + S.zs;
    zzzcube::Cell C;
    zzzcube::make_cell(S, ix, iy, iz, q, C);
    for (int v = 0; v < 4; ++v)
      cells[4 * c + v] = C.verts[v];
    for (int i = 0; i < S.nd; ++i)
    {
      const int32_t l = C.dofs[i];
      cell_dofs[(int64_t)S.nd * c + i] = l;
      // every cell that holds the dof writes the same bits (same lattice formula): benign
      dof_x[3 * (int64_t)l + 0] = C.dof_x[i][0];
      dof_x[3 * (int64_t)l + 1] = C.dof_x[i][1];
      dof_x[3 * (int64_t)l + 2] = C.dof_x[i][2];
    }
    facet_mask[c] = (uint8_t)C.facet_mask;
  }
}

__global__ void k_cube_dofs(int problem, int bs, int64_t nloc, const double* __restrict__ dof_x, uint8_t* __restrict__ bc,
                            double* __restrict__ f, double* __restrict__ g)
{
  for (int64_t l = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; l < nloc; l += (int64_t)gridDim.x * blockDim.x)
  {
    const double X[3] = {dof_x[3 * l], dof_x[3 * l + 1], dof_x[3 * l + 2]};
    const uint8_t m = zzzcube::is_dirichlet(problem, X) ? 1 : 0;
    for (int k = 0; k < bs; ++k)
      bc[l * bs + k] = m;
    if (problem == 0)
    {
      f[l] = zzzcube::poisson_f(X);
      g[l] = zzzcube::poisson_g(X);
    }
    else
    {
      double o[3];
      zzzcube::elasticity_f(X, o);
      f[3 * l + 0] = o[0];
      f[3 * l + 1] = o[1];
      f[3 * l + 2] = o[2];
    }
  }
}

static int gridfor(int64_t n)
{
  int64_t g = (n + 255) / 256;
  if (g > 8192)
    g = 8192;
  if (g < 1)
    g = 1;
  return (int)g;
}
ZZZ_PRELOAD_TU(cubegen)
} // namespace zzz

using namespace zzz;

extern "C" int zzz_cube_generate(zzz_ctx* ctx, int problem, int order, int64_t nx, int64_t ny, int64_t nz, int nparts,
                                 int part, int64_t* info)
{
  if (!ctx)
    return fail(nullptr, ZZZ_ERR_ARG, "NULL context");
  ZZZ_HIP(ctx, hipSetDevice(ctx->device));
  if (problem != ZZZ_FORM_POISSON && problem != ZZZ_FORM_ELASTICITY)
    return fail(ctx, ZZZ_ERR_ARG, "unknown problem %d", problem);
  if (order < 1 || order > 3) // form_*.at(order-1), src/poisson_problem.cpp:117
    return fail(ctx, ZZZ_ERR_ARG, "order %d not supported (1..3)", order);
  if (nx < 1 || ny < 1 || nz < 1 || nparts < 1 || part < 0 || part >= nparts)
    return fail(ctx, ZZZ_ERR_ARG, "bad mesh size %lldx%lldx%lld or partition %d/%d", (long long)nx, (long long)ny,
                (long long)nz, part, nparts);
  if (nz < nparts)
    return fail(ctx, ZZZ_ERR_ARG, "z-slab partition needs nz >= number of parts (%lld < %d)", (long long)nz, nparts);
  const int bs = problem == ZZZ_FORM_ELASTICITY ? 3 : 1;
  const Slab S(nx, ny, nz, order, bs, nparts, part);
  if (S.nloc * bs > INT32_MAX - 8 || S.ncells * S.nd > INT32_MAX - 8 || S.nverts > INT32_MAX / 4)
    return fail(ctx, ZZZ_ERR_LIMIT, "partition too large for int32 local indexing (%lld scalar dofs, %lld cells): use more parts",
                (long long)(S.nloc * bs), (long long)S.ncells);
  hipStream_t s = ctx->stream;
  ctx->nverts = S.nverts;
  ctx->ncells = S.ncells;
  ctx->order = order;
  ctx->bs = bs;
  ctx->nd = S.nd;
  ctx->n_owned = S.n_owned;
  ctx->n_ghost = S.n_lower + S.n_upper;
  ctx->h_cell_verts.clear();
  ctx->h_cell_dofs.clear();
  renumber_clear(ctx); // this feed is generated in the internal order (host/cube_layout.h)
  ZZZ_HIP(ctx, ctx->x.alloc((size_t)(3 * S.nverts)));
  ZZZ_HIP(ctx, ctx->cell_verts.alloc((size_t)(4 * S.ncells)));
  ZZZ_HIP(ctx, ctx->cell_dofs.alloc((size_t)(S.nd * S.ncells)));
  ZZZ_HIP(ctx, ctx->facet_mask.alloc((size_t)S.ncells));
  int rc = alloc_problem_vectors(ctx);
  if (rc)
    return rc;
  DevBuf<double> dof_x;
  ZZZ_HIP(ctx, dof_x.alloc((size_t)(3 * S.nloc)));
  hipLaunchKernelGGL(k_cube_vertices, dim3(gridfor(S.nverts)), dim3(256), 0, s, S, ctx->x.p);
  hipLaunchKernelGGL(k_cube_cells, dim3(gridfor(S.ncells)), dim3(256), 0, s, S, ctx->cell_verts.p, ctx->cell_dofs.p,
                     ctx->facet_mask.p, dof_x.p);
  hipLaunchKernelGGL(k_cube_dofs, dim3(gridfor(S.nloc)), dim3(256), 0, s, problem, bs, S.nloc, dof_x.p, ctx->bc.p,
                     ctx->coeff[0].p, ctx->coeff[1].p);
  ZZZ_HIP(ctx, hipGetLastError());
  ZZZ_HIP(ctx, hipStreamSynchronize(s));
  ctx->nfacets = -1; // not counted: the mask is what the kernels read
  ctx->have_bc = true;
  ctx->have_coeff[0] = true;
  ctx->have_coeff[1] = problem == ZZZ_FORM_POISSON;
  ctx->have_pattern = ctx->have_matrix = false;
  ctx->xq_valid = false;
  ctx->mf.valid = ctx->mf.failed = false;
  ctx->adj_runs_n = -1;
  pattern_reserve(ctx);
  rc = ensure_p1_coords(ctx);
  if (rc)
    return rc;

  // forward-scatter plan (neighbours in ghost order: lower, then upper), same as host/mesh_part.cpp
  std::vector<int32_t> neigh, send_idx;
  std::vector<int64_t> send_off(1, 0), recv_cnt;
  if (S.lower)
  {
    neigh.push_back(part - 1);
    for (int64_t i = 0; i < S.L.NL + S.L.NP; ++i)
      send_idx.push_back((int32_t)i);
    send_off.push_back((int64_t)send_idx.size());
    recv_cnt.push_back(S.n_lower);
  }
  if (S.upper)
  {
    neigh.push_back(part + 1);
    for (int64_t i = S.n_owned - S.L.NP; i < S.n_owned; ++i)
      send_idx.push_back((int32_t)i);
    send_off.push_back((int64_t)send_idx.size());
    recv_cnt.push_back(S.n_upper);
  }
  const int32_t zero32 = 0;
  const int64_t zero64 = 0;
  rc = zzz_halo_upload(ctx, (int)neigh.size(), neigh.empty() ? &zero32 : neigh.data(), send_off.data(),
                       send_idx.empty() ? &zero32 : send_idx.data(), recv_cnt.empty() ? &zero64 : recv_cnt.data());
  if (rc)
    return rc;
  if (info)
  {
    info[0] = S.L.total() * bs;      // index_map.size_global() * bs (src/main.cpp:178-180)
    info[1] = 6 * nx * ny * nz;      // global cells
    info[2] = S.n_owned;             // owned block dofs
    info[3] = S.n_lower + S.n_upper; // ghost block dofs
    info[4] = S.own_lo;              // global block index of local dof 0
    info[5] = S.ncells;              // local cells (own layers + ghost layer)
  }
  return ZZZ_OK;
}

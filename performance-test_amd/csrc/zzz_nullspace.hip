// `ZZZ Create near-nullspace` (src/elasticity_problem.cpp:36-94, called at :233-244): the six rigid-body modes of the
// vector-valued space as la::Vector basis -- translations e_0, e_1, e_2 and the rotations (-x1, x0, 0), (x2, 0, -x0),
// (0, -x2, x1) at the dof coordinates -- orthonormalised by la::orthonormalize (modified Gram-Schmidt in basis order:
// x_i -= <x_i, x_k> x_k for k < i, then x_i /= |x_i|; inner products over the OWNED entries, summed over the ranks) and
// checked with la::is_orthonormal ("Space not orthonormal" otherwise).  The reference hands the result to
// MatSetNearNullSpace for GAMG; with Jacobi-CG nothing consumes it, the phase is built for the surface and for parity.
//
// Dof coordinates (V.tabulate_dof_coordinates): the library keeps vertex coordinates only, so every (cell, local dof)
// pushes its reference node through the cell's affine map (the stores of a dof shared by several cells carry the same
// value up to round-off of the map; the LOWEST cell wins, deterministically, through a first pass that records it).
#include <algorithm>
#include <cmath>
#include <vector>

#include "zzz_device.h"
#include "zzz_internal.h"

namespace zzz
{
namespace
{
// barycentric coordinates (weights of the cell's vertices 0..3) of the gll_warped Lagrange nodes in Basix's local order
// (include/zzz_abi.h): vertices; edges e0 = (2,3), e1 = (1,3), e2 = (1,2), e3 = (0,3), e4 = (0,2), e5 = (0,1); faces opposite
// vertex f.  P3 edge nodes at t = (1 -+ 1/sqrt 5) / 2 from the edge's first to its second vertex.
__device__ inline void ref_node(int order, int i, double w[4])
{
  const int EV[6][2] = {{2, 3}, {1, 3}, {1, 2}, {0, 3}, {0, 2}, {0, 1}};
  w[0] = w[1] = w[2] = w[3] = 0.0;
  if (i < 4)
  {
    w[i] = 1.0;
    return;
  }
  if (order == 2)
  {
    const int e = i - 4;
    w[EV[e][0]] = 0.5;
    w[EV[e][1]] = 0.5;
    return;
  }
  if (i < 16)
  {
    const int e = (i - 4) / 2, k = (i - 4) % 2;
    const double t = k == 0 ? 0.5 * (1.0 - 1.0 / sqrt(5.0)) : 0.5 * (1.0 + 1.0 / sqrt(5.0));
    w[EV[e][0]] = 1.0 - t;
    w[EV[e][1]] = t;
    return;
  }
  const int f = i - 16;
  for (int k = 0; k < 4; ++k)
    w[k] = k == f ? 0.0 : 1.0 / 3.0;
}

__global__ __launch_bounds__(256) void k_nn_first_cell(const int32_t* __restrict__ cell_dofs, int nd, int64_t ncells,
                                                       int32_t* __restrict__ first)
{
  for (int64_t t = blockIdx.x * 256ll + threadIdx.x; t < ncells * nd; t += gridDim.x * 256ll)
    atomicMin(&first[cell_dofs[t]], (int32_t)(t / nd));
}
__global__ __launch_bounds__(256) void k_nn_dof_coords(const double* __restrict__ x, const int32_t* __restrict__ cell_verts,
                                                       const int32_t* __restrict__ cell_dofs, int order, int nd, int64_t ncells,
                                                       const int32_t* __restrict__ first, double* __restrict__ dofx)
{
  for (int64_t t = blockIdx.x * 256ll + threadIdx.x; t < ncells * nd; t += gridDim.x * 256ll)
  {
    const int64_t c = t / nd;
    const int i = (int)(t - c * nd);
    const int32_t d = cell_dofs[t];
    if (first[d] != (int32_t)c)
      continue;
    double w[4];
    ref_node(order, i, w);
    for (int a = 0; a < 3; ++a)
    {
      double s = 0.0;
      for (int k = 0; k < 4; ++k)
        s += w[k] * x[3ll * cell_verts[4 * c + k] + a];
      dofx[3ll * d + a] = s;
    }
  }
}
// the six modes before orthonormalisation (src/elasticity_problem.cpp:42-71), over owned and ghost block dofs
__global__ __launch_bounds__(256) void k_nn_modes(const double* __restrict__ dofx, int64_t nblock, double* __restrict__ B, int64_t ld)
{
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < nblock; i += gridDim.x * 256ll)
  {
    const double x0 = dofx[3 * i], x1 = dofx[3 * i + 1], x2 = dofx[3 * i + 2];
    for (int k = 0; k < 6; ++k)
      for (int c = 0; c < 3; ++c)
        B[k * ld + 3 * i + c] = 0.0;
    B[0 * ld + 3 * i + 0] = 1.0;
    B[1 * ld + 3 * i + 1] = 1.0;
    B[2 * ld + 3 * i + 2] = 1.0;
    B[3 * ld + 3 * i + 0] = -x1;
    B[3 * ld + 3 * i + 1] = x0;
    B[4 * ld + 3 * i + 0] = x2;
    B[4 * ld + 3 * i + 2] = -x0;
    B[5 * ld + 3 * i + 2] = x1;
    B[5 * ld + 3 * i + 1] = -x2;
  }
}
__global__ __launch_bounds__(256) void k_nn_dot(const double* __restrict__ a, const double* __restrict__ b, int64_t n,
                                                double* __restrict__ parts)
{
  __shared__ double sh[4];
  double s = 0.0;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += gridDim.x * 256ll)
    s += a[i] * b[i];
  const double t = block_reduce_sum(s, sh);
  if (threadIdx.x == 0)
    parts[blockIdx.x] = t;
}
// y = alpha x + y (alpha = -<y, x>), or y *= alpha when x is null; over the whole array, ghosts included
__global__ __launch_bounds__(256) void k_nn_axpy(double alpha, const double* __restrict__ x, double* __restrict__ y, int64_t n)
{
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += gridDim.x * 256ll)
    y[i] = x ? alpha * x[i] + y[i] : alpha * y[i];
}

constexpr int NN_GRID = 512;

int nn_dot(zzz_ctx* ctx, const double* a, const double* b, int64_t n_owned_scalars, DevBuf<double>& parts, double* out)
{
  hipLaunchKernelGGL(k_nn_dot, dim3(NN_GRID), dim3(256), 0, ctx->stream, a, b, n_owned_scalars, parts.p);
  std::vector<double> h(NN_GRID);
  ZZZ_HIP(ctx, hipMemcpyAsync(h.data(), parts.p, NN_GRID * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  ZZZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  double s = 0.0;
  for (double v : h)
    s += v;
  if (ctx->comm) // MPI_Allreduce of la::inner_product
  {
    ZZZ_HIP(ctx, hipMemcpyAsync(parts.p, &s, sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    if (int rc = comm_allreduce_sum(ctx, parts.p, 1))
      return rc;
    ZZZ_HIP(ctx, hipMemcpyAsync(&s, parts.p, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    ZZZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  *out = s;
  return ZZZ_OK;
}
} // namespace
ZZZ_PRELOAD_TU(nullspace)
} // namespace zzz

namespace zzz
{
// coordinates of every block dof (owned + ghost), [dof][3]: V.tabulate_dof_coordinates as described above.  Also used by the
// block-window product (zzz_sellp_win.hip) to order rows in space.
int dof_coords_device(zzz_ctx* ctx, DevBuf<double>& dofx, DevBuf<int32_t>& first)
{
  hipStream_t s = ctx->stream;
  const int64_t nblock = ctx->n_owned + ctx->n_ghost, nc = ctx->ncells;
  const int g = (int)std::max<int64_t>(1, std::min<int64_t>((nc * ctx->nd + 255) / 256, 8192));
  ZZZ_HIP(ctx, first.alloc((size_t)nblock));
  ZZZ_HIP(ctx, dofx.alloc((size_t)(3 * nblock)));
  ZZZ_HIP(ctx, hipMemsetAsync(first.p, 0x7f, (size_t)nblock * sizeof(int32_t), s));
  ZZZ_HIP(ctx, hipMemsetAsync(dofx.p, 0, (size_t)(3 * nblock) * sizeof(double), s));
  hipLaunchKernelGGL(k_nn_first_cell, dim3(g), dim3(256), 0, s, ctx->cell_dofs.p, ctx->nd, nc, first.p);
  hipLaunchKernelGGL(k_nn_dof_coords, dim3(g), dim3(256), 0, s, ctx->x.p, ctx->cell_verts.p, ctx->cell_dofs.p, ctx->order, ctx->nd, nc,
                     first.p, dofx.p);
  ZZZ_HIP(ctx, hipGetLastError());
  return ZZZ_OK;
}
} // namespace zzz

using namespace zzz;

extern "C" {

int zzz_near_nullspace_build(zzz_ctx* ctx, double* max_deviation)
{
  if (!ctx)
    return fail(nullptr, ZZZ_ERR_ARG, "NULL context");
  ZZZ_HIP(ctx, hipSetDevice(ctx->device));
  if (ctx->order == 0 || ctx->bs != 3)
    return fail(ctx, ZZZ_ERR_ARG, "zzz_near_nullspace_build: needs the vector-valued space of the elasticity problem (block size 3)");
  hipStream_t s = ctx->stream;
  const int64_t nblock = ctx->n_owned + ctx->n_ghost, ld = 3 * nblock, nown = 3 * ctx->n_owned;
  const int64_t nc = ctx->ncells;
  const int g = (int)std::min<int64_t>((nc * ctx->nd + 255) / 256, 8192);
  DevBuf<int32_t> first;
  DevBuf<double> dofx, parts;
  ZZZ_HIP(ctx, first.alloc((size_t)nblock));
  ZZZ_HIP(ctx, dofx.alloc((size_t)(3 * nblock)));
  ZZZ_HIP(ctx, parts.alloc(NN_GRID));
  ZZZ_HIP(ctx, ctx->near_null.alloc((size_t)(6 * ld)));
  ZZZ_HIP(ctx, hipMemsetAsync(first.p, 0x7f, (size_t)nblock * sizeof(int32_t), s));
  ZZZ_HIP(ctx, hipMemsetAsync(dofx.p, 0, (size_t)(3 * nblock) * sizeof(double), s));
  hipLaunchKernelGGL(k_nn_first_cell, dim3(g), dim3(256), 0, s, ctx->cell_dofs.p, ctx->nd, nc, first.p);
  hipLaunchKernelGGL(k_nn_dof_coords, dim3(g), dim3(256), 0, s, ctx->x.p, ctx->cell_verts.p, ctx->cell_dofs.p, ctx->order, ctx->nd, nc,
                     first.p, dofx.p);
  double* B = ctx->near_null.p;
  const int gb = (int)std::min<int64_t>((nblock + 255) / 256, 4096), gv = (int)std::min<int64_t>((ld + 255) / 256, 4096);
  hipLaunchKernelGGL(k_nn_modes, dim3(gb), dim3(256), 0, s, dofx.p, nblock, B, ld);
  ZZZ_HIP(ctx, hipGetLastError());
  // la::orthonormalize
  for (int i = 0; i < 6; ++i)
  {
    for (int k = 0; k < i; ++k)
    {
      double d = 0.0;
      if (int rc = nn_dot(ctx, B + i * ld, B + k * ld, nown, parts, &d))
        return rc;
      hipLaunchKernelGGL(k_nn_axpy, dim3(gv), dim3(256), 0, s, -d, B + k * ld, B + i * ld, ld);
    }
    double nn = 0.0;
    if (int rc = nn_dot(ctx, B + i * ld, B + i * ld, nown, parts, &nn))
      return rc;
    hipLaunchKernelGGL(k_nn_axpy, dim3(gv), dim3(256), 0, s, 1.0 / std::sqrt(nn), (const double*)nullptr, B + i * ld, ld);
  }
  // la::is_orthonormal: |<x_i, x_j> - delta_ij| beyond the tolerance => "Space not orthonormal" (src/elasticity_problem.cpp:76-81)
  double dev = 0.0;
  for (int i = 0; i < 6; ++i)
    for (int k = 0; k <= i; ++k)
    {
      double d = 0.0;
      if (int rc = nn_dot(ctx, B + i * ld, B + k * ld, nown, parts, &d))
        return rc;
      dev = std::max(dev, std::abs(d - (i == k ? 1.0 : 0.0)));
    }
  if (max_deviation)
    *max_deviation = dev;
  ctx->near_null_ld = ld;
  if (!(dev <= 1.0e-10)) // [EXT] dolfinx::la::is_orthonormal's default tolerance is the scalar type's epsilon scaled; 1e-10 here
    return fail(ctx, ZZZ_ERR_ARG, "Space not orthonormal (largest deviation %g)", dev);
  return ZZZ_OK;
}

int zzz_near_nullspace_download(zzz_ctx* ctx, int k, double* out)
{
  if (!ctx || !out || k < 0 || k > 5 || ctx->near_null_ld == 0)
    return fail(ctx, ZZZ_ERR_ARG, "zzz_near_nullspace_download: no basis built, or bad arguments");
  ZZZ_HIP(ctx, hipSetDevice(ctx->device));
  const size_t n = (size_t)(3 * ctx->n_owned);
  std::vector<double> tmp(n);
  ZZZ_HIP(ctx, hipMemcpy(tmp.data(), ctx->near_null.p + (size_t)k * ctx->near_null_ld, n * sizeof(double), hipMemcpyDeviceToHost));
  if (ctx->renumbered)
    to_caller(ctx, tmp.data(), out, true);
  else
    std::copy(tmp.begin(), tmp.end(), out);
  return ZZZ_OK;
}

} // extern "C"

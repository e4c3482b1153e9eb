// CSR SpMV for gfx950: y = A x, fp64 values / int32 columns (K6 of SURVEY.md 2c).
//
// Replaces PETSc MatMult inside KSPSolve (src/poisson_problem.cpp:177) and the `action` of
// linalg::cg (src/cg.h:62).  HBM-bound: 12 B per nonzero streamed once, x gathered through L2.
//
// Layout of the work: the nonzero stream is cut into row-aligned tiles of <= TILE_NNZ entries and
// <= BLOCK rows.  A workgroup streams one tile's values and columns with 16-B / 8-B per-lane
// contiguous loads (every 128-B line of the matrix is fetched exactly once, fully coalesced),
// multiplies by the gathered x and parks the products in LDS; then one thread per row adds its
// products in column order -- the same order as a scalar CPU loop, so y is reproducible bit for bit.
// Workgroups are persistent (grid ~ 8 per CU) and walk the tiles XCD-aware: the workgroups that
// land on one XCD (blockIdx % 8, round-robin dispatch) sweep one contiguous eighth of the rows, so
// the x window they share stays in that XCD's 4 MiB L2 instead of being fetched by all eight.
#include "zzz_device.h"
#include "zzz_internal.h"

namespace zzz
{
constexpr int SPMV_BLOCK = 256;
constexpr int SPMV_TILE_NNZ = 2048;
typedef double dbl2 __attribute__((ext_vector_type(2)));
typedef int int2v __attribute__((ext_vector_type(2)));

// tile index for (workgroup b, step i): XCD x = b % 8 owns tiles [x*T/8, (x+1)*T/8)
__device__ inline int64_t xcd_tile(int64_t ntiles, int b, int nb, int i)
{
  const int xcd = b & 7;
  const int64_t lo = ntiles * xcd / 8, hi = ntiles * (xcd + 1) / 8;
  const int wg_in_xcd = b >> 3, n_in_xcd = (nb + 7 - xcd) >> 3;
  const int64_t t = lo + wg_in_xcd + (int64_t)i * n_in_xcd;
  return t < hi ? t : -1;
}

// One tile descriptor = {first row, end row, first nonzero, end nonzero}: one 16-B load per tile.
template <bool DOT>
__global__ __launch_bounds__(SPMV_BLOCK) void spmv_tile_kernel(const int32_t* __restrict__ rowptr,
                                                               const int32_t* __restrict__ cols,
                                                               const double* __restrict__ vals,
                                                               const double* __restrict__ x, double* __restrict__ y,
                                                               const int4* __restrict__ tiles, int64_t ntiles,
                                                               int nnz_even, double* __restrict__ partials,
                                                               const int* __restrict__ stop_flag)
{
  if (stop_flag && *stop_flag) // CG already converged: the host is a few iterations ahead
    return;
  __shared__ __attribute__((aligned(16))) double prod[SPMV_TILE_NNZ + 16];
  __shared__ double red[SPMV_BLOCK / 64];
  constexpr int NPASS = SPMV_TILE_NNZ / (2 * SPMV_BLOCK);
  double dot = 0.0;
  for (int i = 0;; ++i)
  {
    const int64_t t = xcd_tile(ntiles, blockIdx.x, gridDim.x, i);
    if (t < 0)
      break;
    const int4 td = tiles[t];
    const int r0 = td.x, r1 = td.y, s = td.z, e = td.w;
    const int s_al = s & ~1;
    // stream the tile: every load below is unconditional (indices clamped into the padded arrays),
    // so all 2*NPASS matrix loads, then all 2*NPASS gathers, are in flight together
    dbl2 v[NPASS];
    int2v c[NPASS];
    int kk[NPASS];
#pragma unroll
    for (int j = 0; j < NPASS; ++j)
    {
      kk[j] = s_al + 2 * (int)threadIdx.x + j * 2 * SPMV_BLOCK;
      const int kc = min(kk[j], nnz_even);
      v[j] = __builtin_nontemporal_load(reinterpret_cast<const dbl2*>(vals + kc));
      c[j] = __builtin_nontemporal_load(reinterpret_cast<const int2v*>(cols + kc));
    }
    double xa[NPASS], xb[NPASS];
#pragma unroll
    for (int j = 0; j < NPASS; ++j)
    {
      xa[j] = x[c[j].x];
      xb[j] = x[c[j].y];
    }
    // this thread's row bounds (used after the barrier): issue the loads now
    const int r = r0 + (int)threadIdx.x;
    const int rc = min(r, r1 - 1);
    const int ra = rowptr[rc] - s_al, rb = rowptr[rc + 1] - s_al;
    const double xr = DOT ? x[rc] : 0.0;
#pragma unroll
    for (int j = 0; j < NPASS; ++j)
      if (kk[j] < e)
      {
        dbl2 pr;
        pr.x = (kk[j] >= s) ? v[j].x * xa[j] : 0.0;
        pr.y = (kk[j] + 1 < e) ? v[j].y * xb[j] : 0.0;
        *reinterpret_cast<dbl2*>(prod + (kk[j] - s_al)) = pr;
      }
    __syncthreads();
    if (r < r1)
    {
      // products are added in column order (the serial CPU order); 8 LDS reads in flight at a time
      double sum = 0.0;
      for (int k = ra; k < rb; k += 8)
      {
        double q[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
          q[u] = prod[k + u]; // may run past the row: within the padded LDS array, masked below
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (k + u < rb)
            sum += q[u];
      }
      y[r] = sum;
      if (DOT)
        dot += sum * xr;
    }
    __syncthreads();
  }
  if (DOT)
  {
    const double sres = block_reduce_sum(dot, red);
    if (threadIdx.x == 0)
      partials[blockIdx.x] = sres;
  }
}

int build_spmv_tiles(zzz_ctx* ctx, const std::vector<int32_t>& h_rowptr)
{
  const int64_t n = (int64_t)h_rowptr.size() - 1;
  std::vector<int32_t> tiles; // 4 ints per tile: r0, r1, s, e
  int64_t r = 0;
  while (r < n)
  {
    const int64_t s_al = h_rowptr[r] & ~1;
    int64_t q = r;
    while (q < n && q - r < SPMV_BLOCK && h_rowptr[q + 1] - s_al <= SPMV_TILE_NNZ)
      ++q;
    if (q == r)
      return fail(ctx, ZZZ_ERR_LIMIT, "matrix row %lld has more than %d nonzeros", (long long)r, SPMV_TILE_NNZ - 1);
    tiles.push_back((int32_t)r);
    tiles.push_back((int32_t)q);
    tiles.push_back(h_rowptr[r]);
    tiles.push_back(h_rowptr[q]);
    r = q;
  }
  ctx->ntiles = (int64_t)tiles.size() / 4;
  ZZZ_HIP(ctx, ctx->tile_row.alloc(tiles.size()));
  ZZZ_HIP(ctx, hipMemcpyAsync(ctx->tile_row.p, tiles.data(), tiles.size() * sizeof(int32_t), hipMemcpyHostToDevice,
                              ctx->stream));
  ZZZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return ZZZ_OK;
}

static int spmv_grid(const zzz_ctx* ctx)
{
  // 8 workgroups of 256 threads per CU; always a multiple of 8 so that every XCD residue
  // (blockIdx % 8) that owns tiles in xcd_tile() has at least one workgroup
  int64_t g = 256 * 8;
  const int64_t need = (ctx->ntiles + 7) / 8 * 8;
  if (g > need)
    g = need;
  if (g < 8)
    g = 8;
  return (int)g;
}

int launch_spmv(zzz_ctx* ctx, const double* x, double* y, double* partials, int* npartials)
{
  const int grid = spmv_grid(ctx);
  const int nnz_even = (int)((ctx->nnz + 1) & ~(int64_t)1); // last valid clamped index (arrays are padded by 8)
  const int* stop = partials ? reinterpret_cast<const int*>(ctx->state.p) : nullptr; // CgState::converged
  if (partials)
  {
    if ((size_t)grid > ctx->part_a.n)
      return fail(ctx, ZZZ_ERR_ARG, "partials buffer too small");
    hipLaunchKernelGGL(spmv_tile_kernel<true>, dim3(grid), dim3(SPMV_BLOCK), 0, ctx->stream, ctx->rowptr.p,
                       ctx->cols.p, ctx->vals.p, x, y, reinterpret_cast<const int4*>(ctx->tile_row.p), ctx->ntiles, nnz_even, partials, stop);
    if (npartials)
      *npartials = grid;
  }
  else
    hipLaunchKernelGGL(spmv_tile_kernel<false>, dim3(grid), dim3(SPMV_BLOCK), 0, ctx->stream, ctx->rowptr.p,
                       ctx->cols.p, ctx->vals.p, x, y, reinterpret_cast<const int4*>(ctx->tile_row.p), ctx->ntiles, nnz_even, (double*)nullptr, stop);
  ZZZ_HIP(ctx, hipGetLastError());
  return ZZZ_OK;
}
} // namespace zzz

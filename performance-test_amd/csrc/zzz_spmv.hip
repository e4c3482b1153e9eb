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

// LDS-only workgroup barrier: orders the LDS writes/reads of the tile without draining the
// global loads already issued for the NEXT tile (a plain __syncthreads() waits for vmcnt(0)).
__device__ inline void lds_barrier()
{
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

template <bool NT, typename T>
__device__ inline T stream_load(const T* p)
{
  return NT ? __builtin_nontemporal_load(p) : *p;
}

// One tile descriptor = {first row, end row, first nonzero, end nonzero}: one 16-B load per tile.
// TILE = nonzeros per tile; PIPE = issue the next tile's matrix loads before reducing this one.
template <bool DOT, bool NT, bool PIPE, int TILE>
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
  __shared__ __attribute__((aligned(16))) double prod[TILE + 16];
  __shared__ double red[SPMV_BLOCK / 64];
  constexpr int NPASS = TILE / (2 * SPMV_BLOCK);
  double dot = 0.0;
  dbl2 v[NPASS];
  int2v c[NPASS];
  int4 td = make_int4(0, 0, 0, 0);
  int64_t t = xcd_tile(ntiles, blockIdx.x, gridDim.x, 0);
  // every matrix load is unconditional (indices clamped into the padded arrays), so all 2*NPASS
  // loads of a tile are in flight together, ahead of the 2*NPASS gathers that depend on them
  auto issue = [&](const int4 d) {
    const int s_al = d.z & ~1;
#pragma unroll
    for (int j = 0; j < NPASS; ++j)
    {
      const int kc = min(s_al + 2 * (int)threadIdx.x + j * 2 * SPMV_BLOCK, nnz_even);
      v[j] = stream_load<NT>(reinterpret_cast<const dbl2*>(vals + kc));
      c[j] = stream_load<NT>(reinterpret_cast<const int2v*>(cols + kc));
    }
  };
  if (t >= 0)
  {
    td = tiles[t];
    issue(td);
  }
  for (int i = 0; t >= 0; ++i)
  {
    const int r0 = td.x, r1 = td.y, s = td.z, e = td.w;
    const int s_al = s & ~1;
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
    {
      const int kk = s_al + 2 * (int)threadIdx.x + j * 2 * SPMV_BLOCK;
      if (kk < e)
      {
        dbl2 pr;
        pr.x = (kk >= s) ? v[j].x * xa[j] : 0.0;
        pr.y = (kk + 1 < e) ? v[j].y * xb[j] : 0.0;
        *reinterpret_cast<dbl2*>(prod + (kk - s_al)) = pr;
      }
    }
    const int64_t tn = xcd_tile(ntiles, blockIdx.x, gridDim.x, i + 1);
    if (PIPE && tn >= 0)
    {
      td = tiles[tn];
      issue(td);
    }
    if (PIPE)
      lds_barrier();
    else
      __syncthreads();
    for (int rr = r; rr < r1; rr += SPMV_BLOCK) // one row per thread unless the rows are very short
    {
      const int ra_ = (rr == r) ? ra : rowptr[rr] - s_al, rb_ = (rr == r) ? rb : rowptr[rr + 1] - s_al;
      const double xr_ = (rr == r) ? xr : (DOT ? x[rr] : 0.0);
      // products are added in column order (the serial CPU order); 8 LDS reads in flight at a time
      double sum = 0.0;
      for (int k = ra_; k < rb_; k += 8)
      {
        double q[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
          q[u] = prod[k + u]; // may run past the row: within the padded LDS array, masked below
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (k + u < rb_)
            sum += q[u];
      }
      y[rr] = sum;
      if (DOT)
        dot += sum * xr_;
    }
    if (PIPE)
      lds_barrier();
    else
      __syncthreads();
    t = tn;
    if (!PIPE && t >= 0)
    {
      td = tiles[t];
      issue(td);
    }
  }
  if (DOT)
  {
    const double sres = block_reduce_sum(dot, red);
    if (threadIdx.x == 0)
      partials[blockIdx.x] = sres;
  }
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

template <bool DOT>
static void launch_variant(zzz_ctx* ctx, int grid, const double* x, double* y, double* partials, const int* stop,
                           int nnz_even)
{
  const int4* tiles = reinterpret_cast<const int4*>(ctx->tile_row.p);
#define ZZZ_SPMV_GO(NT, PIPE, TILE)                                                                                   \
  hipLaunchKernelGGL((spmv_tile_kernel<DOT, NT, PIPE, TILE>), dim3(grid), dim3(SPMV_BLOCK), 0, ctx->stream,           \
                     ctx->rowptr.p, ctx->cols.p, ctx->vals.p, x, y, tiles, ctx->ntiles, nnz_even, partials, stop)
  const int var = ctx->spmv_variant; // bit 0: non-temporal matrix loads, bit 1: pipelined tiles
  if (ctx->spmv_tile == 4096)
  {
    switch (var & 3)
    {
    case 0: ZZZ_SPMV_GO(false, false, 4096); break;
    case 1: ZZZ_SPMV_GO(true, false, 4096); break;
    case 2: ZZZ_SPMV_GO(false, true, 4096); break;
    default: ZZZ_SPMV_GO(true, true, 4096); break;
    }
  }
  else
  {
    switch (var & 3)
    {
    case 0: ZZZ_SPMV_GO(false, false, 2048); break;
    case 1: ZZZ_SPMV_GO(true, false, 2048); break;
    case 2: ZZZ_SPMV_GO(false, true, 2048); break;
    default: ZZZ_SPMV_GO(true, true, 2048); break;
    }
  }
#undef ZZZ_SPMV_GO
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
    launch_variant<true>(ctx, grid, x, y, partials, stop, nnz_even);
    if (npartials)
      *npartials = grid;
  }
  else
    launch_variant<false>(ctx, grid, x, y, nullptr, stop, nnz_even);
  ZZZ_HIP(ctx, hipGetLastError());
  return ZZZ_OK;
}
} // namespace zzz

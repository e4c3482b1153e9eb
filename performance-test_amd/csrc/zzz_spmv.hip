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
#include <cstring>

#include <vector>

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
__global__ __launch_bounds__(SPMV_BLOCK, 8) void spmv_tile_kernel(const rp_t* __restrict__ rowptr,
                                                               const int32_t* __restrict__ cols,
                                                               const double* __restrict__ vals,
                                                               const double* __restrict__ x, double* __restrict__ y,
                                                               const int4* __restrict__ tiles, int64_t ntiles,
                                                               int nnz_even, double* __restrict__ partials,
                                                               const int* __restrict__ stop_flag,
                                                               const int32_t* __restrict__ tile_list,
                                                               const double* __restrict__ rvec, int pstride,
                                                               int nn_is_rr, const uint16_t* __restrict__ cols16,
                                                               const int32_t* __restrict__ tile_base, int offb,
                                                               int col_max, int lpr_shift)
{
  // lpr_shift > 0 (long rows: high order, vector-valued): 2^lpr_shift lanes share a row; lane j adds the
  // j-th contiguous chunk of ceil(len / lanes) products in column order and the chunk sums are combined by a
  // butterfly ((c0+c1)+(c2+c3))+...: a fixed order, restated by the oracle's zo_spmv_chunked.
  // cols16 != nullptr: 16-bit column codes (zzz_pattern.hip, k_tile_encode_cols): column = band base
  // of the tile [code >> offb] + (code & mask); the 2^(16-offb) band bases of a tile sit one per lane
  // in a register and are looked up with ds_bpermute.  Tiles whose columns do not fit (descriptor
  // .y < 0, holding ~r1) read the int32 columns.  Same columns, same arithmetic: results identical.
  // rvec != nullptr (single-reduction CG, x = z): besides <x,y> also leave the partials of <r,x> at
  // partials[pstride + b] and of the test norm (<x,x>, or <r,r> when nn_is_rr) at partials[2 pstride + b]
  // tile_list != nullptr: this launch covers only the listed tiles (interior or boundary subset of a
  // partitioned matrix); ntiles is then the length of the list
  if (stop_flag && *stop_flag) // CG already converged: the host is a few iterations ahead
    return;
  __shared__ __attribute__((aligned(16))) double prod[TILE + 16];
  __shared__ double red[SPMV_BLOCK / 64];
  constexpr int NPASS = TILE / (2 * SPMV_BLOCK);
  double dot = 0.0, dot_rx = 0.0, dot_nn = 0.0;
  dbl2 v[NPASS];
  int2v c[NPASS];
  int4 td = make_int4(0, 0, 0, 0);
  int bands = 0;
  const int nbands = 1 << (16 - offb);
  int64_t t = xcd_tile(ntiles, blockIdx.x, gridDim.x, 0);
  // every matrix load is unconditional (indices clamped into the padded arrays), so all 2*NPASS
  // loads of a tile are in flight together, ahead of the 2*NPASS gathers that depend on them
  auto issue = [&](const int4 d, const int64_t tt) {
    const int s_al = d.z & ~1;
#pragma unroll
    for (int j = 0; j < NPASS; ++j)
    {
      const int kc = min(s_al + 2 * (int)threadIdx.x + j * 2 * SPMV_BLOCK, nnz_even);
      v[j] = stream_load<NT>(reinterpret_cast<const dbl2*>(vals + kc));
      if (cols16 && d.y >= 0) // packed columns: 4 B per pair instead of 8
        c[j].x = (int)stream_load<NT>(reinterpret_cast<const unsigned*>(cols16 + kc));
      else
        c[j] = stream_load<NT>(reinterpret_cast<const int2v*>(cols + kc));
    }
    if (cols16 && d.y >= 0)
      bands = tile_base[tt * nbands + (threadIdx.x & (nbands - 1))];
  };
  if (t >= 0)
  {
    const int64_t ti = tile_list ? tile_list[t] : t;
    td = tiles[ti];
    issue(td, ti);
  }
  for (int i = 0; t >= 0; ++i)
  {
    const int r0 = td.x, r1 = td.y < 0 ? ~td.y : td.y, s = td.z, e = td.w;
    const int s_al = s & ~1;
    double xa[NPASS], xb[NPASS];
    if (cols16 && td.y >= 0)
    {
      const unsigned mask = (1u << offb) - 1u;
#pragma unroll
      for (int j = 0; j < NPASS; ++j)
      {
        const unsigned u = (unsigned)c[j].x, lo = u & 0xffffu, hi = u >> 16;
        // window entries outside [s, e) carry another tile's codes: clamp, they are masked below
        c[j].x = min(__builtin_amdgcn_ds_bpermute((int)((lo >> offb) << 2), bands) + (int)(lo & mask), col_max);
        c[j].y = min(__builtin_amdgcn_ds_bpermute((int)((hi >> offb) << 2), bands) + (int)(hi & mask), col_max);
      }
    }
#pragma unroll
    for (int j = 0; j < NPASS; ++j)
    {
      xa[j] = x[c[j].x];
      xb[j] = x[c[j].y];
    }
    // this thread's row bounds (used after the barrier): issue the loads now
    const int r = r0 + (int)threadIdx.x;
    const int rc = min(r, r1 - 1);
    const int ra = (int)rowptr[rc] - s_al, rb = (int)rowptr[rc + 1] - s_al; // < 2^31 nonzeros on this path
    const double xr = DOT ? x[rc] : 0.0;
    const double rv = (DOT && rvec) ? rvec[rc] : 0.0;
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
      const int64_t ti = tile_list ? tile_list[tn] : tn;
      td = tiles[ti];
      issue(td, ti);
    }
    if (PIPE)
      lds_barrier();
    else
      __syncthreads();
    if (lpr_shift == 0)
    {
      for (int rr = r; rr < r1; rr += SPMV_BLOCK) // one row per thread unless the rows are very short
      {
        const int ra_ = (rr == r) ? ra : (int)rowptr[rr] - s_al, rb_ = (rr == r) ? rb : (int)rowptr[rr + 1] - s_al;
        const double xr_ = (rr == r) ? xr : (DOT ? x[rr] : 0.0);
        // products are added in column order (the serial CPU order); 8 LDS reads in flight at a time
        double sum = 0.0;
        constexpr int QB = PIPE ? 4 : 8; // the pipelined form holds the next tile's loads in registers meanwhile
        for (int k = ra_; k < rb_; k += QB)
        {
          double q[QB];
  #pragma unroll
          for (int u = 0; u < QB; ++u)
            q[u] = prod[k + u]; // may run past the row: within the padded LDS array, masked below
  #pragma unroll
          for (int u = 0; u < QB; ++u)
            if (k + u < rb_)
              sum += q[u];
        }
        y[rr] = sum;
        if (DOT)
        {
          dot += sum * xr_;
          if (rvec)
          {
            const double rr_ = (rr == r) ? rv : rvec[rr];
            dot_rx += rr_ * xr_;
            dot_nn += nn_is_rr ? rr_ * rr_ : xr_ * xr_;
          }
        }
      }
    }
    else
    {
      const int lanes = 1 << lpr_shift, sub = (int)threadIdx.x & (lanes - 1);
      for (int rr = r0 + ((int)threadIdx.x >> lpr_shift); rr < r1; rr += SPMV_BLOCK >> lpr_shift)
      {
        const int ra_ = (int)rowptr[rr] - s_al, rb_ = (int)rowptr[rr + 1] - s_al;
        const int chunk = (rb_ - ra_ + lanes - 1) >> lpr_shift;
        const int ka = ra_ + sub * chunk, kb = min(ka + chunk, rb_);
        double sum = 0.0;
        for (int k = ka; k < kb; k += 4)
        {
          double q[4];
#pragma unroll
          for (int u = 0; u < 4; ++u)
            q[u] = prod[min(k + u, TILE + 15)];
#pragma unroll
          for (int u = 0; u < 4; ++u)
            if (k + u < kb)
              sum += q[u];
        }
        for (int o = 1; o < lanes; o <<= 1)
          sum += __shfl_xor(sum, o);
        if (sub == 0)
        {
          y[rr] = sum;
          if (DOT)
          {
            const double xr_ = x[rr];
            dot += sum * xr_;
            if (rvec)
            {
              const double rr_ = rvec[rr];
              dot_rx += rr_ * xr_;
              dot_nn += nn_is_rr ? rr_ * rr_ : xr_ * xr_;
            }
          }
        }
      }
    }
    if (PIPE)
      lds_barrier();
    else
      __syncthreads();
    t = tn;
    if (!PIPE && t >= 0)
    {
      const int64_t ti = tile_list ? tile_list[t] : t;
      td = tiles[ti];
      issue(td, ti);
    }
  }
  if (DOT)
  {
    const double sres = block_reduce_sum(dot, red);
    if (threadIdx.x == 0)
      partials[blockIdx.x] = sres;
    if (rvec)
    {
      const double s1 = block_reduce_sum(dot_rx, red);
      const double s2 = block_reduce_sum(dot_nn, red);
      if (threadIdx.x == 0)
      {
        partials[pstride + blockIdx.x] = s1;
        partials[2 * pstride + blockIdx.x] = s2;
      }
    }
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
                           int nnz_even, const int32_t* tile_list = nullptr, int64_t nlist = 0, const double* rvec = nullptr,
                           int nn_is_rr = 0)
{
  const int4* tiles = reinterpret_cast<const int4*>(ctx->tile_row.p);
  const int64_t nt = tile_list ? nlist : ctx->ntiles;
  // bit 4 of the variant: ignore the packed columns (A/B and parity of the two index streams)
  const uint16_t* c16 = (ctx->have_cols16 && !(ctx->spmv_variant & 16)) ? ctx->cols16.p : nullptr;
  const int col_max = (int)ctx->nloc() - 1; // nloc() counts scalar entries (a clamp bs times too far read past the end of x)
#define ZZZ_SPMV_GO(NT, PIPE, TILE)                                                                                   \
  hipLaunchKernelGGL((spmv_tile_kernel<DOT, NT, PIPE, TILE>), dim3(grid), dim3(SPMV_BLOCK), 0, ctx->stream,           \
                     ctx->rowptr.p, ctx->cols.p, ctx->vals.p, x, y, tiles, nt, nnz_even, partials, stop, tile_list,    \
                     rvec, SPMV_PSTRIDE, nn_is_rr, c16, ctx->tile_base.p, ctx->cols16_offb, col_max, ctx->spmv_lpr_shift)
  // bit 0: non-temporal matrix loads, bit 1: pipelined tiles.  Unless a variant was forced, the load
  // policy follows the matrix size: a matrix that fits the 256 MiB Infinity Cache is re-read from it
  // every CG iteration, and non-temporal loads would throw that away (measured, 1.25 M-dof P1 matrix,
  // 221 MB: 44 us plain vs 55 us nt; 2.5 M dofs and up: nt 2-8 % faster).
  int var = ctx->spmv_variant;
  if (ctx->spmv_auto)
    var = (var & ~1) | (12.0 * (double)ctx->nnz > 300.0e6 ? 1 : 0);
#ifdef ZZZ_EXPERIMENTS
  // the tools build keeps the variants that were measured and lost: tiles of 4096 nonzeros (they miss the register target
  // of eight wavefronts per SIMD) and the software-pipelined tile loop (no gain: eight workgroups per CU cover the latency)
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
  else if (var & 2)
  {
    if (var & 1)
      ZZZ_SPMV_GO(true, true, 2048);
    else
      ZZZ_SPMV_GO(false, true, 2048);
  }
  else
#endif
  {
    if (var & 1)
      ZZZ_SPMV_GO(true, false, 2048);
    else
      ZZZ_SPMV_GO(false, false, 2048);
  }
#undef ZZZ_SPMV_GO
}

int launch_spmv(zzz_ctx* ctx, const double* x, double* y, double* partials, int* npartials, const double* rvec,
                int nn_is_rr)
{
  const int grid = spmv_grid(ctx);
  const int nnz_even = ctx->tiles_ok ? (int)((ctx->nnz + 1) & ~(int64_t)1) : 0; // last valid clamped index (arrays are padded by 8)
  const int* stop = partials ? reinterpret_cast<const int*>(ctx->state.p) : nullptr; // CgState::converged
  if (sellp_active(ctx))
    return launch_sellp(ctx, x, y, partials, npartials, rvec, nn_is_rr);
  if (!ctx->tiles_ok)
    return fail(ctx, ZZZ_ERR_LIMIT, "%lld nonzeros: the product of a matrix of 2^31 nonzeros or more needs the operator stream "
                "(zzz_sellp.hip), which this matrix / these settings do not use", (long long)ctx->nnz);
  if (int rc = ensure_cols16(ctx))
    return rc;
  if (partials)
  {
    if ((size_t)grid > ctx->part_a.n)
      return fail(ctx, ZZZ_ERR_ARG, "partials buffer too small");
    launch_variant<true>(ctx, grid, x, y, partials, stop, nnz_even, nullptr, 0, rvec, nn_is_rr);
    if (npartials)
      *npartials = grid;
  }
  else
    launch_variant<false>(ctx, grid, x, y, nullptr, stop, nnz_even);
  ZZZ_HIP(ctx, hipGetLastError());
  return ZZZ_OK;
}
static int grid_for_tiles(int64_t nt)
{
  int64_t g = 256 * 8;
  const int64_t need = (nt + 7) / 8 * 8;
  if (g > need)
    g = need;
  if (g < 8)
    g = 8;
  return (int)g;
}

// Partitioned matrix: y = A x with the forward halo of x overlapped with the interior tiles.
//   comm stream : (waits for x) pack + send/recv of the ghost entries        -- issued FIRST so that
//   main stream : SpMV over the tiles that reference no ghost column            its kernels get CUs
//   main stream : (waits for the halo) SpMV over the boundary tiles
// Partials of <x,y>: interior workgroups first, then the boundary ones.
int launch_spmv_overlapped(zzz_ctx* ctx, double* x, double* y, double* partials, int* npartials, const double* rvec,
                           int nn_is_rr)
{
  const int nnz_even = (int)((ctx->nnz + 1) & ~(int64_t)1);
  const int* stop = partials ? reinterpret_cast<const int*>(ctx->state.p) : nullptr;
  if (sellp_active(ctx) && ctx->have_group_split)
    return launch_sellp_overlapped(ctx, x, y, partials, npartials, rvec, nn_is_rr);
  if (int rc = ensure_cols16(ctx))
    return rc;
  const int64_t n_in = ctx->n_tiles_interior, n_bd = ctx->n_tiles_boundary;
  // The interior launch leaves one workgroup slot per CU free (7 of 8): at full occupancy the persistent
  // SpMV workgroups hold every wave slot until the launch ends and RCCL's send/recv kernel, although
  // enqueued first on a high-priority stream, could not start beside them -- no overlap at all.
  int g_in = n_in ? grid_for_tiles(n_in) : 0;
  if (g_in > 256 * 7 && ctx->nneigh > 0)
    g_in = 256 * 7;
  const int g_bd = n_bd ? grid_for_tiles(n_bd) : 0;
  if (partials && (size_t)(g_in + g_bd) > ctx->part_a.n)
    return fail(ctx, ZZZ_ERR_ARG, "partials buffer too small");
  int rc = comm_halo_begin(ctx, x);
  if (rc)
    return rc;
  if (n_in)
  {
    if (partials)
      launch_variant<true>(ctx, g_in, x, y, partials, stop, nnz_even, ctx->tiles_interior.p, n_in, rvec, nn_is_rr);
    else
      launch_variant<false>(ctx, g_in, x, y, nullptr, stop, nnz_even, ctx->tiles_interior.p, n_in);
  }
  rc = comm_halo_end(ctx);
  if (rc)
    return rc;
  if (n_bd)
  {
    if (partials)
      launch_variant<true>(ctx, g_bd, x, y, partials + g_in, stop, nnz_even, ctx->tiles_boundary.p, n_bd, rvec, nn_is_rr);
    else
      launch_variant<false>(ctx, g_bd, x, y, nullptr, stop, nnz_even, ctx->tiles_boundary.p, n_bd);
  }
  if (npartials)
    *npartials = g_in + g_bd;
  ZZZ_HIP(ctx, hipGetLastError());
  return ZZZ_OK;
}
ZZZ_PRELOAD_TU(spmv)
} // namespace zzz

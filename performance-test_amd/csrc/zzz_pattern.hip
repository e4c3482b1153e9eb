// Sparsity pattern, dof->cell adjacency and kernel tile descriptors, built on the GPU.
//
// Replaces fem::petsc::create_matrix(*a) (src/poisson_problem.cpp:122-123, src/elasticity_problem.cpp:196-197),
// which sits inside the reference's `ZZZ Assemble` umbrella timer: once assembly and solve are fast,
// a host-side pattern build dominates that timer.
//
//   1. (dof, cell) pairs of all local cells, stable radix sort by dof  -> adjacency lists with
//      ascending cell ids (the order the row-gather assembly sums in: reproducible);
//   2. one wavefront per owned block row: gather the dofs of the row's cells into LDS, bitonic sort,
//      count / emit the unique ones (two passes around an exclusive scan) -> CSR with ascending
//      columns, block size expanded;
//   3. tile descriptors for the SpMV and assembly kernels by binary search of fixed-width windows
//      of the nonzero stream in rowptr (fully parallel, no host loop).
// rocPRIM supplies the generic scan / radix-sort primitives; everything pattern-specific is below.
#include <cstring>

#include "zzz_device.h"
#include "zzz_internal.h"

#include <rocprim/rocprim.hpp>

#include <climits>
#include <cstdlib>

namespace zzz
{
constexpr int PAT_CAP = 1024; // candidate columns per block row handled on the device

// The (dof, cell) pairs of the dof -> cell adjacency are not written out before the sort: its keys ARE the
// connectivity (ghost dofs are numbered behind the owned ones, so they sort behind them and k_adj_bounds ignores them)
// and its values are the entries' cell numbers, position / nd, kept from one pattern build to the next (they depend on
// the array's length only).  Round 1-2 wrote both arrays every time (k_make_pairs, 1.7 ms at 10 M dofs, with an atomic
// per pair for the valence); the valence now comes from the sorted keys (k_adj_bounds).
__global__ void k_cell_of(int64_t n, int nd, int32_t* __restrict__ out)
{
  for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x)
    out[k] = (int32_t)(k / nd);
}

// onesweep geometry of the adjacency sort: rocPRIM's gfx950 default for (int, int) is 1024 threads x 16 items for both
// kernels; with 8 items per thread in the SORT kernel (the histogram kernel keeps 16) the 237 M-pair sort of the 10 M-dof
// pattern takes 4.70 instead of 5.2 ms (tools/micro/sort_cfg.hip: 4, 6, 10, 12, 16 items and 256/512 threads are slower
// for the sort kernel, 8 items 0.28 ms slower for the histogram kernel; wider digits do not fit the LDS)
using AdjSortConfig
    = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                 rocprim::radix_sort_onesweep_config<rocprim::kernel_config<1024, 16>, rocprim::kernel_config<1024, 8>, 8,
                                                                     rocprim::block_radix_rank_algorithm::match>>;

// adj_off[d] = first position of key d in the sorted keys (d = 0 .. nb; a dof without cells gets an empty range)
__global__ void k_adj_bounds(const int32_t* __restrict__ keys, int64_t n, int32_t nb, int32_t* __restrict__ adj_off)
{
  // four keys per thread (one 16-B load; the array is a hipMalloc allocation, so aligned)
  const int64_t n4 = (n + 1 + 3) / 4; // positions 0 .. n
  for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < n4; q += (int64_t)gridDim.x * blockDim.x)
  {
    const int64_t k0 = 4 * q;
    int32_t kk[4];
    if (k0 + 4 <= n)
    {
      const int4 v = *reinterpret_cast<const int4*>(keys + k0);
      kk[0] = v.x, kk[1] = v.y, kk[2] = v.z, kk[3] = v.w;
    }
    else
#pragma unroll
      for (int e = 0; e < 4; ++e)
        kk[e] = k0 + e < n ? keys[k0 + e] : nb + 1;
    int32_t prev = k0 ? keys[k0 - 1] : -1;
#pragma unroll
    for (int e = 0; e < 4; ++e)
    {
      const int64_t k = k0 + e;
      if (k > n)
        break;
      const int32_t cur = kk[e];
      for (int32_t d = prev + 1; d <= cur && d <= nb; ++d)
        adj_off[d] = (int32_t)k;
      prev = cur;
    }
  }
}

// Bitonic sort of 64 or 128 keys held one or two per lane (key index = 64 kk + lane), every stage unrolled: the
// partner lane through ds_swizzle (lane ^ j, no address register), which of min / max a lane keeps from a compile-time
// 64-bit lane mask fed to v_cndmask -- four instructions per stage and register instead of ten.
template <int J>
__device__ inline int32_t xor_lane(int32_t v)
{
  if constexpr (J < 32)
    return __builtin_amdgcn_ds_swizzle(v, (J << 10) | 0x1f); // bit mode: and 0x1f, or 0, xor J
  else
    return __shfl_xor(v, J);
}
constexpr unsigned long long bitonic_min_mask(int k, int j, int kk)
{
  unsigned long long m = 0;
  for (int l = 0; l < 64; ++l)
  {
    const int idx = kk * 64 + l;
    const bool lower = (l & j) == 0, up = (idx & k) == 0;
    if (lower == up)
      m |= 1ull << l;
  }
  return m;
}
__device__ inline int32_t select_by_lane_mask(unsigned long long m, int32_t if_set, int32_t if_clear)
{
  int32_t r;
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(if_clear), "v"(if_set), "s"(m));
  return r;
}
template <int P, int K, int J>
__device__ inline void bitonic_stage(int32_t (&x)[P / 64])
{
  constexpr int NR = P / 64;
  if constexpr (J >= 64) // partner in another register: kk ^ (J / 64); the direction depends on the register only
  {
#pragma unroll
    for (int kk = 0; kk < NR; ++kk)
      if ((kk & (J / 64)) == 0)
      {
        const int hh = kk | (J / 64);
        const int32_t lo = min(x[kk], x[hh]), hi = max(x[kk], x[hh]);
        const bool up = ((kk * 64) & K) == 0;
        x[kk] = up ? lo : hi;
        x[hh] = up ? hi : lo;
      }
  }
  else
  {
#pragma unroll
    for (int kk = 0; kk < NR; ++kk)
    {
      const int32_t y = xor_lane<J>(x[kk]);
      const int32_t mn = min(x[kk], y), mx = max(x[kk], y);
      x[kk] = select_by_lane_mask(bitonic_min_mask(K, J, kk), mn, mx);
    }
  }
}
template <int P, int K, int J>
__device__ inline void bitonic_merge(int32_t (&x)[P / 64])
{
  bitonic_stage<P, K, J>(x);
  if constexpr (J > 1)
    bitonic_merge<P, K, J / 2>(x);
}
template <int P, int K>
__device__ inline void bitonic_sort_regs(int32_t (&x)[P / 64])
{
  bitonic_merge<P, K, K / 2>(x);
  if constexpr (K < P)
    bitonic_sort_regs<P, K * 2>(x);
}

// One row of the pattern from P / 64 keys per lane: gather the candidates, sort, emit the unique ones (see k_row_pattern).
// POS: every candidate carries its origin (a * nd + j: which cell of the row, which local dof) through the sort in the
// low 9 bits of its key -- key = ((dof - lowest dof of the row) << 9 | origin) with the sign bit flipped, so that the
// network's signed compares order it as an unsigned number: at most 512 candidates -- and its rank among the unique columns,
// which the scan of the "new column" flags yields anyway, is written to pos[adj entry * nd + j]: the position of that
// element-matrix entry inside the row, which the matrix assembly would otherwise find by binary search, 545 M times
// per assembly of the 6.2 M-dof P3 problem.
template <int P, bool FILL, bool POS>
__device__ inline int row_unique_regs(const int32_t* __restrict__ cell_dofs, int nd, int bs, const int32_t* __restrict__ adj_cells,
                                      int a0, int n, int lane, int64_t rp, int32_t nu, int32_t* __restrict__ cols,
                                      int32_t* __restrict__ stage, uint16_t* __restrict__ pos_out, int32_t* __restrict__ pos_missed)
{
  constexpr int NR = P / 64;
  int32_t x[NR];
#pragma unroll
  for (int kk = 0; kk < NR; ++kk)
    x[kk] = INT_MAX;
  if (n > 0) // (wave-uniform)
  {
    // the candidates in two rounds of UNCONDITIONAL loads at clamped indices -- all cells, then all dofs: `if (idx < n) x =
    // dofs[cells[..]]` per register compiled to a branch per load with the wait inside it, up to sixteen round trips one after
    // the other for a vertex row of P3 (disassembly, last third of round 6)
    int32_t cv[NR];
#pragma unroll
    for (int kk = 0; kk < NR; ++kk)
      cv[kk] = adj_cells[a0 + min(kk * 64 + lane, n - 1) / nd];
    int32_t dv[NR];
#pragma unroll
    for (int kk = 0; kk < NR; ++kk)
    {
      const int idx = min(kk * 64 + lane, n - 1);
      dv[kk] = cell_dofs[(int64_t)cv[kk] * nd + (idx - idx / nd * nd)];
    }
#pragma unroll
    for (int kk = 0; kk < NR; ++kk)
      if (kk * 64 + lane < n)
        x[kk] = dv[kk];
  }
  int32_t dmin = 0;
  if (POS)
  {
    // the key holds the dof RELATIVE to the row's lowest column, so that 23 bits suffice at any problem size as long
    // as a row's columns lie within 8 M dofs of each other (a row that does not: *pos_missed, the assembly searches)
    int32_t mn = INT_MAX, mx = 0;
#pragma unroll
    for (int kk = 0; kk < NR; ++kk)
    {
      mn = min(mn, x[kk]);
      mx = max(mx, x[kk] == INT_MAX ? 0 : x[kk]);
    }
    dmin = wave_min_i(mn);
    const int32_t dmax = wave_max_i(mx);
    if (dmax - dmin >= (1 << 23))
    {
      if (lane == 0)
        *pos_missed = 1;
      dmin = dmax - ((1 << 23) - 1); // keys stay in range; the ranks of this row are not used
    }
#pragma unroll
    for (int kk = 0; kk < NR; ++kk)
    {
      const int idx = kk * 64 + lane;
      if (idx < n)
        x[kk] = (int32_t)(((((uint32_t)max(x[kk] - dmin, 0)) << 9) | (uint32_t)idx) ^ 0x80000000u);
    }
  }
  bitonic_sort_regs<P, 2>(x);
  int base = 0;
#pragma unroll
  for (int kk = 0; kk < NR; ++kk)
  {
    if (kk * 64 >= n) // wave-uniform
      break;
    const int idx = kk * 64 + lane;
    const int32_t cur = POS ? (int32_t)(((uint32_t)x[kk] ^ 0x80000000u) >> 9) + dmin : x[kk]; // (the padding keys are never looked at)
    int32_t prev = __shfl_up(cur, 1);
    if (lane == 0)
    {
      const int32_t last = __builtin_amdgcn_readlane(x[kk ? kk - 1 : 0], 63);
      prev = kk ? (POS ? (int32_t)(((uint32_t)last ^ 0x80000000u) >> 9) + dmin : last) : INT_MIN;
    }
    const bool flag = idx < n && (idx == 0 || cur != prev);
    const unsigned long long m = __ballot(flag);
    const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
    if (POS && idx < n) // rank of this candidate's column: heads up to and including this lane, minus one
      pos_out[(int64_t)a0 * nd + (int)(((uint32_t)x[kk] ^ 0x80000000u) & 511u)] = (uint16_t)(pos - (flag ? 0 : 1));
    if (!FILL && stage && flag)
      stage[(int64_t)a0 * nd + pos] = cur;
    if (FILL && flag)
    {
      const int32_t col = cur;
      for (int a = 0; a < bs; ++a)
        for (int d = 0; d < bs; ++d)
          cols[(int64_t)bs * bs * rp + (int64_t)a * bs * nu + (int64_t)pos * bs + d] = col * bs + d;
    }
    base += __popcll(m);
  }
  return base;
}

// One wavefront per block row: sorted unique dofs of the row's cells.
// FILL = false: cnt[r] = number of unique columns (and the maximum over rows); when `stage` is given
//               the sorted unique columns are also parked at stage[adj_off[r]*nd ...] so that the
//               fill pass (k_row_copy) need not sort again;
// FILL = true : scalar CSR columns of the bs rows of block r (used when no staging area is available).
template <bool FILL>
__global__ __launch_bounds__(256) void k_row_pattern(const int32_t* __restrict__ cell_dofs, int nd, int bs,
                                                     const int32_t* __restrict__ adj_off,
                                                     const int32_t* __restrict__ adj_cells, int32_t nb,
                                                     int32_t* __restrict__ cnt, int32_t* __restrict__ maxcnt,
                                                     const int64_t* __restrict__ bptr, int32_t* __restrict__ cols,
                                                     int32_t* __restrict__ overflow, int32_t* __restrict__ stage,
                                                     uint16_t* __restrict__ pos_out, int32_t* __restrict__ pos_missed)
{
  __shared__ int32_t lds[4][PAT_CAP];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  volatile int32_t* c = lds[wv];
  int wmax = 0;
  for (int64_t r = blockIdx.x * 4 + wv; r < nb; r += (int64_t)gridDim.x * 4)
  {
    const int a0 = adj_off[r], na = adj_off[r + 1] - a0;
    const int n = na * nd;
    if (n > PAT_CAP)
    {
      if (lane == 0)
        atomicMax(overflow, n);
      continue;
    }
    int P = 64;
    while (P < n)
      P <<= 1;
    if (P <= 512)
    {
      // up to 512 candidates (every row of P2 and of P3 Poisson on the Kuhn mesh): the keys stay in P / 64 registers
      // per lane (key index = 64 kk + lane) and the bitonic network runs on cross-lane exchanges and register swaps --
      // no LDS round trip per stage (the LDS version below spent ~6 us per row at one row per wavefront)
      const int64_t rp = FILL ? bptr[r] : 0;
      const int32_t nu = FILL ? cnt[r] : 0;
      int base;
#define ZZZ_ROW_REGS(PP)                                                                                                       \
  base = (!FILL && pos_out)                                                                                                    \
             ? row_unique_regs<PP, FILL, true>(cell_dofs, nd, bs, adj_cells, a0, n, lane, rp, nu, cols, stage, pos_out, pos_missed) \
             : row_unique_regs<PP, FILL, false>(cell_dofs, nd, bs, adj_cells, a0, n, lane, rp, nu, cols, stage, nullptr, nullptr)
      if (P == 64)
        ZZZ_ROW_REGS(64);
      else if (P == 128)
        ZZZ_ROW_REGS(128);
      else if (P == 256)
        ZZZ_ROW_REGS(256);
      else
        ZZZ_ROW_REGS(512);
#undef ZZZ_ROW_REGS
      if (!FILL && lane == 0)
        cnt[r] = base;
      wmax = max(wmax, base);
      continue;
    }
    if (!FILL && pos_out && lane == 0)
      *pos_missed = 1; // a row of more than 512 candidates: no positions, the assembly searches
    for (int idx = lane; idx < P; idx += 64)
    {
      int32_t v = INT_MAX;
      if (idx < n)
      {
        const int a = idx / nd, j = idx - a * nd;
        v = cell_dofs[(int64_t)adj_cells[a0 + a] * nd + j];
      }
      c[idx] = v;
    }
    __builtin_amdgcn_wave_barrier();
    // bitonic sort of P keys by one wavefront (LDS operations of a wave complete in order)
    for (int k = 2; k <= P; k <<= 1)
      for (int j = k >> 1; j > 0; j >>= 1)
      {
        for (int t = lane; t < (P >> 1); t += 64)
        {
          const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)); // index with bit j clear
          const int l = i | j;
          const bool up = (i & k) == 0;
          const int32_t x = c[i], y = c[l];
          if ((x > y) == up)
          {
            c[i] = y;
            c[l] = x;
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
    // unique
    int base = 0;
    const int64_t rp = FILL ? bptr[r] : 0;
    const int32_t nu = FILL ? cnt[r] : 0;
    for (int s = 0; s < n; s += 64)
    {
      const int idx = s + lane;
      const bool flag = idx < n && (idx == 0 || c[idx] != c[idx - 1]);
      const unsigned long long m = __ballot(flag);
      if (!FILL && stage && flag)
        stage[(int64_t)a0 * nd + base + __popcll(m & ((1ull << lane) - 1ull))] = c[idx];
      if (FILL && flag)
      {
        const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
        const int32_t col = c[idx];
        // block (r, col) -> bs x bs scalar entries; scalar row (r, a) starts at bs*bs*rp + a*bs*nu
        for (int a = 0; a < bs; ++a)
          for (int d = 0; d < bs; ++d)
            cols[(int64_t)bs * bs * rp + (int64_t)a * bs * nu + (int64_t)pos * bs + d] = col * bs + d;
      }
      base += __popcll(m);
    }
    if (!FILL && lane == 0)
      cnt[r] = base;
    wmax = max(wmax, base);
    __builtin_amdgcn_wave_barrier();
  }
  // one atomic per wavefront, not per row: ~10^7 same-address atomics (or sc1 loads) serialise at
  // one L2 channel and cost ~100 ms
  if (!FILL)
  {
    __shared__ int32_t wmax_s[4]; // ... and one per workgroup is a quarter of that
    if (lane == 0)
      wmax_s[threadIdx.x >> 6] = wmax;
    __syncthreads();
    const int bm = max(max(wmax_s[0], wmax_s[1]), max(wmax_s[2], wmax_s[3]));
    if (threadIdx.x == 0 && bm > 0)
      atomicMax(maxcnt, bm);
  }
}

// P1 (4 dofs per cell, rows of ~15 unique columns out of ~100 candidates): one THREAD per block row keeps its
// sorted unique columns in a private LDS column (u[k][thread], stride 129: conflict-free both for the owner
// and for the transposed read-out) and inserts the candidates as they come -- they arrive nearly ascending, so
// an insertion moves a few entries.  A wavefront per row (k_row_pattern) spends 28 bitonic stages on 128 keys
// with most lanes idle.  Everything that is row-major in memory moves through LDS so that the global accesses
// are dense: the block's slice of adj_cells is loaded cooperatively, and the unique columns are written out one
// row per wavefront instruction (contiguous), not one row per lane.  Same outputs as k_row_pattern<false>:
// cnt[r], the sorted unique columns at stage[adj_off[r]*4 ...], the maximum count.  The kernel has every
// (row, cell, local index) triple in hand, so it also writes the transposed adjacency of the assembly kernels
// (zzz_assemble.hip: entry a of row 64 s + lane at adjT_off[s] + 64 a + lane), which saves a separate gather pass.
// A row with more than ROW_T_CAP unique columns or a block with more than ROW_T_ADJ adjacency entries raises
// `overflow` and the caller uses the wavefront kernel.
constexpr int ROW_T_BLOCK = 128, ROW_T_LD = ROW_T_BLOCK + 1;
// ROW_T_ADJ: adjacency entries of a block's 128 rows staged in LDS (4096 = 32 per row on average).
// ROW_T_CAP: unique columns a row may have (16: the sorted list lives in 16 registers; 32: in 32).  Round 1 kept the
// list in a private LDS column and inserted there: dependent LDS reads at 2 wavefronts per SIMD made that 5 of the
// kernel's 7 ms at 10 M dofs; a branch-free insertion into a register array (every slot recomputed by two compares)
// needs no memory at all.
template <int ROW_T_CAP, int ROW_T_ADJ>
__global__ __launch_bounds__(ROW_T_BLOCK) void k_row_pattern_thread4(const int32_t* __restrict__ cell_dofs,
                                                                     const int32_t* __restrict__ adj_off,
                                                                     const int32_t* __restrict__ adj_cells, int32_t nb,
                                                                     int32_t* __restrict__ cnt, int32_t* __restrict__ maxcnt,
                                                                     int32_t* __restrict__ overflow,
                                                                     int32_t* __restrict__ stage,
                                                                     const int32_t* __restrict__ adjT_off,
                                                                     int32_t* __restrict__ cellT, uint8_t* __restrict__ liT)
{
  // the staged adjacency (read while the rows are built) and the transposed read-out buffer (written after) share
  // their LDS: 13.8 KB per workgroup instead of 22, so that registers, not LDS, set the occupancy (8 workgroups per CU)
  constexpr int ROW_T_SH = ROW_T_ADJ > ROW_T_CAP * ROW_T_LD ? ROW_T_ADJ : ROW_T_CAP * ROW_T_LD;
  __shared__ int32_t sh[ROW_T_SH];
  int32_t* const u = sh;
  int32_t* const adj_s = sh;
  __shared__ int32_t a0_s[ROW_T_BLOCK], m_s[ROW_T_BLOCK];
  int wmax = 0;
  const int64_t nblk = ((int64_t)nb + ROW_T_BLOCK - 1) / ROW_T_BLOCK; // nb64 <= nblk * ROW_T_BLOCK
  // XCD-aware walk: the blocks of one XCD cover one contiguous eighth of the rows, so the cells that
  // neighbouring blocks share (rows one mesh line apart) are fetched into ONE L2, not into all eight
  for (int it = 0;; ++it)
  {
    const int64_t blk = xcd_stride_item(nblk, it);
    if (blk < 0)
      break;
    const int64_t rbase = blk * ROW_T_BLOCK, r = rbase + threadIdx.x;
    const int64_t rend = min((int64_t)nb, rbase + ROW_T_BLOCK);
    const int ab0 = adj_off[rbase], nadj = adj_off[rend] - ab0;
    if (nadj > ROW_T_ADJ) // uniform over the block
    {
      if (threadIdx.x == 0)
        atomicMax(overflow, 1);
      continue;
    }
    for (int k = threadIdx.x; k < nadj; k += ROW_T_BLOCK)
      adj_s[k] = adj_cells[ab0 + k];
    __syncthreads();
    int m = 0, a0 = 0;
    bool over = false;
    int32_t cs[ROW_T_CAP];
#pragma unroll
    for (int k = 0; k < ROW_T_CAP; ++k)
      cs[k] = INT_MAX;
    const bool padrow = r >= nb && r < ((int64_t)nb + 63) / 64 * 64; // the last slice is padded to 64 rows
    if (r < nb || padrow)
    {
      const int to = adjT_off[r >> 6], tlen = (adjT_off[(r >> 6) + 1] - to) >> 6;
      int32_t* ct = cellT + to + (r & 63);
      uint8_t* lt = liT + to + (r & 63);
      int na = 0;
      if (r < nb)
      {
        a0 = adj_off[r];
        na = adj_off[r + 1] - a0;
        const int32_t* myadj = adj_s + (a0 - ab0);
        // eight cells at a time: their dof quadruples are in flight together
        for (int ab = 0; ab < na && !over; ab += 8)
        {
          int32_t cell[8];
          int4 d[8];
#pragma unroll
          for (int q = 0; q < 8; ++q)
            cell[q] = myadj[min(ab + q, na - 1)];
#pragma unroll
          for (int q = 0; q < 8; ++q)
            d[q] = *reinterpret_cast<const int4*>(cell_dofs + 4 * (int64_t)cell[q]);
#pragma unroll
          for (int q = 0; q < 8; ++q)
          {
            if (ab + q >= na)
              break;
            const int32_t v4[4] = {d[q].x, d[q].y, d[q].z, d[q].w};
            ct[(ab + q) * 64] = cell[q];
            lt[(ab + q) * 64] = (uint8_t)(v4[1] == (int32_t)r ? 1 : (v4[2] == (int32_t)r ? 2 : (v4[3] == (int32_t)r ? 3 : 0)));
#pragma unroll
            for (int j = 0; j < 4; ++j)
            {
              const int32_t v = v4[j];
              bool dup = false;
#pragma unroll
              for (int k = 0; k < ROW_T_CAP; ++k)
                dup |= cs[k] == v;
              if (dup)
                continue;
              if (m == ROW_T_CAP)
              {
                over = true;
                break;
              }
              // slot k of the new list: the old entry if it is smaller than v; v if it is the first that is not;
              // else the old entry one slot down (unused slots hold INT_MAX)
#pragma unroll
              for (int k = ROW_T_CAP - 1; k > 0; --k)
                cs[k] = cs[k] < v ? cs[k] : (cs[k - 1] < v ? v : cs[k - 1]);
              cs[0] = cs[0] < v ? cs[0] : v;
              ++m;
            }
            if (over)
              break;
          }
        }
      }
      for (int a = na; a < tlen; ++a)
      {
        ct[a * 64] = -1;
        lt[a * 64] = 0;
      }
      if (r < nb && !over)
        cnt[r] = m;
    }
    if (over)
      atomicMax(overflow, 1);
    __syncthreads(); // every row has read its adjacency: the buffer changes hands
#pragma unroll
    for (int k = 0; k < ROW_T_CAP; ++k)
      u[k * ROW_T_LD + threadIdx.x] = cs[k]; // transposed read-out below: one row per wavefront instruction
    a0_s[threadIdx.x] = a0;
    m_s[threadIdx.x] = (r < nb && !over) ? m : 0;
    wmax = max(wmax, m);
    __syncthreads();
    // read-out: one row per wavefront instruction, lanes = entries (contiguous in `stage`).  (Dense 64-B records per
    // row, written four rows per instruction as whole 256-B stores, made this kernel 0.7 ms SLOWER -- 1.97 -> 2.69 ms --
    // for 0.15 ms saved in k_row_copy; measured on one box, not understood; the sparse layout stays.)
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int t = wv * 64; t < wv * 64 + 64; ++t)
    {
      const int mt = m_s[t];
      if (lane < mt)
        stage[4 * (int64_t)a0_s[t] + lane] = u[lane * ROW_T_LD + t];
    }
    __syncthreads();
  }
  for (int o = 32; o; o >>= 1)
    wmax = max(wmax, __shfl_xor(wmax, o));
  if ((threadIdx.x & 63) == 0 && wmax > 0)
    atomicMax(maxcnt, wmax);
}

// fill pass when the count pass staged the sorted unique columns: expand block (r, col) to bs x bs
__global__ __launch_bounds__(256) void k_row_copy(const int32_t* __restrict__ stage, const int32_t* __restrict__ adj_off,
                                                  int nd, int bs, int32_t nb, const int32_t* __restrict__ cnt,
                                                  const int64_t* __restrict__ bptr, int32_t* __restrict__ cols)
{
  // 16 lanes per row: FE rows have tens of columns, not hundreds
  const int sub = threadIdx.x & 15;
  for (int64_t r = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 4; r < nb; r += ((int64_t)gridDim.x * blockDim.x) >> 4)
  {
    const int nu = cnt[r];
    const int64_t rp = bptr[r];
    const int32_t* src = stage + (int64_t)adj_off[r] * nd;
    for (int k = sub; k < nu; k += 16)
    {
      const int32_t col = src[k];
      for (int a = 0; a < bs; ++a)
        for (int d = 0; d < bs; ++d)
          cols[(int64_t)bs * bs * rp + (int64_t)a * bs * nu + (int64_t)k * bs + d] = col * bs + d;
    }
  }
}

__global__ void k_scalar_rowptr(const int64_t* __restrict__ bptr, const int32_t* __restrict__ cnt, int32_t nb, int bs,
                                rp_t* __restrict__ rowptr)
{
  for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r <= nb; r += (int64_t)gridDim.x * blockDim.x)
  {
    if (r == nb)
    {
      rowptr[(int64_t)nb * bs] = (int64_t)bs * bs * bptr[nb];
      continue;
    }
    for (int a = 0; a < bs; ++a)
      rowptr[r * bs + a] = (int64_t)bs * bs * bptr[r] + (int64_t)a * bs * cnt[r];
  }
}

__device__ inline int lower_bound_rows(const rp_t* __restrict__ rowptr, int stride, int n, int64_t target)
{
  // first i in [0, n] with rowptr[i*stride] >= target (rowptr[n*stride] = nnz closes the range)
  int lo = 0, hi = n;
  while (lo < hi)
  {
    const int mid = (lo + hi) >> 1;
    if (rowptr[(int64_t)mid * stride] < target)
      lo = mid + 1;
    else
      hi = mid;
  }
  return lo;
}

// SpMV tiles: window t = rows whose first nonzero lies in [t*W, (t+1)*W); descriptor {r0, r1, s, e}
__global__ void k_spmv_tiles(const rp_t* __restrict__ rowptr, int nrows, int64_t W, int64_t ntiles,
                             int4* __restrict__ tiles)
{
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < ntiles; t += (int64_t)gridDim.x * blockDim.x)
  {
    const int r0 = lower_bound_rows(rowptr, 1, nrows, t * W);
    const int r1 = lower_bound_rows(rowptr, 1, nrows, (t + 1) * W);
    tiles[t] = make_int4(r0, r1, (int)rowptr[r0], (int)rowptr[r1]); // only built below 2^31 nonzeros
  }
}

// 16-bit column stream for the SpMV.  Per tile the columns are covered greedily by up to NB = 2^(16-offb)
// bands of width 2^offb (band 0 starts at the smallest column, band k+1 at the smallest column beyond
// band k); a column is stored as (band << offb) | (column - base[band]).  FE matrices with any locality in
// their numbering need a handful of narrow bands per 2048-nonzero tile, so the column stream shrinks from
// 4 to 2 bytes per nonzero (12 -> 10 B per nonzero of SpMV traffic).  A tile that does not fit keeps its
// int32 columns: its descriptor gets .y = ~r1 and `nfallback` counts it.  One workgroup per tile.
template <int TILE>
__global__ __launch_bounds__(256) void k_tile_encode_cols(int4* __restrict__ tiles, int64_t ntiles,
                                                          const int32_t* __restrict__ cols, int offb,
                                                          uint16_t* __restrict__ cols16, int32_t* __restrict__ tile_base,
                                                          int32_t* __restrict__ nfallback)
{
  constexpr int PER = TILE / 256;
  const int NB = 1 << (16 - offb), W = 1 << offb;
  __shared__ int sbase[64];
  __shared__ int red[4];
  for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x)
  {
    int4 d = tiles[t];
    if (d.y < 0)
      d.y = ~d.y; // re-encoding with another width
    int c[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j)
    {
      const int k = d.z + (int)threadIdx.x + j * 256;
      c[j] = k < d.w ? cols[k] : INT32_MAX;
    }
    int thr = 0, nb = 0;
    bool more = true;
    while (more && nb < NB)
    {
      int m = INT32_MAX;
#pragma unroll
      for (int j = 0; j < PER; ++j)
        m = min(m, c[j] >= thr ? c[j] : INT32_MAX);
      for (int o = 32; o; o >>= 1)
        m = min(m, __shfl_xor(m, o));
      __syncthreads();
      if ((threadIdx.x & 63) == 0)
        red[threadIdx.x >> 6] = m;
      __syncthreads();
      m = min(min(red[0], red[1]), min(red[2], red[3]));
      if (m == INT32_MAX)
        more = false;
      else
      {
        if (threadIdx.x == 0)
          sbase[nb] = m;
        ++nb;
        thr = (m > INT32_MAX - W) ? INT32_MAX : m + W;
      }
    }
    // anything left beyond the last band?
    int left = 0;
    if (more)
#pragma unroll
      for (int j = 0; j < PER; ++j)
        left |= (c[j] != INT32_MAX && c[j] >= thr) ? 1 : 0;
    const int fb = __syncthreads_or(left);
    if (threadIdx.x < NB)
      tile_base[t * NB + threadIdx.x] = fb ? 0 : sbase[min((int)threadIdx.x, max(nb - 1, 0))] * (nb > 0 ? 1 : 0);
    if (threadIdx.x == 0)
    {
      tiles[t] = make_int4(d.x, fb ? ~d.y : d.y, d.z, d.w);
      if (fb)
        atomicAdd(nfallback, 1);
    }
    if (!fb)
    {
#pragma unroll
      for (int j = 0; j < PER; ++j)
      {
        const int k = d.z + (int)threadIdx.x + j * 256;
        if (k < d.w)
        {
          int b = 0;
          while (b + 1 < nb && sbase[b + 1] <= c[j])
            ++b;
          cols16[k] = (uint16_t)((b << offb) | (c[j] - sbase[b]));
        }
      }
    }
    __syncthreads();
  }
}

// assembly tiles: boundaries in block dofs; tile t = block dofs whose first nonzero lies in window t
__global__ void k_asm_tiles(const rp_t* __restrict__ rowptr, int nb, int bs, int64_t W, int64_t ntiles,
                            int32_t* __restrict__ tiles)
{
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t <= ntiles; t += (int64_t)gridDim.x * blockDim.x)
    tiles[t] = (t == ntiles) ? nb : lower_bound_rows(rowptr, bs, nb, t * W);
}

static int grid_for(int64_t n, int block = 256, int cap = 4096)
{
  int64_t g = (n + block - 1) / block;
  if (g > cap)
    g = cap;
  if (g < 1)
    g = 1;
  return (int)g;
}

// packed column stream of the SpMV tiles; the offset width is the first of four candidates (10..13 bits)
// that leaves no tile on int32 columns, else the one that leaves the fewest.  Encoded when the CSR tile kernel is
// first used on this pattern (matrices that run on the operator stream of zzz_sellp.hip never need it).
int ensure_cols16(zzz_ctx* ctx)
{
  if (!ctx->cols16_pending)
    return ZZZ_OK;
  ctx->cols16_pending = false;
  ctx->have_cols16 = false;
  ctx->cols16_fallback_tiles = 0;
  if (!ctx->cols16_enabled || ctx->ntiles == 0)
    return ZZZ_OK;
  hipStream_t s = ctx->stream;
  ZZZ_HIP(ctx, ctx->cols16.alloc((size_t)ctx->nnz + 16));
  ZZZ_HIP(ctx, ctx->tile_base.alloc((size_t)ctx->ntiles * 64));
  ZZZ_HIP(ctx, ctx->scr_c16.alloc(4));
  // the padding is read by the clamped tail loads of the last tile
  ZZZ_HIP(ctx, hipMemsetAsync(ctx->cols16.p + ctx->nnz, 0, 16 * sizeof(uint16_t), s));
  // long rows (high order, vector-valued) spread a tile over more, narrower clusters: start narrower
  const bool long_rows = ctx->max_row_nnz > 64;
  const int order[4] = {long_rows ? 11 : 12, long_rows ? 12 : 11, long_rows ? 10 : 13, long_rows ? 13 : 10};
  const int ntry = ctx->cols16_offb_forced ? 1 : 4;
  int best = -1;
  int32_t best_fb = INT32_MAX;
  auto run = [&](int offb, int32_t* nfb) -> int {
    ZZZ_HIP(ctx, hipMemsetAsync(ctx->scr_c16.p, 0, sizeof(int32_t), s));
    const int g = grid_for(ctx->ntiles, 1, 256 * 8);
    if (ctx->spmv_tile == 4096)
      hipLaunchKernelGGL(k_tile_encode_cols<4096>, dim3(g), dim3(256), 0, s, reinterpret_cast<int4*>(ctx->tile_row.p),
                         ctx->ntiles, ctx->cols.p, offb, ctx->cols16.p, ctx->tile_base.p, ctx->scr_c16.p);
    else
      hipLaunchKernelGGL(k_tile_encode_cols<2048>, dim3(g), dim3(256), 0, s, reinterpret_cast<int4*>(ctx->tile_row.p),
                         ctx->ntiles, ctx->cols.p, offb, ctx->cols16.p, ctx->tile_base.p, ctx->scr_c16.p);
    ZZZ_HIP(ctx, hipGetLastError());
    ZZZ_HIP(ctx, hipMemcpyAsync(nfb, ctx->scr_c16.p, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    ZZZ_HIP(ctx, hipStreamSynchronize(s));
    return ZZZ_OK;
  };
  int last = -1;
  for (int i = 0; i < ntry; ++i)
  {
    const int offb = ctx->cols16_offb_forced ? ctx->cols16_offb_forced : order[i];
    int32_t nfb = 0;
    int rc = run(offb, &nfb);
    if (rc)
      return rc;
    last = offb;
    if (nfb < best_fb)
    {
      best_fb = nfb;
      best = offb;
    }
    if (nfb == 0)
      break;
  }
  if (best != last)
  {
    int32_t nfb = 0;
    int rc = run(best, &nfb);
    if (rc)
      return rc;
  }
  ctx->cols16_offb = best;
  ctx->cols16_fallback_tiles = best_fb;
  ctx->have_cols16 = true;
  return ZZZ_OK;
}

// tile descriptors from a device rowptr (used by both the device and the host pattern builders)
int build_tiles_device(zzz_ctx* ctx, int max_block_cols)
{
  const int bs = ctx->bs;
  hipStream_t s = ctx->stream;
  const int maxrow = max_block_cols * bs;           // longest scalar row
  const int maxblock = max_block_cols * bs * bs;    // nonzeros of the bs rows of one block dof
  ctx->max_row_nnz = maxrow;
  // lanes per row of the SpMV row phase: one (the serial CPU order) unless the rows are very long --
  // measured: 8 lanes pay from ~128 nonzeros per row on average (P3 elasticity 0.47 -> 0.40 ms), cost below
  {
    const double avg = ctx->nrows > 0 ? (double)ctx->nnz / (double)ctx->nrows : 0.0;
    ctx->spmv_lpr_shift = ctx->spmv_lpr_forced >= 0 ? ctx->spmv_lpr_forced : (avg >= 128.0 ? 3 : 0);
  }
  const int64_t Ws = (int64_t)ctx->spmv_tile - maxrow - 2;
  const int64_t Wa = (int64_t)asm_tile_nnz(ctx) - maxblock;
  if (Ws < maxrow || Wa < maxblock || Wa < 1)
    return fail(ctx, ZZZ_ERR_LIMIT, "matrix rows too long for the kernel tiles (%d nonzeros per row)", maxrow);
  // the SpMV tile windows are 32-bit offsets into the nonzero stream (head-room of one tile: the kernel forms
  // indices up to tile start + tile size); beyond that the product runs on the operator stream only
  ctx->tiles_ok = ctx->nnz <= (int64_t)INT32_MAX - 16384;
  ctx->ntiles = ctx->tiles_ok ? (ctx->nnz + Ws - 1) / Ws : 0;
  ctx->n_asm_tiles = (ctx->nnz + Wa - 1) / Wa;
  if (ctx->n_asm_tiles > INT32_MAX - 8)
    return fail(ctx, ZZZ_ERR_LIMIT, "%lld assembly tiles: use more parts", (long long)ctx->n_asm_tiles);
  ZZZ_HIP(ctx, ctx->tile_row.alloc((size_t)ctx->ntiles * 4 + 4));
  ZZZ_HIP(ctx, ctx->asm_tile.alloc((size_t)ctx->n_asm_tiles + 1));
  if (ctx->ntiles)
    hipLaunchKernelGGL(k_spmv_tiles, dim3(grid_for(ctx->ntiles)), dim3(256), 0, s, ctx->rowptr.p, (int)ctx->nrows, Ws,
                       ctx->ntiles, reinterpret_cast<int4*>(ctx->tile_row.p));
  hipLaunchKernelGGL(k_asm_tiles, dim3(grid_for(ctx->n_asm_tiles + 1)), dim3(256), 0, s, ctx->rowptr.p,
                     (int)ctx->n_owned, bs, Wa, ctx->n_asm_tiles, ctx->asm_tile.p);
  ZZZ_HIP(ctx, hipGetLastError());
  int rc = build_tile_split(ctx);
  if (rc)
    return rc;
  ctx->have_cols16 = false;
  ctx->cols16_pending = true;
  return ZZZ_OK;
}

// ---- dof -> cell adjacency WITHOUT a sort, for connectivities made of few monotone runs ---------------------------
// create_matrix needs, for every owned dof, the ascending list of its cells.  Sorting the (dof, cell) incidences costs
// three radix passes over 237 M pairs at the headline size (rocPRIM onesweep: 4.9 of create_matrix's 7.7 ms).  But the
// connectivity of a structured feed -- native, or put into lattice order by zzz_renumber.hip -- is a handful of
// MONOTONE RUNS: inside a block of cells (one simplex type of one slab, typically) the dof number at a fixed local
// index k grows strictly with the cell number.  The incidences of a WINDOW of 256 consecutive dofs are then, in every
// run, one contiguous range of at most 256 cells, found by binary search; one workgroup (one thread per cell of a
// range) counts the window's incidences per dof in LDS, scans, places them, orders each dof's cells, and writes its part
// of adj_off / adj_cells with dense stores.  No global atomics, no library sort.  (Reading whole connectivity records over
// the hull of the ranges of all k was tried: the ranges of different k lie a mesh plane apart -- 28 ms.)  Connectivities
// with more runs than ADJ_MAX_RUNS (a caller's arbitrary cell order) or a window beyond the LDS budget take the radix
// sort as before.
constexpr int ADJ_W = 256;          // dofs per window = threads per workgroup
constexpr int ADJ_CAP = 7168;       // incidences a window may hold in LDS
constexpr int ADJ_MAX_RUNS = 512;   // (block, local index) pairs
constexpr int ADJ_BREAK_CAP = 4096; // break records collected before giving up

// cells c where some sequence k stops growing (cd[c][k] <= cd[c-1][k]); out: the cells, count in nbreaks[0]
__global__ void k_run_breaks(const int32_t* __restrict__ cd, int64_t ncells, int nd, int cap, int32_t* __restrict__ nbreaks,
                             int32_t* __restrict__ out)
{
  const int64_t total = (ncells - 1) * nd;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x)
  {
    const int64_t c = 1 + t / nd;
    const int k = (int)(t % nd);
    if (cd[c * nd + k] <= cd[(c - 1) * nd + k])
    {
      const int slot = atomicAdd(nbreaks, 1);
      if (slot < cap)
        out[slot] = (int32_t)c;
    }
  }
}

// lo[r][w] = first cell of run r = {k, first cell, end cell} whose key is >= w * ADJ_W   (w = 0 .. nwin)
__global__ void k_run_window_bounds(const int32_t* __restrict__ cd, int nd, const int32_t* __restrict__ runs, int nruns,
                                    int nwin, int32_t* __restrict__ lo)
{
  const int64_t total = (int64_t)nruns * (nwin + 1);
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x)
  {
    const int r = (int)(t / (nwin + 1)), w = (int)(t % (nwin + 1));
    const int k = runs[3 * r];
    int a = runs[3 * r + 1], b = runs[3 * r + 2];
    const int64_t key = (int64_t)w * ADJ_W;
    while (a < b)
    {
      const int m = (a + b) >> 1;
      if (cd[(int64_t)m * nd + k] < key)
        a = m + 1;
      else
        b = m;
    }
    lo[t] = a;
  }
}

// tot[w] = incidences of window w (one thread per window; lo[r][.] is contiguous in w: dense reads)
__global__ void k_window_tot(const int32_t* __restrict__ lo, int nruns, int nwin, int32_t* __restrict__ tot)
{
  for (int w = blockIdx.x * blockDim.x + threadIdx.x; w < nwin; w += gridDim.x * blockDim.x)
  {
    int t = 0;
    for (int r = 0; r < nruns; ++r)
      t += lo[(int64_t)r * (nwin + 1) + w + 1] - lo[(int64_t)r * (nwin + 1) + w];
    tot[w] = t;
  }
}

// base = exclusive scan of tot, in place (one workgroup: a few tens of values per thread); flag[0] |= a window does
// not fit the LDS of k_adj_window
__global__ __launch_bounds__(1024) void k_window_base(int nwin, int32_t* __restrict__ base, int32_t* __restrict__ flag)
{
  __shared__ int64_t part[1024];
  const int per = (nwin + 1023) / 1024;
  const int w0 = min(nwin, (int)threadIdx.x * per), w1 = min(nwin, w0 + per);
  int64_t sum = 0;
  bool over = false;
  for (int w = w0; w < w1; ++w)
  {
    sum += base[w];
    over |= base[w] > ADJ_CAP;
  }
  if (over)
    flag[0] = 1;
  part[threadIdx.x] = sum;
  __syncthreads();
  if (threadIdx.x == 0)
  {
    int64_t acc = 0;
    for (int i = 0; i < 1024; ++i)
    {
      const int64_t v = part[i];
      part[i] = acc;
      acc += v;
    }
    if (acc > INT32_MAX)
      flag[0] = 1;
    base[nwin] = (int32_t)acc;
  }
  __syncthreads();
  int64_t acc = part[threadIdx.x];
  for (int w = w0; w < w1; ++w)
  {
    const int32_t v = base[w];
    base[w] = (int32_t)acc;
    acc += v;
  }
}

__global__ __launch_bounds__(ADJ_W) void k_adj_window(const int32_t* __restrict__ cd, int nd, const int32_t* __restrict__ runs,
                                                      int nruns, const int32_t* __restrict__ lo, int nwin,
                                                      const int32_t* __restrict__ base, int32_t nb,
                                                      int32_t* __restrict__ adj_off, int32_t* __restrict__ adj_cells)
{
  __shared__ int32_t seg0[ADJ_MAX_RUNS], segn[ADJ_MAX_RUNS], segk[ADJ_MAX_RUNS];
  __shared__ int32_t out_s[ADJ_CAP];
  __shared__ int32_t cnt[ADJ_W], off[ADJ_W + 1], cur[ADJ_W];
  __shared__ int wsum[ADJ_W / 64];
  const int w = blockIdx.x, tid = threadIdx.x;
  if (base[w + 1] - base[w] > ADJ_CAP)
  {
    // k_window_base has raised the flag and the host will build again with the sort; until it looks, the kernels behind
    // this one must find something harmless here: all the window's entries, valid cell numbers, on its first dof
    const int32_t gb = base[w], ge = base[w + 1];
    if (w * ADJ_W + tid <= nb)
      adj_off[w * ADJ_W + tid] = tid == 0 ? gb : ge;
    for (int t = gb + tid; t < ge; t += ADJ_W)
      adj_cells[t] = 0;
    return;
  }
  for (int r = tid; r < nruns; r += ADJ_W)
  {
    const int a = lo[(int64_t)r * (nwin + 1) + w], b = lo[(int64_t)r * (nwin + 1) + w + 1];
    // (an empty range may start at the run's END, i.e. at cell `ncells` for the last run: the sweeps below load the first
    // cell of every range unconditionally -- one entry past the connectivity, a memory fault when the array ends on a
    // page boundary: 2^k cells, found in round 6 at 64 x 32 x 32 sub-cubes.  An empty range points at the run's first cell.)
    seg0[r] = b > a ? a : runs[3 * r + 1];
    segn[r] = b - a; // <= 256: the keys of a run grow strictly
    segk[r] = runs[3 * r];
  }
  cnt[tid] = 0;
  cur[tid] = 0;
  __syncthreads();
  const int32_t key0 = w * ADJ_W;
  // Thread tid owns the tid-th cell of every run's range.  Two sweeps (count, place); eight runs' loads are requested
  // together (unconditional, clamped), then used; the second sweep's reads come from L2.
  auto sweep = [&](auto&& hit) {
    for (int r0 = 0; r0 < nruns; r0 += 8)
    {
      int key[8], cell[8];
      bool in[8];
#pragma unroll
      for (int u = 0; u < 8; ++u)
      {
        const int r = min(r0 + u, nruns - 1);
        in[u] = r0 + u < nruns && tid < segn[r];
        cell[u] = seg0[r] + (in[u] ? tid : 0);
        key[u] = cd[(int64_t)cell[u] * nd + segk[r]];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (in[u])
          hit(key[u] - key0, cell[u]);
    }
  };
  sweep([&](int key, int) { atomicAdd(&cnt[key], 1); });
  __syncthreads();
  // exclusive scan of the counts (256 values: wavefront scans + carry)
  {
    int v = cnt[tid];
#pragma unroll
    for (int o = 1; o < 64; o <<= 1)
    {
      const int u = __shfl_up(v, o, 64);
      if ((tid & 63) >= o)
        v += u;
    }
    if ((tid & 63) == 63)
      wsum[tid >> 6] = v;
    __syncthreads();
    int carry = 0;
    for (int q = 0; q < (tid >> 6); ++q)
      carry += wsum[q];
    off[tid] = carry + v - cnt[tid];
    if (tid == ADJ_W - 1)
      off[ADJ_W] = carry + v;
  }
  __syncthreads();
  const int T = off[ADJ_W];
  // place (arrival order), then order each dof's cells ascending = the serial assembly order
  sweep([&](int key, int c) { out_s[off[key] + atomicAdd(&cur[key], 1)] = c; });
  __syncthreads();
  {
    const int a = off[tid], n = cnt[tid];
    for (int i = 1; i < n; ++i)
    {
      const int32_t v = out_s[a + i];
      int j = i - 1;
      while (j >= 0 && out_s[a + j] > v)
      {
        out_s[a + j + 1] = out_s[a + j];
        --j;
      }
      out_s[a + j + 1] = v;
    }
  }
  __syncthreads();
  // the window's part of the adjacency, dense stores
  const int32_t gb = base[w];
  if (key0 + tid <= nb)
    adj_off[key0 + tid] = gb + off[tid];
  if (w == nwin - 1 && tid == 0 && nb == nwin * ADJ_W) // (otherwise written above: the last window also holds ghost keys)
    adj_off[nb] = gb + T;
  for (int t = tid; t < T; t += ADJ_W)
    adj_cells[gb + t] = out_s[t];
}

// finds the monotone runs of the current connectivity (once per dofmap): ctx->adj_runs_n > 0 when the sort-free
// path applies
static void adjacency_find_runs(zzz_ctx* ctx)
{
  ctx->adj_runs_n = 0;
#ifdef ZZZ_EXPERIMENTS
  if (getenv("ZZZ_ADJ_SORT")) // A/B knob (tools build): always the radix sort
    return;
#endif
  const int nd = ctx->nd;
  const int64_t nc = ctx->ncells;
  if (nc >= INT32_MAX)
    return;
  hipStream_t s = ctx->stream;
  DevBuf<int32_t> nbr, brk;
  if (nbr.alloc(1) != hipSuccess || brk.alloc((size_t)ADJ_BREAK_CAP) != hipSuccess
      || hipMemsetAsync(nbr.p, 0, sizeof(int32_t), s) != hipSuccess)
  {
    (void)hipGetLastError();
    return;
  }
  if (nc > 1)
    hipLaunchKernelGGL(k_run_breaks, dim3(grid_for((nc - 1) * nd)), dim3(256), 0, s, ctx->cell_dofs.p, nc, nd, ADJ_BREAK_CAP,
                       nbr.p, brk.p);
  int32_t n = 0;
  if (hipMemcpyAsync(&n, nbr.p, sizeof(n), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
  {
    (void)hipGetLastError();
    return;
  }
  if (n > ADJ_BREAK_CAP)
    return; // not a connectivity of few monotone runs
  std::vector<int32_t> cuts((size_t)n);
  if (n && hipMemcpy(cuts.data(), brk.p, cuts.size() * sizeof(int32_t), hipMemcpyDeviceToHost) != hipSuccess)
  {
    (void)hipGetLastError();
    return;
  }
  // blocks between ALL break cells (whatever k broke): inside a block every local index grows strictly
  cuts.push_back(0);
  cuts.push_back((int32_t)nc);
  std::sort(cuts.begin(), cuts.end());
  cuts.erase(std::unique(cuts.begin(), cuts.end()), cuts.end());
  const int nblocks = (int)cuts.size() - 1;
  if (nblocks < 1 || (int64_t)nblocks * nd > ADJ_MAX_RUNS)
    return;
  std::vector<int32_t> runs;
  for (int j = 0; j < nblocks; ++j)
    for (int k = 0; k < nd; ++k)
      runs.insert(runs.end(), {k, cuts[(size_t)j], cuts[(size_t)j + 1]});
  const int nruns = nblocks * nd;
  const int nwin = (int)((ctx->n_owned + ADJ_W - 1) / ADJ_W);
  if (ctx->adj_runs.alloc(runs.size()) != hipSuccess
      || hipMemcpy(ctx->adj_runs.p, runs.data(), runs.size() * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess
      || ctx->adj_run_lo.alloc((size_t)nruns * (nwin + 1)) != hipSuccess || ctx->adj_win_base.alloc((size_t)nwin + 2) != hipSuccess)
  {
    (void)hipGetLastError();
    return;
  }
  ctx->adj_runs_n = nruns;
}

#define ZZZ_DBG(name)                                                                                                  \
  do                                                                                                                   \
  {                                                                                                                    \
    if (getenv("ZZZ_DEBUG_SYNC"))                                                                                      \
    {                                                                                                                  \
      hipError_t e_ = hipStreamSynchronize(ctx->stream);                                                               \
      fprintf(stderr, "[zzz dbg] %s: %s\n", name, hipGetErrorString(e_));                                              \
      fflush(stderr);                                                                                                  \
    }                                                                                                                  \
  } while (0)
// adjacency through the runs, enqueued without a host wait: whether every window fitted arrives in
// ctx->adj_flag_host behind these kernels and is looked at the next time the build waits for the device anyway
static int adjacency_by_runs(zzz_ctx* ctx)
{
  const int nruns = ctx->adj_runs_n, nd = ctx->nd;
  const int32_t nb = (int32_t)ctx->n_owned;
  const int nwin = (nb + ADJ_W - 1) / ADJ_W;
  hipStream_t s = ctx->stream;
  if (!ctx->adj_flag_host)
    ZZZ_HIP(ctx, hipHostMalloc((void**)&ctx->adj_flag_host, sizeof(int32_t), hipHostMallocDefault));
  int32_t* flag = ctx->adj_win_base.p + nwin + 1;
  ZZZ_HIP(ctx, hipMemsetAsync(flag, 0, sizeof(int32_t), s));
  hipLaunchKernelGGL(k_run_window_bounds, dim3(grid_for((int64_t)nruns * (nwin + 1), 256, 16384)), dim3(256), 0, s,
                     ctx->cell_dofs.p, nd, ctx->adj_runs.p, nruns, nwin, ctx->adj_run_lo.p);
  if (getenv("ZZZ_DEBUG_SYNC"))
    fprintf(stderr, "[zzz dbg] nruns %d nwin %d nb %d\n", nruns, nwin, nb);
  ZZZ_DBG("k_run_window_bounds");
  hipLaunchKernelGGL(k_window_tot, dim3(grid_for(nwin)), dim3(256), 0, s, ctx->adj_run_lo.p, nruns, nwin, ctx->adj_win_base.p);
  ZZZ_DBG("k_window_tot");
  hipLaunchKernelGGL(k_window_base, dim3(1), dim3(1024), 0, s, nwin, ctx->adj_win_base.p, flag);
  ZZZ_DBG("k_window_base");
  hipLaunchKernelGGL(k_adj_window, dim3(nwin), dim3(ADJ_W), 0, s, ctx->cell_dofs.p, nd, ctx->adj_runs.p, nruns, ctx->adj_run_lo.p,
                     nwin, ctx->adj_win_base.p, nb, ctx->adj_off.p, ctx->adj_cells.p);
  ZZZ_HIP(ctx, hipMemcpyAsync(ctx->adj_flag_host, flag, sizeof(int32_t), hipMemcpyDeviceToHost, s));
  ZZZ_HIP(ctx, hipGetLastError());
  return ZZZ_OK;
}

// The staging area of the sorted unique columns (4 B per candidate: 17 GB for Poisson P3 at 50 M dofs) is worth a
// quarter of whatever memory is free: without it the fill pass sorts every row a second time (30 of 75 ms there).
static bool stage_fits(const zzz_ctx* ctx, int64_t nstage)
{
  if (ctx->scr_stage.p && (int64_t)ctx->scr_stage.cap >= nstage)
    return true;
  size_t fr = 0, tot = 0;
  if (hipMemGetInfo(&fr, &tot) != hipSuccess)
  {
    (void)hipGetLastError();
    return nstage < ((int64_t)3 << 30);
  }
  return nstage < ((int64_t)3 << 30) || (double)nstage * 4.0 < 0.25 * (double)fr;
}

// Scratch of the pattern build whose size follows from the dofmap alone (sort buffers, the entry -> cell map, the
// staging area): reserved when the dofmap arrives, so that a one-shot `ZZZ Assemble` (the reference driver runs every
// phase once) does not pay ~18 ms of multi-GB hipMalloc calls inside its timer.  Failure to reserve is not an error: the
// build allocates what it needs.
void pattern_reserve(zzz_ctx* ctx)
{
  if (ctx->ncells <= 0 || ctx->nd <= 0 || ctx->n_owned <= 0)
    return;
  const int nd = ctx->nd;
  const int64_t N = ctx->ncells * nd, nb = ctx->n_owned;
  if (N > INT32_MAX - 8)
    return;
  hipStream_t s = ctx->stream;
  if (ctx->scr_keys_out.alloc((size_t)N) != hipSuccess || ctx->adj_cells.alloc((size_t)N) != hipSuccess
      || ctx->scr_cnt.alloc((size_t)nb + 1) != hipSuccess || ctx->scr_bptr.alloc((size_t)nb + 1) != hipSuccess
      || ctx->adj_off.alloc((size_t)nb + 1) != hipSuccess || ctx->scr_vals_in.alloc((size_t)N) != hipSuccess)
  {
    (void)hipGetLastError();
    return;
  }
  if (ctx->scr_cell_of_n != N || ctx->scr_cell_of_nd != nd)
  {
    hipLaunchKernelGGL(k_cell_of, dim3(grid_for(N)), dim3(256), 0, s, N, nd, ctx->scr_vals_in.p);
    ctx->scr_cell_of_n = N;
    ctx->scr_cell_of_nd = nd;
  }
  int end_bit = 1;
  while ((1ll << end_bit) <= (long long)(ctx->n_owned + ctx->n_ghost))
    ++end_bit;
  size_t tb = 0, tb2 = 0;
  if (rocprim::radix_sort_pairs<AdjSortConfig>(nullptr, tb, ctx->cell_dofs.p, ctx->scr_keys_out.p, ctx->scr_vals_in.p,
                                               ctx->adj_cells.p, (size_t)N, 0, (unsigned)end_bit, s)
          == hipSuccess
      && rocprim::exclusive_scan(nullptr, tb2, ctx->scr_cnt.p, ctx->scr_bptr.p, (int64_t)0, (size_t)nb + 1,
                                 rocprim::plus<int64_t>(), s)
             == hipSuccess)
    (void)ctx->scr_tmp.alloc(tb > tb2 ? tb : tb2);
  const int64_t nstage = N * nd;
  if (stage_fits(ctx, nstage))
    (void)ctx->scr_stage.alloc((size_t)nstage);
  if (nd > 4)
    (void)ctx->asm_pos.alloc((size_t)nstage);
  (void)hipGetLastError();
  adjacency_find_runs(ctx); // a property of the dofmap, like the sizes above
}

// returns ZZZ_OK, or ZZZ_ERR_LIMIT with *fallback = true when a row has more candidates than the
// device kernel holds (the caller then uses the host builder)
int pattern_build_device(zzz_ctx* ctx, bool* fallback)
{
  *fallback = false;
  const int nd = ctx->nd, bs = ctx->bs;
  const int32_t nb = (int32_t)ctx->n_owned;
  const int64_t N = ctx->ncells * nd;
  hipStream_t s = ctx->stream;
  if (N > INT32_MAX - 8)
    return fail(ctx, ZZZ_ERR_LIMIT, "dof->cell adjacency exceeds int32");

  DevBuf<int32_t>&keys_out = ctx->scr_keys_out, &cnt = ctx->scr_cnt;
  DevBuf<int64_t>& bptr = ctx->scr_bptr;
  DevBuf<unsigned char>& tmp = ctx->scr_tmp;
  DevBuf<int32_t> scal;
  ZZZ_HIP(ctx, keys_out.alloc((size_t)N));
  ZZZ_HIP(ctx, ctx->adj_cells.alloc((size_t)N));
  ZZZ_HIP(ctx, cnt.alloc((size_t)nb + 1));
  ZZZ_HIP(ctx, bptr.alloc((size_t)nb + 1));
  ZZZ_HIP(ctx, ctx->adj_off.alloc((size_t)nb + 1));
  ZZZ_HIP(ctx, scal.alloc(4)); // [0] max unique cols, [1] overflow
  ZZZ_HIP(ctx, hipMemsetAsync(scal.p, 0, 4 * sizeof(int32_t), s));
  ZZZ_HIP(ctx, hipMemsetAsync(cnt.p, 0, ((size_t)nb + 1) * sizeof(int32_t), s)); // cnt[nb] closes the scans

  // 1. adjacency: sort the (dof, cell) incidences by dof
  int end_bit = 1;
  while ((1ll << end_bit) <= (long long)nb)
    ++end_bit;
  while ((1ll << end_bit) <= (long long)(ctx->n_owned + ctx->n_ghost)) // ghost dofs are keys too
    ++end_bit;
  DevBuf<int32_t>& cell_of = ctx->scr_vals_in;
  if (ctx->scr_cell_of_n != N || ctx->scr_cell_of_nd != nd)
  {
    ZZZ_HIP(ctx, cell_of.alloc((size_t)N));
    hipLaunchKernelGGL(k_cell_of, dim3(grid_for(N)), dim3(256), 0, s, N, nd, cell_of.p);
    ctx->scr_cell_of_n = N;
    ctx->scr_cell_of_nd = nd;
  }
  size_t tb = 0, tb2 = 0;
  ZZZ_HIP(ctx, rocprim::radix_sort_pairs<AdjSortConfig>(nullptr, tb, ctx->cell_dofs.p, keys_out.p, cell_of.p, ctx->adj_cells.p, (size_t)N, 0,
                                         (unsigned)end_bit, s));
  ZZZ_HIP(ctx, rocprim::exclusive_scan(nullptr, tb2, cnt.p, bptr.p, (int64_t)0, (size_t)nb + 1, rocprim::plus<int64_t>(), s));
  ZZZ_HIP(ctx, tmp.alloc(tb > tb2 ? tb : tb2));
  if (ctx->adj_runs_n < 0)
    adjacency_find_runs(ctx);
  const bool by_runs = ctx->adj_runs_n > 0;
  if (by_runs)
  {
    if (int rc = adjacency_by_runs(ctx))
      return rc;
    ZZZ_DBG("adjacency_by_runs");
  }
  else
  {
    ZZZ_HIP(ctx, rocprim::radix_sort_pairs<AdjSortConfig>(tmp.p, tb, ctx->cell_dofs.p, keys_out.p, cell_of.p, ctx->adj_cells.p, (size_t)N,
                                           0, (unsigned)end_bit, s));
    hipLaunchKernelGGL(k_adj_bounds, dim3(grid_for((N + 4) / 4)), dim3(256), 0, s, keys_out.p, N, nb, ctx->adj_off.p);
    ZZZ_DBG("radix sort + k_adj_bounds");
  }

  // 2. pattern: count, scan, fill
  const int rgrid = grid_for((int64_t)nb, 4, 256 * 16);
  // staging area for the sorted unique columns (upper bound: every candidate distinct); skipped when
  // it would not fit comfortably -- then the fill pass sorts again
  const int64_t nstage = N * nd;
  int32_t* stage = nullptr;
  if (stage_fits(ctx, nstage) && ctx->scr_stage.alloc((size_t)nstage) == hipSuccess)
    stage = ctx->scr_stage.p;
  int32_t h[4] = {0, 0, 0, 0};
  bool counted = false;
  // (called after a device wait) a window of the sort-free adjacency did not fit: build again, sorting
  auto adjacency_overflowed = [&]() { return by_runs && ctx->adj_flag_host && *ctx->adj_flag_host != 0; };
#ifdef ZZZ_EXPERIMENTS
  const bool p1_thread_rows = !getenv("ZZZ_PATTERN_WAVE"); // A/B knob (tools build): P1 rows by wavefronts like P2 / P3
#else
  const bool p1_thread_rows = true;
#endif
  if (nd == 4 && stage && p1_thread_rows)
  {
    // P1: one thread per row; scal[2] = "a row has more than ROW_T_CAP unique columns"
    int rc = build_adjT_offsets(ctx);
    if (rc)
      return rc;
    ZZZ_DBG("build_adjT_offsets");
    // 8 workgroups per CU = 4 wavefronts per SIMD, what the kernel's 108 registers allow (a cap of 4 left half of that
    // occupancy unused: 2.98 -> ~2.0 ms)
    const dim3 tg((grid_for((int64_t)nb, ROW_T_BLOCK, 256 * 8) + 7) / 8 * 8);
    // rows of up to 16 unique columns first (the sorted list in 16 registers), then up to 32
    for (int cap = 16; cap <= 32; cap *= 2)
    {
      if (cap == 16)
        hipLaunchKernelGGL((k_row_pattern_thread4<16, 4096>), tg, dim3(ROW_T_BLOCK), 0, s, ctx->cell_dofs.p, ctx->adj_off.p,
                           ctx->adj_cells.p, nb, cnt.p, scal.p, scal.p + 2, stage, ctx->adjT_off.p, ctx->adjT_cells.p, ctx->adj_li.p);
      else
        hipLaunchKernelGGL((k_row_pattern_thread4<32, 4096>), tg, dim3(ROW_T_BLOCK), 0, s, ctx->cell_dofs.p, ctx->adj_off.p,
                           ctx->adj_cells.p, nb, cnt.p, scal.p, scal.p + 2, stage, ctx->adjT_off.p, ctx->adjT_cells.p, ctx->adj_li.p);
      ZZZ_DBG("k_row_pattern_thread4");
      ZZZ_HIP(ctx, hipMemcpyAsync(h, scal.p, sizeof(h), hipMemcpyDeviceToHost, s));
      ZZZ_HIP(ctx, hipStreamSynchronize(s));
      if (adjacency_overflowed())
      {
        ctx->adj_runs_n = 0;
        return pattern_build_device(ctx, fallback);
      }
      if (h[2] == 0 || cap == 32)
        break;
      ZZZ_HIP(ctx, hipMemsetAsync(scal.p, 0, 4 * sizeof(int32_t), s));
    }
    counted = h[2] == 0;
    ctx->have_adj_li = counted; // complete only if no row overflowed
    if (!counted)
      ZZZ_HIP(ctx, hipMemsetAsync(scal.p, 0, 4 * sizeof(int32_t), s));
  }
  if (!counted)
  {
    // P2/P3: the positions of the element-matrix entries inside their rows, for the assembly (row_unique_regs)
    uint16_t* pos_out = nullptr;
    ctx->have_asm_pos = false;
    if (nd > 4 && !getenv("ZZZ_ASM_SEARCH") && ctx->asm_pos.alloc((size_t)nstage) == hipSuccess)
      pos_out = ctx->asm_pos.p;
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_row_pattern<false>, dim3(rgrid), dim3(256), 0, s, ctx->cell_dofs.p, nd, bs, ctx->adj_off.p,
                       ctx->adj_cells.p, nb, cnt.p, scal.p, (const int64_t*)nullptr, (int32_t*)nullptr, scal.p + 1, stage, pos_out,
                       scal.p + 3);
    ZZZ_HIP(ctx, hipMemcpyAsync(h, scal.p, sizeof(h), hipMemcpyDeviceToHost, s));
    ZZZ_HIP(ctx, hipStreamSynchronize(s));
    ctx->have_asm_pos = pos_out != nullptr && h[3] == 0;
    if (adjacency_overflowed())
    {
      ctx->adj_runs_n = 0;
      return pattern_build_device(ctx, fallback);
    }
  }
  if (h[1] > 0)
  {
    *fallback = true;
    return fail(ctx, ZZZ_ERR_LIMIT, "a row gathers %d candidate columns (device limit %d)", h[1], PAT_CAP);
  }
  ZZZ_HIP(ctx, rocprim::exclusive_scan(tmp.p, tb2, cnt.p, bptr.p, (int64_t)0, (size_t)nb + 1, rocprim::plus<int64_t>(), s));
  int64_t nblk = 0;
  ZZZ_HIP(ctx, hipMemcpyAsync(&nblk, bptr.p + nb, sizeof(int64_t), hipMemcpyDeviceToHost, s));
  ZZZ_HIP(ctx, hipStreamSynchronize(s));
  const int64_t nnz = nblk * bs * bs;
  if (nblk < 0 || nnz > ((int64_t)1 << 40))
    return fail(ctx, ZZZ_ERR_LIMIT, "%lld nonzeros: use more parts", (long long)nnz);
  ctx->nrows = (int64_t)nb * bs;
  ctx->ncols = ctx->nloc();
  ctx->nnz = nnz;
  ZZZ_HIP(ctx, ctx->rowptr.alloc((size_t)ctx->nrows + 1));
  ZZZ_HIP(ctx, ctx->cols.alloc((size_t)nnz + 8));
  ZZZ_HIP(ctx, ctx->vals.alloc((size_t)nnz + 8));
  ZZZ_HIP(ctx, hipMemsetAsync(ctx->cols.p + nnz, 0, 8 * sizeof(int32_t), s));
  ZZZ_HIP(ctx, hipMemsetAsync(ctx->vals.p + nnz, 0, 8 * sizeof(double), s)); // the assembly writes every entry itself
  hipLaunchKernelGGL(k_scalar_rowptr, dim3(grid_for((int64_t)nb + 1)), dim3(256), 0, s, bptr.p, cnt.p, nb, bs,
                     ctx->rowptr.p);
  if (stage)
    hipLaunchKernelGGL(k_row_copy, dim3(grid_for((int64_t)nb * 16, 256, 8192)), dim3(256), 0, s, stage, ctx->adj_off.p, nd, bs,
                       nb, cnt.p, bptr.p, ctx->cols.p);
  else
    hipLaunchKernelGGL(k_row_pattern<true>, dim3(rgrid), dim3(256), 0, s, ctx->cell_dofs.p, nd, bs, ctx->adj_off.p,
                       ctx->adj_cells.p, nb, cnt.p, scal.p, bptr.p, ctx->cols.p, scal.p + 1, (int32_t*)nullptr, (uint16_t*)nullptr,
                       (int32_t*)nullptr);
  ZZZ_HIP(ctx, hipGetLastError());
  ZZZ_DBG("k_row_copy");
  // 3. tiles
  int rc = build_tiles_device(ctx, h[0]);
  if (rc)
    return rc;
  ZZZ_DBG("build_tiles_device");
  ZZZ_HIP(ctx, hipStreamSynchronize(s));
  return ZZZ_OK;
}
// Which SpMV tiles reference a ghost column (>= nrows)?  One wavefront per tile scans its columns.
__global__ __launch_bounds__(256) void k_tile_ghost_flag(const int4* __restrict__ tiles, int64_t ntiles,
                                                         const int32_t* __restrict__ cols, int32_t nrows,
                                                         uint8_t* __restrict__ flag)
{
  const int lane = threadIdx.x & 63;
  for (int64_t t = blockIdx.x * 4 + (threadIdx.x >> 6); t < ntiles; t += (int64_t)gridDim.x * 4)
  {
    const int4 d = tiles[t];
    bool g = false;
    for (int k = d.z + lane; k < d.w; k += 64)
      g |= cols[k] >= nrows;
    const unsigned long long m = __ballot(g);
    if (lane == 0)
      flag[t] = m != 0ull;
  }
}

// interior / boundary tile lists for the halo-compute overlap of a partitioned matrix
int build_tile_split(zzz_ctx* ctx)
{
  ctx->have_tile_split = false;
  ctx->n_tiles_interior = ctx->n_tiles_boundary = 0;
  if (ctx->n_ghost == 0 || ctx->ntiles == 0)
    return ZZZ_OK;
  DevBuf<uint8_t> flag;
  ZZZ_HIP(ctx, flag.alloc((size_t)ctx->ntiles));
  hipLaunchKernelGGL(k_tile_ghost_flag, dim3(grid_for(ctx->ntiles, 4, 4096)), dim3(256), 0, ctx->stream,
                     reinterpret_cast<const int4*>(ctx->tile_row.p), ctx->ntiles, ctx->cols.p, (int32_t)ctx->nrows, flag.p);
  std::vector<uint8_t> h((size_t)ctx->ntiles);
  ZZZ_HIP(ctx, hipMemcpyAsync(h.data(), flag.p, h.size(), hipMemcpyDeviceToHost, ctx->stream));
  ZZZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  std::vector<int32_t> in, bd;
  for (int64_t t = 0; t < ctx->ntiles; ++t)
    (h[(size_t)t] ? bd : in).push_back((int32_t)t);
  ZZZ_HIP(ctx, ctx->tiles_interior.alloc(in.size()));
  ZZZ_HIP(ctx, ctx->tiles_boundary.alloc(bd.size()));
  if (!in.empty())
    ZZZ_HIP(ctx, hipMemcpy(ctx->tiles_interior.p, in.data(), in.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  if (!bd.empty())
    ZZZ_HIP(ctx, hipMemcpy(ctx->tiles_boundary.p, bd.data(), bd.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  ctx->n_tiles_interior = (int64_t)in.size();
  ctx->n_tiles_boundary = (int64_t)bd.size();
  ctx->have_tile_split = true;
  return ZZZ_OK;
}
ZZZ_PRELOAD_TU(pattern)
} // namespace zzz

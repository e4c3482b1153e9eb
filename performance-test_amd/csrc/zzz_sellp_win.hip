// The CG product for LONG SCALAR ROWS (Poisson P2 / P3: BASELINE configs[4]) with x taken from LDS (round 6).
//
// Replaces PETSc MatMult inside KSPSolve (src/poisson_problem.cpp:168-177) like the other products of this library, with the
// same arithmetic: a row's products are added in ascending column order, mul and add rounded separately -- bit-identical to
// the serial CSR loop (zo_spmv) for finite x.
//
// Why.  On the operator stream (zzz_sellp.hip) a P3 row of ~48 entries costs one gather of x per entry and wavefront
// instruction; slot e of 64 consecutive rows reaches 4-10 cache lines, and the product runs at the rate the CU's address path
// retires those gathers (profiles/r05_sq_c5rank.json: 9 vector-memory instructions per chunk of 8 entries per row, 39 cycles
// each, wavefronts waiting for memory 86 % of their cycles) -- 0.43 ms for 1.55 GB at 6.2 M rows, 0.45 of the HBM peak.  Fewer
// bytes do not help it; fewer vector-memory instructions per entry do.  The columns a SET of rows reaches are few when the rows
// are neighbours in SPACE (all entity types of a small brick of the mesh), not neighbours in the type-major numbering: 2 048
// rows of P3 around one point reach ~4 500 columns.  So, privately to this product:
//   * the rows are put into the Morton order of their nodes' coordinates (no lattice is assumed: whatever mesh) and cut into
//     BLOCKS of 4 096; inside a block they are ordered by length, 64 to a slice, so that a slice pads little;
//   * per block: the WINDOW = the distinct columns its rows reach (a list; at most 11 264), and a DICTIONARY of its distinct
//     values (at most 8 192); per entry a 16-bit window index and a 16-bit value code: 4 B per entry, as on the stream;
//   * one workgroup of 1 024 lanes per block: the window's x values and the dictionary go into LDS (two barriers per block),
//     then a lane per row: per chunk of 8 entries two 16-B loads, eight LDS reads of x, eight of the dictionary, eight mul + add.
//     Two vector-memory instructions per chunk instead of ten; y leaves through the row permutation.
// Every entry of the pattern is kept (exact zeros too: 3-4 % at P3): the structure then depends on the PATTERN only and is
// built once per zzz_csr_pattern_build's matrix; the value codes are refreshed after every assembly (sell_update; for a
// partitioned matrix at the first product after it).
//
// When it applies: scalar matrices and those of block size 3 the block-row form declined (elasticity P2 / P3: their scalar rows
// are long rows like any), natural row order of the stream, at least BW_MIN_AVG entries per row on average, every block
// within the LDS budget, no folded all-reduce on the launch (tools build).  Chebyshev-Jacobi terms ride on it as epilogues (CHEB).
// ZZZ_SELLP_BWIN: 0 never, 1 (default) by size (rows of ~48 entries from 300 000 rows on, of ~27 from 800 000: below, the generic
// product is as fast or faster -- a few blocks for 256 CUs), 2 always.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "zzz_sellp.h"

#include <rocprim/rocprim.hpp>

namespace zzz
{
constexpr int BW_R = 4096;        // rows per block
constexpr int BW_THREADS = 1024;  // one workgroup per block: four rows per lane in the builders, four slices per wavefront in the product
constexpr int BW_WCAP = 11264;    // window: distinct columns of a block at most (88 KiB of LDS in the product; P3: 7 600 on average,
                                  // 10 500 the largest met)
constexpr int BW_WBITS = 14;     // (load factor at most 0.69, 0.46 on average at P3)
constexpr int BW_WHASH = 1 << BW_WBITS; // ... and the slots of the set that finds them (build only)
constexpr int BW_DCAP = 8192;     // distinct values of a block at most (+0.0 = code 0 included; 64 KiB; P3: 2 600 on average, 4 100 the
                                  // largest at 61^3 sub-cubes, 7 100 at 122^3)
constexpr int BW_DBITS = 14;
constexpr int BW_DHASH = 1 << BW_DBITS;
constexpr int BW_MIN_AVG = 24;    // average row length from which the form is considered (P2: ~27, 0.187 -> 0.172 ms at 5 M dofs; P3: ~48,
                                  // 0.44 -> 0.30 ms at 6.2 M; P1's 15 never)
constexpr int BW_SLICES = BW_R / 64;

// ---- structure -----------------------------------------------------------------------------------------------------------
// bounding box of the dof coordinates (one workgroup)
__global__ __launch_bounds__(1024) void k_bw_bbox(const double* __restrict__ dofx, int64_t n, double* __restrict__ out)
{
  __shared__ double sh[6][16];
  double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
  for (int64_t i = threadIdx.x; i < n; i += 1024)
    for (int a = 0; a < 3; ++a)
    {
      const double v = dofx[3 * i + a];
      lo[a] = fmin(lo[a], v);
      hi[a] = fmax(hi[a], v);
    }
  for (int a = 0; a < 3; ++a)
    for (int o = 32; o; o >>= 1)
    {
      lo[a] = fmin(lo[a], __shfl_xor(lo[a], o));
      hi[a] = fmax(hi[a], __shfl_xor(hi[a], o));
    }
  if ((threadIdx.x & 63) == 0)
    for (int a = 0; a < 3; ++a)
    {
      sh[a][threadIdx.x >> 6] = lo[a];
      sh[3 + a][threadIdx.x >> 6] = hi[a];
    }
  __syncthreads();
  if (threadIdx.x < 3)
  {
    double l = 1e300, h = -1e300;
    for (int w = 0; w < 16; ++w)
    {
      l = fmin(l, sh[threadIdx.x][w]);
      h = fmax(h, sh[3 + threadIdx.x][w]);
    }
    out[threadIdx.x] = l;
    out[3 + threadIdx.x] = h;
  }
}

__device__ inline unsigned bw_part1by2(unsigned a)
{
  a &= 0x3ffu;
  a = (a | (a << 16)) & 0x30000ffu;
  a = (a | (a << 8)) & 0x300f00fu;
  a = (a | (a << 4)) & 0x30c30c3u;
  a = (a | (a << 2)) & 0x9249249u;
  return a;
}

// sum over the cells of their extent along each axis (max - min of the four vertices): out[6..8]; the mean is the axis' cell size
__global__ __launch_bounds__(256) void k_bw_cellsize(const double* __restrict__ x, const int32_t* __restrict__ cell_verts, int64_t ncells,
                                                     double* __restrict__ part)
{
  __shared__ double sh[4];
  double e[3] = {0.0, 0.0, 0.0};
  for (int64_t c = blockIdx.x * 256ll + threadIdx.x; c < ncells; c += gridDim.x * 256ll)
  {
    const int4 v = *reinterpret_cast<const int4*>(cell_verts + 4 * c);
    const int vv[4] = {v.x, v.y, v.z, v.w};
    for (int a = 0; a < 3; ++a)
    {
      double lo = 1e300, hi = -1e300;
      for (int k = 0; k < 4; ++k)
      {
        const double t = x[3 * (int64_t)vv[k] + a];
        lo = fmin(lo, t);
        hi = fmax(hi, t);
      }
      e[a] += hi - lo;
    }
  }
  // per-workgroup partials, added in a FIXED order by k_bw_cellsum: an atomicAdd of doubles here made the mean cell size -- and
  // with it the Morton keys of nodes on a key boundary, the rows' order inside their blocks and so the order in which a lane adds
  // its dot-product terms -- depend on which workgroup came first: the product stayed bit-identical, but the CG history of a
  // matrix in block-window form differed from run to run at rounding level (found by the early / late form test, elasticity P2)
  for (int a = 0; a < 3; ++a)
  {
    const double t = block_reduce_sum(e[a], sh);
    if (threadIdx.x == 0)
      part[3 * blockIdx.x + a] = t;
  }
}

__global__ __launch_bounds__(64) void k_bw_cellsum(const double* __restrict__ part, int nparts, double* __restrict__ out)
{
  if (threadIdx.x < 3)
  {
    double t = 0.0;
    for (int k = 0; k < nparts; ++k)
      t += part[3 * k + threadIdx.x];
    out[6 + threadIdx.x] = t;
  }
}

// (row i belongs to node i / bs: the scalar rows of a vector-valued space are ordered by their nodes, components side by side)
__global__ __launch_bounds__(256) void k_bw_keys(const double* __restrict__ dofx, const double* __restrict__ bbox, int64_t ncells, int32_t n,
                                                 int bs,
                                                 uint32_t* __restrict__ key, int32_t* __restrict__ val)
{
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256)
  {
    // Coordinates in units of the axis' mean CELL SIZE (bbox[6..8], summed over ncells), then ONE scale for the three axes: the
    // cells of the Morton curve are cubes in the mesh's own metric, i.e. a block's rows are neighbours in the GRAPH.  (With each
    // axis scaled to the box, or with physical lengths on a mesh of flat cells -- 122 x 122 x 16 sub-cubes of the unit cube --
    // the blocks are 7.6 x wider than high and their windows do not fit: 14 700 columns against 7 600.)
    double units[3], most = 0.0;
    for (int a = 0; a < 3; ++a)
    {
      const double h = bbox[6 + a] / (double)ncells;
      units[a] = h > 0.0 ? 1.0 / h : 0.0;
      most = fmax(most, (bbox[3 + a] - bbox[a]) * units[a]);
    }
    unsigned q[3];
    for (int a = 0; a < 3; ++a)
    {
      const double t = most > 0.0 ? (dofx[3 * (int64_t)(i / bs) + a] - bbox[a]) * units[a] / most : 0.0;
      q[a] = (unsigned)fmin(1023.0, fmax(0.0, t * 1024.0));
    }
    key[i] = bw_part1by2(q[0]) | (bw_part1by2(q[1]) << 1) | (bw_part1by2(q[2]) << 2);
    val[i] = i;
  }
}

// info: [0] a block does not fit (window, values, a row of 2^11 entries or more), [1] blocks that reach a ghost column,
// [6..7] (one 64-bit counter) window entries of all blocks

// Pass 1, one workgroup per block: the block's rows by (length descending, position ascending) -- a bitonic sort, deterministic
// --, kept in memory for pass 2 (skey_all); the block's chunks (a slice's first row is its longest); the start of its window list
// (every block gets BW_WCAP entries of room: no pass over the columns is needed to place it).
__global__ __launch_bounds__(BW_THREADS) void k_bw_sort(const int32_t* __restrict__ order, int32_t nrows, int32_t nblk,
                                                        const rp_t* __restrict__ rowptr, uint32_t* __restrict__ skey_all,
                                                        int32_t* __restrict__ blk_chunks, int64_t* __restrict__ woff, int* __restrict__ info)
{
  __shared__ uint32_t skey[BW_R];
  __shared__ int32_t sl_n[BW_SLICES];
  __shared__ int bad;
  const int tid = threadIdx.x;
  for (int b = blockIdx.x; b < nblk; b += gridDim.x)
  {
    const int r0 = b * BW_R, nb = min(BW_R, nrows - r0);
    if (tid == 0)
      bad = 0;
    __syncthreads();
    for (int p = tid; p < BW_R; p += BW_THREADS)
    {
      unsigned len = 0;
      if (p < nb)
      {
        const int32_t r = order[r0 + p];
        len = (unsigned)(rowptr[r + 1] - rowptr[r]);
        if (len > 2047u)
        {
          bad = 1;
          len = 2047u;
        }
      }
      skey[p] = ((2047u - len) << 12) | (unsigned)p; // (rows beyond the block's last: length 0, they sort last)
    }
    __syncthreads();
    for (int k = 2; k <= BW_R; k <<= 1)
      for (int j = k >> 1; j > 0; j >>= 1)
      {
        for (int i = tid; i < BW_R; i += BW_THREADS)
        {
          const int l = i ^ j;
          if (l > i)
          {
            const uint32_t a = skey[i], c = skey[l];
            const bool up = (i & k) == 0;
            if ((a > c) == up)
            {
              skey[i] = c;
              skey[l] = a;
            }
          }
        }
        __syncthreads();
      }
    for (int q = tid; q < BW_R; q += BW_THREADS)
      skey_all[(int64_t)b * BW_R + q] = skey[q];
    if (tid < BW_SLICES)
    {
      const unsigned len = 2047u - (skey[tid * 64] >> 12);
      sl_n[tid] = (int)((len + 7u) / 8u);
    }
    __syncthreads();
    if (tid == 0)
    {
      int acc = 0;
      for (int s = 0; s < BW_SLICES; ++s)
        acc += sl_n[s];
      blk_chunks[b] = acc;
      woff[b] = (int64_t)b * BW_WCAP;
      if (b == nblk - 1)
        woff[nblk] = (int64_t)nblk * BW_WCAP;
      if (bad)
        info[0] = 1;
    }
    __syncthreads();
  }
}

// a lane's next eight entries of a row (k0 .. k0 + 7; those at or beyond len come back as `fill`): two 16-B loads where the row
// has them, single loads at its end
template <typename T>
__device__ inline void bw_load8(const T* __restrict__ p, int k0, int len, T fill, T (&v)[8])
{
  if (k0 + 8 <= len)
  {
    __builtin_memcpy(v, p + k0, 8 * sizeof(T));
    return;
  }
#pragma unroll
  for (int e = 0; e < 8; ++e)
    v[e] = k0 + e < len ? p[k0 + e] : fill;
}

// Pass 2, one workgroup per block: the block's part of the structure -- perm, slice descriptors, window list, the window index of
// every entry.  The columns are read ONCE: a lane walks its rows chunk by chunk (16-B loads), puts each column into the set and
// parks the SLOT it landed in -- 16 bits, in the entry's own place in ccode; when the set is complete its slots are sorted by
// column (indirectly: 16-bit slot numbers, compared through the set), which numbers the window in ascending column order --
// neighbouring lanes of a slice then read neighbouring window slots, the product's window load gathers x along runs of columns,
// and a row's window indices ascend (cpack's differences are positive) --, and a second walk over the parked slots (its own
// 16-B pieces, no column is read again) turns them into window indices.  Everything the lookups need is in LDS:
// set 64 KiB + slot -> index 32 KiB + sort buffer 32 KiB.  (Until round 6's last third: three walks over the columns by 4-B
// loads, the slot -> index map in memory: 3.0 + 9.5 ms per pattern at 6.2 M rows of P3.)
using BwSort = rocprim::block_radix_sort<int32_t, BW_THREADS, BW_WHASH / BW_THREADS, uint16_t>;
// (out of line: inlined into k_bw_block the sort took hipcc 7.2's instruction selection down)
__device__ __noinline__ void bw_sort_pairs(int32_t (&keys)[BW_WHASH / BW_THREADS], uint16_t (&slots)[BW_WHASH / BW_THREADS],
                                           typename BwSort::storage_type& st, unsigned key_bits)
{
  BwSort().sort(keys, slots, st, 0, key_bits);
}
__global__ __launch_bounds__(BW_THREADS) void k_bw_block(const int32_t* __restrict__ order, int32_t nrows, int32_t nblk,
                                                         const rp_t* __restrict__ rowptr, const int32_t* __restrict__ cols,
                                                         const uint32_t* __restrict__ skey_all, int32_t* __restrict__ blk_wn,
                                                         const int64_t* __restrict__ chunk0, int32_t* __restrict__ perm,
                                                         int2* __restrict__ desc, int32_t* __restrict__ wlist,
                                                         uint16_t* __restrict__ ccode, uint8_t* __restrict__ gflag, int* __restrict__ info,
                                                         uint32_t* __restrict__ cpack, uint8_t* __restrict__ cflag, int phases, int32_t ncols)
{
  // (phases: 4 = everything; the tools build stops a block after the first walk (1), the compaction (2), the sort (3) to time them)
  __shared__ union
  {
    int32_t hcol[BW_WHASH]; // the set of columns (-1: empty)
    typename BwSort::storage_type sort;
  } u;
  __shared__ uint16_t hid[BW_WHASH]; // slot -> window index
  __shared__ int32_t sl_c0[BW_SLICES + 1];
  __shared__ int n_win, any_ghost, bad, max_delta;
  const int tid = threadIdx.x;
  // keys of the sort: the columns, and one value beyond them for the empty slots; key_bits = the bits that tell them apart
  const int32_t pad_key = ncols;
  int key_bits = 1;
  while (key_bits < 31 && ((int64_t)1 << key_bits) <= (int64_t)ncols)
    ++key_bits;
  for (int b = blockIdx.x; b < nblk; b += gridDim.x)
  {
    const int r0 = b * BW_R, nb = min(BW_R, nrows - r0);
    const uint32_t* __restrict__ skey = skey_all + (int64_t)b * BW_R;
    for (int k = tid; k < BW_WHASH; k += BW_THREADS)
      u.hcol[k] = -1;
    if (tid == 0)
      n_win = 0, any_ghost = 0, bad = 0, max_delta = 0;
    if (tid < BW_SLICES)
    {
      const unsigned len = 2047u - (skey[tid * 64] >> 12);
      sl_c0[tid] = (int)((len + 7u) / 8u);
    }
    __syncthreads();
    if (tid == 0)
    {
      int acc = 0;
      for (int s = 0; s < BW_SLICES; ++s)
      {
        const int v = sl_c0[s];
        sl_c0[s] = acc;
        acc += v;
      }
      sl_c0[BW_SLICES] = acc;
    }
    __syncthreads();
    const int64_t cb = chunk0[b];
    // first walk: every column into the set, its slot parked in ccode
    // (1.6 of the kernel's 2.9 ms at 6.2 M rows of P3.  Not the probes' LDS latency: a chunk's eight lookups side by side, one
    // LDS round trip per probe round instead of one per entry, took 1.96 ms.  The rows of a block are neighbours in space, not
    // in the CSR arrays: 4 096 separate segments of ~200 B per block, 0.7 TB/s.)
    for (int q = tid; q < BW_R; q += BW_THREADS)
    {
      const int p = (int)(skey[q] & 4095u);
      const bool real = p < nb;
      const int32_t r = real ? order[r0 + p] : -1;
      perm[(int64_t)b * BW_R + q] = r;
      if (!real)
        continue;
      const int s = q >> 6, ln = q & 63;
      const int64_t c0 = cb + sl_c0[s];
      const rp_t a = rowptr[r];
      const int len = (int)(rowptr[r + 1] - a);
      for (int k0 = 0; k0 < len; k0 += 8)
      {
        int32_t c[8];
        bw_load8(cols + a, k0, len, (int32_t)-1, c);
        unsigned slot[8];
#pragma unroll
        for (int e = 0; e < 8; ++e)
        {
          slot[e] = 0xffffu;
          if (c[e] < 0)
            continue;
          if (c[e] >= nrows)
            any_ghost = 1;
          unsigned h = ((unsigned)c[e] * 2654435761u) >> (32 - BW_WBITS);
          for (int probe = 0; probe < BW_WHASH; ++probe)
          {
            int32_t cur = u.hcol[h];
            if (cur == -1)
            {
              cur = atomicCAS(&u.hcol[h], -1, c[e]);
              if (cur == -1)
              {
                if (atomicAdd(&n_win, 1) >= BW_WCAP)
                  bad = 1;
                break;
              }
            }
            if (cur == c[e] || bad)
              break;
            h = (h + 1) & (BW_WHASH - 1);
          }
          slot[e] = h;
        }
        uint4v pk;
        pk.x = slot[0] | (slot[1] << 16);
        pk.y = slot[2] | (slot[3] << 16);
        pk.z = slot[4] | (slot[5] << 16);
        pk.w = slot[6] | (slot[7] << 16);
        *reinterpret_cast<uint4v*>(ccode + ((c0 + (k0 >> 3)) * 64 + ln) * 8) = pk;
      }
    }
    __syncthreads();
    if (tid == 0)
    {
      atomicMax(&info[3], n_win); // (diagnostics: the largest window met, blocks that did not fit)
      if (bad)
        atomicAdd(&info[4], 1);
    }
    if (bad)
    {
      if (tid == 0)
        info[0] = 1;
      __syncthreads();
      continue;
    }
    if (phases < 2)
      continue;
    // The window in ascending column order: every thread takes its sixteen slots of the set as (column, slot) pairs -- an empty
    // slot's key is beyond every column --, the workgroup sorts the 16 384 pairs by key (rocPRIM's block radix sort, in the set's
    // own LDS: the columns are in registers by then), and pair number i < n_win is window entry i.  (An indirect bitonic sort of
    // the slot numbers through the set took 320 us per block, more than both walks: dependent LDS round trips per compare.)
    const int nw = n_win;
    {
      constexpr int PER = BW_WHASH / BW_THREADS;
      int32_t keys[PER];
      uint16_t slots[PER];
#pragma unroll
      for (int k = 0; k < PER; ++k)
      {
        const int h = tid * PER + k;
        const int32_t c = u.hcol[h];
        keys[k] = c == -1 ? pad_key : c;
        slots[k] = (uint16_t)h;
      }
      __syncthreads(); // (the set's LDS becomes the sort's)
      if (phases < 3)
        continue;
      bw_sort_pairs(keys, slots, u.sort, (unsigned)key_bits);
      if (phases < 4)
      {
        __syncthreads();
        continue;
      }
      const int64_t w0 = (int64_t)b * BW_WCAP;
#pragma unroll
      for (int k = 0; k < PER; ++k)
      {
        const int i = tid * PER + k;
        if (i < nw)
        {
          wlist[w0 + i] = keys[k];
          hid[slots[k]] = (uint16_t)i;
        }
      }
    }
    if (tid < BW_SLICES)
      desc[(int64_t)b * BW_SLICES + tid] = make_int2((int)(cb + sl_c0[tid]), sl_c0[tid + 1] - sl_c0[tid]);
    if (tid == 0)
    {
      blk_wn[b] = nw;
      gflag[b] = any_ghost ? 1 : 0;
      if (any_ghost)
        atomicAdd(&info[1], 1);
      atomicAdd(reinterpret_cast<unsigned long long*>(info + 6), (unsigned long long)nw);
    }
    __syncthreads();
    // second walk: the parked slots become window indices.  Every entry's index twice: as 16 bits (ccode), and PACKED, 12 B per
    // lane and chunk (cpack: the chunk's first index in 16 bits, then seven differences of 10 bits -- the window is in ascending
    // column order, as a row's entries are: differences are positive and small, 69 % of P3's chunks below 256, all below 1 024).
    // The product reads the packed form where every difference of the block fits (cflag); entries beyond the row's last have
    // index 0 in ccode and repeat the last index (difference 0) in cpack.
    for (int q = tid; q < BW_R; q += BW_THREADS)
    {
      const int p = (int)(skey[q] & 4095u);
      if (p >= nb)
        continue;
      const int32_t r = order[r0 + p];
      const int s = q >> 6, ln = q & 63;
      const int64_t c0 = cb + sl_c0[s];
      const int len = (int)(rowptr[r + 1] - rowptr[r]);
      int dmax = 0;
      for (int k0 = 0; k0 < len; k0 += 8)
      {
        uint16_t* const cc = ccode + ((c0 + (k0 >> 3)) * 64 + ln) * 8;
        const uint4v pk = *reinterpret_cast<const uint4v*>(cc);
        const unsigned sl[8] = {pk.x & 0xffffu, pk.x >> 16, pk.y & 0xffffu, pk.y >> 16, pk.z & 0xffffu, pk.z >> 16, pk.w & 0xffffu, pk.w >> 16};
        unsigned id[8];
#pragma unroll
        for (int e = 0; e < 8; ++e)
          id[e] = sl[e] == 0xffffu ? 0xffffu : (unsigned)hid[sl[e]];
        unsigned long long lo = id[0]; // bits 0..15 first index, then seven differences of 10 bits: bits 16..85
        unsigned w2 = 0;
        unsigned prev = id[0];
#pragma unroll
        for (int e = 1; e < 8; ++e)
        {
          const unsigned cur = id[e] == 0xffffu ? prev : id[e];
          const int d = (int)cur - (int)prev;
          dmax = max(dmax, d);
          const unsigned long long dd = (unsigned long long)(unsigned)(d & 1023);
          const int bit = 16 + 10 * (e - 1); // 16, 26, 36, 46, 56, 66, 76
          if (bit < 64)
          {
            lo |= dd << bit;
            if (bit + 10 > 64)
              w2 |= (unsigned)(dd >> (64 - bit));
          }
          else
            w2 |= (unsigned)(dd << (bit - 64));
          prev = cur;
        }
        uint4v out;
        out.x = (id[0] == 0xffffu ? 0u : id[0]) | ((id[1] == 0xffffu ? 0u : id[1]) << 16);
        out.y = (id[2] == 0xffffu ? 0u : id[2]) | ((id[3] == 0xffffu ? 0u : id[3]) << 16);
        out.z = (id[4] == 0xffffu ? 0u : id[4]) | ((id[5] == 0xffffu ? 0u : id[5]) << 16);
        out.w = (id[6] == 0xffffu ? 0u : id[6]) | ((id[7] == 0xffffu ? 0u : id[7]) << 16);
        *reinterpret_cast<uint4v*>(cc) = out;
        uint32_t* dst = cpack + ((c0 + (k0 >> 3)) * 64 + ln) * 3;
        dst[0] = (unsigned)lo;
        dst[1] = (unsigned)(lo >> 32);
        dst[2] = w2;
      }
      if (dmax)
        atomicMax(&max_delta, dmax);
    }
    __syncthreads();
    if (tid == 0)
      cflag[b] = max_delta < 1024 ? 1 : 0;
    __syncthreads();
  }
}

// ---- values (after every assembly) ---------------------------------------------------------------------------------------
// One workgroup per block: the block's distinct values into a set in LDS; the table and every entry's code (code 0 = +0.0).
// The values are read ONCE (16-B loads): a lane parks the set's SLOT of each entry in the entry's place in vcode; the occupied
// slots are then numbered in slot order (deterministic: the dictionary's order does not depend on which lane came first), the
// table written, the numbers put into the set's own slots, and a second walk over the parked slots turns them into codes.
// info[2]: a block holds more values than the table.
__global__ __launch_bounds__(BW_THREADS) void k_bw_values(const int32_t* __restrict__ perm, int32_t nblk, const rp_t* __restrict__ rowptr,
                                                          const unsigned long long* __restrict__ vals, const int2* __restrict__ desc,
                                                          uint16_t* __restrict__ vcode, unsigned long long* __restrict__ dict,
                                                          int32_t* __restrict__ dnum, int* __restrict__ info,
                                                          uint32_t* __restrict__ vpack, uint8_t* __restrict__ vflag)
{
  __shared__ unsigned long long hval[BW_DHASH];
  __shared__ int wsum[BW_THREADS / 64];
  __shared__ int n_val, bad;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  for (int b = blockIdx.x; b < nblk; b += gridDim.x)
  {
    for (int k = tid; k < BW_DHASH; k += BW_THREADS)
      hval[k] = ~0ull;
    if (tid == 0)
      n_val = 1, bad = 0; // (entry 0 is +0.0)
    __syncthreads();
    unsigned long long* const tab = dict + (int64_t)b * BW_DCAP;
    for (int q = tid; q < BW_R; q += BW_THREADS)
    {
      const int32_t r = perm[(int64_t)b * BW_R + q];
      if (r < 0)
        continue;
      const int s = q >> 6, ln = q & 63;
      const int64_t c0 = desc[(int64_t)b * BW_SLICES + s].x;
      const rp_t a = rowptr[r];
      const int len = (int)(rowptr[r + 1] - a);
      unsigned long long last = 0ull;
      unsigned last_slot = 0xffffu;
      for (int k0 = 0; k0 < len; k0 += 8)
      {
        unsigned long long v[8];
        bw_load8(vals + a, k0, len, 0ull, v);
        unsigned slot[8];
#pragma unroll
        for (int e = 0; e < 8; ++e)
        {
          slot[e] = 0xffffu; // (+0.0: code 0)
          if (v[e] == 0ull)
            continue;
          if (v[e] == last)
          {
            slot[e] = last_slot;
            continue;
          }
          if (v[e] == ~0ull)
          {
            bad = 1;
            continue;
          }
          unsigned h = (unsigned)(((v[e] ^ (v[e] >> 29)) * 0x9E3779B97F4A7C15ull) >> (64 - BW_DBITS));
          for (int probe = 0; probe < BW_DHASH; ++probe)
          {
            unsigned long long cur = hval[h];
            if (cur == ~0ull)
            {
              cur = atomicCAS(&hval[h], ~0ull, v[e]);
              if (cur == ~0ull)
              {
                if (atomicAdd(&n_val, 1) >= BW_DCAP)
                  bad = 1;
                break;
              }
            }
            if (cur == v[e] || bad)
              break;
            h = (h + 1) & (BW_DHASH - 1);
          }
          slot[e] = h;
          last = v[e];
          last_slot = h;
        }
        uint4v pk;
        pk.x = slot[0] | (slot[1] << 16);
        pk.y = slot[2] | (slot[3] << 16);
        pk.z = slot[4] | (slot[5] << 16);
        pk.w = slot[6] | (slot[7] << 16);
        *reinterpret_cast<uint4v*>(vcode + ((c0 + (k0 >> 3)) * 64 + ln) * 8) = pk;
      }
    }
    __syncthreads();
    const int nv = n_val;
    if (tid == 0)
    {
      atomicMax(&info[3], nv); // (diagnostics: the largest dictionary met, blocks that did not fit)
      if (bad)
        atomicAdd(&info[4], 1);
    }
    if (bad)
    {
      if (tid == 0)
        info[2] = 1;
      __syncthreads();
      continue;
    }
    // codes: the occupied slots numbered in slot order from 1; the table; the code into the slot itself
    {
      constexpr int PER = BW_DHASH / BW_THREADS;
      int mine = 0;
      for (int k = 0; k < PER; ++k)
        mine += hval[tid * PER + k] != ~0ull ? 1 : 0;
      int incl = mine;
      for (int d = 1; d < 64; d <<= 1)
      {
        const int t = __shfl_up(incl, d);
        if (lane >= d)
          incl += t;
      }
      if (lane == 63)
        wsum[wv] = incl;
      __syncthreads();
      int off = 0;
      for (int q = 0; q < wv; ++q)
        off += wsum[q];
      int code = 1 + off + incl - mine;
      for (int k = 0; k < PER; ++k)
      {
        const int h = tid * PER + k;
        const unsigned long long v = hval[h];
        if (v != ~0ull)
        {
          tab[code] = v;
          hval[h] = (unsigned long long)code;
          ++code;
        }
      }
      if (tid == 0)
      {
        tab[0] = 0ull;
        dnum[b] = nv;
        vflag[b] = nv <= 4096 ? 1 : 0; // the codes fit 12 bits: the product reads them packed, 12 B per lane and chunk (vpack)
      }
    }
    __syncthreads();
    const bool vp = nv <= 4096;
    for (int q = tid; q < BW_R; q += BW_THREADS)
    {
      const int32_t r = perm[(int64_t)b * BW_R + q];
      if (r < 0)
        continue;
      const int s = q >> 6, ln = q & 63;
      const int64_t c0 = desc[(int64_t)b * BW_SLICES + s].x;
      const int len = (int)(rowptr[r + 1] - rowptr[r]);
      for (int k0 = 0; k0 < len; k0 += 8)
      {
        uint16_t* const vc = vcode + ((c0 + (k0 >> 3)) * 64 + ln) * 8;
        const uint4v pk = *reinterpret_cast<const uint4v*>(vc);
        const unsigned sl[8] = {pk.x & 0xffffu, pk.x >> 16, pk.y & 0xffffu, pk.y >> 16, pk.z & 0xffffu, pk.z >> 16, pk.w & 0xffffu, pk.w >> 16};
        unsigned code[8];
#pragma unroll
        for (int e = 0; e < 8; ++e)
          code[e] = sl[e] == 0xffffu ? 0u : (unsigned)hval[sl[e]];
        uint4v out;
        out.x = code[0] | (code[1] << 16);
        out.y = code[2] | (code[3] << 16);
        out.z = code[4] | (code[5] << 16);
        out.w = code[6] | (code[7] << 16);
        *reinterpret_cast<uint4v*>(vc) = out;
        if (vp)
        {
          // eight codes of 12 bits: bits 0..95
          const unsigned long long lo = (unsigned long long)code[0] | ((unsigned long long)code[1] << 12) | ((unsigned long long)code[2] << 24)
                                        | ((unsigned long long)code[3] << 36) | ((unsigned long long)code[4] << 48)
                                        | ((unsigned long long)code[5] << 60);
          const unsigned hi = (code[5] >> 4) | (code[6] << 8) | (code[7] << 20);
          uint32_t* dst = vpack + ((c0 + (k0 >> 3)) * 64 + ln) * 3;
          dst[0] = (unsigned)lo;
          dst[1] = (unsigned)(lo >> 32);
          dst[2] = hi;
        }
      }
    }
    __syncthreads();
  }
}

// ---- the product -----------------------------------------------------------------------------------------------------------
struct WinArgs
{
  int nblk, nrows;
  double* partials;
  const int* stop_flag;
  int64_t nlist;
  int pstride, nn_is_rr;
};

template <bool DOT, bool SR, bool NT, bool CHEB = false>
__global__ __launch_bounds__(BW_THREADS) void spmv_win_kernel(const int32_t* __restrict__ p_perm, const int2* __restrict__ p_desc,
                                                              const int64_t* __restrict__ p_woff, const int32_t* __restrict__ p_wn,
                                                              const int32_t* __restrict__ p_wlist,
                                                              const uint16_t* __restrict__ p_ccode, const uint16_t* __restrict__ p_vcode,
                                                              const uint32_t* __restrict__ p_cpack, const uint32_t* __restrict__ p_vpack,
                                                              const uint8_t* __restrict__ p_cflag, const uint8_t* __restrict__ p_vflag,
                                                              const double* __restrict__ p_dict, const int32_t* __restrict__ p_dnum,
                                                              const double* __restrict__ p_x, double* __restrict__ p_y,
                                                              const double* __restrict__ p_rvec, const int32_t* __restrict__ p_list,
                                                              WinArgs a, ChebEpi epi)
{
  static_assert(!(CHEB && SR), "a Chebyshev term has no residual vector of its own");
  extern __shared__ __attribute__((aligned(16))) double bw_lds[]; // [0, BW_WCAP): the window's x; behind it the dictionary
  __shared__ double red[BW_THREADS / 64];
  double* const xwin = bw_lds;
  double* const dtab = bw_lds + BW_WCAP;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (a.stop_flag && *a.stop_flag) // CG already converged: the host is a few iterations ahead
    return;
  const int64_t nitems = p_list ? a.nlist : (int64_t)a.nblk;
  double dot = 0.0, dot_rx = 0.0, dot_nn = 0.0;
  for (int i = 0;; ++i)
  {
    const int64_t t = xcd_stride_item(nitems, i);
    if (t < 0)
      break;
    const int b = p_list ? p_list[t] : (int)t;
    const int64_t w0 = p_woff[b];
    const int wn = p_wn[b], dn = p_dnum[b];
    const bool cpk = __builtin_amdgcn_readfirstlane((int)p_cflag[b]) != 0, vpk = __builtin_amdgcn_readfirstlane((int)p_vflag[b]) != 0;
    // window and dictionary into LDS: every thread's (at most eleven + eight) entries requested before any is stored, and before
    // the barrier that waits for the previous block's lookups (the loads touch no LDS).
    // (Loads at CLAMPED indices, not loads under a condition: `i < wn ? p[i] : 0` compiles to a branch per load with the wait
    // for the load inside it -- eleven round trips one after the other for the window list, found in the disassembly in the
    // last third of round 6; the entries beyond wn / dn are loaded twice and stored nowhere.  Worth nothing measurable: 0.2345-0.2398
    // against 0.2338-0.2439 ms at 6.2 M rows of P3 -- those loads hit the L2.  Also measured there and dropped: the wavefront's four
    // slices as ONE sequence of chunks, every request two unconditional 16-B loads so that the compiler's waits become
    // vmcnt(4) / vmcnt(5) instead of vmcnt(0) and the look-ahead runs across slice ends and the window load: 0.2465 ms -- the
    // cursors' scalar bookkeeping costs what the overlap gains; the kernel sits at the rate the memory system gives this mix.)
    {
      int32_t wc[BW_WCAP / BW_THREADS];
      double xv[BW_WCAP / BW_THREADS], dv[BW_DCAP / BW_THREADS];
#pragma unroll
      for (int k = 0; k < BW_WCAP / BW_THREADS; ++k)
        wc[k] = p_wlist[w0 + min(tid + k * BW_THREADS, wn - 1)];
#pragma unroll
      for (int k = 0; k < BW_DCAP / BW_THREADS; ++k)
        dv[k] = p_dict[(int64_t)b * BW_DCAP + min(tid + k * BW_THREADS, dn - 1)];
#pragma unroll
      for (int k = 0; k < BW_WCAP / BW_THREADS; ++k)
        xv[k] = p_x[wc[k]];
      __syncthreads(); // the previous block's lookups are done
#pragma unroll
      for (int k = 0; k < BW_DCAP / BW_THREADS; ++k)
        if (tid + k * BW_THREADS < dn)
          dtab[tid + k * BW_THREADS] = dv[k];
#pragma unroll
      for (int k = 0; k < BW_WCAP / BW_THREADS; ++k)
        if (tid + k * BW_THREADS < wn)
          xwin[tid + k * BW_THREADS] = xv[k];
    }
    __syncthreads();
    for (int s = wv; s < BW_SLICES; s += BW_THREADS / 64)
    {
      const int2 ds = p_desc[(int64_t)b * BW_SLICES + s];
      const int64_t c0 = ds.x;
      const int nch = __builtin_amdgcn_readfirstlane(ds.y);
      if (nch == 0)
        continue; // (slices are ordered by length: the rest of the block is empty too, but other wavefronts' are not)
      const int32_t r = p_perm[(int64_t)b * BW_R + s * 64 + lane];
      double xr = 0.0, rr = 0.0;
      if ((DOT || CHEB) && r >= 0)
      {
        xr = p_x[r];
        if (SR)
          rr = p_rvec[r];
      }
      double sum = 0.0;
      // The chunk's eight value codes and eight window indices: packed (12 B per lane each: 12-bit codes; a 16-bit first index
      // and seven 10-bit differences) where the block allows, else 16 bits each.  Two chunks of look-ahead: the raw words of
      // chunk j + 2 are requested before chunk j is decoded and summed (three named stages, the body three times: no register that a
      // load is still writing is copied) -- a slice is otherwise a chain load -> decode -> LDS -> sums per chunk.
      struct Raw
      {
        uint4v v, c; // (.w unused by the packed forms)
      };
      auto request = [&](int jj, Raw& R) {
        if (vpk)
        {
          const uint32_t* __restrict__ vp = p_vpack + ((c0 + jj) * 64 + lane) * 3;
          R.v.x = NT ? __builtin_nontemporal_load(vp) : vp[0];
          R.v.y = NT ? __builtin_nontemporal_load(vp + 1) : vp[1];
          R.v.z = NT ? __builtin_nontemporal_load(vp + 2) : vp[2];
        }
        else
          R.v = NT ? __builtin_nontemporal_load(reinterpret_cast<const uint4v*>(p_vcode + (c0 + jj) * 512) + lane)
                   : reinterpret_cast<const uint4v*>(p_vcode + (c0 + jj) * 512)[lane];
        if (cpk)
        {
          const uint32_t* __restrict__ cp = p_cpack + ((c0 + jj) * 64 + lane) * 3;
          R.c.x = NT ? __builtin_nontemporal_load(cp) : cp[0];
          R.c.y = NT ? __builtin_nontemporal_load(cp + 1) : cp[1];
          R.c.z = NT ? __builtin_nontemporal_load(cp + 2) : cp[2];
        }
        else
          R.c = NT ? __builtin_nontemporal_load(reinterpret_cast<const uint4v*>(p_ccode + (c0 + jj) * 512) + lane)
                   : reinterpret_cast<const uint4v*>(p_ccode + (c0 + jj) * 512)[lane];
      };
      auto consume = [&](const Raw& R) {
        unsigned vc[8], cc[8];
        if (vpk)
        {
          const unsigned a0 = R.v.x, a1 = R.v.y, a2 = R.v.z;
          vc[0] = a0 & 4095u, vc[1] = (a0 >> 12) & 4095u, vc[2] = ((a0 >> 24) | (a1 << 8)) & 4095u, vc[3] = (a1 >> 4) & 4095u;
          vc[4] = (a1 >> 16) & 4095u, vc[5] = ((a1 >> 28) | (a2 << 4)) & 4095u, vc[6] = (a2 >> 8) & 4095u, vc[7] = a2 >> 20;
        }
        else
        {
          vc[0] = R.v.x & 0xffffu, vc[1] = R.v.x >> 16, vc[2] = R.v.y & 0xffffu, vc[3] = R.v.y >> 16;
          vc[4] = R.v.z & 0xffffu, vc[5] = R.v.z >> 16, vc[6] = R.v.w & 0xffffu, vc[7] = R.v.w >> 16;
        }
        if (cpk)
        {
          const unsigned b0 = R.c.x, b1 = R.c.y, b2 = R.c.z;
          cc[0] = b0 & 0xffffu;
          cc[1] = cc[0] + ((b0 >> 16) & 1023u);
          cc[2] = cc[1] + (((b0 >> 26) | (b1 << 6)) & 1023u);
          cc[3] = cc[2] + ((b1 >> 4) & 1023u);
          cc[4] = cc[3] + ((b1 >> 14) & 1023u);
          cc[5] = cc[4] + (((b1 >> 24) | (b2 << 8)) & 1023u);
          cc[6] = cc[5] + ((b2 >> 2) & 1023u);
          cc[7] = cc[6] + ((b2 >> 12) & 1023u);
        }
        else
        {
          cc[0] = R.c.x & 0xffffu, cc[1] = R.c.x >> 16, cc[2] = R.c.y & 0xffffu, cc[3] = R.c.y >> 16;
          cc[4] = R.c.z & 0xffffu, cc[5] = R.c.z >> 16, cc[6] = R.c.w & 0xffffu, cc[7] = R.c.w >> 16;
        }
        double xe[8], ve[8];
#pragma unroll
        for (int e = 0; e < 8; ++e)
        {
          xe[e] = xwin[cc[e]];
          ve[e] = dtab[vc[e]];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e)
          sum += ve[e] * xe[e]; // (entries beyond a row's last carry code 0 = +0.0 and repeat an index: + 0 * x)
      };
      {
        Raw A, B, C;
        A.v = A.c = B.v = B.c = C.v = C.c = uint4v{0u, 0u, 0u, 0u};
        request(0, A);
        if (nch > 1)
          request(1, B);
        for (int jj = 0; jj < nch; jj += 3)
        {
          if (jj + 2 < nch)
            request(jj + 2, C);
          consume(A);
          if (jj + 1 < nch)
          {
            if (jj + 3 < nch)
              request(jj + 3, A);
            consume(B);
          }
          if (jj + 2 < nch)
          {
            if (jj + 4 < nch)
              request(jj + 4, B);
            consume(C);
          }
        }
      }
      if (CHEB)
      {
        // a term of the Chebyshev-Jacobi polynomial as the epilogue (ChebEpi; the generic kernel's arithmetic): x is d, y the next d
        if (r >= 0)
        {
          const double gi = -1.0 * (epi.dinv[r] * sum) + epi.g[r];
          const double dn = epi.c1 * xr + epi.c2 * gi;
          const double zi = epi.z[r] + dn;
          epi.z[r] = zi;
          if (DOT)
          {
            const double ri = epi.r[r];
            dot_rx += ri * zi;
            dot_nn += a.nn_is_rr ? ri * ri : zi * zi;
          }
          else
          {
            epi.g[r] = gi;
            p_y[r] = dn;
          }
        }
      }
      else if (r >= 0)
      {
        p_y[r] = sum;
        if (DOT)
        {
          dot += sum * xr;
          if (SR)
          {
            dot_rx += rr * xr;
            dot_nn += a.nn_is_rr ? rr * rr : xr * xr;
          }
        }
      }
    }
  }
  if (DOT)
  {
    const double sres = CHEB ? 0.0 : block_reduce_sum(dot, red);
    double s1 = 0.0, s2 = 0.0;
    if (SR || CHEB)
    {
      s1 = block_reduce_sum(dot_rx, red);
      s2 = block_reduce_sum(dot_nn, red);
    }
    if (threadIdx.x == 0)
    {
      if (!CHEB)
        a.partials[blockIdx.x] = sres;
      if (SR || CHEB)
      {
        a.partials[a.pstride + blockIdx.x] = s1;
        a.partials[2 * a.pstride + blockIdx.x] = s2;
      }
    }
  }
}

// ---- host side ---------------------------------------------------------------------------------------------------------
static int bw_structure(zzz_ctx* ctx)
{
  hipStream_t s = ctx->stream;
  const int32_t nrows = (int32_t)ctx->nrows;
  const int32_t nblk = (nrows + BW_R - 1) / BW_R;
  ctx->bw_struct_ok = false;
  // Morton order of the rows' nodes
  // (scratch kept in the context: hipFree waits for the whole device -- other ranks' kernels on a shared GPU included)
  DevBuf<double>&dofx = ctx->bw_dofx, &bbox = ctx->bw_bbox;
  DevBuf<uint32_t>&key = ctx->bw_key, &key2 = ctx->bw_key2;
  DevBuf<int32_t>& val = ctx->bw_val;
  if (int rc = dof_coords_device(ctx, dofx, ctx->bw_first))
    return rc;
  ZZZ_HIP(ctx, bbox.alloc(16 + 3 * 2048)); // [0..5] box, [6..8] summed cell extents, [16..) the workgroups' partial sums of those
  ZZZ_HIP(ctx, hipMemsetAsync(bbox.p, 0, 9 * sizeof(double), s));
  ZZZ_HIP(ctx, key.alloc((size_t)nrows));
  ZZZ_HIP(ctx, key2.alloc((size_t)nrows));
  ZZZ_HIP(ctx, val.alloc((size_t)nrows));
  ZZZ_HIP(ctx, ctx->bw_order.alloc((size_t)nrows));
  // (the nodes of a Lagrange space lie in the convex hull of their cells' vertices: the vertices' box will do, and they are few)
  hipLaunchKernelGGL(k_bw_bbox, dim3(1), dim3(1024), 0, s, ctx->x.p, ctx->nverts, bbox.p);
  const unsigned cs_grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>((ctx->ncells + 255) / 256, 2048));
  hipLaunchKernelGGL(k_bw_cellsize, dim3(cs_grid), dim3(256), 0, s, ctx->x.p, ctx->cell_verts.p, ctx->ncells, bbox.p + 16);
  hipLaunchKernelGGL(k_bw_cellsum, dim3(1), dim3(64), 0, s, bbox.p + 16, (int)cs_grid, bbox.p);
  hipLaunchKernelGGL(k_bw_keys, dim3((unsigned)std::min<int64_t>(((int64_t)nrows + 255) / 256, 4096)), dim3(256), 0, s, dofx.p, bbox.p,
                     ctx->ncells, nrows, ctx->bs, key.p, val.p);
  {
    size_t tb = 0;
    ZZZ_HIP(ctx, rocprim::radix_sort_pairs(nullptr, tb, key.p, key2.p, val.p, ctx->bw_order.p, (size_t)nrows, 0, 30, s));
    ZZZ_HIP(ctx, ctx->scr_tmp.grow_keep(tb, ctx->retired));
    ZZZ_HIP(ctx, rocprim::radix_sort_pairs(ctx->scr_tmp.p, tb, key.p, key2.p, val.p, ctx->bw_order.p, (size_t)nrows, 0, 30, s));
  }
  DevBuf<int32_t>& info = ctx->bw_info;
  ZZZ_HIP(ctx, info.reserve(8));
  ZZZ_HIP(ctx, hipMemsetAsync(info.p, 0, 8 * sizeof(int32_t), s));
  ZZZ_HIP(ctx, ctx->bw_blk_chunks.alloc((size_t)nblk + 1));
  ZZZ_HIP(ctx, ctx->bw_blk_wn.alloc((size_t)nblk + 1));
  ZZZ_HIP(ctx, ctx->bw_chunk0.alloc((size_t)nblk + 1));
  ZZZ_HIP(ctx, ctx->bw_woff.alloc((size_t)nblk + 1));
  ZZZ_HIP(ctx, ctx->bw_gflag.alloc((size_t)nblk));
  ZZZ_HIP(ctx, ctx->bw_skey.alloc((size_t)nblk * BW_R));
  ZZZ_HIP(ctx, hipMemsetAsync(ctx->bw_blk_chunks.p + nblk, 0, sizeof(int32_t), s));
  ZZZ_HIP(ctx, hipMemsetAsync(ctx->bw_blk_wn.p, 0, ((size_t)nblk + 1) * sizeof(int32_t), s));
  const unsigned grid = (unsigned)std::min<int32_t>(nblk, 256 * 2);
  hipLaunchKernelGGL(k_bw_sort, dim3(grid), dim3(BW_THREADS), 0, s, ctx->bw_order.p, nrows, nblk, ctx->rowptr.p, ctx->bw_skey.p,
                     ctx->bw_blk_chunks.p, ctx->bw_woff.p, info.p);
  {
    size_t tb = 0;
    ZZZ_HIP(ctx, rocprim::exclusive_scan(nullptr, tb, ctx->bw_blk_chunks.p, ctx->bw_chunk0.p, (int64_t)0, (size_t)nblk + 1, rocprim::plus<int64_t>(), s));
    ZZZ_HIP(ctx, ctx->scr_tmp.grow_keep(tb, ctx->retired));
    ZZZ_HIP(ctx, rocprim::exclusive_scan(ctx->scr_tmp.p, tb, ctx->bw_blk_chunks.p, ctx->bw_chunk0.p, (int64_t)0, (size_t)nblk + 1, rocprim::plus<int64_t>(), s));
  }
  ZZZ_HIP(ctx, hipGetLastError());
  int32_t h[8];
  int64_t tot[2] = {0, 0};
  ZZZ_HIP(ctx, hipMemcpyAsync(h, info.p, sizeof(h), hipMemcpyDeviceToHost, s));
  ZZZ_HIP(ctx, hipMemcpyAsync(&tot[0], ctx->bw_chunk0.p + nblk, sizeof(int64_t), hipMemcpyDeviceToHost, s));
  ZZZ_HIP(ctx, hipStreamSynchronize(s));
  if (getenv("ZZZ_DEBUG_SYNC"))
    fprintf(stderr, "[zzz dbg] bw_structure pass 1: flag %d chunks %lld nblk %d\n", h[0], (long long)tot[0], nblk);
  if (h[0] || tot[0] <= 0)
    return ZZZ_OK; // declined: a row of 2^11 entries or more
  ZZZ_HIP(ctx, ctx->bw_perm.alloc((size_t)nblk * BW_R));
  ZZZ_HIP(ctx, ctx->bw_desc.alloc((size_t)nblk * BW_SLICES * 2));
  ZZZ_HIP(ctx, ctx->bw_wlist.alloc((size_t)nblk * BW_WCAP + 8));
  ZZZ_HIP(ctx, ctx->bw_ccode.alloc((size_t)tot[0] * 512));
  ZZZ_HIP(ctx, ctx->bw_vcode.alloc((size_t)tot[0] * 512));
  ZZZ_HIP(ctx, hipMemsetAsync(ctx->bw_ccode.p, 0, (size_t)tot[0] * 1024, s));
  ZZZ_HIP(ctx, ctx->bw_cpack.alloc((size_t)tot[0] * 192)); // the packed forms: 12 B per lane and chunk
  ZZZ_HIP(ctx, ctx->bw_vpack.alloc((size_t)tot[0] * 192));
  ZZZ_HIP(ctx, ctx->bw_cflag.alloc((size_t)nblk));
  ZZZ_HIP(ctx, ctx->bw_vflag.alloc((size_t)nblk));
  ZZZ_HIP(ctx, hipMemsetAsync(ctx->bw_cpack.p, 0, (size_t)tot[0] * 768, s));
  int phases = 4;
#ifdef ZZZ_EXPERIMENTS
  if (const char* e = getenv("ZZZ_BW_PHASES")) // timing of the builder's phases (an incomplete structure: the form then declines)
    phases = atoi(e);
#endif
  hipLaunchKernelGGL(k_bw_block, dim3(grid), dim3(BW_THREADS), 0, s, ctx->bw_order.p, nrows, nblk, ctx->rowptr.p, ctx->cols.p,
                     ctx->bw_skey.p, ctx->bw_blk_wn.p, ctx->bw_chunk0.p, ctx->bw_perm.p, reinterpret_cast<int2*>(ctx->bw_desc.p),
                     ctx->bw_wlist.p, ctx->bw_ccode.p, ctx->bw_gflag.p, info.p, ctx->bw_cpack.p, ctx->bw_cflag.p, phases, (int32_t)ctx->nloc());
  ZZZ_HIP(ctx, hipGetLastError());
  ZZZ_HIP(ctx, hipMemcpyAsync(h, info.p, sizeof(h), hipMemcpyDeviceToHost, s));
  ZZZ_HIP(ctx, hipStreamSynchronize(s));
  {
    unsigned long long we = 0;
    memcpy(&we, h + 6, sizeof(we));
    tot[1] = (int64_t)we;
  }
  if (getenv("ZZZ_DEBUG_SYNC"))
    fprintf(stderr, "[zzz dbg] bw_structure pass 2: flag %d ghost blocks %d window %lld max window %d bad blocks %d\n", h[0], h[1],
            (long long)tot[1], h[3], h[4]);
  if (h[0] || tot[1] <= 0 || phases < 4)
    return ZZZ_OK; // declined: a block beyond the LDS budget
  // interior / boundary blocks for the halo-compute overlap of a partitioned matrix
  ctx->bw_n_interior = ctx->bw_n_boundary = 0;
  ctx->bw_have_split = false;
  if (ctx->n_ghost > 0 || ctx->have_group_split)
  {
    std::vector<uint8_t> gf((size_t)nblk);
    ZZZ_HIP(ctx, hipMemcpyAsync(gf.data(), ctx->bw_gflag.p, gf.size(), hipMemcpyDeviceToHost, s));
    ZZZ_HIP(ctx, hipStreamSynchronize(s));
    std::vector<int32_t> in, bd;
    for (int32_t q = 0; q < nblk; ++q)
      (gf[(size_t)q] ? bd : in).push_back(q);
    ZZZ_HIP(ctx, ctx->bw_list_interior.alloc(in.size()));
    ZZZ_HIP(ctx, ctx->bw_list_boundary.alloc(bd.size()));
    if (!in.empty())
      ZZZ_HIP(ctx, hipMemcpyAsync(ctx->bw_list_interior.p, in.data(), in.size() * sizeof(int32_t), hipMemcpyHostToDevice, s));
    if (!bd.empty())
      ZZZ_HIP(ctx, hipMemcpyAsync(ctx->bw_list_boundary.p, bd.data(), bd.size() * sizeof(int32_t), hipMemcpyHostToDevice, s));
    ZZZ_HIP(ctx, hipStreamSynchronize(s));
    ctx->bw_n_interior = (int64_t)in.size();
    ctx->bw_n_boundary = (int64_t)bd.size();
    ctx->bw_have_split = true;
  }
  ctx->bw_nblk = nblk;
  ctx->bw_chunks = tot[0];
  ctx->bw_window_entries = tot[1];
  ctx->bw_struct_ok = true;
  return ZZZ_OK;
}

// Called when the matrix is assembled (sell_update) or at the stream's first use after an assembly (sellp_active).  Declined (bw_on stays false, nothing else changes): block
// size 3, short rows, sorted stream, a block beyond the LDS budget, ZZZ_SELLP_BWIN=0.
int sellp_win_build(zzz_ctx* ctx)
{
  ctx->bw_on = false;
  // (ZZZ_SELLP_BWIN=2 also takes P2's rows of ~27 -- the tests of the form at small sizes -- but never P1's 15)
  // (block size 3 and rows of ~44: elasticity P1 where the block-row form declined -- the generic product with its x windows is
  // faster there, 324 against 427 ms per C4 solve; its P2 / P3, rows of 80-170, take the form)
  const int64_t min_avg = ctx->sellp_bwin == 2 ? 20 : (ctx->bs == 3 ? 64 : BW_MIN_AVG);
  // (block size 3: the block-row form first -- elasticity P1, 2.4 x; where it declines, P2 / P3, the scalar rows are long rows like any)
  if (!ctx->sellp_bwin || ctx->bk_on || ctx->sp_sorted || ctx->nrows <= 0 || ctx->nnz < min_avg * ctx->nrows)
    return ZZZ_OK;
  // size rule (one MI355X, P3 / P2 cubes): rows of ~48 win from 390 k rows on (0.028 against 0.046 ms; 0.033 against 0.072 at 0.9 M;
  // about equal at 118 k), rows of ~27 from 0.9 M on (0.027 against 0.031; 0.022 against 0.016 at 275 k)
  if (ctx->sellp_bwin == 1 && ctx->nrows < (ctx->nnz >= 40 * ctx->nrows ? 300000 : 800000))
    return ZZZ_OK;
  if (getenv("ZZZ_DEBUG_SYNC"))
    fprintf(stderr, "[zzz dbg] sellp_win_build: rows %lld nnz %lld knob %d\n", (long long)ctx->nrows, (long long)ctx->nnz, ctx->sellp_bwin);
  if (ctx->nrows >= (int64_t)1 << 31 || ctx->order == 0 || ctx->ncells <= 0)
    return ZZZ_OK;
  hipStream_t s = ctx->stream;
  if (ctx->bw_struct_version != ctx->pattern_version || !ctx->bw_struct_ok)
  {
    ctx->bw_struct_version = ctx->pattern_version;
    if (int rc = bw_structure(ctx))
      return rc;
  }
  if (getenv("ZZZ_DEBUG_SYNC"))
    fprintf(stderr, "[zzz dbg] bw structure ok %d\n", (int)ctx->bw_struct_ok);
  if (!ctx->bw_struct_ok)
    return ZZZ_OK;
  const int32_t nblk = ctx->bw_nblk;
  ZZZ_HIP(ctx, ctx->bw_dict.alloc((size_t)nblk * BW_DCAP));
  ZZZ_HIP(ctx, ctx->bw_dnum.alloc((size_t)nblk));
  DevBuf<int32_t>& info = ctx->bw_info;
  ZZZ_HIP(ctx, hipMemsetAsync(info.p, 0, 8 * sizeof(int32_t), s));
  ZZZ_HIP(ctx, hipMemsetAsync(ctx->bw_vcode.p, 0, (size_t)ctx->bw_chunks * 1024, s));
  ZZZ_HIP(ctx, hipMemsetAsync(ctx->bw_vpack.p, 0, (size_t)ctx->bw_chunks * 768, s));
  const unsigned vgrid = (unsigned)std::min<int32_t>(nblk, 256 * 2);
  hipLaunchKernelGGL(k_bw_values, dim3(vgrid), dim3(BW_THREADS), 0, s, ctx->bw_perm.p, nblk,
                     ctx->rowptr.p, reinterpret_cast<const unsigned long long*>(ctx->vals.p), reinterpret_cast<const int2*>(ctx->bw_desc.p),
                     ctx->bw_vcode.p, reinterpret_cast<unsigned long long*>(ctx->bw_dict.p), ctx->bw_dnum.p, info.p, ctx->bw_vpack.p,
                     ctx->bw_vflag.p);
  ZZZ_HIP(ctx, hipGetLastError());
  int32_t h[8];
  ZZZ_HIP(ctx, hipMemcpyAsync(h, info.p, sizeof(h), hipMemcpyDeviceToHost, s));
  ZZZ_HIP(ctx, hipStreamSynchronize(s));
  if (getenv("ZZZ_DEBUG_SYNC"))
    fprintf(stderr, "[zzz dbg] bw values: flag %d max dictionary %d bad blocks %d of %d\n", h[2], h[3], h[4], nblk);
  if (h[2])
    return ZZZ_OK; // a block with more distinct values than the table holds (an irregular mesh): the stream serves the product
  if (!ctx->bw_lds_attr)
  {
#define ZZZ_BW_ATTR(DOT, SR, NT)                                                                                                   \
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&spmv_win_kernel<DOT, SR, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                            (BW_WCAP + BW_DCAP) * 8)
    ZZZ_BW_ATTR(true, true, true);
    ZZZ_BW_ATTR(true, true, false);
    ZZZ_BW_ATTR(true, false, true);
    ZZZ_BW_ATTR(true, false, false);
    ZZZ_BW_ATTR(false, false, true);
    ZZZ_BW_ATTR(false, false, false);
#define ZZZ_BW_ATTRC(DOT, NT)                                                                                                      \
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&spmv_win_kernel<DOT, false, NT, true>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                            (BW_WCAP + BW_DCAP) * 8)
    ZZZ_BW_ATTRC(true, true);
    ZZZ_BW_ATTRC(true, false);
    ZZZ_BW_ATTRC(false, true);
    ZZZ_BW_ATTRC(false, false);
#undef ZZZ_BW_ATTRC
#undef ZZZ_BW_ATTR
    ZZZ_HIP(ctx, hipGetLastError());
    ctx->bw_lds_attr = true;
  }
  // bytes a product reads: codes, descriptors, permutation, window lists and their x (L2), dictionaries
  ctx->bw_bytes = (int64_t)nblk * BW_SLICES * 8 + (int64_t)nblk * BW_R * 4 + ctx->bw_window_entries * 4 + (int64_t)nblk * 18;
  {
    std::vector<int32_t> dn((size_t)nblk), bc((size_t)nblk);
    std::vector<uint8_t> cf((size_t)nblk), vf((size_t)nblk);
    ZZZ_HIP(ctx, hipMemcpyAsync(dn.data(), ctx->bw_dnum.p, dn.size() * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    ZZZ_HIP(ctx, hipMemcpyAsync(bc.data(), ctx->bw_blk_chunks.p, bc.size() * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    ZZZ_HIP(ctx, hipMemcpyAsync(cf.data(), ctx->bw_cflag.p, cf.size(), hipMemcpyDeviceToHost, s));
    ZZZ_HIP(ctx, hipMemcpyAsync(vf.data(), ctx->bw_vflag.p, vf.size(), hipMemcpyDeviceToHost, s));
    ZZZ_HIP(ctx, hipStreamSynchronize(s));
    int64_t t = 0, codes = 0, packed = 0;
    for (int32_t q = 0; q < nblk; ++q)
    {
      t += dn[(size_t)q];
      codes += (int64_t)bc[(size_t)q] * ((cf[(size_t)q] ? 768 : 1024) + (vf[(size_t)q] ? 768 : 1024)); // bytes of codes per chunk
      packed += (cf[(size_t)q] ? 1 : 0) + (vf[(size_t)q] ? 1 : 0);
    }
    ctx->bw_bytes += t * 8 + codes;
    ctx->bw_dict_entries = t;
    ctx->bw_packed_planes = packed; // of 2 nblk
  }
  ctx->bw_on = true;
  return ZZZ_OK;
}

bool sellp_win_serves(const zzz_ctx* ctx) { return ctx->bw_on && ctx->sellp_bwin && !ctx->sp_sorted; }

int sellp_win_grid(const zzz_ctx* ctx, int64_t items)
{
  (void)ctx;
  const int64_t g = (items + 7) / 8 * 8;
  return (int)std::max<int64_t>(8, std::min<int64_t>(g, 256));
}

bool launch_sellp_win(zzz_ctx* ctx, bool dot, bool nt, int grid, const double* x, double* y, double* partials, const int* stop,
                      const int32_t* list, int64_t nlist, const double* rvec, int nn_is_rr, const ChebEpi* epi)
{
  if (!sellp_win_serves(ctx))
    return false;
  WinArgs a;
  a.nblk = ctx->bw_nblk;
  a.nrows = (int)ctx->nrows;
  a.partials = partials;
  a.stop_flag = stop;
  a.nlist = nlist;
  a.pstride = SPMV_PSTRIDE;
  a.nn_is_rr = nn_is_rr;
  const size_t lds = (size_t)(BW_WCAP + BW_DCAP) * 8;
#define ZZZ_BW_GO4(DOT, SR, NT, CHEB, EPI)                                                                                         \
  hipLaunchKernelGGL((spmv_win_kernel<DOT, SR, NT, CHEB>), dim3(grid), dim3(BW_THREADS), lds, ctx->stream, ctx->bw_perm.p,         \
                     reinterpret_cast<const int2*>(ctx->bw_desc.p), ctx->bw_woff.p, ctx->bw_blk_wn.p, ctx->bw_wlist.p, ctx->bw_ccode.p, ctx->bw_vcode.p, \
                     ctx->bw_cpack.p, ctx->bw_vpack.p, ctx->bw_cflag.p, ctx->bw_vflag.p, ctx->bw_dict.p, ctx->bw_dnum.p, x, y, rvec, list, a, EPI)
#define ZZZ_BW_GO(DOT, SR, NT) ZZZ_BW_GO4(DOT, SR, NT, false, ChebEpi())
  if (epi)
  {
    if (dot)
    {
      if (nt)
        ZZZ_BW_GO4(true, false, true, true, *epi);
      else
        ZZZ_BW_GO4(true, false, false, true, *epi);
    }
    else
    {
      if (nt)
        ZZZ_BW_GO4(false, false, true, true, *epi);
      else
        ZZZ_BW_GO4(false, false, false, true, *epi);
    }
  }
  else if (dot && rvec)
  {
    if (nt)
      ZZZ_BW_GO(true, true, true);
    else
      ZZZ_BW_GO(true, true, false);
  }
  else if (dot)
  {
    if (nt)
      ZZZ_BW_GO(true, false, true);
    else
      ZZZ_BW_GO(true, false, false);
  }
  else
  {
    if (nt)
      ZZZ_BW_GO(false, false, true);
    else
      ZZZ_BW_GO(false, false, false);
  }
#undef ZZZ_BW_GO4
#undef ZZZ_BW_GO
  return true;
}
ZZZ_PRELOAD_TU(sellp_win)
} // namespace zzz

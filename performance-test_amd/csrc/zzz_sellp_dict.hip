// The CG operator stream, part 2 of 3: the VALUE DICTIONARIES of a packed stream (matrix-wide, copied into LDS by the product's
// workgroups; per slice for long rows).  Packer: zzz_sellp_pack.hip; product: zzz_sellp.hip / zzz_sellp_pipe.hip.
#include <climits>
#include <cstring>
#include <cstdlib>

#include "zzz_sellp.h"

#include <rocprim/rocprim.hpp>

namespace zzz
{
struct Even2
{
  __host__ __device__ int64_t operator()(int32_t n) const { return ((int64_t)n + 1) & ~(int64_t)1; }
};
// ---- value dictionary -----------------------------------------------------------------------------------
// The entries of the packed stream as (slice, chunk, lane, slot) with the values the product would load: slots of a
// slice's last chunk beyond its width and lanes without a row are never loaded (and hold anything).
constexpr int SP_DICT_BITS = 18;                        // table of 2^18 slots for at most 65 535 values
constexpr unsigned long long SP_DICT_EMPTY = ~0ull;     // (a NaN pattern no assembled value has; met all the same: no dictionary)
constexpr int SP_DICT_MAX = 65535;
constexpr int SP_DICT_LDS_MAX = SP_DICT_LDS_ENTRIES;

__device__ inline unsigned sp_dict_hash(unsigned long long b)
{
  b ^= b >> 29;
  b *= 0x9E3779B97F4A7C15ull;
  return (unsigned)(b >> (64 - SP_DICT_BITS));
}

// value of entry (chunk c of width w, lane, slot e) in the value blocks: [4][64 lanes][2]; the last entry of an odd width
// sits alone, 8 B per lane (emit_chunk)
__device__ inline unsigned long long sp_value_bits(const double* __restrict__ svals, int64_t c, int w, int lane, int e)
{
  const int64_t at = ((w & 1) && e == w - 1) ? 128 * (e >> 1) + lane : 128 * (e >> 1) + 2 * lane + (e & 1);
  return reinterpret_cast<const unsigned long long*>(svals)[c * 512 + at];
}

// info[0] distinct values so far, info[1] overflow / unusable.  The lanes of a slice mostly hold the same value in a slot:
// one lane per distinct value of the wavefront goes to the table; the table is read past the L1 cache (a line cached as
// empty before another CU's insertion would send every later occurrence of that value to the atomic: 3.4 ms at 1.25 M rows
// instead of 0.05).
template <bool PERM>
__global__ __launch_bounds__(256) void k_sp_dict_insert(const int2* __restrict__ desc, const int32_t* __restrict__ perm,
                                                         const double* __restrict__ svals, int nrows, int64_t nslices,
                                                         unsigned long long* __restrict__ table, int* __restrict__ info, int limit)
{
  const int lane = threadIdx.x & 63;
  for (int64_t s = blockIdx.x * 4ll + (threadIdx.x >> 6); s < nslices; s += gridDim.x * 4ll)
  {
    if (__hip_atomic_load(&info[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
      return; // more distinct values than the dictionary may hold (an unstructured mesh): nothing left to find out
    const int r = PERM ? perm[s * 64 + lane] : (int)(s * 64 + lane);
    const bool row = r >= 0 && r < nrows;
    const int2 ds = desc[s];
    const int c0 = ds.x, nch = ds.y & 0xffffff, wl = ds.y >> 24;
    unsigned long long last = 0ull; // (+0.0 is code 0 without the table)
    for (int j = 0; j < nch; ++j)
    {
      const int w = j + 1 < nch ? 8 : wl;
      for (int e = 0; e < w; ++e)
      {
        const unsigned long long b = row ? sp_value_bits(svals, c0 + j, w, lane, e) : 0ull;
        bool need = b != last && b != 0ull; // (a row repeats its values: the previous one is in the table already)
        last = b;
        unsigned long long todo = __ballot(need);
        while (todo)
        {
          if (__hip_atomic_load(&info[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            return; // (the waves in flight when the limit is met would fill the table up otherwise)
          const int src = __ffsll((long long)todo) - 1;
          const unsigned long long bb = ((unsigned long long)(unsigned)__shfl((int)(b >> 32), src) << 32)
                                        | (unsigned)__shfl((int)(unsigned)b, src);
          if (lane == src)
          {
            if (bb == SP_DICT_EMPTY)
              info[1] = 1;
            else
            {
              unsigned h = sp_dict_hash(bb);
              for (int probe = 0; probe < (1 << SP_DICT_BITS); ++probe)
              {
                const unsigned long long cur = __hip_atomic_load(&table[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (cur == bb)
                  break;
                if (cur == SP_DICT_EMPTY)
                {
                  const unsigned long long old = atomicCAS(&table[h], SP_DICT_EMPTY, bb);
                  if (old == SP_DICT_EMPTY)
                  {
                    if (atomicAdd(&info[0], 1) >= limit - 1)
                      info[1] = 1;
                    break;
                  }
                  if (old == bb)
                    break;
                }
                h = (h + 1) & ((1u << SP_DICT_BITS) - 1);
                if (__hip_atomic_load(&info[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                  break; // too many distinct values: the table may be filling up, stop looking
              }
            }
          }
          need = need && b != bb;
          todo = __ballot(need);
        }
      }
    }
  }
}

// codes: every thread numbers the occupied slots it meets (slot = k * 1024 + thread), threads in order; code 0 = +0.0
__global__ __launch_bounds__(1024) void k_sp_dict_number(const unsigned long long* __restrict__ table, int32_t* __restrict__ slot_code,
                                                         double* __restrict__ dict, int* __restrict__ info, int lds_max, int forced)
{
  __shared__ int wsum[16];
  if (info[1])
    return;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int mine = 0;
  for (int k = threadIdx.x; k < (1 << SP_DICT_BITS); k += 1024)
    mine += table[k] != SP_DICT_EMPTY ? 1 : 0;
  // exclusive scan of `mine` over the 1024 threads
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
  int off = 1; // (code 0 is +0.0)
  for (int q = 0; q < wv; ++q)
    off += wsum[q];
  int code = off + incl - mine;
  for (int k = threadIdx.x; k < (1 << SP_DICT_BITS); k += 1024)
  {
    const unsigned long long b = table[k];
    if (b != SP_DICT_EMPTY)
    {
      slot_code[k] = code;
      if (code <= SP_DICT_MAX)
        dict[code] = __longlong_as_double((long long)b);
      ++code;
    }
  }
  if (threadIdx.x == 1023)
  {
    dict[0] = 0.0;
    info[2] = code; // entries of the dictionary, +0.0 included
    if (code > lds_max && !forced)
      info[1] = 2; // too large for the LDS copy: the stream stays as doubles, the encoding pass has nothing to do
  }
}

// the stream's values as codes, [chunk][lane][8] (16 B per lane and chunk); info[4..5]: bytes the product reads in this form
template <bool PERM>
__global__ __launch_bounds__(256) void k_sp_dict_encode(const int2* __restrict__ desc, const int32_t* __restrict__ perm,
                                                         const double* __restrict__ svals, const int32_t* __restrict__ meta,
                                                         int nrows, int64_t nslices, const unsigned long long* __restrict__ table,
                                                         const int32_t* __restrict__ slot_code, uint16_t* __restrict__ vcode,
                                                         int* __restrict__ info)
{
  if (info[1])
    return;
  const int lane = threadIdx.x & 63;
  unsigned long long bytes = 0;
  for (int64_t s = blockIdx.x * 4ll + (threadIdx.x >> 6); s < nslices; s += gridDim.x * 4ll)
  {
    const int r = PERM ? perm[s * 64 + lane] : (int)(s * 64 + lane);
    const bool row = r >= 0 && r < nrows;
    const int2 ds = desc[s];
    const int c0 = ds.x, nch = ds.y & 0xffffff, wl = ds.y >> 24;
    unsigned long long last = 0ull;
    unsigned last_code = 0;
    for (int j = 0; j < nch; ++j)
    {
      const int w = j + 1 < nch ? 8 : wl;
      unsigned code[8];
#pragma unroll
      for (int e = 0; e < 8; ++e)
      {
        code[e] = 0;
        if (row && e < w)
        {
          const unsigned long long b = sp_value_bits(svals, c0 + j, w, lane, e);
          if (b == 0ull)
            continue;
          if (b != last)
          {
            unsigned h = sp_dict_hash(b);
            while (table[h] != b)
              h = (h + 1) & ((1u << SP_DICT_BITS) - 1);
            last = b;
            last_code = (unsigned)slot_code[h];
          }
          code[e] = last_code;
        }
      }
      uint4v q;
      q.x = code[0] | (code[1] << 16);
      q.y = code[2] | (code[3] << 16);
      q.z = code[4] | (code[5] << 16);
      q.w = code[6] | (code[7] << 16);
      reinterpret_cast<uint4v*>(vcode + (size_t)(c0 + j) * 512)[lane] = q;
      if (lane == 0)
      {
        // what the product reads of this chunk: 1 KiB of value codes, the slot bases, the column codes by the chunk's mode
        const int m0 = meta[(size_t)(c0 + j) * 8];
        unsigned cb = 0;
        if (m0 < 0 && (m0 & 0x40000000))
          cb = 100; // periodic: 25 scalar words
        else if (m0 < 0)
          cb = 2048; // int32 columns
        else if ((m0 & 0x60000000) == 0x20000000)
          cb = 0; // affine
        else if (m0 & 0x40000000)
          cb = 512; // 8-bit codes
        else
          cb = 1024; // 16-bit codes
        bytes += 1024 + 32 + cb;
      }
    }
  }
  if (lane == 0 && bytes)
    atomicAdd(reinterpret_cast<unsigned long long*>(info + 4), bytes);
}

// ---- per-slice value dictionaries (long rows: P3) ----------------------------------------------------------------
// A slice of 64 rows of one entity type holds a few hundred distinct values even where the whole matrix holds thousands
// (P3 at 30^3 sub-cubes: median 296 per slice, all slices below 1 024; 7 400 in the matrix; 8 270 at 61^3).  One wavefront
// per slice: the slice's distinct values into a hash set in LDS (at most 1 023 besides +0.0), numbered as they arrive; then
// every value of the slice as a 16-bit code in the layout of the matrix-wide dictionary's codes ([chunk][lane][8], 16 B per
// lane and chunk), and the table beside it (sd_info[slice] = entries, 0 = this slice stays doubles; the tables back to back in
// sd_vals, slice s at sd_off[s]: a first pass (COUNT) finds the sizes, a scan the offsets -- 1 024 doubles reserved per slice
// were 6.4 GB at 49.8 M rows for 1.8 GB of tables).
// The product copies a slice's table into its wavefront's part of LDS (8 KiB per wavefront: five workgroups per CU).
// Tried: 8-bit codes and tables of 256 (a third of P3's slices qualify: product 0.67 -> 0.54 ms at 6.2 M dofs), tables of 512
// (0.46 ms there, 4.19 ms at 49.8 M dofs), tables of 1 024 (0.47 / 3.68 ms: kept).
// (SD_SLOTS = 2048, SD_MAX = 1024: zzz_sellp.h)
template <bool PERM, bool COUNT>
__global__ __launch_bounds__(128) void k_sp_sd_build(const int2* __restrict__ desc, const int32_t* __restrict__ perm,
                                                      const double* __restrict__ svals, const int32_t* __restrict__ meta,
                                                      int nrows, int64_t nslices, uint16_t* __restrict__ vcode8,
                                                      double* __restrict__ sd_vals, const int64_t* __restrict__ sd_off,
                                                      int32_t* __restrict__ sd_info, unsigned long long* __restrict__ bytes_out)
{
  __shared__ unsigned long long keys_s[2][SD_SLOTS];
  __shared__ uint16_t code_s[2][SD_SLOTS];
  __shared__ int cnt_s[2];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  unsigned long long* const keys = keys_s[wv];
  uint16_t* const codes = code_s[wv];
  unsigned long long bytes = 0;
  for (int64_t s = blockIdx.x * 2ll + wv; s < nslices; s += gridDim.x * 2ll)
  {
    for (int k = lane; k < SD_SLOTS; k += 64)
      keys[k] = ~0ull;
    if (lane == 0)
      cnt_s[wv] = 1; // (entry 0 is +0.0)
    __builtin_amdgcn_wave_barrier();
    const int r = PERM ? perm[s * 64 + lane] : (int)(s * 64 + lane);
    const bool row = r >= 0 && r < nrows;
    const int2 ds = desc[s];
    const int c0 = ds.x, nch = ds.y & 0xffffff, wl = ds.y >> 24;
    if (!COUNT && sd_info[s] == 0) // (the first pass found more values than a table holds: this slice stays doubles)
    {
      if (lane == 0)
      {
        for (int j = 0; j < nch; ++j)
        {
          const int w = j + 1 < nch ? 8 : wl;
          const int m0 = meta[(size_t)(c0 + j) * 8];
          const unsigned cb = (m0 < 0 && (m0 & 0x40000000)) ? 100u : (m0 < 0 ? 2048u : ((m0 & 0x60000000) == 0x20000000 ? 0u : ((m0 & 0x40000000) ? 512u : 1024u)));
          bytes += 32 + cb + (unsigned)((w >> 1) * 1024 + (w & 1) * 512);
        }
        bytes += 4;
        atomicAdd(reinterpret_cast<int*>(bytes_out) + 2, 1); // slices that stay doubles
      }
      continue;
    }
    double* const tab = COUNT ? nullptr : sd_vals + sd_off[s];
    if (!COUNT && lane == 0)
      tab[0] = 0.0;
    unsigned long long last = 0ull;
    for (int j = 0; j < nch; ++j)
    {
      if (cnt_s[wv] > SD_MAX)
        break; // (more values than the table holds: this slice stays doubles)
      const int w = j + 1 < nch ? 8 : wl;
      for (int e = 0; e < w; ++e)
      {
        const unsigned long long b = row ? sp_value_bits(svals, c0 + j, w, lane, e) : 0ull;
        if (b != 0ull && b != last && b != ~0ull)
        {
          unsigned h = sp_dict_hash(b) & (SD_SLOTS - 1);
          for (int probe = 0; probe < SD_SLOTS; ++probe)
          {
            const unsigned long long cur = keys[h];
            if (cur == b)
              break;
            if (cur == ~0ull)
            {
              const unsigned long long old = atomicCAS(&keys[h], ~0ull, b);
              if (old == ~0ull)
              {
                const int c = atomicAdd(&cnt_s[wv], 1);
                codes[h] = (uint16_t)c;
                if (!COUNT && c < SD_MAX)
                  tab[c] = __longlong_as_double((long long)b);
                break;
              }
              if (old == b)
                break;
            }
            h = (h + 1) & (SD_SLOTS - 1);
            if (cnt_s[wv] > SD_MAX)
              break;
          }
        }
        if (b == ~0ull)
          cnt_s[wv] = SD_MAX + 1;
        last = b;
      }
    }
    __builtin_amdgcn_wave_barrier();
    const int n = cnt_s[wv];
    const bool ok = n <= SD_MAX;
    if (COUNT)
    {
      if (lane == 0)
        sd_info[s] = ok ? n : 0;
      __builtin_amdgcn_wave_barrier();
      continue;
    }
    if (ok)
    {
      last = 0ull;
      unsigned last_code = 0;
      for (int j = 0; j < nch; ++j)
      {
        const int w = j + 1 < nch ? 8 : wl;
        unsigned code[8];
#pragma unroll
        for (int e = 0; e < 8; ++e)
        {
          code[e] = 0;
          if (row && e < w)
          {
            const unsigned long long b = sp_value_bits(svals, c0 + j, w, lane, e);
            if (b == 0ull)
              continue;
            if (b != last)
            {
              unsigned h = sp_dict_hash(b) & (SD_SLOTS - 1);
              while (keys[h] != b)
                h = (h + 1) & (SD_SLOTS - 1);
              last = b;
              last_code = codes[h];
            }
            code[e] = last_code;
          }
        }
        uint4v q;
        q.x = code[0] | (code[1] << 16);
        q.y = code[2] | (code[3] << 16);
        q.z = code[4] | (code[5] << 16);
        q.w = code[6] | (code[7] << 16);
        reinterpret_cast<uint4v*>(vcode8 + (size_t)(c0 + j) * 512)[lane] = q;
      }
    }
    if (lane == 0)
    {
      // what the product reads of this slice: the table and per chunk 512 B of codes, or the values as before
      for (int j = 0; j < nch; ++j)
      {
        const int w = j + 1 < nch ? 8 : wl;
        const int m0 = meta[(size_t)(c0 + j) * 8];
        unsigned cb = 0;
        if (m0 < 0 && (m0 & 0x40000000))
          cb = 100;
        else if (m0 < 0)
          cb = 2048;
        else if ((m0 & 0x60000000) == 0x20000000)
          cb = 0;
        else if (m0 & 0x40000000)
          cb = 512;
        else
          cb = 1024;
        bytes += 32 + cb + (ok ? 1024 : (unsigned)((w >> 1) * 1024 + (w & 1) * 512));
      }
      bytes += ok ? (unsigned)n * 8 + 4 : 4;
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (lane == 0 && bytes)
    atomicAdd(bytes_out, bytes);
}

// ---- host side ------------------------------------------------------------------------------------------
// The value dictionary of the finished stream (see zzz_internal.h): distinct values into a hash set, numbered, every
// value of the stream replaced by its code.  Synchronous (once per assembly, at the first use of the stream): 1-2 ms at
// 10 M rows.  More than 65 535 distinct values, or ZZZ_SELLP_DICT=0: the stream keeps being read as values.
int sp_dict_build(zzz_ctx* ctx)
{
  ctx->sp_dict_done = true;
  ctx->sp_dict_on = false;
  ctx->sp_dict_n = 0;
  // ZZZ_SELLP_DICT: 0 never, 2 always (tests at small sizes), 1: for streams of more than 48 MB of values -- below that the
  // whole loop sits in the Infinity Cache, bytes are not what the product waits for, and building the dictionary (three
  // passes over the stream and a synchronisation, ~0.4 ms at 500 k rows) costs more than a solve gains
  if (!ctx->sellp_dict || ctx->sp_chunks <= 0 || (ctx->sellp_dict == 1 && ctx->sp_bytes < 48ll << 20))
    return ZZZ_OK;
  hipStream_t s = ctx->stream;
  const int64_t nsl = ctx->nslices;
  ZZZ_HIP(ctx, ctx->sp_dict_table.alloc((size_t)1 << SP_DICT_BITS));
  ZZZ_HIP(ctx, ctx->sp_dict_slot.alloc((size_t)1 << SP_DICT_BITS));
  ZZZ_HIP(ctx, ctx->sp_dict.alloc((size_t)SP_DICT_MAX + 1));
  ZZZ_HIP(ctx, ctx->sp_vcode.alloc((size_t)ctx->sp_chunks * 512));
  DevBuf<int32_t>& info = ctx->sp_dict_info;
  ZZZ_HIP(ctx, info.reserve(8));
  ZZZ_HIP(ctx, hipMemsetAsync(info.p, 0, 8 * sizeof(int32_t), s));
  ZZZ_HIP(ctx, hipMemsetAsync(ctx->sp_dict_table.p, 0xff, sizeof(unsigned long long) << SP_DICT_BITS, s));
  const int2* desc = reinterpret_cast<const int2*>(ctx->sp_desc.p);
  const unsigned grid = (unsigned)std::min<int64_t>((nsl + 3) / 4, 256 * 16);
  // (the search stops at the first value beyond what will be used: the LDS copy's capacity, unless the memory form is forced)
  const int limit = ctx->sellp_dict == 2 ? SP_DICT_MAX : SP_DICT_LDS_MAX - 1;
  if (ctx->sp_sorted)
    hipLaunchKernelGGL(k_sp_dict_insert<true>, dim3(grid), dim3(256), 0, s, desc, ctx->sp_perm.p, ctx->sp_vals.p, (int)ctx->nrows, nsl,
                       ctx->sp_dict_table.p, info.p, limit);
  else
    hipLaunchKernelGGL(k_sp_dict_insert<false>, dim3(grid), dim3(256), 0, s, desc, (const int32_t*)nullptr, ctx->sp_vals.p,
                       (int)ctx->nrows, nsl, ctx->sp_dict_table.p, info.p, limit);
  hipLaunchKernelGGL(k_sp_dict_number, dim3(1), dim3(1024), 0, s, ctx->sp_dict_table.p, ctx->sp_dict_slot.p, ctx->sp_dict.p, info.p,
                     SP_DICT_LDS_MAX, ctx->sellp_dict == 2 ? 1 : 0);
  if (ctx->sp_sorted)
    hipLaunchKernelGGL(k_sp_dict_encode<true>, dim3(grid), dim3(256), 0, s, desc, ctx->sp_perm.p, ctx->sp_vals.p, ctx->sp_meta.p,
                       (int)ctx->nrows, nsl, ctx->sp_dict_table.p, ctx->sp_dict_slot.p, ctx->sp_vcode.p, info.p);
  else
    hipLaunchKernelGGL(k_sp_dict_encode<false>, dim3(grid), dim3(256), 0, s, desc, (const int32_t*)nullptr, ctx->sp_vals.p,
                       ctx->sp_meta.p, (int)ctx->nrows, nsl, ctx->sp_dict_table.p, ctx->sp_dict_slot.p, ctx->sp_vcode.p, info.p);
  ZZZ_HIP(ctx, hipGetLastError());
  int32_t h[8];
  ZZZ_HIP(ctx, hipMemcpyAsync(h, info.p, sizeof(h), hipMemcpyDeviceToHost, s));
  ZZZ_HIP(ctx, hipStreamSynchronize(s));
  if (h[1] || h[2] <= 0 || h[2] > SP_DICT_MAX + 1)
    return ZZZ_OK;
  // A dictionary too large for the LDS copy is gathered from memory: 2.3x fewer bytes at P3 6.2 M dofs (8 270 values) and the
  // same 0.61-0.63 ms per product -- the gathers, not the bytes, are what the kernel waits for -- so that form is not used
  // unless ZZZ_SELLP_DICT=2 asks for it (tests)
  if (h[2] > SP_DICT_LDS_MAX && ctx->sellp_dict != 2)
    return ZZZ_OK;
  unsigned long long bytes = 0;
  memcpy(&bytes, h + 4, sizeof(bytes));
  ctx->sp_dict_n = h[2];
  ctx->sp_dict_bytes = (int64_t)bytes + (int64_t)h[2] * 8;
  ctx->sp_dict_on = true;
  return ZZZ_OK;
}

// Per-slice dictionaries (k_sp_sd_build) for streams whose global dictionary does not fit LDS: long rows (P3).  Kept when
// they take the stream below 60 % of its bytes.  ZZZ_SELLP_DICT: 0 none of this, 3 slice dictionaries whenever they apply.
int sp_sd_build(zzz_ctx* ctx)
{
  ctx->sp_sd_on = ctx->sp_sd_all = false;
  if (!ctx->sellp_dict || ctx->sp_dict_on || ctx->sp_chunks <= 0 || ctx->sp_win_max > 0)
    return ZZZ_OK;
  if (ctx->sellp_dict != 3 && (ctx->sp_bytes < 48ll << 20 || ctx->sp_chunks < 4 * ctx->nslices))
    return ZZZ_OK; // (small streams: bytes do not matter; short rows -- P1: the table would cost as much as it saves)
  hipStream_t s = ctx->stream;
  const int64_t nsl = ctx->nslices;
  ZZZ_HIP(ctx, ctx->sp_vcode8.alloc((size_t)ctx->sp_chunks * 512));
  ZZZ_HIP(ctx, ctx->sp_sd_info.alloc((size_t)nsl + 1));
  ZZZ_HIP(ctx, ctx->sp_sd_off.alloc((size_t)nsl + 1));
  ZZZ_HIP(ctx, ctx->sp_dict_info.reserve(8));
  ZZZ_HIP(ctx, hipMemsetAsync(ctx->sp_dict_info.p, 0, 8 * sizeof(int32_t), s));
  ZZZ_HIP(ctx, hipMemsetAsync(ctx->sp_sd_info.p + nsl, 0, sizeof(int32_t), s)); // closes the scan
  const int2* desc = reinterpret_cast<const int2*>(ctx->sp_desc.p);
  const unsigned grid = (unsigned)std::min<int64_t>((nsl + 1) / 2, 256 * 8);
  unsigned long long* bytes = reinterpret_cast<unsigned long long*>(ctx->sp_dict_info.p + 4);
  // first pass: the tables' sizes; scan: where each starts; second pass: tables and codes
  if (ctx->sp_sorted)
    hipLaunchKernelGGL((k_sp_sd_build<true, true>), dim3(grid), dim3(128), 0, s, desc, ctx->sp_perm.p, ctx->sp_vals.p, ctx->sp_meta.p,
                       (int)ctx->nrows, nsl, (uint16_t*)nullptr, (double*)nullptr, (const int64_t*)nullptr, ctx->sp_sd_info.p, bytes);
  else
    hipLaunchKernelGGL((k_sp_sd_build<false, true>), dim3(grid), dim3(128), 0, s, desc, (const int32_t*)nullptr, ctx->sp_vals.p,
                       ctx->sp_meta.p, (int)ctx->nrows, nsl, (uint16_t*)nullptr, (double*)nullptr, (const int64_t*)nullptr,
                       ctx->sp_sd_info.p, bytes);
  {
    const auto even = rocprim::make_transform_iterator(ctx->sp_sd_info.p, Even2{}); // (tables start at even entries: 16-B aligned)
    size_t tb = 0;
    ZZZ_HIP(ctx, rocprim::exclusive_scan(nullptr, tb, even, ctx->sp_sd_off.p, (int64_t)0, (size_t)nsl + 1, rocprim::plus<int64_t>(), s));
    ZZZ_HIP(ctx, ctx->scr_tmp.grow_keep(tb, ctx->retired));
    ZZZ_HIP(ctx, rocprim::exclusive_scan(ctx->scr_tmp.p, tb, even, ctx->sp_sd_off.p, (int64_t)0, (size_t)nsl + 1, rocprim::plus<int64_t>(), s));
  }
  int64_t total = 0;
  ZZZ_HIP(ctx, hipMemcpyAsync(&total, ctx->sp_sd_off.p + nsl, sizeof(total), hipMemcpyDeviceToHost, s));
  ZZZ_HIP(ctx, hipStreamSynchronize(s));
  ZZZ_HIP(ctx, ctx->sp_sd_vals.alloc((size_t)total + 2));
  if (ctx->sp_sorted)
    hipLaunchKernelGGL((k_sp_sd_build<true, false>), dim3(grid), dim3(128), 0, s, desc, ctx->sp_perm.p, ctx->sp_vals.p, ctx->sp_meta.p,
                       (int)ctx->nrows, nsl, ctx->sp_vcode8.p, ctx->sp_sd_vals.p, ctx->sp_sd_off.p, ctx->sp_sd_info.p, bytes);
  else
    hipLaunchKernelGGL((k_sp_sd_build<false, false>), dim3(grid), dim3(128), 0, s, desc, (const int32_t*)nullptr, ctx->sp_vals.p,
                       ctx->sp_meta.p, (int)ctx->nrows, nsl, ctx->sp_vcode8.p, ctx->sp_sd_vals.p, ctx->sp_sd_off.p, ctx->sp_sd_info.p,
                       bytes);
  ZZZ_HIP(ctx, hipGetLastError());
  int32_t h[8];
  ZZZ_HIP(ctx, hipMemcpyAsync(h, ctx->sp_dict_info.p, sizeof(h), hipMemcpyDeviceToHost, s));
  ZZZ_HIP(ctx, hipStreamSynchronize(s));
  unsigned long long b = 0;
  memcpy(&b, h + 4, sizeof(b));
  if (ctx->sellp_dict != 3 && (double)b > 0.6 * (double)ctx->sp_bytes)
    return ZZZ_OK;
  ctx->sp_sd_bytes = (int64_t)b;
  ctx->sp_sd_on = true;
  ctx->sp_sd_all = h[6] == 0;
  return ZZZ_OK;
}
ZZZ_PRELOAD_TU(sellp_dict)
} // namespace zzz

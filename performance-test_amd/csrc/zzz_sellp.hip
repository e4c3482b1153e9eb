// The CG operator stream, part 3 of 3: the PRODUCT on the stream the packer (zzz_sellp_pack.hip) and the dictionary builders
// (zzz_sellp_dict.hip) leave; streams of one-chunk slices on coded values have a kernel of their own (zzz_sellp_pipe.hip).
// The stream's format is described at the head of zzz_sellp_pack.hip.
#include <climits>
#include <cstring>
#include <cstdlib>
#include <vector>

#include "zzz_sellp.h"
#include "zzz_cg_device.h"

namespace zzz
{
// ---- the product --------------------------------------------------------------------------------------
template <bool NT, typename T>
__device__ inline T sp_load(const T* p)
{
  return NT ? __builtin_nontemporal_load(p) : *p;
}

// Values and columns of chunk c for this lane.  FULL: all eight slots (every chunk but the last of a slice);
// otherwise only the first w: unused value blocks are not loaded (an odd w loads its last entry with one 8-B load).
// Columns by the chunk's mode (meta[c][0]: bit 31 int32, bit 30 8-bit codes, else 16-bit codes on the slot bases).
// DICT: the values are 16-bit codes into the stream's dictionary (vcode: [chunk][lane][8], one 16-B load; code 0 = +0.0,
// which is also what the slots beyond a last chunk's width hold)
// FOLD (the x window in LDS): a periodic chunk leaves the lane's own part of its columns, 3 (row div 3 - first), in `loff`
// instead of adding it to each of the eight: the caller adds it to the window's address once.
template <bool NT, bool FULL, int DICT = 0, bool FOLD = false>
__device__ inline void read_chunk(int c, int w, int lane, const double* __restrict__ svals, const uint16_t* __restrict__ c16,
                                  const int32_t* __restrict__ c32, const int32_t* __restrict__ meta, dbl2 (&v)[4], int (&cl)[8],
                                  const uint16_t* __restrict__ vcode = nullptr, const double* __restrict__ dict = nullptr,
                                  int* __restrict__ loff = nullptr)
{
  if (FOLD)
    *loff = 0;
  const double* __restrict__ sp = svals + (size_t)c * 512;
  if (DICT == 1 || DICT == 2 || (DICT == 3 && dict != nullptr))
  {
    const uint4v q = sp_load<NT>(reinterpret_cast<const uint4v*>(vcode + (size_t)c * 512) + lane);
    // DICT == 2: `dict` is the workgroup's copy in LDS (a small dictionary: the lanes of a slice mostly hold the same
    // code in a slot -- broadcast reads); DICT == 1: gathers from memory; DICT == 3: the slice's own table in the
    // wavefront's LDS (dict == nullptr: this slice stays doubles, below)
    auto look = [&](unsigned code) -> double { return DICT == 1 ? gather(dict, (int)code) : dict[code]; };
    v[0].x = look(q.x & 0xffffu);
    v[0].y = look(q.x >> 16);
    v[1].x = look(q.y & 0xffffu);
    v[1].y = look(q.y >> 16);
    v[2].x = look(q.z & 0xffffu);
    v[2].y = look(q.z >> 16);
    v[3].x = look(q.w & 0xffffu);
    v[3].y = look(q.w >> 16);
  }
  else
#pragma unroll
  for (int j = 0; j < 4; ++j)
  {
    if (FULL || 2 * j + 1 < w)
      v[j] = sp_load<NT>(reinterpret_cast<const dbl2*>(sp + 128 * j) + lane);
    else if (2 * j < w)
    {
      v[j].x = sp_load<NT>(sp + 128 * j + lane);
      v[j].y = 0.0;
    }
    else
    {
      v[j].x = 0.0;
      v[j].y = 0.0;
    }
  }
  const int32_t* __restrict__ mp = meta + (size_t)c * 8;
  const int m0 = mp[0];
  if (m0 < 0 && (m0 & 0x40000000))
  {
    // periodic chunk (block size 3): column = T[slot][row mod 3] + 3 (row div 3 - first), nothing per lane to load
    const int32_t* __restrict__ tp = reinterpret_cast<const int32_t*>(c16 + (size_t)c * 512);
    // (round 5 probe: with every chunk reading the table of one of eight chunks -- scalar-cache hits -- C4's product takes the
    // same 123-124 us: it does not wait for these loads)
    // the 25 words are wave-uniform: pinned into scalar registers, the selects below stay register selects.  (Left to
    // itself the compiler folds them into a per-lane ADDRESS select, tp + 3 e + k, behind divergent branches -- and
    // that code decoded slot 0 of the third class wrongly when the chunk sits inside the chunk loop.)
    // (All 25 requested first, pinned after: pinning each word as it is loaded made 25 DEPENDENT scalar round trips of
    // them -- `s_load_dword; s_waitcnt lgkmcnt(0)` 25 times per periodic chunk, 2-3 us of a 3.5-us chunk at C4.)
    int t[25];
#pragma unroll
    for (int i = 0; i < 25; ++i)
      t[i] = tp[i];
#pragma unroll
    for (int i = 0; i < 25; ++i)
      asm volatile("" : "+s"(t[i]));
    const int l = lane + t[24];
    const int q = l / 3, k = l - 3 * q, q3 = 3 * q;
    // T[slot][k] without a select per class: T0 + (k >= 1) (T1 - T0) + (k == 2) (T2 - T1), the brackets as masks.  (The nested
    // selects compiled into divergent control flow, ~15 scalar instructions per slot: 163 scalar instructions per chunk at C4,
    // more than the CU's scalar unit issues in the time the chunk's bytes take.)
    int k1 = k >= 1 ? -1 : 0, k2 = k == 2 ? -1 : 0;
    // (the masks made opaque: left visible, the compiler turns `mask & scalar` back into a select, which needs the scalar
    // in a vector register first -- two moves, two selects and two adds per slot; this way it is two `v_and` with a scalar
    // operand and one three-operand add: 48 -> 24 vector instructions per periodic chunk, a quarter of all it issues at C4)
    asm volatile("" : "+v"(k1), "+v"(k2));
    if (FOLD)
      *loff = q3;
#pragma unroll
    for (int e = 0; e < 8; ++e)
    {
      const int a = k1 & (t[3 * e + 1] - t[3 * e]), b = k2 & (t[3 * e + 2] - t[3 * e + 1]);
      cl[e] = FOLD ? (t[3 * e] + a) + b : ((t[3 * e] + q3) + a) + b;
    }
  }
  else if (m0 < 0)
  {
    const int4v* __restrict__ cp = reinterpret_cast<const int4v*>(c32 + (size_t)c * 512) + 2 * lane;
    const int4v q0 = sp_load<NT>(cp), q1 = sp_load<NT>(cp + 1);
    cl[0] = q0.x, cl[1] = q0.y, cl[2] = q0.z, cl[3] = q0.w;
    cl[4] = q1.x, cl[5] = q1.y, cl[6] = q1.z, cl[7] = q1.w;
  }
  else if ((m0 & 0x60000000) == 0x20000000)
  {
    // affine chunk: column = slot base + lane, nothing to load
    cl[0] = (m0 & 0x1fffffff) + lane;
#pragma unroll
    for (int e = 1; e < 8; ++e)
      cl[e] = mp[e] + lane;
  }
  else if (m0 & 0x40000000)
  {
    const uint2v q = (m0 & 0x20000000) ? sp_load<NT>(reinterpret_cast<const uint2v*>(sp + 448) + lane)
                                       : sp_load<NT>(reinterpret_cast<const uint2v*>(c16 + (size_t)c * 512) + lane);
    cl[0] = (m0 & 0x1fffffff) + (int)(q.x & 0xffu);
    cl[1] = mp[1] + (int)((q.x >> 8) & 0xffu);
    cl[2] = mp[2] + (int)((q.x >> 16) & 0xffu);
    cl[3] = mp[3] + (int)(q.x >> 24);
    cl[4] = mp[4] + (int)(q.y & 0xffu);
    cl[5] = mp[5] + (int)((q.y >> 8) & 0xffu);
    cl[6] = mp[6] + (int)((q.y >> 16) & 0xffu);
    cl[7] = mp[7] + (int)(q.y >> 24);
  }
  else
  {
    const uint4v q = sp_load<NT>(reinterpret_cast<const uint4v*>(c16 + (size_t)c * 512) + lane);
    cl[0] = m0 + (int)(q.x & 0xffffu);
    cl[1] = mp[1] + (int)(q.x >> 16);
    cl[2] = mp[2] + (int)(q.y & 0xffffu);
    cl[3] = mp[3] + (int)(q.y >> 16);
    cl[4] = mp[4] + (int)(q.z & 0xffffu);
    cl[5] = mp[5] + (int)(q.z >> 16);
    cl[6] = mp[6] + (int)(q.w & 0xffffu);
    cl[7] = mp[7] + (int)(q.w >> 16);
  }
}

// sum += the chunk's products in ascending column order, mul and add rounded separately (the scalar CPU loop's bits)
template <bool NT, bool FULL, bool LDS = false, int DICT = 0>
__device__ inline void chunk_product(int c, int w, int lane, const double* __restrict__ svals, const uint16_t* __restrict__ c16,
                                     const int32_t* __restrict__ c32, const int32_t* __restrict__ meta,
                                     const double* __restrict__ x, double& sum, const uint16_t* __restrict__ vcode = nullptr,
                                     const double* __restrict__ dict = nullptr)
{
  dbl2 v[4];
  int cl[8];
  int loff = 0;
  read_chunk<NT, FULL, DICT, LDS>(c, w, lane, svals, c16, c32, meta, v, cl, vcode, dict, &loff);
  const double* __restrict__ xl = LDS ? x + loff : x;
  double xv[8];
#pragma unroll
  for (int e = 0; e < 8; ++e)
    xv[e] = (FULL || e < w) ? (LDS ? xl[cl[e]] : gather(x, cl[e])) : 0.0; // LDS: x is the group's window, cl its index
#pragma unroll
  for (int e = 0; e < 8; ++e)
    if (FULL || e < w)
      sum += ((e & 1) ? v[e >> 1].y : v[e >> 1].x) * xv[e];
}

template <bool DOT, bool NT, bool PERM, bool CHEB = false, bool WIN = false, int DICT = 0>
__global__ __launch_bounds__(SP_BLOCK, 8) void spmv_sellp_kernel(const int2* __restrict__ desc,
                                                              const double* __restrict__ svals,
                                                              const uint16_t* __restrict__ c16,
                                                              const int32_t* __restrict__ c32,
                                                              const int32_t* __restrict__ meta,
                                                              const int32_t* __restrict__ perm,
                                                              const double* __restrict__ x, double* __restrict__ y,
                                                              int nrows, int64_t nslices, double* __restrict__ partials,
                                                              const int* __restrict__ stop_flag,
                                                              const int32_t* __restrict__ group_list, int64_t nlist,
                                                              const double* __restrict__ rvec, int pstride, int nn_is_rr,
                                                              TailArgs tail, ChebEpi epi, const int2* __restrict__ win_info,
                                                              const int2* __restrict__ win_seg, const uint16_t* __restrict__ vcode,
                                                              const double* __restrict__ dict_g, int dict_n,
                                                              const int32_t* __restrict__ sd_info,
                                                              const int64_t* __restrict__ sd_off)
{
  // dynamic LDS: DICT == 2: the value dictionary (dict_n doubles, rounded up to 2); WIN: the group's x window behind it
  // (launch: sp_win_max doubles)
  extern __shared__ __attribute__((aligned(16))) double sp_lds[];
  double* const xwin = sp_lds + (DICT == 2 ? ((dict_n + 1) & ~1) : 0);
  // DICT == 3: dict_g holds the slices' tables (SD_MAX entries each), sp_lds one table per wavefront
  double* const sdl = sp_lds + (threadIdx.x >> 6) * SD_MAX;
  const double* dict = DICT == 2 ? sp_lds : dict_g;
  // group_list != nullptr: only the listed groups of 4 slices (interior or boundary subset of a partitioned
  // matrix); rvec != nullptr: also the partials of <r,x> and of the test norm (single-reduction CG), as in
  // spmv_tile_kernel
  __shared__ double red[SP_BLOCK / 64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t ngroups = group_list ? nlist : (nslices + 3) / 4; // a workgroup takes 4 consecutive slices
  double dot = 0.0, dot_rx = 0.0, dot_nn = 0.0;
  // A slice is a chain of dependent round trips: (group list ->) descriptor -> values / codes / bases -> gathers.  At the
  // 8-GPU per-rank size a wavefront has two or three slices in all, so the kernel's length is that chain times three;
  // the slice number and descriptor of the NEXT slice are therefore requested (scalar loads) before the current slice's
  // chunks, and those of the first slice before the stop word is looked at.
  auto next_slice = [&](int i, int& s_out, int2& ds_out) -> int { // 1 valid, 0 no slice for this wavefront, -1 done
    const int64_t gi = sp_xcd_item(ngroups, blockIdx.x, gridDim.x, i);
    if (gi < 0)
      return -1;
    const int64_t g = group_list ? group_list[gi] : gi;
    const int s = __builtin_amdgcn_readfirstlane((int)(4 * g + wv));
    s_out = s;
    if (s >= nslices)
      return 0;
    ds_out = desc[s];
    return 1;
  };
  int s_n = 0;
  int2 ds_n = make_int2(0, 0);
  int st_n = next_slice(0, s_n, ds_n);
  if (stop_flag && *stop_flag) // CG already converged: the host is a few iterations ahead
    return;
  if (DICT == 2)
  {
    // the dictionary into LDS: every thread's (at most eight) entries requested together, one round trip (five dependent ones
    // at C2's 1 204 values cost the 8-GPU per-rank product ~2 of its 12 us)
    double t[SP_DICT_LDS_ENTRIES / SP_BLOCK];
#pragma unroll
    for (int i = 0; i < SP_DICT_LDS_ENTRIES / SP_BLOCK; ++i)
    {
      const int k = (int)threadIdx.x + i * SP_BLOCK;
      t[i] = k < dict_n ? dict_g[k] : 0.0;
    }
#pragma unroll
    for (int i = 0; i < SP_DICT_LDS_ENTRIES / SP_BLOCK; ++i)
    {
      const int k = (int)threadIdx.x + i * SP_BLOCK;
      if (k < dict_n)
        sp_lds[k] = t[i];
    }
    __syncthreads();
  }
  for (int i = 0; st_n >= 0; ++i)
  {
    const int st = st_n, s = s_n;
    const int2 ds = ds_n;
    st_n = next_slice(i + 1, s_n, ds_n);
    // WIN: the four wavefronts work on one group; where the group has a window its segments of x are loaded into LDS
    // first (every wavefront takes part, also one without a slice of its own at the end of the matrix)
    int nwin = 0;
    if (WIN)
    {
      const int2 wi = win_info[s >> 2];
      nwin = __builtin_amdgcn_readfirstlane(wi.x);
      if (nwin > 0)
      {
        const int2* __restrict__ sg = win_seg + (int64_t)(s >> 2) * SP_WIN_NSEG;
        __syncthreads(); // the previous group's window is done with
        // Every entry of the window is REQUESTED before any is stored (a loop of load -> store per segment was five to
        // seven dependent round trips per group, a quarter of the group's time at C4): the segments are cut into blocks of
        // 256 entries, block i is thread-uniformly one segment's, thread t takes entry t of each; at most WIN_BLOCKS blocks
        // in registers at a time.
        constexpr int WIN_BLOCKS = 12;
        int q = 0, seg_first = 0, off = 0; // the segment of the current block, the first block of it, where it starts in the window
        int2 sq = sg[0];
        int c0s = __builtin_amdgcn_readfirstlane(sq.x), len = __builtin_amdgcn_readfirstlane(sq.y);
        for (int b0 = 0; q < nwin; b0 += WIN_BLOCKS)
        {
          double t[WIN_BLOCKS];
          int at[WIN_BLOCKS]; // where the entry goes (-1: none)
#pragma unroll
          for (int i = 0; i < WIN_BLOCKS; ++i)
          {
            at[i] = -1;
            t[i] = 0.0;
            while (q < nwin && (b0 + i - seg_first) * SP_BLOCK >= len) // (uniform: on to the segment this block belongs to)
            {
              off += len;
              seg_first = b0 + i;
              ++q;
              if (q < nwin)
              {
                sq = sg[q];
                c0s = __builtin_amdgcn_readfirstlane(sq.x);
                len = __builtin_amdgcn_readfirstlane(sq.y);
              }
            }
            if (q < nwin)
            {
              const int k = (b0 + i - seg_first) * SP_BLOCK + (int)threadIdx.x;
              if (k < len)
              {
                t[i] = x[c0s + k];
                at[i] = off + k;
              }
            }
          }
#pragma unroll
          for (int i = 0; i < WIN_BLOCKS; ++i)
            if (at[i] >= 0)
              xwin[at[i]] = t[i];
        }
        __syncthreads();
      }
    }
    if (st == 0)
      continue;
    if (DICT == 3)
    {
      // the slice's table into the wavefront's LDS (entries 0 .. n - 1; n == 0: the slice's values are doubles)
      const int n_sd = __builtin_amdgcn_readfirstlane(sd_info[s]);
      const int64_t sd_tab = sd_off[s];
      if (n_sd > 0)
      {
        __builtin_amdgcn_wave_barrier(); // (the previous slice's lookups are done)
        // the table requested 512 entries at a time before any of them is stored, 16 B per lane and request (the tables start
        // at even entries) -- not one dependent round trip per 64 entries (five for a median table of 296)
        const dbl2* __restrict__ tsrc = reinterpret_cast<const dbl2*>(dict_g + sd_tab);
        const int n2 = (n_sd + 1) >> 1;
        for (int b0 = 0; b0 * 64 < n2; b0 += 4) // (four requests = 512 entries at a time: registers)
        {
          dbl2 t[4];
#pragma unroll
          for (int i = 0; i < 4; ++i)
            t[i] = lane + 64 * (b0 + i) < n2 ? tsrc[lane + 64 * (b0 + i)] : dbl2{0.0, 0.0};
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (lane + 64 * (b0 + i) < n2)
              reinterpret_cast<dbl2*>(sdl)[lane + 64 * (b0 + i)] = t[i];
        }
        __builtin_amdgcn_wave_barrier();
        dict = sdl;
      }
      else
        dict = nullptr;
    }
    const int c0 = ds.x, nch = ds.y & 0xffffff, wl = ds.y >> 24;
    int r = PERM ? perm[(int64_t)s * 64 + lane] : s * 64 + lane;
    if (!PERM && r >= nrows)
      r = -1;
    const double xr = ((DOT || CHEB) && r >= 0) ? x[r] : 0.0;
    double sum = 0.0;
    if (WIN && nwin > 0)
    {
      for (int j = 0; j + 1 < nch; ++j)
        chunk_product<NT, true, true, DICT>(c0 + j, 8, lane, svals, c16, c32, meta, xwin, sum, vcode, dict);
      if (nch)
      {
        if (wl == 8)
          chunk_product<NT, true, true, DICT>(c0 + nch - 1, 8, lane, svals, c16, c32, meta, xwin, sum, vcode, dict);
        else
          chunk_product<NT, false, true, DICT>(c0 + nch - 1, wl, lane, svals, c16, c32, meta, xwin, sum, vcode, dict);
      }
    }
    else
    {
      for (int j = 0; j + 1 < nch; ++j)
        chunk_product<NT, true, false, DICT>(c0 + j, 8, lane, svals, c16, c32, meta, x, sum, vcode, dict);
      if (nch)
      {
        if (wl == 8)
          chunk_product<NT, true, false, DICT>(c0 + nch - 1, 8, lane, svals, c16, c32, meta, x, sum, vcode, dict);
        else
          chunk_product<NT, false, false, DICT>(c0 + nch - 1, wl, lane, svals, c16, c32, meta, x, sum, vcode, dict);
      }
    }
    if (CHEB)
    {
      // a term of the Chebyshev-Jacobi polynomial (ChebEpi): x is d, y the next d; DOT marks the last term
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
          dot_nn += nn_is_rr ? ri * ri : zi * zi;
        }
        else
        {
          epi.g[r] = gi;
          y[r] = dn;
        }
      }
    }
    else if (r >= 0)
    {
      y[r] = sum;
      if (DOT)
      {
        dot += sum * xr;
        if (rvec)
        {
          const double rr_ = rvec[r];
          dot_rx += rr_ * xr;
          dot_nn += nn_is_rr ? rr_ * rr_ : xr * xr;
        }
      }
    }
  }
  if (DOT)
  {
    const double sres = CHEB ? 0.0 : block_reduce_sum(dot, red);
    double s1 = 0.0, s2 = 0.0;
    if (rvec || CHEB)
    {
      s1 = block_reduce_sum(dot_rx, red);
      s2 = block_reduce_sum(dot_nn, red);
    }
    if (!CHEB && tail.parts)
    {
      // multi-GPU: the all-reduce of these sums happens in the tail of this launch (zzz_tail.h); output order
      // (<r,x>, norm, <x,y>) for the single-reduction form, <x,y> alone otherwise
      if (rvec)
        tail_arrive(tail, s1, s2, sres);
      else
        tail_arrive(tail, sres, 0.0, 0.0);
      return;
    }
    if (threadIdx.x == 0)
    {
      if (!CHEB)
        partials[blockIdx.x] = sres;
      if (rvec || CHEB)
      {
        partials[pstride + blockIdx.x] = s1;
        partials[2 * pstride + blockIdx.x] = s2;
      }
    }
  }
}

#ifdef ZZZ_EXPERIMENTS // measurement-only kernels: the tools build (libzzz_hip_exp.so), never the product library
// ---- TIMING PROBE (ZZZ_EXP_WIN=<doubles>): what would an x window in LDS buy? ------------------------------------
// The cost structure of a windowed product without its packer: per group of four slices the workgroup loads <doubles>
// consecutive entries of x into LDS (coalesced 16-B loads), and every gather of the chunk loop reads LDS at a
// pseudo-random index instead of global memory.  The RESULT IS WRONG by construction; only zzz_spmv_time may run it.
template <bool NT>
__global__ __launch_bounds__(SP_BLOCK) void spmv_sellp_win_probe_kernel(const int2* __restrict__ desc,
                                                                    const double* __restrict__ svals,
                                                                    const uint16_t* __restrict__ c16,
                                                                    const int32_t* __restrict__ c32,
                                                                    const int32_t* __restrict__ meta,
                                                                    const double* __restrict__ x, double* __restrict__ y,
                                                                    int nrows, int64_t nslices, double* __restrict__ partials,
                                                                    int wlen, int lds_slots)
{
  extern __shared__ __attribute__((aligned(16))) double win[];
  __shared__ double red[SP_BLOCK / 64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t ngroups = (nslices + 3) / 4;
  double dot = 0.0;
  for (int i = 0;; ++i)
  {
    const int64_t g = sp_xcd_item(ngroups, blockIdx.x, gridDim.x, i);
    if (g < 0)
      break;
    __syncthreads();
    {
      int64_t base = g * 256 - wlen / 2;
      if (base < 0)
        base = 0;
      if (base + wlen > nrows)
        base = nrows > wlen ? nrows - wlen : 0;
      base &= ~(int64_t)1;
      const dbl2* __restrict__ src = reinterpret_cast<const dbl2*>(x + base);
      dbl2* dst = reinterpret_cast<dbl2*>(win);
      for (int k = threadIdx.x; k < wlen / 2; k += SP_BLOCK)
        dst[k] = src[k];
    }
    __syncthreads();
    const int s = __builtin_amdgcn_readfirstlane((int)(4 * g + wv));
    if (s >= nslices)
      continue;
    const int2 ds = desc[s];
    const int c0 = ds.x, nch = ds.y & 0xffffff, wl = ds.y >> 24;
    const int r = s * 64 + lane < nrows ? s * 64 + lane : -1;
    double sum = 0.0;
    for (int j = 0; j < nch; ++j)
    {
      dbl2 v[4];
      int cl[8];
      if (j + 1 < nch || wl == 8)
        read_chunk<NT, true>(c0 + j, 8, lane, svals, c16, c32, meta, v, cl);
      else
        read_chunk<NT, false>(c0 + j, wl, lane, svals, c16, c32, meta, v, cl);
#pragma unroll
      for (int e = 0; e < 8; ++e)
      {
        // the first lds_slots slots of every chunk gather from the window, the others from memory (a PARTIAL window)
        const unsigned idx = (unsigned)cl[e] % (unsigned)wlen;
        const double xv = e < lds_slots ? win[idx] : gather(x, min(cl[e], nrows - 1));
        sum += ((e & 1) ? v[e >> 1].y : v[e >> 1].x) * xv;
      }
    }
    if (r >= 0)
    {
      y[r] = sum;
      dot += sum;
    }
  }
  const double sres = block_reduce_sum(dot, red);
  if (threadIdx.x == 0 && partials)
    partials[blockIdx.x] = sres;
}

// ---- product fused with the direction update -------------------------------------------------------------
// One CG iteration as TWO kernels instead of three: the head of iteration `it` (convergence test, k_update_p of
// zzz_cg.hip) and the product w = A p, with p = z + b p_old formed on the fly where the product gathers it
// (the same two roundings as k_update_p, so p, w and every scalar keep their bits), written once per owned row
// into the OTHER p buffer (the gathers of other workgroups still read p_old), together with the pending
// solution update x += alpha_{it-1} p_old (src/cg.h:68,82).  Ghost entries of p follow the same recurrence from the
// ghost values of z, so the halo exchange of an iteration moves z instead of p.  An A/B variant (ZZZ_CG_FUSED=2):
// measured slower than the three-kernel form at every size tried (cg_solve has the numbers), kept because it
// pins the iteration's arithmetic from a second side -- tests demand identical bits from both forms.
template <bool NT, bool PERM>
__global__ __launch_bounds__(SP_BLOCK, 4) void spmv_sellp_dir_kernel(
    const int2* __restrict__ desc, const double* __restrict__ svals, const uint16_t* __restrict__ c16,
    const int32_t* __restrict__ c32, const int32_t* __restrict__ meta, const int32_t* __restrict__ perm,
    const double* __restrict__ z, const double* __restrict__ p_old, double* __restrict__ p_new, double* __restrict__ xsol,
    double* __restrict__ y, int nrows, int ncols, int64_t nslices, double* __restrict__ partials,
    const int32_t* __restrict__ group_list, int64_t nlist, int ghost_update, CgState* __restrict__ st,
    double* __restrict__ beta_hist, double* __restrict__ dp_hist, const double* __restrict__ alpha_hist, int it, CgParams P,
    const double* __restrict__ pa, const double* __restrict__ pb, int np)
{
  __shared__ double red[SP_BLOCK / 64];
  DirScalars S;
  if (!cg_direction_scalars(st, beta_hist, dp_hist, it, P, pa, pb, np, red, S))
    return;
  const double bcoef = S.rz / S.bprev;
  const double alpha = it > 0 ? alpha_hist[it - 1] : 0.0;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t ngroups = group_list ? nlist : (nslices + 3) / 4;
  double dot = 0.0;
  for (int i = 0;; ++i)
  {
    const int64_t gi = sp_xcd_item(ngroups, blockIdx.x, gridDim.x, i);
    if (gi < 0)
      break;
    const int64_t g = group_list ? group_list[gi] : gi;
    const int s = __builtin_amdgcn_readfirstlane((int)(4 * g + wv));
    if (s >= nslices)
      continue;
    int r = PERM ? perm[(int64_t)s * 64 + lane] : s * 64 + lane;
    if (!PERM && r >= nrows)
      r = -1;
    const double po = r >= 0 ? p_old[r] : 0.0;
    if (r >= 0 && it > 0)
      xsol[r] = alpha * po + xsol[r]; // the previous iteration's solution update, also by the launch that stops
    if (S.conv)
      continue;
    const double pn = r >= 0 ? bcoef * po + z[r] : 0.0;
    const int2 ds = desc[s];
    const int c0 = ds.x, nch = ds.y & 0xffffff, wl = ds.y >> 24;
    double sum = 0.0;
    for (int j = 0; j < nch; ++j)
    {
      const int w = j + 1 < nch ? 8 : wl;
      dbl2 v[4];
      int cl[8];
      read_chunk<NT, false>(c0 + j, w, lane, svals, c16, c32, meta, v, cl);
      double zv[8], pv[8];
#pragma unroll
      for (int e = 0; e < 8; ++e)
      {
        zv[e] = e < w ? gather(z, cl[e]) : 0.0;
        pv[e] = e < w ? gather(p_old, cl[e]) : 0.0;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (e < w)
          sum += ((e & 1) ? v[e >> 1].y : v[e >> 1].x) * (bcoef * pv[e] + zv[e]);
    }
    if (r >= 0)
    {
      p_new[r] = pn;
      y[r] = sum;
      dot += sum * pn;
    }
  }
  if (S.conv)
    return;
  if (ghost_update) // ghost entries of the new direction (the launch that runs behind the halo of z)
    for (int64_t k = nrows + blockIdx.x * (int64_t)SP_BLOCK + threadIdx.x; k < ncols; k += (int64_t)gridDim.x * SP_BLOCK)
      p_new[k] = bcoef * p_old[k] + z[k];
  const double sres = block_reduce_sum(dot, red);
  if (threadIdx.x == 0)
    partials[blockIdx.x] = sres;
}

#endif // ZZZ_EXPERIMENTS

// ---- host side ------------------------------------------------------------------------------------------
// A failed build leaves its form off (and the stream, where there is one, as it is).
bool sellp_special_build(zzz_ctx* ctx)
{
  if (sellp_blk_build(ctx) != ZZZ_OK)
  {
    if (getenv("ZZZ_DEBUG_SYNC"))
      fprintf(stderr, "[zzz dbg] sellp_blk_build: %s\n", ctx->err.c_str());
    ctx->bk_on = false;
    (void)hipGetLastError();
  }
  if (sellp_win_build(ctx) != ZZZ_OK)
  {
    if (getenv("ZZZ_DEBUG_SYNC"))
      fprintf(stderr, "[zzz dbg] sellp_win_build: %s\n", ctx->err.c_str());
    ctx->bw_on = false;
    (void)hipGetLastError();
  }
  return sellp_blk_serves(ctx) || sellp_win_serves(ctx);
}

bool sellp_active(zzz_ctx* ctx)
{
  if (ctx->sp_pending)
    (void)sellp_resolve(ctx);
  if (!ctx->have_sell || !ctx->sell_current)
    return false;
  if (!(ctx->spmv_auto || (ctx->spmv_variant & 8) != 0))
    return false;
  if (!ctx->sp_dict_done)
  {
    // a FAILED build (a hipMalloc that did not fit) leaves the stream on doubles -- valid -- but must not leave its partial
    // allocations or HIP's sticky last error behind (the next hipGetLastError after a product launch would report it).
    // A build that was DECLINED (too many distinct values, the 60 % rule) keeps its buffers for the next assembly: this runs
    // at the first product of a solve, and hipFree waits for the whole device -- with two ranks on ONE GPU (the tests' and
    // the driver's --comm local) the other rank may already be polling its all-reduce mailbox for this one: a release here
    // was a dead wait until the poll's time-out (round 5, found by tools/soak_driver.sh: elasticity P2, two ranks).
    // The special forms first (block rows for block size 3, zzz_sellp_blk.hip; block windows for long scalar rows,
    // zzz_sellp_win.hip): where one of them serves the product, the stream's value dictionaries are not built -- the launches
    // that still take the generic kernel (a folded all-reduce: tools build only) read its values as doubles.  7.6 ms per
    // assembly at 6.2 M rows of P3, 3 ms at C4.  A failed build leaves the stream as it is.
    ctx->sp_dict_done = true;
    ctx->sp_dict_on = ctx->sp_sd_on = ctx->sp_sd_all = false;
    ctx->sp_dict_n = 0;
    if (!ctx->sp_special_tried)
    {
      (void)sellp_special_build(ctx);
      ctx->sp_special_tried = true;
    }
    if (!sellp_blk_serves(ctx) && !sellp_win_serves(ctx))
    {
      if (sp_dict_build(ctx) != ZZZ_OK)
      {
        ctx->sp_dict_on = false;
        (void)hipGetLastError();
        ctx->sp_vcode.release();
        ctx->sp_dict_table.release();
        ctx->sp_dict_slot.release();
      }
      if (sp_sd_build(ctx) != ZZZ_OK)
      {
        ctx->sp_sd_on = ctx->sp_sd_all = false;
        (void)hipGetLastError();
        ctx->sp_vcode8.release();
        ctx->sp_sd_vals.release();
        ctx->sp_sd_info.release();
        ctx->sp_sd_off.release();
      }
    }
    ctx->sp_pairs_ok = false;
    if (ctx->sp_one_chunk && ctx->sp_dict_on && ctx->bs == 1 && !ctx->sp_sorted)
      ctx->sp_pairs_ok = sellp_pairs_build(ctx) == ZZZ_OK;
  }
  return true;
}

// bytes one product reads from the stream (values or value codes + dictionary, column codes, bases; int32 chunks are not
// counted separately)
int64_t sellp_stream_bytes(const zzz_ctx* ctx)
{
  if (sellp_blk_serves(ctx))
    return ctx->bk_bytes; // (descriptors and table included)
  if (sellp_win_serves(ctx))
    return ctx->bw_bytes;
  return (ctx->sp_sd_on ? ctx->sp_sd_bytes : (ctx->sp_dict_on ? ctx->sp_dict_bytes : ctx->sp_bytes)) + ctx->nslices * 8; // (x windows: sp_win_bytes, reported apart)
}

// Load policy of the product: non-temporal stream loads when one CG iteration (the stream and six vectors) cannot stay in
// the 256 MiB Infinity Cache anyway -- then the stream should not push the vectors' lines out; plain loads when it is
// resident.  (Until the value dictionary the stream alone decided, at 300 MB; a coded stream of 167 MB beside 480 MB of
// vectors reads 4 % faster non-temporally: 0.091 -> 0.087 ms at C2.)
static bool sp_stream_nt(const zzz_ctx* ctx)
{
  return (double)sellp_stream_bytes(ctx) + 48.0 * (double)(ctx->n_owned + ctx->n_ghost) * ctx->bs > 200.0e6;
}

// plain: the launch carries no Chebyshev epilogue and no folded all-reduce, so the specialised kernels (one-chunk slices,
// block rows) may serve it; otherwise the generic kernel runs and keeps its eight workgroups per CU
static int sp_grid(const zzz_ctx* ctx, int64_t ngroups, bool sr, bool plain)
{
  // persistent workgroups: as many as are resident at once (a second round of a grid that is not would run on part of the chip)
  const int pw = plain ? sellp_pipe_wgs(ctx, sr) : 0;
  int64_t gs = 256 * (pw ? pw : 8); // (1024 or 1536 workgroups at the per-rank size: no faster)
#ifdef ZZZ_EXPERIMENTS
  if (const char* e = getenv("ZZZ_SP_WGS_PER_CU")) // how the product's time depends on the wavefronts in flight
    gs = 256 * std::max(1, std::min(8, atoi(e)));
#endif
  const int64_t need = (ngroups + 7) / 8 * 8;
  if (gs > need)
    gs = need;
  if (gs < 8)
    gs = 8;
  return (int)gs;
}

template <bool DOT>
static void launch_one(zzz_ctx* ctx, int grid, const double* x, double* y, double* partials, const int* stop,
                       const int32_t* group_list, int64_t nlist, const double* rvec, int nn_is_rr,
                       const TailArgs& tail = TailArgs(), const ChebEpi* epi = nullptr, int special = 0)
{
  // load policy by stream size, as for the tile kernel: a stream that stays in the 256 MiB Infinity Cache from
  // one CG iteration to the next is read with plain loads, a larger one with non-temporal loads
  bool nt = sp_stream_nt(ctx);
  if (!ctx->spmv_auto)
    nt = (ctx->spmv_variant & 1) != 0;
  // special: the caller sized the grid and chose the list for the block-row kernel (1: block size 3) or the block-window kernel
  // (2: long scalar rows)
  if (special == 1 && launch_sellp_blk(ctx, DOT, nt, grid, x, y, partials, stop, group_list, nlist, rvec, nn_is_rr, epi))
    return;
  if (special == 2 && launch_sellp_win(ctx, DOT, nt, grid, x, y, partials, stop, group_list, nlist, rvec, nn_is_rr, epi))
    return;
  if (!epi && !tail.parts && launch_sellp_pipe(ctx, DOT, nt, grid, x, y, partials, stop, group_list, nlist, rvec, nn_is_rr))
    return;
  const int2* off = reinterpret_cast<const int2*>(ctx->sp_desc.p);
  const int2* winfo = reinterpret_cast<const int2*>(ctx->sp_win_info.p);
  const int2* wseg = reinterpret_cast<const int2*>(ctx->sp_win_seg.p);
#define ZZZ_SP_GO6(NT, PERM, WIN, LDSB, DICT)                                                                          \
  do                                                                                                                   \
  {                                                                                                                    \
    const uint16_t* VC_ = (DICT) == 3 ? ctx->sp_vcode8.p : ctx->sp_vcode.p;         \
    const double* DG_ = (DICT) == 3 ? ctx->sp_sd_vals.p : ctx->sp_dict.p;                                               \
    if (epi)                                                                                                           \
      hipLaunchKernelGGL((spmv_sellp_kernel<DOT, NT, PERM, true, WIN, DICT>), dim3(grid), dim3(SP_BLOCK), LDSB,         \
                         ctx->stream, off, ctx->sp_vals.p, ctx->sp_codes16.p, ctx->sp_codes32.p, ctx->sp_meta.p,           \
                         ctx->sp_perm.p, x, y, (int)ctx->nrows, ctx->nslices, partials, stop, group_list, nlist, rvec,     \
                         SPMV_PSTRIDE, nn_is_rr, TailArgs(), *epi, winfo, wseg, VC_, DG_, ctx->sp_dict_n,                  \
                         ctx->sp_sd_info.p, ctx->sp_sd_off.p);                                                         \
    else                                                                                                               \
      hipLaunchKernelGGL((spmv_sellp_kernel<DOT, NT, PERM, false, WIN, DICT>), dim3(grid), dim3(SP_BLOCK), LDSB,        \
                         ctx->stream, off, ctx->sp_vals.p, ctx->sp_codes16.p, ctx->sp_codes32.p, ctx->sp_meta.p,           \
                         ctx->sp_perm.p, x, y, (int)ctx->nrows, ctx->nslices, partials, stop, group_list, nlist, rvec,     \
                         SPMV_PSTRIDE, nn_is_rr, tail, ChebEpi(), winfo, wseg, VC_, DG_, ctx->sp_dict_n,                   \
                         ctx->sp_sd_info.p, ctx->sp_sd_off.p);                                                         \
  } while (0)
#define ZZZ_SP_GO5(NT, PERM, WIN, LDSB)                                                                                \
  do                                                                                                                   \
  {                                                                                                                    \
    if (ctx->sp_sd_on && !(WIN))                                                                                       \
      ZZZ_SP_GO6(NT, PERM, false, (size_t)4 * SD_MAX * sizeof(double), 3);                                             \
    else if (ctx->sp_dict_on && ctx->sp_dict_n <= SP_DICT_LDS_ENTRIES)                                                     \
      ZZZ_SP_GO6(NT, PERM, WIN, (LDSB) + (size_t)((ctx->sp_dict_n + 1) & ~1) * sizeof(double), 2);                     \
    else if (ctx->sp_dict_on)                                                                                          \
      ZZZ_SP_GO6(NT, PERM, WIN, LDSB, 1);                                                                              \
    else                                                                                                               \
      ZZZ_SP_GO6(NT, PERM, WIN, LDSB, 0);                                                                              \
  } while (0)
#define ZZZ_SP_GO(NT, PERM) ZZZ_SP_GO5(NT, PERM, false, 0)
  if (ctx->sp_win_max > 0 && !ctx->sp_sorted)
  {
    // windowed groups: their codes index the LDS window the kernel loads per group
    const size_t ldsb = (size_t)ctx->sp_win_max * sizeof(double);
    if (nt)
      ZZZ_SP_GO5(true, false, true, ldsb);
    else
      ZZZ_SP_GO5(false, false, true, ldsb);
  }
  else if (ctx->sp_sorted)
  {
    if (nt)
      ZZZ_SP_GO(true, true);
    else
      ZZZ_SP_GO(false, true);
  }
  else
  {
    if (nt)
      ZZZ_SP_GO(true, false);
    else
      ZZZ_SP_GO(false, false);
  }
#undef ZZZ_SP_GO
#undef ZZZ_SP_GO5
#undef ZZZ_SP_GO6
}

int launch_sellp(zzz_ctx* ctx, const double* x, double* y, double* partials, int* npartials, const double* rvec, int nn_is_rr,
                 const ChebEpi* epi)
{
  const int* stop = partials ? reinterpret_cast<const int*>(ctx->state.p) : nullptr; // CgState::converged
  const bool plain = !epi && !(partials && ctx->tail_armed);
  const bool no_tail = !(partials && ctx->tail_armed); // (the special forms carry the Chebyshev epilogue, not the folded all-reduce)
  const int blk = no_tail ? (sellp_blk_serves(ctx) ? 1 : (sellp_win_serves(ctx) ? 2 : 0)) : 0;
  if (!blk)
    if (int rc = sellp_need_generic(ctx))
      return rc;
  const int gs = blk == 1 ? sellp_blk_grid(ctx, ctx->bk_slices) : blk == 2 ? sellp_win_grid(ctx, ctx->bw_nblk)
                          : sp_grid(ctx, (ctx->nslices + 3) / 4, (partials && rvec), plain);
#ifdef ZZZ_EXPERIMENTS
  const char* e = ctx->timing_only ? getenv("ZZZ_EXP_WIN") : nullptr; // timing probe, wrong results by construction (see
  if (e)                                                               // the kernel): inside zzz_spmv_time only
  {
    const int wlen = atoi(e) & ~1;
    if (wlen >= 256 && wlen <= 8192 && !ctx->sp_sorted && !epi && wlen < ctx->nrows && ctx->sp_win_max == 0 && !blk)
    {
      const int lds_slots = getenv("ZZZ_EXP_WIN_SLOTS") ? atoi(getenv("ZZZ_EXP_WIN_SLOTS")) : 8;
      const int per_cu = std::max(1, std::min(8, (int)(160 * 1024 / ((size_t)wlen * 8 + 512))));
      const int grid = std::min(gs, 256 * per_cu);
      const bool nt = sp_stream_nt(ctx);
      const int2* off = reinterpret_cast<const int2*>(ctx->sp_desc.p);
      if (nt)
        hipLaunchKernelGGL(spmv_sellp_win_probe_kernel<true>, dim3(grid), dim3(SP_BLOCK), (size_t)wlen * 8, ctx->stream, off,
                           ctx->sp_vals.p, ctx->sp_codes16.p, ctx->sp_codes32.p, ctx->sp_meta.p, x, y, (int)ctx->nrows,
                           ctx->nslices, partials, wlen, lds_slots);
      else
        hipLaunchKernelGGL(spmv_sellp_win_probe_kernel<false>, dim3(grid), dim3(SP_BLOCK), (size_t)wlen * 8, ctx->stream, off,
                           ctx->sp_vals.p, ctx->sp_codes16.p, ctx->sp_codes32.p, ctx->sp_meta.p, x, y, (int)ctx->nrows,
                           ctx->nslices, partials, wlen, lds_slots);
      if (npartials)
        *npartials = grid;
      ZZZ_HIP(ctx, hipGetLastError());
      return ZZZ_OK;
    }
  }
#endif
  if (partials)
  {
    TailArgs T;
    if (ctx->tail_armed)
    {
      T = ctx->tail;
      T.expected = gs;
      T.base = 0;
      ctx->tail_armed = false;
      ctx->tail_used = true;
    }
    launch_one<true>(ctx, gs, x, y, partials, stop, nullptr, 0, rvec, nn_is_rr, T, epi, blk);
    if (npartials)
      *npartials = gs;
  }
  else
    launch_one<false>(ctx, gs, x, y, nullptr, stop, nullptr, 0, nullptr, 0, TailArgs(), epi, blk);
  ZZZ_HIP(ctx, hipGetLastError());
  return ZZZ_OK;
}

// Partitioned matrix: forward halo of x overlapped with the groups that reference no ghost column
// (scheme and the 7-of-8 workgroup slots: launch_spmv_overlapped in zzz_spmv.hip)
int launch_sellp_overlapped(zzz_ctx* ctx, double* x, double* y, double* partials, int* npartials, const double* rvec,
                            int nn_is_rr, const ChebEpi* epi)
{
  const int* stop = partials ? reinterpret_cast<const int*>(ctx->state.p) : nullptr;
  const bool plain = !epi && !(partials && ctx->tail_armed);
  const bool no_tail = !(partials && ctx->tail_armed);
  const int blk = !no_tail ? 0 : (sellp_blk_serves(ctx) && ctx->bk_have_split) ? 1 : (sellp_win_serves(ctx) && ctx->bw_have_split) ? 2 : 0;
  if (!blk)
    if (int rc = sellp_need_generic(ctx))
      return rc;
  const int64_t gi = blk == 1 ? ctx->bk_n_interior : blk == 2 ? ctx->bw_n_interior : ctx->n_groups_interior;
  const int64_t gb = blk == 1 ? ctx->bk_n_boundary : blk == 2 ? ctx->bw_n_boundary : ctx->n_groups_boundary;
  const int32_t* list_in = blk == 1 ? ctx->bk_list_interior.p : blk == 2 ? ctx->bw_list_interior.p : ctx->groups_interior.p;
  const int32_t* list_bd = blk == 1 ? ctx->bk_list_boundary.p : blk == 2 ? ctx->bw_list_boundary.p : ctx->groups_boundary.p;
  auto special_grid = [&](int64_t items) { return blk == 1 ? sellp_blk_grid(ctx, items) : sellp_win_grid(ctx, items); };
  int g_in = gi ? (blk ? special_grid(gi) : sp_grid(ctx, gi, (partials && rvec), plain)) : 0;
  const int pw = plain ? sellp_pipe_wgs(ctx, partials && rvec) : 0;
  const int room = 256 * ((pw ? pw : 8) - 1); // (one workgroup slot per CU left to the exchange's kernel; the block-row
  if (!blk && g_in > room && ctx->nneigh > 0) //  kernel's one workgroup per CU leaves half the CU's wavefront slots free)
    g_in = room;
  const int g_bd = gb ? (blk ? special_grid(gb) : sp_grid(ctx, gb, (partials && rvec), plain)) : 0;
  if (partials && (size_t)(g_in + g_bd) > (size_t)SPMV_PSTRIDE)
    return fail(ctx, ZZZ_ERR_ARG, "partials buffer too small");
  TailArgs Ti, Tb;
  if (partials && ctx->tail_armed)
  {
    // one ticket over both launches: the workgroup that arrives last (in the boundary launch, or in the interior
    // one when no group touches a ghost column) finishes the reduction
    Ti = ctx->tail;
    Ti.expected = g_in + g_bd;
    Ti.base = 0;
    Tb = Ti;
    Tb.base = g_in;
    ctx->tail_armed = false;
    ctx->tail_used = true;
  }
  int rc = comm_halo_begin(ctx, x);
  if (rc)
    return rc;
  if (gi)
  {
    if (partials)
      launch_one<true>(ctx, g_in, x, y, partials, stop, list_in, gi, rvec, nn_is_rr, Ti, epi, blk);
    else
      launch_one<false>(ctx, g_in, x, y, nullptr, stop, list_in, gi, nullptr, 0, TailArgs(), epi, blk);
  }
  rc = comm_halo_end(ctx);
  if (rc)
    return rc;
  if (gb)
  {
    if (partials)
      launch_one<true>(ctx, g_bd, x, y, partials + g_in, stop, list_bd, gb, rvec, nn_is_rr, Tb, epi, blk);
    else
      launch_one<false>(ctx, g_bd, x, y, nullptr, stop, list_bd, gb, nullptr, 0, TailArgs(), epi, blk);
  }
  if (npartials)
    *npartials = g_in + g_bd;
  ZZZ_HIP(ctx, hipGetLastError());
  return ZZZ_OK;
}
#ifdef ZZZ_EXPERIMENTS
// the fused product + direction kernel on the whole matrix, or (partitioned matrix) interior groups, halo of z,
// boundary groups.  Partials of <p,w>: interior workgroups first.
int launch_sellp_dir(zzz_ctx* ctx, double* z, const double* p_old, double* p_new, double* xsol, double* y, double* partials,
                     int* npartials, int it, const CgParams& P, const double* pa, const double* pb, int np, bool overlap)
{
  if (int rc = sellp_need_generic(ctx))
    return rc;
  const bool nt = ctx->spmv_auto ? sp_stream_nt(ctx) : (ctx->spmv_variant & 1) != 0;
  const int2* off = reinterpret_cast<const int2*>(ctx->sp_desc.p);
  const int ncols = (int)ctx->nloc();
  auto go = [&](int grid, const int32_t* list, int64_t nlist, double* parts, int ghost) {
#define ZZZ_SPD_GO(NT, PERM)                                                                                           \
  hipLaunchKernelGGL((spmv_sellp_dir_kernel<NT, PERM>), dim3(grid), dim3(SP_BLOCK), 0, ctx->stream, off, ctx->sp_vals.p,  \
                     ctx->sp_codes16.p, ctx->sp_codes32.p, ctx->sp_meta.p, ctx->sp_perm.p, z, p_old, p_new, xsol, y,   \
                     (int)ctx->nrows, ncols, ctx->nslices, parts, list, nlist, ghost, ctx->state.p, ctx->beta_hist.p, \
                     ctx->dp_hist.p, ctx->alpha_hist.p, it, P, pa, pb, np)
    if (ctx->sp_sorted)
    {
      if (nt)
        ZZZ_SPD_GO(true, true);
      else
        ZZZ_SPD_GO(false, true);
    }
    else
    {
      if (nt)
        ZZZ_SPD_GO(true, false);
      else
        ZZZ_SPD_GO(false, false);
    }
#undef ZZZ_SPD_GO
  };
  if (overlap && ctx->have_group_split)
  {
    const int64_t gi = ctx->n_groups_interior, gb = ctx->n_groups_boundary;
    int g_in = gi ? sp_grid(ctx, gi, false, false) : 0;
    if (g_in > 256 * 7 && ctx->nneigh > 0)
      g_in = 256 * 7; // room for the exchange's kernel beside the persistent workgroups (launch_spmv_overlapped)
    const int g_bd = gb ? sp_grid(ctx, gb, false, false) : 8;
    if ((size_t)(g_in + g_bd) > (size_t)SPMV_PSTRIDE)
      return fail(ctx, ZZZ_ERR_ARG, "partials buffer too small");
    int rc = comm_halo_begin(ctx, z);
    if (rc)
      return rc;
    if (gi)
      go(g_in, ctx->groups_interior.p, gi, partials, 0);
    rc = comm_halo_end(ctx);
    if (rc)
      return rc;
    go(g_bd, ctx->groups_boundary.p, gb, partials + g_in, 1); // also with no boundary group: the ghost entries of p
    *npartials = g_in + g_bd;
  }
  else
  {
    if (ctx->comm)
    {
      int rc = comm_halo_forward(ctx, z);
      if (rc)
        return rc;
    }
    const int gs = sp_grid(ctx, (ctx->nslices + 3) / 4, false, false);
    go(gs, nullptr, 0, partials, ctx->n_ghost > 0 ? 1 : 0);
    *npartials = gs;
  }
  ZZZ_HIP(ctx, hipGetLastError());
  return ZZZ_OK;
}
#endif // ZZZ_EXPERIMENTS
ZZZ_PRELOAD_TU(sellp)
} // namespace zzz

// The CG product on the operator stream, software-pipelined (round 5).
//
// Replaces PETSc MatMult inside KSPSolve (src/poisson_problem.cpp:177, src/elasticity_problem.cpp:259) and the `action` of
// linalg::cg (src/cg.h:62), like spmv_sellp_kernel of zzz_sellp.hip, on the same stream and with the same arithmetic (a row's
// products added in ascending column order, mul and add rounded separately: bit-identical to the serial CSR loop).
//
// Why a second kernel.  Once the stream's values are 16-bit codes (zzz_sellp.hip, value dictionaries) a chunk is 1-2 KiB and
// the generic kernel is no longer bound by bytes but by its CHAIN: per chunk it requests codes and bases, waits (an HBM round
// trip), looks the values up in LDS, requests the eight gathers of x, waits (an L2 round trip), multiplies and adds -- 3.5-4 us
// per chunk and wavefront whatever the chunk holds (rocprofv3, round 5: wavefronts wait in s_waitcnt 72-88 % of their cycles;
// C2 0.083 ms = 0.49 of the HBM peak, C4 0.169 ms = 0.35, P3 6.2 M dofs 0.467 ms = 0.42), with one chunk in flight per
// wavefront.  Here a wavefront walks the chunks of ALL its slices as one flat sequence and keeps the stream data of the next
// two chunks in flight while it works on the current one:
//
//   * what it will meet is known without a memory round trip: the descriptors and mode words (zzz_sellp.h: two bits per chunk
//     = what there is to load for the chunk's columns) of its next 64 slices sit in five vector registers, lane i = step i,
//     read out with v_readlane;
//   * per chunk three vector loads, all counted by vmcnt, issued right BEHIND the gathers of the chunk two places earlier:
//     the 8 slot bases (or the 25 words of a periodic chunk) as one dword per lane, the eight 16-bit value codes, the column
//     codes (a buffer load whose descriptor has zero records where the chunk has none: the same count on every path, so the
//     compiler's `s_waitcnt vmcnt(3)` in front of the first product lets exactly these three stay in flight);
//   * vmcnt retires in order, so a prefetch lives from "behind the gathers of chunk k" to "the gathers of chunk k + 1 are
//     back": one iteration plus one L2 round trip -- enough to cover an HBM round trip once the iteration itself is short;
//   * the two stages are two named register sets and the loop body exists twice (consume A / refill A, consume B / refill B):
//     no register that a load is still writing is ever copied.
//
// Serves: natural row order, the value dictionary in LDS (DICT = 2) or per-slice tables (DICT = 3), x windows (block size 3),
// group lists of a partitioned matrix, the single-reduction form's extra sums.  Everything else (sorted rows, doubles, int32
// columns, slices of more than 32 chunks, the Chebyshev epilogue, the folded all-reduce) stays on the generic kernel.
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "zzz_sellp.h"

namespace zzz
{
struct PipeArgs
{
  const int2* desc;
  const unsigned long long* smode;
  const double* svals;
  const uint16_t* c16;
  const int32_t* meta;
  const uint16_t* vcode;
  const double* dict_g;
  int dict_n;
  const int32_t* sd_info;
  const double* x;
  double* y;
  int nrows;
  int nslices;
  double* partials;
  const int* stop_flag;
  const int32_t* group_list;
  int64_t nlist;
  const double* rvec;
  int pstride, nn_is_rr;
  const int2* win_info;
  const int2* win_seg;
  int dbg; // TEMPORARY timing probe
  unsigned long long* stamps; // ZZZ_PIPE_STAMPS build: per-segment cycle sums
};
#ifdef ZZZ_PIPE_STAMPS
#define ZZZ_STAMP(i)                                                                                                              \
  do                                                                                                                             \
  {                                                                                                                              \
    unsigned long long t_;                                                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                                           \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                                           \
    seg[i] += t_ - t_prev;                                                                                                       \
    t_prev = t_;                                                                                                                 \
  } while (0)
#else
#define ZZZ_STAMP(i)
#endif

// stream registers of one chunk
struct Stage
{
  int hv;    // lane e < 8: slot base e (word 0 with the mode bits); block size 3, lanes 8..32: the periodic chunk's 25 words
  uint4v vq; // eight 16-bit value codes
  uint4v cq; // column codes (16-bit: all of it; 8-bit: x, y)
  double xr; // DOT, the last chunk of a slice: x of the lane's own row (and, single-reduction form, r of it)
  double rr;
};

struct Ticket
{
  int c;    // chunk
  int info; // bits 0-3 entries per row in use (0: an empty slice), 4 first chunk of its slice, 5 last, 6 the slice exists,
            // 8-9 column-code class, 10 a ticket (clear: the sequence has ended)
  int s;    // slice
};
constexpr int TK_FIRST = 16, TK_LAST = 32, TK_LIVE = 64, TK_VALID = 1024;

template <bool NT, typename T>
__device__ inline T pipe_load(const T* p)
{
  return NT ? __builtin_nontemporal_load(p) : *p;
}

template <bool DOT, bool SR, bool NT, bool WIN, int DICT, bool BS3>
__global__ __launch_bounds__(SP_BLOCK, SP_PIPE_WGS) void spmv_pipe_kernel(PipeArgs a)
{
  // dynamic LDS: DICT == 2: the value dictionary (dict_n doubles, rounded up to 2), WIN: the group's x window behind it;
  // DICT == 3: one table of SD_MAX doubles per wavefront
  extern __shared__ __attribute__((aligned(16))) double pp_lds[];
  __shared__ double red[SP_BLOCK / 64];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  double* const xwin = pp_lds + (DICT == 2 ? ((a.dict_n + 1) & ~1) : 0);
  double* const sdl = pp_lds + wv * SD_MAX;
  const double* dict = DICT == 2 ? pp_lds : sdl;

  // the steps of this workgroup: groups first, first + n_in_xcd, ... of its XCD's eighth (sp_xcd_item)
  const int64_t ngroups = a.group_list ? a.nlist : ((int64_t)a.nslices + 3) / 4;
  const int xcd = blockIdx.x & 7;
  const int64_t lo = ngroups * xcd / 8, hi = ngroups * (xcd + 1) / 8;
  const int n_in_xcd = ((int)gridDim.x + 7 - xcd) >> 3;
  const int64_t first_item = lo + (blockIdx.x >> 3);
  const int n_steps = first_item < hi ? (int)((hi - first_item + n_in_xcd - 1) / n_in_xcd) : 0;

  // look-ahead registers: lane l <-> step regs_base + l
  int r_c0 = 0, r_y = 0, r_s = -1;
  unsigned r_smlo = 0, r_smhi = 0;
  int regs_base = 0;
  auto load_regs = [&](int base) {
    const int i = base + lane;
    const bool valid = i < n_steps;
    const int64_t t = first_item + (int64_t)(valid ? i : 0) * n_in_xcd;
    const int64_t g = (valid && a.group_list) ? a.group_list[t] : t;
    const int64_t s64 = 4 * g + wv;
    const bool live = valid && s64 < a.nslices;
    const int2 d = live ? a.desc[s64] : make_int2(0, 0);
    const unsigned long long sm = live ? a.smode[s64] : 0ull;
    r_c0 = d.x;
    r_y = d.y;
    r_s = valid ? (int)s64 : -1;
    r_smlo = (unsigned)sm;
    r_smhi = (unsigned)(sm >> 32);
  };
  load_regs(0);
  if (a.stop_flag && *a.stop_flag) // CG already converged: the host is a few iterations ahead
    return;
  if (DICT == 2)
  {
    for (int k = threadIdx.x; k < a.dict_n; k += SP_BLOCK)
      pp_lds[k] = a.dict_g[k];
    __syncthreads();
  }

  // the far cursor: (step, chunk of its slice) of the next ticket
  int f_i = 0, f_j = 0, f_nch = 0, f_c0 = 0, f_wl = 8, f_s = -1;
  unsigned f_smlo = 0, f_smhi = 0;
  auto step_fetch = [&]() {
    if (f_i - regs_base >= 64)
    {
      regs_base = f_i;
      load_regs(regs_base);
      __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0) HERE, on the rare path: the compiler would otherwise put it at the
                                          // join, in front of every step's v_readlane, and drain the prefetches there
    }
    const int l = f_i - regs_base;
    const int yy = __builtin_amdgcn_readlane(r_y, l);
    f_c0 = __builtin_amdgcn_readlane(r_c0, l);
    f_nch = yy & 0xffffff;
    f_wl = (int)((unsigned)yy >> 24);
    f_s = __builtin_amdgcn_readlane(r_s, l);
    f_smlo = (unsigned)__builtin_amdgcn_readlane((int)r_smlo, l);
    f_smhi = (unsigned)__builtin_amdgcn_readlane((int)r_smhi, l);
    f_j = 0;
  };
  if (n_steps > 0)
    step_fetch();
  auto next_ticket = [&]() -> Ticket {
    Ticket t;
    if (f_i >= n_steps)
    {
      t.c = 0; // (loads of a ticket that is none go to chunk 0 and are dropped)
      t.info = 0;
      t.s = -1;
      return t;
    }
    const bool live = f_s < a.nslices;
    const int nch = live ? f_nch : 0;
    const unsigned smw = f_j < 16 ? f_smlo >> (2 * f_j) : f_smhi >> (2 * (f_j - 16));
    const bool last = f_j + 1 >= nch;
    t.c = nch ? f_c0 + f_j : 0;
    t.info = (nch ? (last ? f_wl : 8) : 0) | (f_j == 0 ? TK_FIRST : 0) | (last ? TK_LAST : 0) | (live ? TK_LIVE : 0)
             | (nch ? (int)(smw & 3u) << 8 : 0) | TK_VALID;
    t.s = f_s;
    if (last)
    {
      ++f_i;
      if (f_i < n_steps)
        step_fetch();
    }
    else
      ++f_j;
    return t;
  };

  // the three stream loads of a chunk
  auto prefetch = [&](Stage& S, const Ticket& t) {
    const int64_t c = (a.dbg & 2) ? (t.c & 63) : t.c;
    const int cls = (t.info >> 8) & 3;
    const int32_t* hp = a.meta + c * 8 + (lane & 7);
    if (BS3)
    {
      const int32_t* tp = reinterpret_cast<const int32_t*>(a.c16 + c * 512);
      const int q = lane - 8;
      hp = lane < 8 ? hp : tp + (q < 24 ? q : 24);
    }
    S.hv = pipe_load<NT>(hp);
    S.vq = pipe_load<NT>(reinterpret_cast<const uint4v*>(a.vcode + c * 512) + lane);
    // column codes: ONE buffer load on every path, 16 B per lane at lane * 16 (16-bit codes) or lane * 8 (8-bit codes: the
    // lane's eight codes are the first 8 B of what it loads; the 8 B behind the last lane's lie inside the chunk's code
    // block, or, for codes in a value block's tail, in the next chunk's block or the array's slack).  Class none: a
    // descriptor without records -- nothing is fetched, the instruction still counts.
    const char* cbase = cls == SP_CLS_C8T ? reinterpret_cast<const char*>(a.svals + c * 512 + 448)
                                          : reinterpret_cast<const char*>(a.c16 + c * 512);
    const int crec = cls == SP_CLS_NONE ? 0 : 1024;
    const int csh = cls == SP_CLS_C16 ? 4 : 3;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(cbase), 0, crec, 0x00020000);
    const auto q = __builtin_amdgcn_raw_buffer_load_b128(rs, lane << csh, 0, NT ? 2 : 0);
    S.cq.x = q[0], S.cq.y = q[1], S.cq.z = q[2], S.cq.w = q[3];
    // the lane's own x (and r) for the sums of a slice's last chunk: first touches of those lines, i.e. as far away as the
    // stream -- requested with it (every chunk, the same count: a chunk that needs none reads entry 0)
    if (DOT)
    {
      const int r = t.s * 64 + lane;
      const bool need = (t.info & (TK_LAST | TK_LIVE)) == (TK_LAST | TK_LIVE) && r < a.nrows;
      S.xr = a.x[need ? r : 0];
      if (SR)
        S.rr = a.rvec[need ? r : 0];
    }
  };

  double sum = 0.0, dot = 0.0, dot_rx = 0.0, dot_nn = 0.0;
  int nwin = 0;
  Ticket T0 = next_ticket(), T1 = next_ticket(), T2;
  Stage A, B;
  prefetch(A, T0);
  prefetch(B, T1);

  // one chunk: consume stage S (ticket T), refill it for the chunk two places on (ticket Tn)
  auto body = [&](Stage& S, const Ticket& T, const Ticket& Tn) {
    const int w = T.info & 15;
    const bool first = (T.info & TK_FIRST) != 0, last = (T.info & TK_LAST) != 0, live = (T.info & TK_LIVE) != 0;
    if (first)
    {
      sum = 0.0;
      if (WIN)
      {
        // the four wavefronts work on one group; where the group has a window its segments of x are loaded into LDS first
        // (every wavefront takes part, also one without a slice of its own at the end of the matrix)
        const int g = T.s >> 2;
        const int2 wi = a.win_info[g];
        nwin = __builtin_amdgcn_readfirstlane(wi.x);
        if (nwin > 0)
        {
          const int2* __restrict__ sg = a.win_seg + (int64_t)g * SP_WIN_NSEG;
          __syncthreads(); // the previous group's window is done with
          int off = 0;
          for (int q = 0; q < nwin; ++q)
          {
            const int2 sq = sg[q];
            const int c0s = __builtin_amdgcn_readfirstlane(sq.x), len = __builtin_amdgcn_readfirstlane(sq.y);
            for (int k = threadIdx.x; k < len; k += SP_BLOCK)
              xwin[off + k] = a.x[c0s + k];
            off += len;
          }
          __syncthreads();
        }
      }
      if (DICT == 3)
      {
        // the slice's table into the wavefront's LDS (n == 0: no live slice here; the packer keeps every slice of a
        // stream this kernel serves on codes)
        const int n_sd = live ? __builtin_amdgcn_readfirstlane(a.sd_info[T.s]) : 0;
        __builtin_amdgcn_wave_barrier();
        for (int k = lane; k < n_sd; k += 64)
          sdl[k] = a.dict_g[(int64_t)T.s * SD_MAX + k];
        __builtin_amdgcn_wave_barrier();
      }
    }
    // columns
    const int hv = S.hv;
    const uint4v vq = S.vq, cq = S.cq;
    const int m0 = __builtin_amdgcn_readlane(hv, 0);
    int cl[8];
    if (BS3 && m0 < 0 && (m0 & 0x40000000))
    {
      // periodic chunk (block size 3): column = T[slot][row mod 3] + 3 (row div 3 - first); the 25 words sit in lanes
      // 8..32 of hv: lane l takes word 3 e + (l + phase) mod 3 of every slot e through the LDS crossbar
      const int l = lane + __builtin_amdgcn_readlane(hv, 32);
      const int q = l / 3, k = l - 3 * q, q3 = 3 * q;
#pragma unroll
      for (int e = 0; e < 8; ++e)
        cl[e] = __builtin_amdgcn_ds_bpermute(4 * (8 + 3 * e + k), hv) + q3;
    }
    else if ((m0 & 0x60000000) == 0x20000000 && m0 >= 0)
    {
      cl[0] = (m0 & 0x1fffffff) + lane;
#pragma unroll
      for (int e = 1; e < 8; ++e)
        cl[e] = __builtin_amdgcn_readlane(hv, e) + lane;
    }
    else if (m0 & 0x40000000)
    {
      cl[0] = (m0 & 0x1fffffff) + (int)(cq.x & 0xffu);
      cl[1] = __builtin_amdgcn_readlane(hv, 1) + (int)((cq.x >> 8) & 0xffu);
      cl[2] = __builtin_amdgcn_readlane(hv, 2) + (int)((cq.x >> 16) & 0xffu);
      cl[3] = __builtin_amdgcn_readlane(hv, 3) + (int)(cq.x >> 24);
      cl[4] = __builtin_amdgcn_readlane(hv, 4) + (int)(cq.y & 0xffu);
      cl[5] = __builtin_amdgcn_readlane(hv, 5) + (int)((cq.y >> 8) & 0xffu);
      cl[6] = __builtin_amdgcn_readlane(hv, 6) + (int)((cq.y >> 16) & 0xffu);
      cl[7] = __builtin_amdgcn_readlane(hv, 7) + (int)(cq.y >> 24);
    }
    else
    {
      cl[0] = m0 + (int)(cq.x & 0xffffu);
      cl[1] = __builtin_amdgcn_readlane(hv, 1) + (int)(cq.x >> 16);
      cl[2] = __builtin_amdgcn_readlane(hv, 2) + (int)(cq.y & 0xffffu);
      cl[3] = __builtin_amdgcn_readlane(hv, 3) + (int)(cq.y >> 16);
      cl[4] = __builtin_amdgcn_readlane(hv, 4) + (int)(cq.z & 0xffffu);
      cl[5] = __builtin_amdgcn_readlane(hv, 5) + (int)(cq.z >> 16);
      cl[6] = __builtin_amdgcn_readlane(hv, 6) + (int)(cq.w & 0xffffu);
      cl[7] = __builtin_amdgcn_readlane(hv, 7) + (int)(cq.w >> 16);
    }
    if (a.dbg & 1)
    {
#pragma unroll
      for (int e = 0; e < 8; ++e)
        cl[e] = lane + 64 * e;
    }
    // values: the codes' doubles from the dictionary in LDS (code 0 = +0.0: padding, slots beyond the chunk's width)
    double v[8];
    v[0] = dict[vq.x & 0xffffu];
    v[1] = dict[vq.x >> 16];
    v[2] = dict[vq.y & 0xffffu];
    v[3] = dict[vq.y >> 16];
    v[4] = dict[vq.z & 0xffffu];
    v[5] = dict[vq.z >> 16];
    v[6] = dict[vq.w & 0xffffu];
    v[7] = dict[vq.w >> 16];
    // x: all eight slots on every path (a slot beyond the width reads the column the packer left there, 0: a valid entry)
    double xv[8];
    if (WIN && nwin > 0)
    {
#pragma unroll
      for (int e = 0; e < 8; ++e)
        xv[e] = xwin[cl[e]];
    }
    else
    {
#pragma unroll
      for (int e = 0; e < 8; ++e)
        xv[e] = gather(a.x, cl[e]);
    }
    const int r = T.s * 64 + lane;
    const bool row = last && live && r < a.nrows;
    const double xr = DOT ? S.xr : 0.0, rr = (DOT && SR) ? S.rr : 0.0;
    // the stream of the chunk two places on: BEHIND the gathers (vmcnt retires in order: a prefetch the scheduler hoists
    // above them would have to be back before the first product)
    __builtin_amdgcn_sched_barrier(0);
    prefetch(S, Tn);
    __builtin_amdgcn_sched_barrier(0);
    // the row sums, in column order
    if (w == 8)
    {
#pragma unroll
      for (int e = 0; e < 8; ++e)
        sum += v[e] * xv[e];
    }
    else
    {
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (e < w)
          sum += v[e] * xv[e];
    }
    if (row)
    {
      a.y[r] = sum;
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
  };

  for (;;)
  {
    if (!(T0.info & TK_VALID))
      break;
    T2 = next_ticket();
    body(A, T0, T2);
    T0 = T1;
    T1 = T2;
    if (!(T0.info & TK_VALID))
      break;
    T2 = next_ticket();
    body(B, T0, T2);
    T0 = T1;
    T1 = T2;
  }

  if (DOT)
  {
    const double sres = block_reduce_sum(dot, red);
    double s1 = 0.0, s2 = 0.0;
    if (SR)
    {
      s1 = block_reduce_sum(dot_rx, red);
      s2 = block_reduce_sum(dot_nn, red);
    }
    if (threadIdx.x == 0)
    {
      a.partials[blockIdx.x] = sres;
      if (SR)
      {
        a.partials[a.pstride + blockIdx.x] = s1;
        a.partials[2 * a.pstride + blockIdx.x] = s2;
      }
    }
  }
}

// ---- streams of one-chunk slices (scalar P1 on a regular mesh: BASELINE configs 2 and 3) ---------------------------------
// Counters of the kernels above and of the generic one on C2 (rocprofv3, round 5): ~260 instructions per chunk and wavefront
// (110 of them scalar), and with the memory side taken out of the way (every load served from L2) the product still takes
// 90 us -- the CUs' instruction issue, not bytes or latency, is then the bound.  A slice of this stream IS a chunk: no chunk
// loop, no slice prologue, the step's scalars (chunk, width, column-code class) packed into one look-ahead register, every
// stream load a buffer load with a scalar offset (no per-lane address arithmetic), rows beyond the matrix dropped by the
// descriptors' range check instead of by branches, one stage of look-ahead (the next step's stream behind this step's
// gathers).  ~100 instructions per slice.
// (the stream's arrays as __restrict__ parameters of their own: only then does the compiler know that the stores to y do not
// touch them, and reads the slot bases with scalar loads inside the loop)
template <bool DOT, bool SR, bool NT>
__global__ __launch_bounds__(SP_BLOCK, SR ? SP_ONE_WGS_SR : SP_ONE_WGS) void spmv_one_kernel(
    const int2* __restrict__ p_desc, const unsigned long long* __restrict__ p_smode, const double* __restrict__ p_svals,
    const uint16_t* __restrict__ p_c16, const int32_t* __restrict__ p_meta, const uint16_t* __restrict__ p_vcode,
    const double* __restrict__ p_dict, const double* __restrict__ p_x, double* __restrict__ p_y, const double* __restrict__ p_rvec,
    const int32_t* __restrict__ p_list, PipeArgs a)
{
  extern __shared__ __attribute__((aligned(16))) double pp_lds[]; // the value dictionary
  __shared__ double red[SP_BLOCK / 64];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const double* dict = pp_lds;

  const int64_t ngroups = p_list ? a.nlist : ((int64_t)a.nslices + 3) / 4;
  const int xcd = blockIdx.x & 7;
  const int64_t lo = ngroups * xcd / 8, hi = ngroups * (xcd + 1) / 8;
  const int n_in_xcd = ((int)gridDim.x + 7 - xcd) >> 3;
  const int64_t first_item = lo + (blockIdx.x >> 3);
  const int n_steps = first_item < hi ? (int)((hi - first_item + n_in_xcd - 1) / n_in_xcd) : 0;

  // look-ahead registers, lane l <-> step regs_base + l: info = chunk | width << 25 | class << 29 | exists << 31; slice
  int r_info = 0, r_s = 0;
  auto load_regs = [&](int base) {
    const int i = base + lane;
    const bool valid = i < n_steps;
    const int64_t t = first_item + (int64_t)(valid ? i : 0) * n_in_xcd;
    const int64_t g = (valid && p_list) ? p_list[t] : t;
    const int64_t s64 = 4 * g + wv;
    const bool live = valid && s64 < a.nslices;
    const int2 d = live ? p_desc[s64] : make_int2(0, 0);
    const unsigned sm = live ? (unsigned)p_smode[s64] : 0u;
    const int nch = d.y & 0xffffff;
    r_info = live ? ((nch ? d.x : 0) | ((nch ? (int)((unsigned)d.y >> 24) : 0) << 25) | ((nch ? (int)(sm & 3u) : 0) << 29) | (int)0x80000000) : 0;
    r_s = live ? (int)s64 : 0;
  };
  load_regs(0);
  if (a.stop_flag && *a.stop_flag)
    return;
  for (int k = threadIdx.x; k < a.dict_n; k += SP_BLOCK)
    pp_lds[k] = p_dict[k];
  __syncthreads();

  // descriptors: every stream array, x, y (and r) -- offsets below 4 GB (a stream of one-chunk slices of < 2^20 slices per GB)
  const int nb8 = a.nrows * 8;
  const __amdgpu_buffer_rsrc_t rs_vq = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p_vcode), 0, a.nslices * 1024, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(p_x), 0, nb8, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(p_y, 0, nb8, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(SR ? p_rvec : p_x), 0, nb8, 0x00020000);
  const int lane8 = lane * 8, lane16 = lane * 16;
  const int aux = NT ? 2 : 0;

  struct One
  {
    int b[8];  // slot bases (scalars; word 0 with the mode bits)
    uint4v vq; // eight 16-bit value codes
    uint4v cq; // column codes (16-bit: all of it; 8-bit: x, y)
    double xr, rr;
  };
  // Every vector-memory instruction costs the CU's address path ~15 clk whatever it moves (tools/micro/gather_rate.hip: one
  // lane or 64, 8 B or 16 B per lane), and seven gathers are 105 of them per slice: nothing else goes through that path that
  // can go elsewhere -- the bases through the scalar cache, no column-code load where a slice has no codes, no gather for a
  // slot beyond the slice's width.
  auto prefetch = [&](One& S, int info, int s) {
    const int c = info & 0x1ffffff;
    const int32_t* __restrict__ mp = p_meta + (int64_t)c * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e)
      S.b[e] = mp[e]; // (s_load_dwordx8)
    const auto q = __builtin_amdgcn_raw_buffer_load_b128(rs_vq, lane16, c << 10, aux);
    S.vq.x = q[0], S.vq.y = q[1], S.vq.z = q[2], S.vq.w = q[3];
    // column codes: ONE load whatever the class -- what is in flight behind the gathers must be the same count on every path
    // (the compiler's counted waits in front of the products); class none: a descriptor without records, nothing is fetched.
    // 8-bit codes are the first 8 B of the 16 a lane loads.
    const int cls = (info >> 29) & 3;
    const char* cbase = cls == SP_CLS_C8T ? reinterpret_cast<const char*>(p_svals) + ((int64_t)c * 4096 + 3584)
                                          : reinterpret_cast<const char*>(p_c16) + (int64_t)c * 1024;
    const __amdgpu_buffer_rsrc_t rs_c = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(cbase), 0, cls == SP_CLS_NONE ? 0 : 1024, 0x00020000);
    const auto cqv = __builtin_amdgcn_raw_buffer_load_b128(rs_c, cls == SP_CLS_C16 ? lane16 : lane8, 0, aux);
    S.cq.x = cqv[0], S.cq.y = cqv[1], S.cq.z = cqv[2], S.cq.w = cqv[3];
    if (DOT)
    {
      const auto u = __builtin_amdgcn_raw_buffer_load_b64(rs_x, lane8, s << 9, 0);
      S.xr = __hiloint2double((int)u[1], (int)u[0]);
      if (SR)
      {
        const auto v = __builtin_amdgcn_raw_buffer_load_b64(rs_r, lane8, s << 9, 0);
        S.rr = __hiloint2double((int)v[1], (int)v[0]);
      }
    }
  };

  double dot = 0.0, dot_rx = 0.0, dot_nn = 0.0;
  int regs_base = 0;
#ifdef ZZZ_PIPE_STAMPS
  unsigned long long seg[6] = {0, 0, 0, 0, 0, 0}, t_prev = 0;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_prev)::"memory");
#endif
  auto step_scalars = [&](int i, int& info, int& s) {
    if (i >= n_steps)
    {
      info = 0;
      s = 0;
      return;
    }
    if (i - regs_base >= 64)
    {
      regs_base = i;
      load_regs(regs_base);
      __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0) here, on the rare path (see spmv_pipe_kernel)
    }
    info = __builtin_amdgcn_readlane(r_info, i - regs_base);
    s = __builtin_amdgcn_readlane(r_s, i - regs_base);
  };

  // one slice: consume stage S (scalars info, s), request the stream of the step two places on into stage N
  auto body = [&](One& S, int info, int s, One& N, int info_n, int s_n) {
    const int w = (info >> 25) & 15, cls = (info >> 29) & 3;
    ZZZ_STAMP(0); // loop overhead, the step's scalars
    const uint4v vq = S.vq, cq = S.cq;
#ifdef ZZZ_PIPE_STAMPS
    asm volatile("" ::"v"(vq.x), "v"(cq.x), "s"(S.b[0]));
    ZZZ_STAMP(1); // waiting for the stage's stream data
#endif
    const double xr = DOT ? S.xr : 0.0, rr = (DOT && SR) ? S.rr : 0.0;
    double sum = 0.0;
    // W entries per row: W gathers, W dictionary look-ups (waited for before the scalar loads of the next stage are requested:
    // both count on lgkmcnt and scalar loads return out of order), the next stage's stream, W products
    auto run = [&](auto wtag) {
      constexpr int W = decltype(wtag)::value;
      unsigned cc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; // the lane's column codes (class none: the lane itself)
      if (cls == SP_CLS_C16)
      {
        cc[0] = cq.x & 0xffffu, cc[1] = cq.x >> 16, cc[2] = cq.y & 0xffffu, cc[3] = cq.y >> 16;
        cc[4] = cq.z & 0xffffu, cc[5] = cq.z >> 16, cc[6] = cq.w & 0xffffu, cc[7] = cq.w >> 16;
      }
      else if (cls != SP_CLS_NONE)
      {
        cc[0] = cq.x & 0xffu, cc[1] = (cq.x >> 8) & 0xffu, cc[2] = (cq.x >> 16) & 0xffu, cc[3] = cq.x >> 24;
        cc[4] = cq.y & 0xffu, cc[5] = (cq.y >> 8) & 0xffu, cc[6] = (cq.y >> 16) & 0xffu, cc[7] = cq.y >> 24;
      }
      else
      {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          cc[e] = lane;
      }
      double xv[W], v[W];
#pragma unroll
      for (int e = 0; e < W; ++e)
      {
        const unsigned base = e == 0 ? (unsigned)(S.b[0] & 0x1fffffff) : (unsigned)S.b[e];
        xv[e] = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(p_x) + ((base + cc[e]) << 3));
      }
      const unsigned vc[8] = {vq.x & 0xffffu, vq.x >> 16, vq.y & 0xffffu, vq.y >> 16, vq.z & 0xffffu, vq.z >> 16, vq.w & 0xffffu, vq.w >> 16};
#pragma unroll
      for (int e = 0; e < W; ++e)
        v[e] = dict[vc[e]];
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_waitcnt(0xC07F); // lgkmcnt(0)
      ZZZ_STAMP(2); // decode, gathers and look-ups issued, look-ups back
      prefetch(N, info_n, s_n);
      __builtin_amdgcn_sched_barrier(0);
      ZZZ_STAMP(3); // requesting the stream two steps on
#pragma unroll
      for (int e = 0; e < W; ++e)
        sum += v[e] * xv[e];
#ifdef ZZZ_PIPE_STAMPS
      asm volatile("" ::"v"(sum));
      ZZZ_STAMP(4); // the gathers' round trip and the products
#endif
    };
    if (w == 7) // (an interior P1 row of the Kuhn mesh)
      run(std::integral_constant<int, 7>());
    else if (w == 8)
      run(std::integral_constant<int, 8>());
    else
    {
      // a narrow slice (the mesh's corners, an empty slice): all eight slots fetched (a slot beyond the width reads the column
      // the packer left there, 0: a valid entry), the products of the first w added
      double xv[8], v[8];
      unsigned cc[8];
      if (cls == SP_CLS_C16)
      {
        cc[0] = cq.x & 0xffffu, cc[1] = cq.x >> 16, cc[2] = cq.y & 0xffffu, cc[3] = cq.y >> 16;
        cc[4] = cq.z & 0xffffu, cc[5] = cq.z >> 16, cc[6] = cq.w & 0xffffu, cc[7] = cq.w >> 16;
      }
      else if (cls != SP_CLS_NONE)
      {
        cc[0] = cq.x & 0xffu, cc[1] = (cq.x >> 8) & 0xffu, cc[2] = (cq.x >> 16) & 0xffu, cc[3] = cq.x >> 24;
        cc[4] = cq.y & 0xffu, cc[5] = (cq.y >> 8) & 0xffu, cc[6] = (cq.y >> 16) & 0xffu, cc[7] = cq.y >> 24;
      }
      else
      {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          cc[e] = lane;
      }
      const unsigned vc[8] = {vq.x & 0xffffu, vq.x >> 16, vq.y & 0xffffu, vq.y >> 16, vq.z & 0xffffu, vq.z >> 16, vq.w & 0xffffu, vq.w >> 16};
#pragma unroll
      for (int e = 0; e < 8; ++e)
      {
        const unsigned base = e == 0 ? (unsigned)(S.b[0] & 0x1fffffff) : (unsigned)S.b[e];
        xv[e] = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(p_x) + ((base + (e < w ? cc[e] : 0u)) << 3));
        v[e] = dict[vc[e]];
      }
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_waitcnt(0xC07F);
      prefetch(N, info_n, s_n);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int e = 0; e < 6; ++e)
        if (e < w)
          sum += v[e] * xv[e];
    }
    if (info < 0) // the slice exists; lanes beyond the last row are dropped by the descriptor's range check
    {
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(uint2v, sum), rs_y, lane8, s << 9, 0);
      if (DOT && (s << 6) + lane < a.nrows) // (no row, no term: a lane without a row may hold 0 * inf)
      {
        dot += sum * xr;
        if (SR)
        {
          dot_rx += rr * xr;
          dot_nn += a.nn_is_rr ? rr * rr : xr * xr;
        }
      }
    }
    ZZZ_STAMP(5); // store, the sums
  };

  // three stages in rotation: consume A (step i) and refill C's successor ... no register that a load is still writing is
  // ever copied (the loop body exists three times)
  One A, B, C;
  int info_a, s_a, info_b, s_b, info_c, s_c;
  step_scalars(0, info_a, s_a);
  step_scalars(1, info_b, s_b);
  prefetch(A, info_a, s_a);
  prefetch(B, info_b, s_b);
  for (int i = 0; i < n_steps;)
  {
    step_scalars(i + 2, info_c, s_c);
    body(A, info_a, s_a, C, info_c, s_c);
    if (++i >= n_steps)
      break;
    step_scalars(i + 2, info_a, s_a);
    body(B, info_b, s_b, A, info_a, s_a);
    if (++i >= n_steps)
      break;
    step_scalars(i + 2, info_b, s_b);
    body(C, info_c, s_c, B, info_b, s_b);
    ++i;
  }

#ifdef ZZZ_PIPE_STAMPS
  if (lane == 0 && a.stamps)
    for (int q = 0; q < 6; ++q)
      atomicAdd(&a.stamps[q], seg[q]);
  if (threadIdx.x == 0 && a.stamps)
    atomicAdd(&a.stamps[7], (unsigned long long)n_steps);
#endif
  if (DOT)
  {
    const double sres = block_reduce_sum(dot, red);
    double s1 = 0.0, s2 = 0.0;
    if (SR)
    {
      s1 = block_reduce_sum(dot_rx, red);
      s2 = block_reduce_sum(dot_nn, red);
    }
    if (threadIdx.x == 0)
    {
      a.partials[blockIdx.x] = sres;
      if (SR)
      {
        a.partials[a.pstride + blockIdx.x] = s1;
        a.partials[2 * a.pstride + blockIdx.x] = s2;
      }
    }
  }
}

// ---- host side -----------------------------------------------------------------------------------------------------
// Do these kernels serve this context's stream, and with how many workgroups per CU?  0: the generic kernel.  (What a launch
// adds: no Chebyshev epilogue, no folded all-reduce.)
static bool pipe_one(const zzz_ctx* ctx)
{
  return ctx->sp_one_chunk && ctx->bs == 1 && ctx->sp_win_max == 0 && ctx->nslices < (1 << 21) && !(ctx->sellp_pipe & 2);
}
int sellp_pipe_wgs(const zzz_ctx* ctx, bool sr)
{
  if (!ctx->sellp_pipe || !ctx->sp_pipe_ok || ctx->sp_sorted || ctx->sp_chunks <= 0)
    return 0;
  const bool d2 = ctx->sp_dict_on && ctx->sp_dict_n <= SP_DICT_LDS_ENTRIES && !ctx->sp_sd_on;
  const bool d3 = ctx->sp_sd_on && ctx->sp_sd_all;
  if (!d2 && !d3)
    return 0;
  if (ctx->sp_win_max > 0 && !d2)
    return 0;
  return d2 && pipe_one(ctx) ? (sr ? SP_ONE_WGS_SR : SP_ONE_WGS) : SP_PIPE_WGS;
}

bool launch_sellp_pipe(zzz_ctx* ctx, bool dot, bool nt, int grid, const double* x, double* y, double* partials, const int* stop,
                       const int32_t* group_list, int64_t nlist, const double* rvec, int nn_is_rr)
{
  const int wgs = sellp_pipe_wgs(ctx, dot && rvec);
  if (!wgs)
    return false;
  const bool d3 = ctx->sp_sd_on;
  const bool win = ctx->sp_win_max > 0;
  PipeArgs a;
  a.desc = reinterpret_cast<const int2*>(ctx->sp_desc.p);
  a.smode = ctx->sp_smode.p;
  a.svals = ctx->sp_vals.p;
  a.c16 = ctx->sp_codes16.p;
  a.meta = ctx->sp_meta.p;
  a.vcode = d3 ? ctx->sp_vcode8.p : ctx->sp_vcode.p;
  a.dict_g = d3 ? ctx->sp_sd_vals.p : ctx->sp_dict.p;
  a.dict_n = ctx->sp_dict_n;
  a.sd_info = ctx->sp_sd_info.p;
  a.x = x;
  a.y = y;
  a.nrows = (int)ctx->nrows;
  a.nslices = (int)ctx->nslices;
  a.partials = partials;
  a.stop_flag = stop;
  a.group_list = group_list;
  a.nlist = nlist;
  a.rvec = rvec;
  a.pstride = SPMV_PSTRIDE;
  a.nn_is_rr = nn_is_rr;
  a.win_info = reinterpret_cast<const int2*>(ctx->sp_win_info.p);
  a.win_seg = reinterpret_cast<const int2*>(ctx->sp_win_seg.p);
  a.dbg = getenv("ZZZ_PIPE_DBG") ? atoi(getenv("ZZZ_PIPE_DBG")) : 0;
  a.stamps = nullptr;
#ifdef ZZZ_PIPE_STAMPS
  static unsigned long long* stamps_dev = nullptr;
  static int stamps_launches = 0;
  if (!stamps_dev)
  {
    (void)hipMalloc((void**)&stamps_dev, 64);
    (void)hipMemset(stamps_dev, 0, 64);
  }
  a.stamps = stamps_dev;
  if (++stamps_launches % 32 == 0)
  {
    unsigned long long h[8];
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipMemcpy(h, stamps_dev, 64, hipMemcpyDeviceToHost);
    (void)hipMemset(stamps_dev, 0, 64);
    const double n = (double)h[7] * 4.0; // wavefront-steps
    fprintf(stderr, "STAMPS per slice and wavefront (clk): overhead %.0f  stream wait %.0f  decode+issue+lookups %.0f  prefetch issue %.0f  gathers+products %.0f  store %.0f  | steps %llu\n",
            h[0] / n, h[1] / n, h[2] / n, h[3] / n, h[4] / n, h[5] / n, h[7]);
  }
#endif
  const size_t lds = d3 ? (size_t)4 * SD_MAX * sizeof(double)
                        : (size_t)((ctx->sp_dict_n + 1) & ~1) * sizeof(double) + (win ? (size_t)ctx->sp_win_max * sizeof(double) : 0);
  const bool bs3 = ctx->bs == 3;
  if (wgs != SP_PIPE_WGS)
  {
#define ZZZ_ONE_GO(DOT, SR, NT)                                                                                                    \
  hipLaunchKernelGGL((spmv_one_kernel<DOT, SR, NT>), dim3(grid), dim3(SP_BLOCK), lds, ctx->stream, a.desc, a.smode, a.svals, a.c16, \
                     a.meta, a.vcode, a.dict_g, a.x, a.y, a.rvec, a.group_list, a)
    if (dot && rvec)
    {
      if (nt)
        ZZZ_ONE_GO(true, true, true);
      else
        ZZZ_ONE_GO(true, true, false);
    }
    else if (dot)
    {
      if (nt)
        ZZZ_ONE_GO(true, false, true);
      else
        ZZZ_ONE_GO(true, false, false);
    }
    else
    {
      if (nt)
        ZZZ_ONE_GO(false, false, true);
      else
        ZZZ_ONE_GO(false, false, false);
    }
#undef ZZZ_ONE_GO
    return true;
  }
#define ZZZ_PIPE_GO(DOT, NT, WIN, DICT, BS3)                                                                                         \
  do                                                                                                                               \
  {                                                                                                                                \
    if (DOT && rvec)                                                                                                               \
      hipLaunchKernelGGL((spmv_pipe_kernel<DOT, DOT, NT, WIN, DICT, BS3>), dim3(grid), dim3(SP_BLOCK), lds, ctx->stream, a);       \
    else                                                                                                                           \
      hipLaunchKernelGGL((spmv_pipe_kernel<DOT, false, NT, WIN, DICT, BS3>), dim3(grid), dim3(SP_BLOCK), lds, ctx->stream, a);     \
  } while (0)
#define ZZZ_PIPE_F(DOT, NT)                                                                                                          \
  do                                                                                                                               \
  {                                                                                                                                \
    if (d3)                                                                                                                        \
    {                                                                                                                              \
      if (bs3)                                                                                                                     \
        ZZZ_PIPE_GO(DOT, NT, false, 3, true);                                                                                      \
      else                                                                                                                         \
        ZZZ_PIPE_GO(DOT, NT, false, 3, false);                                                                                     \
    }                                                                                                                              \
    else if (win)                                                                                                                  \
      ZZZ_PIPE_GO(DOT, NT, true, 2, true);                                                                                         \
    else if (bs3)                                                                                                                  \
      ZZZ_PIPE_GO(DOT, NT, false, 2, true);                                                                                        \
    else                                                                                                                           \
      ZZZ_PIPE_GO(DOT, NT, false, 2, false);                                                                                       \
  } while (0)
  if (dot)
  {
    if (nt)
      ZZZ_PIPE_F(true, true);
    else
      ZZZ_PIPE_F(true, false);
  }
  else
  {
    if (nt)
      ZZZ_PIPE_F(false, true);
    else
      ZZZ_PIPE_F(false, false);
  }
#undef ZZZ_PIPE_F
#undef ZZZ_PIPE_GO
  return true;
}
} // namespace zzz

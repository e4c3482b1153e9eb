// The CG product on operator streams of ONE-CHUNK SLICES (round 5): scalar P1 on a regular mesh, BASELINE configs 2 and 3.
//
// Replaces PETSc MatMult inside KSPSolve (src/poisson_problem.cpp:177) and the `action` of linalg::cg (src/cg.h:62), like
// spmv_sellp_kernel of zzz_sellp.hip, on the same stream and with the same arithmetic (a row's products added in ascending
// column order, mul and add rounded separately: bit-identical to the serial CSR loop and to the generic kernel).
//
// What was measured before this kernel was written (MI355X, C2: 10 M rows, 156 k slices, values as 16-bit codes):
//   * the generic kernel: 0.083 ms = 0.49 of the HBM peak; its wavefronts sit in s_waitcnt 80 % of their cycles;
//   * timing probes with one kind of access after the other taken out (wrong results, time only): without the gathers of x the
//     kernel runs at 5.9 TB/s, i.e. at the speed of its bytes -- the gathers, which add no HBM bytes, cost 36 of 91 us; so do the
//     stream loads when the gathers stay; the parts ADD UP (instruction issue alone: 32 us);
//   * tools/micro/gather_rate.hip: a vector-memory instruction costs the CU's address path ~15 clk whatever it moves -- two
//     lanes or 64, 8 B or 16 B per lane, a dense run or one address 64 times; only scattered lines cost more (31 clk);
//   * variants of a lane-per-row kernel with look-ahead of one or two slices, 5 to 8 workgroups per CU, half the scalar
//     instructions, the bases through the scalar cache: each took what its COUNT of vector-memory instructions per slice
//     predicts (10.3 -> 0.080 ms, 11.3 -> 0.085, 13 -> 0.091): ~25 clk per instruction in the product's mix of hits and misses.
// So this kernel moves the same bytes in fewer instructions (below); what remains (0.074 ms at C2, 0.55 of the peak) is no
// longer explained by that count -- deeper look-ahead, a touch of the farthest run a step ahead and staggered wavefronts
// were measured and changed nothing: see DESIGN.md section 4b.
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "zzz_sellp.h"

namespace zzz
{
// what the launcher hands over beside the stream's arrays (those are __restrict__ parameters of their own: only then does the
// compiler know that the stores to y do not touch them, and reads the slot bases with scalar loads inside the loop)
struct PipeArgs
{
  int dict_n;
  int nrows;
  int nslices;
  double* partials;
  const int* stop_flag;
  int64_t nlist;
  int pstride, nn_is_rr;
};

template <bool NT, typename T>
__device__ inline T pipe_load(const T* p)
{
  return NT ? __builtin_nontemporal_load(p) : *p;
}

// TWO slices per wavefront and step, a lane owning rows 2 l and 2 l + 1 of the 128: where both slices are affine with the same
// slot layout (slot e of the second starts 64 columns behind slot e of the first: an interior slice pair of a regular mesh --
// 99.4 % of the pairs at C2; found once per packing, k_sp_pairs) slot e of the pair is ONE run of 128 doubles, i.e. one 16-B
// load per lane instead of two gathers; x of the own rows and the store are one 16-B access each; the codes of a lane's two
// rows are 32 contiguous bytes of one of the two chunks.  Eleven vector-memory instructions per 128 rows instead of twenty-one.
// A pair that does not qualify (a slice with column codes, another width, the last slice of an odd count) takes its two slices
// one after the other in the lane-per-row form.  Slot bases come through the scalar cache (no vector instruction), a slice's
// column codes are loaded only where it has any, a slot beyond the width costs nothing.  One stage of look-ahead: the next
// step's value codes and bases are requested behind this step's gathers (vmcnt retires in order: behind them, so that the
// products wait for the gathers alone); the two stages are two named register sets and the loop body exists twice -- no
// register that a load is still writing is ever copied.  What a wavefront will meet is known without a memory round trip: the
// descriptors and mode words (zzz_sellp.h) of its next 64 steps sit in three vector registers, lane i = step i.
typedef double dbl2u __attribute__((ext_vector_type(2), aligned(8))); // a 16-B load from an 8-B aligned address

// pairs[p] = 1: slices 2 p and 2 p + 1 form an affine pair (see above)
__global__ __launch_bounds__(256) void k_sp_pairs(const int2* __restrict__ desc, const unsigned long long* __restrict__ smode,
                                                  const int32_t* __restrict__ meta, int64_t nslices, uint8_t* __restrict__ pairs)
{
  const int64_t npairs = (nslices + 1) / 2;
  for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < npairs; p += (int64_t)gridDim.x * blockDim.x)
  {
    bool ok = 2 * p + 1 < nslices;
    if (ok)
    {
      const int2 da = desc[2 * p], db = desc[2 * p + 1];
      ok = (da.y & 0xffffff) == 1 && (db.y & 0xffffff) == 1 && (da.y >> 24) == (db.y >> 24) && (smode[2 * p] & 3ull) == 0ull
           && (smode[2 * p + 1] & 3ull) == 0ull;
      if (ok)
      {
        const int32_t* ma = meta + (int64_t)da.x * 8;
        const int32_t* mb = meta + (int64_t)db.x * 8;
        const int w = da.y >> 24;
        ok = ma[0] >= 0 && mb[0] >= 0 && (ma[0] & 0x60000000) == 0x20000000 && (mb[0] & 0x60000000) == 0x20000000; // affine, both
        for (int e = 0; e < w && ok; ++e)
        {
          const int ba = e ? ma[e] : (ma[0] & 0x1fffffff), bb = e ? mb[e] : (mb[0] & 0x1fffffff);
          ok = bb - ba == 64;
        }
      }
    }
    pairs[p] = ok ? 1 : 0;
  }
}

template <bool DOT, bool SR, bool NT>
__global__ __launch_bounds__(SP_BLOCK, SR ? SP_ONE_WGS_SR : SP_ONE_WGS) void spmv_one_kernel(
    const int2* __restrict__ p_desc, const unsigned long long* __restrict__ p_smode, const uint8_t* __restrict__ p_pairs,
    const double* __restrict__ p_svals, const uint16_t* __restrict__ p_c16, const int32_t* __restrict__ p_meta,
    const uint16_t* __restrict__ p_vcode, const double* __restrict__ p_dict, const double* __restrict__ p_x, double* __restrict__ p_y,
    const double* __restrict__ p_rvec, const int32_t* __restrict__ p_list, PipeArgs a)
{
  extern __shared__ __attribute__((aligned(16))) double pp_lds[]; // the value dictionary
  __shared__ double red[SP_BLOCK / 64];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const double* dict = pp_lds;

  // steps: a workgroup takes two groups of four slices at a time (list entries 2 t, 2 t + 1, or groups 2 t, 2 t + 1), wavefront
  // wv the slices 4 g + 2 (wv & 1), + 1 of group g = the (wv >> 1)-th of the two
  const int64_t ngroups = p_list ? a.nlist : ((int64_t)a.nslices + 3) / 4;
  const int64_t nitems = (ngroups + 1) / 2;
  const int xcd = blockIdx.x & 7;
  const int64_t lo = nitems * xcd / 8, hi = nitems * (xcd + 1) / 8;
  const int n_in_xcd = ((int)gridDim.x + 7 - xcd) >> 3;
  const int64_t first_item = lo + (blockIdx.x >> 3);
  const int n_steps = first_item < hi ? (int)((hi - first_item + n_in_xcd - 1) / n_in_xcd) : 0;

  // look-ahead registers, lane l <-> step regs_base + l: per slice chunk | width << 25 | class << 29 | exists << 31;
  // the first slice's index, bit 30: an affine pair
  int r_ia = 0, r_ib = 0, r_s = 0;
  auto load_regs = [&](int base) {
    const int i = base + lane;
    const bool valid = i < n_steps;
    const int64_t t = first_item + (int64_t)(valid ? i : 0) * n_in_xcd;
    const int64_t gi = 2 * t + (wv >> 1);
    const bool have = valid && gi < ngroups;
    const int64_t g = (have && p_list) ? p_list[gi] : gi;
    const int64_t sa = 4 * g + 2 * (wv & 1);
    auto slice_info = [&](int64_t s64) -> int {
      if (!(have && s64 < a.nslices))
        return 0;
      const int2 d = p_desc[s64];
      const unsigned sm = (unsigned)p_smode[s64];
      const int nch = d.y & 0xffffff;
      return (nch ? d.x : 0) | ((nch ? (int)((unsigned)d.y >> 24) : 0) << 25) | ((nch ? (int)(sm & 3u) : 0) << 29) | (int)0x80000000;
    };
    r_ia = slice_info(sa);
    r_ib = slice_info(sa + 1);
    const bool pr = have && sa + 1 < a.nslices && p_pairs[sa >> 1] != 0;
    r_s = have ? ((int)sa | (pr ? 0x40000000 : 0)) : 0;
  };
  load_regs(0);
  if (a.stop_flag && *a.stop_flag)
    return;
  {
    // the dictionary into LDS: every thread's (at most eight) entries requested together, one round trip
    double t[SP_DICT_LDS_ENTRIES / SP_BLOCK];
#pragma unroll
    for (int i = 0; i < SP_DICT_LDS_ENTRIES / SP_BLOCK; ++i)
    {
      const int k = (int)threadIdx.x + i * SP_BLOCK;
      t[i] = k < a.dict_n ? p_dict[k] : 0.0;
    }
#pragma unroll
    for (int i = 0; i < SP_DICT_LDS_ENTRIES / SP_BLOCK; ++i)
    {
      const int k = (int)threadIdx.x + i * SP_BLOCK;
      if (k < a.dict_n)
        pp_lds[k] = t[i];
    }
  }
  __syncthreads();

  // descriptors: the codes, x, y (and r) -- offsets below 4 GB (the launcher sees to it)
  const int nb8 = a.nrows * 8;
  const __amdgpu_buffer_rsrc_t rs_vq = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p_vcode), 0, a.nslices * 1024, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(p_x), 0, nb8, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(p_y, 0, nb8, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(SR ? p_rvec : p_x), 0, nb8, 0x00020000);
  const int lane8 = lane * 8, lane16 = lane * 16;
  const int aux = NT ? 2 : 0;

  struct Two
  {
    int ba[8];          // slot bases of the first slice (scalars; word 0 with the mode bits; a pair's second: 64 on)
    uint4v c0, c1;      // value codes: pair form, the lane's rows 2 l and 2 l + 1; else the lane's row of either slice
  };
  // requested behind the gathers of the step before: bases (scalar loads), value codes
  auto prefetch = [&](Two& S, int ia, int ib, int sw) {
    const int ca = ia & 0x1ffffff, cb = ib & 0x1ffffff;
    const int32_t* __restrict__ ma = p_meta + (int64_t)ca * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e)
      S.ba[e] = ma[e];
    int off0, off1;
    if (sw & 0x40000000)
    {
      // rows 2 l and 2 l + 1 of the 128: lanes 0..31 in the first chunk, 32..63 in the second, 32 contiguous bytes per lane
      off0 = ((lane < 32 ? ca : cb) << 10) + ((lane & 31) << 5);
      off1 = off0 + 16;
    }
    else
    {
      off0 = (ca << 10) + lane16;
      off1 = (cb << 10) + lane16;
    }
    const auto q0 = __builtin_amdgcn_raw_buffer_load_b128(rs_vq, off0, 0, aux);
    const auto q1 = __builtin_amdgcn_raw_buffer_load_b128(rs_vq, off1, 0, aux);
    S.c0.x = q0[0], S.c0.y = q0[1], S.c0.z = q0[2], S.c0.w = q0[3];
    S.c1.x = q1[0], S.c1.y = q1[1], S.c1.z = q1[2], S.c1.w = q1[3];
  };
  // a slice's column codes, where it has any (the lane-per-row form only; loaded when the slice is worked on: such slices are
  // the mesh's corners and edges on a regular mesh)
  auto load_codes = [&](int info) -> uint4v {
    const int c = info & 0x1ffffff, cls = (info >> 29) & 3;
    uint4v q = {0u, 0u, 0u, 0u};
    if (cls == SP_CLS_C16)
      q = pipe_load<NT>(reinterpret_cast<const uint4v*>(p_c16 + (int64_t)c * 512) + lane);
    else if (cls != SP_CLS_NONE)
    {
      const char* cbase = cls == SP_CLS_C8T ? reinterpret_cast<const char*>(p_svals) + ((int64_t)c * 4096 + 3584)
                                            : reinterpret_cast<const char*>(p_c16) + (int64_t)c * 1024;
      const uint2v t = pipe_load<NT>(reinterpret_cast<const uint2v*>(cbase) + lane);
      q.x = t.x, q.y = t.y;
    }
    return q;
  };

  double dot = 0.0, dot_rx = 0.0, dot_nn = 0.0;
  int regs_base = 0;
  auto step_scalars = [&](int i, int& ia, int& ib, int& sw) {
    if (i >= n_steps)
    {
      ia = ib = sw = 0;
      return;
    }
    if (i - regs_base >= 64)
    {
      regs_base = i;
      load_regs(regs_base);
      __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0) here, on the rare path (see spmv_pipe_kernel)
    }
    ia = __builtin_amdgcn_readlane(r_ia, i - regs_base);
    ib = __builtin_amdgcn_readlane(r_ib, i - regs_base);
    sw = __builtin_amdgcn_readlane(r_s, i - regs_base);
  };

  // one slice in the lane-per-row form: W gathers, W dictionary look-ups, W products; `between`: what is requested behind the gathers
  auto slice = [&](auto wtag, const int (&b)[8], const uint4v vq, int info, int s, auto&& between) {
    constexpr int W = decltype(wtag)::value;
    const int w = (info >> 25) & 15, cls = (info >> 29) & 3;
    const uint4v cq = load_codes(info);
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
    double xv[W > 0 ? W : 1], v[W > 0 ? W : 1];
#pragma unroll
    for (int e = 0; e < W; ++e)
    {
      const unsigned base = e == 0 ? (unsigned)(b[0] & 0x1fffffff) : (unsigned)b[e];
      // (a narrow slice, W = 8 > w: a slot beyond the width reads the column the packer left there, 0: a valid entry)
      xv[e] = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(p_x) + ((base + ((W == 8 && e >= w) ? 0u : cc[e])) << 3));
    }
    double xr = 0.0, rr = 0.0;
    if (DOT)
    {
      const auto u = __builtin_amdgcn_raw_buffer_load_b64(rs_x, lane8, s << 9, 0);
      xr = __hiloint2double((int)u[1], (int)u[0]);
      if (SR)
      {
        const auto t = __builtin_amdgcn_raw_buffer_load_b64(rs_r, lane8, s << 9, 0);
        rr = __hiloint2double((int)t[1], (int)t[0]);
      }
    }
#pragma unroll
    for (int e = 0; e < W; ++e)
      v[e] = dict[vc[e]];
    between();
    double sum = 0.0;
#pragma unroll
    for (int e = 0; e < W; ++e)
      if (W < 8 || e < w)
        sum += v[e] * xv[e];
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
  };
  auto slice_w = [&](const int (&b)[8], const uint4v vq, int info, int s, auto&& between) {
    if (info >= 0) // no such slice
    {
      between();
      return;
    }
    const int w = (info >> 25) & 15;
    if (w == 7)
      slice(std::integral_constant<int, 7>(), b, vq, info, s, between);
    else
      slice(std::integral_constant<int, 8>(), b, vq, info, s, between);
  };

  // one step: consume stage S (scalars ia, ib, sw), request the next step's stream into stage N
  auto body = [&](Two& S, int ia, int ib, int sw, Two& N, int ia_n, int ib_n, int sw_n) {
    const int sa = sw & 0x3fffffff;
    auto request_next = [&]() {
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_waitcnt(0xC07F); // lgkmcnt(0): the look-ups are back before scalar loads are in flight (they return out of order)
      prefetch(N, ia_n, ib_n, sw_n);
      __builtin_amdgcn_sched_barrier(0);
    };
    if (sw & 0x40000000)
    {
      // an affine pair: the lane's rows are 2 l and 2 l + 1 of the 128; slot e is the run of 128 doubles from ba[e]
      const int w = (ia >> 25) & 15;
      auto pair = [&](auto wtag, auto dtag) {
        constexpr int W = decltype(wtag)::value;
        // D >= 1: slot D is the DIAGONAL's (its run starts at the pair's first row) and slots D - 1, D, D + 1 start at
        // consecutive columns (c - 1, c, c + 1: the mesh line's own neighbours) -- the caller has checked both on the slots'
        // bases (scalars).  Two loads are saved per pair then: the middle run is the outer two's inner halves,
        // x[c + 2 l] = .y of slot D - 1, x[c + 2 l + 1] = .x of slot D + 1: not loaded; and it IS the rows' own x for
        // <x, A x>: no 16-B load of x at the rows themselves.  D = -1: every slot loaded, the rows' x too.  (Decided per slot
        // inside one body the same savings cost ~70 selects per pair: 13.4 M against 7.9 M vector instructions per product.)
        constexpr int D = decltype(dtag)::value;
        dbl2u xq[W];
#pragma unroll
        for (int e = 0; e < W; ++e)
        {
          if (e == D)
            continue;
          const unsigned base = e == 0 ? (unsigned)(S.ba[0] & 0x1fffffff) : ((W == 8 && e >= w) ? 0u : (unsigned)S.ba[e]);
          xq[e] = *reinterpret_cast<const dbl2u*>(reinterpret_cast<const char*>(p_x) + ((base << 3) + lane16));
        }
        if (D >= 1)
        {
          xq[D >= 1 ? D : 1].x = xq[D >= 1 ? D - 1 : 0].y;
          xq[D >= 1 ? D : 1].y = xq[D >= 1 ? D + 1 : 2].x;
        }
        double xr0 = 0.0, xr1 = 0.0, rr0 = 0.0, rr1 = 0.0;
        if (DOT)
        {
          if (D >= 1)
          {
            xr0 = xq[D >= 1 ? D : 1].x;
            xr1 = xq[D >= 1 ? D : 1].y;
          }
          else
          {
            const auto u = __builtin_amdgcn_raw_buffer_load_b128(rs_x, lane16, sa << 9, 0);
            xr0 = __hiloint2double((int)u[1], (int)u[0]);
            xr1 = __hiloint2double((int)u[3], (int)u[2]);
          }
          if (SR)
          {
            const auto t = __builtin_amdgcn_raw_buffer_load_b128(rs_r, lane16, sa << 9, 0);
            rr0 = __hiloint2double((int)t[1], (int)t[0]);
            rr1 = __hiloint2double((int)t[3], (int)t[2]);
          }
        }
        // the stream of the next step goes out behind the gathers; the values are looked up as the products need them (the
        // codes stay packed until then: registers)
        const uint4v c0 = S.c0, c1 = S.c1;
        __builtin_amdgcn_sched_barrier(0);
        prefetch(N, ia_n, ib_n, sw_n);
        __builtin_amdgcn_sched_barrier(0);
        auto code = [](const uint4v& c, int e) -> unsigned {
          const unsigned wd = e < 2 ? c.x : (e < 4 ? c.y : (e < 6 ? c.z : c.w));
          return (e & 1) ? wd >> 16 : wd & 0xffffu;
        };
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int e = 0; e < W; ++e)
          if (W < 8 || e < w) // (W = 8 also serves the narrow pairs: a slot beyond the width has read run 0, a valid one)
          {
            s0 += dict[code(c0, e)] * xq[e].x;
            s1 += dict[code(c1, e)] * xq[e].y;
          }
        uint4v out;
        out.x = (unsigned)__double2loint(s0), out.y = (unsigned)__double2hiint(s0);
        out.z = (unsigned)__double2loint(s1), out.w = (unsigned)__double2hiint(s1);
        __builtin_amdgcn_raw_buffer_store_b128(out, rs_y, lane16, sa << 9, 0); // (rows beyond the last: the range check)
        if (DOT)
        {
          const int r0 = (sa << 6) + 2 * lane;
          if (r0 < a.nrows)
          {
            dot += s0 * xr0;
            if (SR)
            {
              dot_rx += rr0 * xr0;
              dot_nn += a.nn_is_rr ? rr0 * rr0 : xr0 * xr0;
            }
          }
          if (r0 + 1 < a.nrows)
          {
            dot += s1 * xr1;
            if (SR)
            {
              dot_rx += rr1 * xr1;
              dot_nn += a.nn_is_rr ? rr1 * rr1 : xr1 * xr1;
            }
          }
        }
      };
      // where the diagonal's slot sits between its two line neighbours (slot 3 of a lattice row's seven or eight entries, 4 where
      // an eighth comes first): the scalar test on the bases picks the body
      const unsigned rb = (unsigned)sa << 6;
      auto tri = [&](int d) {
        const unsigned bl = d - 1 == 0 ? (unsigned)(S.ba[0] & 0x1fffffff) : (unsigned)S.ba[d - 1];
        return d + 1 < w && (unsigned)S.ba[d] == rb && bl + 1u == rb && (unsigned)S.ba[d + 1] == rb + 1u;
      };
      const int dsel = tri(3) ? 3 : (tri(4) ? 4 : -1);
      if (w == 7)
      {
        if (dsel == 3)
          pair(std::integral_constant<int, 7>(), std::integral_constant<int, 3>());
        else if (dsel == 4)
          pair(std::integral_constant<int, 7>(), std::integral_constant<int, 4>());
        else
          pair(std::integral_constant<int, 7>(), std::integral_constant<int, -1>());
      }
      else
      {
        if (dsel == 3)
          pair(std::integral_constant<int, 8>(), std::integral_constant<int, 3>());
        else if (dsel == 4)
          pair(std::integral_constant<int, 8>(), std::integral_constant<int, 4>());
        else
          pair(std::integral_constant<int, 8>(), std::integral_constant<int, -1>());
      }
    }
    else
    {
      // the two slices one after the other, a lane per row; the next step's stream goes out behind the first one's gathers
      slice_w(S.ba, S.c0, ia, sa, request_next);
      int bb[8]; // (the second slice's bases: loaded here, this path is the rare one)
      const int32_t* __restrict__ mb = p_meta + (int64_t)(ib & 0x1ffffff) * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e)
        bb[e] = mb[e];
      slice_w(bb, S.c1, ib, sa + 1, []() {});
    }
  };

  Two A, B;
  int ia_a, ib_a, sw_a, ia_b, ib_b, sw_b;
  step_scalars(0, ia_a, ib_a, sw_a);
  prefetch(A, ia_a, ib_a, sw_a);
  for (int i = 0; i < n_steps;)
  {
    step_scalars(i + 1, ia_b, ib_b, sw_b);
    body(A, ia_a, ib_a, sw_a, B, ia_b, ib_b, sw_b);
    if (++i >= n_steps)
      break;
    step_scalars(i + 1, ia_a, ib_a, sw_a);
    body(B, ia_b, ib_b, sw_b, A, ia_a, ib_a, sw_a);
    ++i;
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

// the pairs of a freshly packed stream of one-chunk slices (called by the stream's first use, behind the dictionary build)
int sellp_pairs_build(zzz_ctx* ctx)
{
  const int64_t npairs = (ctx->nslices + 1) / 2;
  ZZZ_HIP(ctx, ctx->sp_pairs.alloc((size_t)npairs + 1));
  hipLaunchKernelGGL(k_sp_pairs, dim3((unsigned)std::min<int64_t>((npairs + 255) / 256, 4096)), dim3(256), 0, ctx->stream,
                     reinterpret_cast<const int2*>(ctx->sp_desc.p), ctx->sp_smode.p, ctx->sp_meta.p, ctx->nslices, ctx->sp_pairs.p);
  ZZZ_HIP(ctx, hipGetLastError());
  return ZZZ_OK;
}

// ---- host side -----------------------------------------------------------------------------------------------------
// Does the kernel serve this context's stream, and with how many workgroups per CU?  0: the generic kernel.  (What a launch
// adds: no Chebyshev epilogue, no folded all-reduce.)
int sellp_pipe_wgs(const zzz_ctx* ctx, bool sr)
{
  if (!ctx->sellp_pipe || !ctx->sp_pipe_ok || ctx->sp_sorted || ctx->sp_chunks <= 0)
    return 0;
  if (!(ctx->sp_dict_on && ctx->sp_dict_n <= SP_DICT_LDS_ENTRIES && !ctx->sp_sd_on))
    return 0;
  if (!(ctx->sp_one_chunk && ctx->sp_pairs_ok && ctx->bs == 1 && ctx->sp_win_max == 0 && ctx->nslices < (1 << 21)))
    return 0;
  return sr ? SP_ONE_WGS_SR : SP_ONE_WGS;
}

bool launch_sellp_pipe(zzz_ctx* ctx, bool dot, bool nt, int grid, const double* x, double* y, double* partials, const int* stop,
                       const int32_t* group_list, int64_t nlist, const double* rvec, int nn_is_rr)
{
  if (!sellp_pipe_wgs(ctx, dot && rvec))
    return false;
  PipeArgs a;
  a.dict_n = ctx->sp_dict_n;
  a.nrows = (int)ctx->nrows;
  a.nslices = (int)ctx->nslices;
  a.partials = partials;
  a.stop_flag = stop;
  a.nlist = nlist;
  a.pstride = SPMV_PSTRIDE;
  a.nn_is_rr = nn_is_rr;
  const size_t lds = (size_t)((ctx->sp_dict_n + 1) & ~1) * sizeof(double);
#define ZZZ_ONE_GO(DOT, SR, NT)                                                                                                    \
  hipLaunchKernelGGL((spmv_one_kernel<DOT, SR, NT>), dim3(grid), dim3(SP_BLOCK), lds, ctx->stream,                                 \
                     reinterpret_cast<const int2*>(ctx->sp_desc.p), ctx->sp_smode.p, ctx->sp_pairs.p, ctx->sp_vals.p,               \
                     ctx->sp_codes16.p, ctx->sp_meta.p, ctx->sp_vcode.p, ctx->sp_dict.p, x, y, rvec, group_list, a)
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
ZZZ_PRELOAD_TU(sellp_pipe)
} // namespace zzz

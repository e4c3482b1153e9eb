// Conjugate gradients on the GPU: linalg::cg of src/cg.h:38-86 (variant CGH) and PETSc's KSPCG with
// PCJACOBI / PCNONE as the reference selects it at src/poisson_problem.cpp:168-177 (variant PETSC).
//
// All Krylov scalars live in device memory; the host only enqueues.  One iteration is three kernels:
//   k_update_p : sums the partials of <r,z> and of the test norm left by the previous kernel (every
//                workgroup redundantly, in the same fixed order => identical scalars), does the
//                convergence test (squared_norm + test, src/cg.h:74-79), then
//                x += alpha_{k-1} p  (the previous iteration's axpy, src/cg.h:68, applied here because p
//                is read anyway) and p = z + (beta_k / beta_{k-1}) p    (axpy, src/cg.h:82)
//   [halo]     : ghost values of p from their owners                   (scatter_fwd)
//   spmv       : w = A p, per-workgroup partials of <p, w>             (action, src/cg.h:62)
//   k_update_xr: sums the <p,w> partials -> alpha = beta_k / <p,w>     (inner_product, src/cg.h:65);
//                r -= alpha w; z = D^-1 r; partials of <r,z> and the test norm
//                                                                      (axpy, src/cg.h:71)
// With a communicator attached a one-workgroup reduce + ncclAllReduce sits between producer and
// consumer, and the consumer reads the single all-reduced value instead of the partials.
// Every kernel returns at once when the device-side `converged` flag is set; the host polls that
// flag a few iterations behind the queue, so the returned iteration count is exact.
// Reductions use fixed trees => reproducible runs.
// HBM traffic per iteration beyond the SpMV: 40 B/row (x and p update) + 40 B/row (r, z update).
#include "zzz_device.h"
#include "zzz_internal.h"
#include "zzz_cg_device.h"

#include <cmath>
#include <cstdlib>

namespace zzz
{
typedef double dbl2 __attribute__((ext_vector_type(2)));
constexpr int VB = 256;        // threads per workgroup of the vector kernels
constexpr int VGRID_MAX = 2048; // 8 workgroups per CU
// Product launches timed with HIP events when zzz_solver_opts.profile is set: every PROF_STRIDE-th iteration.  An event
// record between two kernels costs ~3.5 us of idle GPU (rocprofv3 timeline at 1.25 M rows: 4.1-4.4 us gaps on both sides
// of the product against 0.5-0.8 us elsewhere), i.e. 7 us per timed iteration -- 13 % of a 52-us iteration when every
// launch was timed.
constexpr int PROF_STRIDE = 8;

// Non-temporal access for data touched once per iteration pays only when the working set of the loop exceeds the
// 256 MiB Infinity Cache; a loop that fits (the 8-GPU per-rank size: ~100 MB of operator stream + 60 MB of vectors)
// keeps everything on-die with plain accesses.
template <bool NT, typename T>
__device__ inline T vload(const T* p)
{
  return NT ? __builtin_nontemporal_load(p) : *p;
}
template <bool NT, typename T>
__device__ inline void vstore(T v, T* p)
{
  if (NT)
    __builtin_nontemporal_store(v, p);
  else
    *p = v;
}


__global__ void k_invert(double* __restrict__ d, int64_t n)
{
  for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < n; r += (int64_t)gridDim.x * blockDim.x)
    d[r] = 1.0 / (d[r] == 0.0 ? 1.0 : d[r]); // PCJACOBI replaces zero diagonal entries by one
}

__global__ void k_extract_dinv(const rp_t* __restrict__ rowptr, const int32_t* __restrict__ cols,
                               const double* __restrict__ vals, double* __restrict__ dinv, int64_t n, int jacobi)
{
  for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < n; r += (int64_t)gridDim.x * blockDim.x)
  {
    double d = 1.0;
    if (jacobi)
    {
      // columns are ascending: binary search for the diagonal (MatGetDiagonal)
      int64_t lo = rowptr[r], hi = rowptr[r + 1] - 1;
      d = 0.0;
      while (lo <= hi)
      {
        const int64_t mid = (lo + hi) >> 1;
        const int32_t cm = cols[mid];
        if (cm == (int32_t)r)
        {
          d = vals[mid];
          break;
        }
        if (cm < (int32_t)r)
          lo = mid + 1;
        else
          hi = mid - 1;
      }
      if (d == 0.0)
        d = 1.0; // PCJACOBI replaces zero diagonal entries by one
    }
    dinv[r] = 1.0 / d;
  }
}

// ---- the inverse diagonal as 16-bit codes (DinvCodes below): distinct values of d[0..n) into a small open-addressing set
constexpr int DD_BITS = 14; // 16 384 slots for at most DZ_MAX = 2 048 values
constexpr unsigned long long DD_EMPTY = ~0ull;
__device__ inline unsigned dd_hash(unsigned long long b)
{
  b ^= b >> 29;
  b *= 0x9E3779B97F4A7C15ull;
  return (unsigned)(b >> (64 - DD_BITS));
}
// info[0] distinct values, info[1] too many (or a value with the bit pattern of the empty marker)
__global__ __launch_bounds__(256) void k_dd_insert(const double* __restrict__ d, int64_t n, unsigned long long* __restrict__ table,
                                                   int* __restrict__ info, int limit)
{
  const int lane = threadIdx.x & 63;
  const int64_t span = (n + 63) & ~(int64_t)63;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < span; i += gridDim.x * 256ll)
  {
    if (__hip_atomic_load(&info[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
      return;
    const bool have = i < n;
    const unsigned long long b = have ? (unsigned long long)__double_as_longlong(d[i]) : 0ull;
    bool need = have;
    unsigned long long todo = __ballot(need);
    while (todo) // one lane per distinct value of the wavefront goes to the table
    {
      const int src = __ffsll((long long)todo) - 1;
      const unsigned long long bb = ((unsigned long long)(unsigned)__shfl((int)(b >> 32), src) << 32) | (unsigned)__shfl((int)(unsigned)b, src);
      if (lane == src)
      {
        if (bb == DD_EMPTY)
          info[1] = 1;
        else
        {
          unsigned h = dd_hash(bb);
          for (int probe = 0; probe < (1 << DD_BITS); ++probe)
          {
            const unsigned long long cur = __hip_atomic_load(&table[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (cur == bb)
              break;
            if (cur == DD_EMPTY)
            {
              const unsigned long long old = atomicCAS(&table[h], DD_EMPTY, bb);
              if (old == DD_EMPTY)
              {
                if (atomicAdd(&info[0], 1) >= limit)
                  info[1] = 1;
                break;
              }
              if (old == bb)
                break;
            }
            h = (h + 1) & ((1u << DD_BITS) - 1);
          }
        }
      }
      need = need && b != bb;
      todo = __ballot(need);
      if (__hip_atomic_load(&info[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        return;
    }
  }
}
// codes in slot order (one workgroup); dict[code] = value
__global__ __launch_bounds__(1024) void k_dd_number(const unsigned long long* __restrict__ table, int32_t* __restrict__ slot_code,
                                                    double* __restrict__ dict, int* __restrict__ info)
{
  __shared__ int wsum[16];
  if (info[1])
    return;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int mine = 0;
  for (int k = threadIdx.x; k < (1 << DD_BITS); k += 1024)
    mine += table[k] != DD_EMPTY ? 1 : 0;
  int incl = mine;
  for (int dd = 1; dd < 64; dd <<= 1)
  {
    const int t = __shfl_up(incl, dd);
    if (lane >= dd)
      incl += t;
  }
  if (lane == 63)
    wsum[wv] = incl;
  __syncthreads();
  int off = 0;
  for (int q = 0; q < wv; ++q)
    off += wsum[q];
  int code = off + incl - mine;
  for (int k = threadIdx.x; k < (1 << DD_BITS); k += 1024)
  {
    const unsigned long long b = table[k];
    if (b != DD_EMPTY)
    {
      slot_code[k] = code;
      dict[code] = __longlong_as_double((long long)b);
      ++code;
    }
  }
  if (threadIdx.x == 1023)
    info[2] = code;
}
__global__ __launch_bounds__(256) void k_dd_encode(const double* __restrict__ d, int64_t n, int64_t npad,
                                                   const unsigned long long* __restrict__ table, const int32_t* __restrict__ slot_code,
                                                   uint16_t* __restrict__ codes, const int* __restrict__ info)
{
  if (info[1])
    return;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < npad; i += gridDim.x * 256ll)
  {
    uint16_t c = 0;
    if (i < n)
    {
      const unsigned long long b = (unsigned long long)__double_as_longlong(d[i]);
      unsigned h = dd_hash(b);
      while (table[h] != b)
        h = (h + 1) & ((1u << DD_BITS) - 1);
      c = (uint16_t)slot_code[h];
    }
    codes[i] = c;
  }
}

// r = b - w (w = A x0, or nothing when x0 == 0); z = dinv r; partials: pa = <r,z>, pb = test norm^2
__global__ __launch_bounds__(VB) void k_init_residual(const double* __restrict__ b, const double* __restrict__ w,
                                                      const double* __restrict__ dinv, double* __restrict__ r,
                                                      double* __restrict__ z, int64_t n, int norm,
                                                      double* __restrict__ pa, double* __restrict__ pb)
{
  __shared__ double sh[VB / 64];
  double sa = 0, sb = 0;
  for (int64_t i = blockIdx.x * (int64_t)VB + threadIdx.x; i < n; i += (int64_t)gridDim.x * VB)
  {
    const double ri = w ? (-1.0 * w[i] + b[i]) : b[i]; // axpy(r, -1, y, b), src/cg.h:47
    const double zi = dinv[i] * ri;
    r[i] = ri;
    z[i] = zi;
    sa += ri * zi;
    sb += (norm == ZZZ_NORM_UNPRECONDITIONED) ? ri * ri : zi * zi;
  }
  const double ta = block_reduce_sum(sa, sh);
  const double tb = block_reduce_sum(sb, sh);
  if (threadIdx.x == 0)
  {
    pa[blockIdx.x] = ta;
    pb[blockIdx.x] = tb;
  }
}

// The convergence flag can be set by workgroup 0 of the SAME launch while other workgroups start: one value per
// workgroup (thread 0's, loaded by the caller ahead of its other requests), so all threads of a workgroup take the same
// branch -- the reductions behind it need every wavefront.
__device__ inline int block_flag(int f)
{
  __shared__ int flag;
  if (threadIdx.x == 0)
    flag = f;
  __syncthreads();
  return flag;
}

// `it` = number of completed iterations.  pa/pb: partials of <r,z> and of the test norm (np each).
// The solution update of the PREVIOUS iteration, x += alpha_{it-1} p_{it-1}, is applied here (p is read
// anyway) instead of in k_update_xr: one vector read less per iteration, same operations on the same
// operands, so x is bit-identical.  It must be applied by the launch that detects convergence too; only
// launches enqueued after that one skip it (conv_it1).  update_dir == 0: the final test after max_it.
// DZ (round 4): no z vector.  Jacobi's inverse diagonal holds few distinct values on a regular mesh (a subset of the
// matrix's: section 3 of DESIGN.md); as 16-bit codes into a table in LDS it costs 2 B per row instead of 8, and with it
// z = D^-1 r is cheaper to RECOMPUTE here from r (the same product of the same two doubles: the same bits) than to
// write in k_update_xr and read back: 80 -> 68 B per row and iteration for the two kernels.  dz.codes = nullptr: off.
struct DinvCodes
{
  const uint32_t* codes; // two 16-bit codes per word, entry pairs as the dbl2 accesses take them
  const double* dict;
  const double* r;
  int ndict;
};
constexpr int DZ_MAX = 2048;

template <bool NT, bool DZ = false>
__global__ __launch_bounds__(VB) void k_update_p(CgState* __restrict__ st, double* __restrict__ beta_hist,
                                                 double* __restrict__ dp_hist, const double* __restrict__ alpha_hist,
                                                 int it, CgParams P, const double* __restrict__ pa,
                                                 const double* __restrict__ pb, int np, const double* __restrict__ z,
                                                 double* __restrict__ p, double* __restrict__ x, int64_t n,
                                                 int update_dir, DinvCodes dz = DinvCodes())
{
  __shared__ double sh[VB / 64];
  __shared__ double dtab[DZ ? DZ_MAX : 1];
  if (DZ)
  {
    for (int k = threadIdx.x; k < dz.ndict; k += VB)
      dtab[k] = dz.dict[k];
    __syncthreads();
  }
  // The first entries of this thread are requested BEFORE the scalar prologue (flag, partial sums, convergence logic: a
  // chain of dependent loads and barriers of 3-5 us that every workgroup walks): at the 8-GPU per-rank size a thread
  // has one or two entries in all, so the kernel was prologue + one memory round trip in sequence (11-12 us for 50 MB
  // that stream in 7).  Clamped index: the loads are unconditional, a thread without entries drops them.
  const int64_t n2 = n >> 1, stride = (int64_t)gridDim.x * VB;
  const int64_t i0 = blockIdx.x * (int64_t)VB + threadIdx.x;
  dbl2* __restrict__ p2 = reinterpret_cast<dbl2*>(p);
  dbl2* __restrict__ x2 = reinterpret_cast<dbl2*>(x);
  const dbl2* __restrict__ z2 = reinterpret_cast<const dbl2*>(DZ ? dz.r : z);
  const int64_t c0 = (i0 < n2) ? i0 : 0;
  dbl2 pi0 = {0, 0}, xi0 = {0, 0}, zi0 = {0, 0};
  uint32_t dc0 = 0;
  if (n2 > 0)
  {
    pi0 = p2[c0];
    xi0 = vload<NT>(x2 + c0);
    zi0 = vload<NT>(z2 + c0); // DZ: r
    if (DZ)
      dc0 = dz.codes[c0];
  }
  const double alpha_h = it > 0 ? alpha_hist[it - 1] : 0.0; // requested with the rest of the prologue's inputs
  DirScalars S;
  if (!cg_direction_scalars(st, beta_hist, dp_hist, it, P, pa, pb, np, sh, S))
    return;
  const int conv = S.conv;
  const double rz = S.rz, bprev = S.bprev;
  const bool dir = !conv && update_dir;
  if (it == 0)
  {
    if (dir)
      for (int64_t i = blockIdx.x * (int64_t)VB + threadIdx.x; i < n; i += (int64_t)gridDim.x * VB)
        p[i] = DZ ? dtab[reinterpret_cast<const uint16_t*>(dz.codes)[i]] * dz.r[i] : z[i];
    return;
  }
  const double alpha = alpha_h;
  const double bcoef = rz / bprev;
  // two entries per lane and load (16-B accesses); the arithmetic per entry is unchanged
  for (int64_t i = i0; i < n2; i += stride)
  {
    dbl2 pi, xi, zi;
    if (i == i0)
    {
      pi = pi0;
      xi = xi0; // x is touched once per iteration: keep the cache for p, z, w
      zi = zi0; // last use of z
    }
    else
    {
      pi = p2[i];
      xi = vload<NT>(x2 + i);
      zi = vload<NT>(z2 + i);
    }
    if (DZ)
    {
      const uint32_t dc = i == i0 ? dc0 : dz.codes[i];
      zi.x = dtab[dc & 0xffffu] * zi.x; // z = D^-1 r, as k_update_xr formed it for its sums
      zi.y = dtab[dc >> 16] * zi.y;
    }
    xi.x = alpha * pi.x + xi.x; // src/cg.h:68, one kernel late
    xi.y = alpha * pi.y + xi.y;
    vstore<NT>(xi, x2 + i);
    if (dir)
    {
      dbl2 pn;
      pn.x = bcoef * pi.x + zi.x;
      pn.y = bcoef * pi.y + zi.y;
      p2[i] = pn;
    }
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0)
  {
    const int64_t i = n - 1;
    const double pi = p[i];
    x[i] = alpha * pi + x[i];
    if (dir)
      p[i] = bcoef * pi + (DZ ? dtab[reinterpret_cast<const uint16_t*>(dz.codes)[i]] * dz.r[i] : z[i]);
  }
}

template <bool NT, bool DZ = false>
__global__ __launch_bounds__(VB) void k_update_xr(CgState* __restrict__ st, const double* __restrict__ beta_hist,
                                                  double* __restrict__ alpha_hist, int it,
                                                  const double* __restrict__ pw_parts, int npw,
                                                  const double* __restrict__ w, const double* __restrict__ dinv,
                                                  double* __restrict__ r, double* __restrict__ z, int64_t n, int norm,
                                                  double* __restrict__ pa, double* __restrict__ pb, int variant,
                                                  TailArgs tail, DinvCodes dz = DinvCodes())
{
  __shared__ double dtab[DZ ? DZ_MAX : 1];
  if (DZ)
  {
    for (int k = threadIdx.x; k < dz.ndict; k += VB)
      dtab[k] = dz.dict[k];
    __syncthreads();
  }
  // first entries requested before the scalar prologue (see k_update_p)
  const int64_t n2 = n >> 1, stride = (int64_t)gridDim.x * VB;
  const int64_t i0 = blockIdx.x * (int64_t)VB + threadIdx.x;
  const dbl2* __restrict__ w2 = reinterpret_cast<const dbl2*>(w);
  const dbl2* __restrict__ d2 = reinterpret_cast<const dbl2*>(dinv);
  dbl2* __restrict__ r2 = reinterpret_cast<dbl2*>(r);
  dbl2* __restrict__ z2 = reinterpret_cast<dbl2*>(z);
  const int64_t c0 = (i0 < n2) ? i0 : 0;
  dbl2 wi0 = {0, 0}, di0 = {0, 0}, ri0 = {0, 0};
  uint32_t dc0 = 0;
  // (the scalar inputs of the prologue are requested in the same breath, below: flag, beta, <p,w>)
  if (n2 > 0)
  {
    wi0 = vload<NT>(w2 + c0); // last use of w
    if (DZ)
      dc0 = dz.codes[c0];
    else
      di0 = vload<NT>(d2 + c0);
    ri0 = vload<NT>(r2 + c0); // r and D^-1 are touched once per iteration
  }
  const int f0 = st->converged;
  const double beta_it = beta_hist[it];
  const double pw1 = pw_parts[0]; // the all-reduced value itself when a communicator is attached (npw == 1)
  if (block_flag(f0))
    return;
  __shared__ double sh[VB / 64];
  const double pw = npw == 1 ? pw1 : reduce_parts_bcast(pw_parts, npw, sh);
  const double alpha = beta_it / pw; // src/cg.h:65
  // KSPCG stops on a non-finite scalar (KSP_DIVERGED_NANORINF / _BREAKDOWN); linalg::cg has no such guard: with
  // rnorm0 == 0 its alpha is 0/0, every comparison with NaN is false and the loop runs kmax times (src/cg.h:58-83)
  if (variant != ZZZ_CG_CGH && !isfinite(alpha))
  {
    if (blockIdx.x == 0 && threadIdx.x == 0)
    {
      st->iters = it;
      st->converged = 2;
      __atomic_store_n(&st->conv_it1, it + 1, __ATOMIC_RELAXED);
    }
    return; // every workgroup sees the same non-finite alpha
  }
  if (blockIdx.x == 0 && threadIdx.x == 0)
    alpha_hist[it] = alpha; // x += alpha p: applied by k_update_p(it + 1)
  double sa = 0, sb = 0;
  for (int64_t i = i0; i < n2; i += stride)
  {
    dbl2 wi, di, ri, zi;
    if (i == i0)
    {
      wi = wi0;
      di = di0;
      ri = ri0;
    }
    else
    {
      wi = vload<NT>(w2 + i);
      if (!DZ)
        di = vload<NT>(d2 + i);
      ri = vload<NT>(r2 + i);
    }
    if (DZ)
    {
      const uint32_t dc = i == i0 ? dc0 : dz.codes[i];
      di.x = dtab[dc & 0xffffu];
      di.y = dtab[dc >> 16];
    }
    ri.x = -alpha * wi.x + ri.x; // src/cg.h:71
    ri.y = -alpha * wi.y + ri.y;
    zi.x = di.x * ri.x;
    zi.y = di.y * ri.y;
    if (DZ)
      r2[i] = ri; // (read again by k_update_p: left in the cache)
    else
    {
      vstore<NT>(ri, r2 + i);
      z2[i] = zi;
    }
    sa += ri.x * zi.x;
    sa += ri.y * zi.y;
    if (norm == ZZZ_NORM_UNPRECONDITIONED)
    {
      sb += ri.x * ri.x;
      sb += ri.y * ri.y;
    }
    else
    {
      sb += zi.x * zi.x;
      sb += zi.y * zi.y;
    }
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0)
  {
    const int64_t i = n - 1;
    const double ri = -alpha * w[i] + r[i];
    const double zi = (DZ ? dtab[reinterpret_cast<const uint16_t*>(dz.codes)[i]] : dinv[i]) * ri;
    r[i] = ri;
    if (!DZ)
      z[i] = zi;
    sa += ri * zi;
    sb += (norm == ZZZ_NORM_UNPRECONDITIONED) ? ri * ri : zi * zi;
  }
  const double ta = block_reduce_sum(sa, sh);
  const double tb = block_reduce_sum(sb, sh);
  if (tail.parts) // multi-GPU: <r,z> and the norm are all-reduced in the tail of this launch (zzz_tail.h)
  {
    tail_arrive(tail, ta, tb, 0.0);
    return;
  }
  if (threadIdx.x == 0)
  {
    pa[blockIdx.x] = ta;
    pb[blockIdx.x] = tb;
  }
}

// ---- PETSc's -ksp_cg_single_reduction (KSPCGUseSingleReduction; Chronopoulos/Gear form) ----------
// One iteration = this kernel + one SpMV (s = A z with the partials of <z,s>, <r,z> and the norm):
//   b = beta_k/beta_{k-1};  <p,w>_k = <z,s> - beta_k^2 <p,w>_{k-1} / beta_{k-1}^2;  a = beta_k / <p,w>_k
//   p = z + b p;  w = s + b w  (= A p by recurrence);  x += a p;  r -= a w;  z = D^-1 r
// so a whole iteration has ONE reduction point (after the SpMV) instead of two.
// pa/pb/pc: partials (or the single all-reduced values) of <r,z>, the test norm^2 and <z,s>.
// CHEB (Chebyshev-Jacobi preconditioner, cg_solve_single_reduction): z = D^-1 r becomes the polynomial's first term --
// g = D^-1 r, d = g / theta, z = d (k_cheb_xr's tail) -- and its further terms follow as product epilogues.
// DZ (round 5): as in k_update_p -- Jacobi's inverse diagonal as 16-bit codes into a table in LDS, and z = D^-1 r of the
// LAST iteration recomputed from the r this kernel reads anyway (the same product of the same two doubles: the same bits)
// instead of read back: 96 -> 82 B per row and iteration.  z is still written: the product gathers it.
template <bool NT, bool CHEB = false, bool DZ = false>
__global__ __launch_bounds__(VB) void k_sr_update(CgState* __restrict__ st, double* __restrict__ beta_hist,
                                                  double* __restrict__ dpi_hist, double* __restrict__ dp_hist, int it,
                                                  CgParams P, const double* __restrict__ pa, const double* __restrict__ pb,
                                                  const double* __restrict__ pc, int np, const double* __restrict__ dinv,
                                                  const double* __restrict__ s, double* __restrict__ z, double* __restrict__ p,
                                                  double* __restrict__ w, double* __restrict__ x, double* __restrict__ r,
                                                  int64_t n, int scalars_only, double theta = 0.0, double* __restrict__ chg = nullptr,
                                                  double* __restrict__ chd = nullptr, DinvCodes dz = DinvCodes())
{
  static_assert(!(CHEB && DZ), "the polynomial's first term is not D^-1 r alone");
  __shared__ double dtab[DZ ? DZ_MAX : 1];
  if (DZ)
  {
    for (int k = threadIdx.x; k < dz.ndict; k += VB)
      dtab[k] = dz.dict[k];
    __syncthreads();
  }
  // first entries requested before the scalar prologue (see k_update_p): seven 16-B loads in flight per thread while
  // the workgroup walks the flag, the three partial sums and the convergence logic
  const int64_t n2 = n >> 1, stride = (int64_t)gridDim.x * VB;
  const int64_t i0 = blockIdx.x * (int64_t)VB + threadIdx.x;
  const dbl2* __restrict__ s2 = reinterpret_cast<const dbl2*>(s);
  const dbl2* __restrict__ d2 = reinterpret_cast<const dbl2*>(dinv);
  dbl2 *__restrict__ z2 = reinterpret_cast<dbl2*>(z), *__restrict__ p2 = reinterpret_cast<dbl2*>(p),
                     *__restrict__ w2 = reinterpret_cast<dbl2*>(w), *__restrict__ x2 = reinterpret_cast<dbl2*>(x),
                     *__restrict__ r2 = reinterpret_cast<dbl2*>(r);
  const int64_t c0 = (i0 < n2) ? i0 : 0;
  dbl2 zi0 = {0, 0}, si0 = {0, 0}, di0 = {0, 0}, xi0 = {0, 0}, ri0 = {0, 0}, po0 = {0, 0}, wo0 = {0, 0};
  uint32_t dc0 = 0;
  if (n2 > 0 && !scalars_only)
  {
    if (DZ)
      dc0 = dz.codes[c0];
    else
    {
      zi0 = z2[c0];
      di0 = vload<NT>(d2 + c0);
    }
    si0 = vload<NT>(s2 + c0);
    xi0 = vload<NT>(x2 + c0);
    ri0 = vload<NT>(r2 + c0);
    po0 = vload<NT>(p2 + c0); // zero before the first iteration (cg_solve_single_reduction clears p and w)
    wo0 = vload<NT>(w2 + c0);
  }
  // ... and so are the scalar inputs: flag, tolerances, last iteration's coefficients, and (communicator attached:
  // np == 1) the three all-reduced sums themselves -- that prologue then has no barrier and one memory round trip
  const int f0 = st->converged;
  const double ttol_st = st->ttol, dp0_st = st->dp0;
  const double bo_h = it > 0 ? beta_hist[it - 1] : 1.0, dpi_h = it > 0 ? dpi_hist[it - 1] : 0.0;
  double rz = pa[0], nn = pb[0], zs = pc[0];
  if (np == 1)
  {
    // every thread holds the same flag unless workgroup 0 of THIS launch is just setting it -- and then every
    // workgroup reaches the same verdict from the same scalars and leaves below: no wavefront updates a vector
    if (f0)
      return;
  }
  else
  {
    if (block_flag(f0))
      return;
    reduce_parts3_bcast(pa, pb, pc, np, rz, nn, zs);
  }
  const double dp = (P.norm == ZZZ_NORM_NATURAL) ? sqrt(fabs(rz)) : sqrt(nn);
  double ttol = ttol_st;
  if (it == 0)
    ttol = fmax(P.rtol * dp, P.atol);
  int conv = 0;
  if (!isfinite(dp))
    conv = 2;
  else if (dp <= ttol) // KSPConvergedDefault
    conv = 1;
  else if (dp >= P.dtol * (it == 0 ? dp : dp0_st)) // ... KSP_DIVERGED_DTOL
    conv = 3;
  double b = 0.0, dpi = zs;
  if (it > 0)
  {
    const double bo = bo_h;
    b = rz / bo;
    dpi = zs - rz * rz * dpi_h / (bo * bo);
  }
  const double a = rz / dpi;
  if (!conv && !scalars_only && !isfinite(a))
    conv = 2;
  if (blockIdx.x == 0 && threadIdx.x == 0)
  {
    beta_hist[it] = rz;
    dpi_hist[it] = dpi;
    dp_hist[it] = dp;
    st->dp = dp;
    if (it == 0)
    {
      st->dp0 = dp;
      st->ttol = ttol;
    }
    if (conv)
    {
      st->iters = it;
      st->converged = conv;
    }
  }
  if (conv || scalars_only)
    return;
  auto one = [&](int64_t i) {
    const double dv = DZ ? dtab[reinterpret_cast<const uint16_t*>(dz.codes)[i]] : dinv[i];
    const double zo = DZ ? dv * r[i] : z[i];
    const double pn = (it == 0) ? zo : b * p[i] + zo;
    const double wn = (it == 0) ? s[i] : b * w[i] + s[i];
    p[i] = pn;
    w[i] = wn;
    x[i] = a * pn + x[i];
    const double ri = -a * wn + r[i];
    r[i] = ri;
    const double zi = dv * ri;
    if (CHEB)
    {
      chg[i] = zi;
      chd[i] = zi / theta;
      z[i] = zi / theta;
    }
    else
      z[i] = zi;
  };
  for (int64_t i = i0; i < n2; i += stride)
  {
    // cache policy: only z (gathered by the next SpMV) and s (its output) are worth keeping; p, w, x, r, D^-1
    // are touched by this kernel alone, once per iteration
    dbl2 zi, si, di, xi, ri, po, wo;
    uint32_t dc = dc0;
    if (i == i0)
    {
      zi = zi0, si = si0, di = di0, xi = xi0, ri = ri0, po = po0, wo = wo0;
    }
    else
    {
      if (DZ)
        dc = dz.codes[i];
      else
        zi = z2[i], di = vload<NT>(d2 + i);
      si = vload<NT>(s2 + i), xi = vload<NT>(x2 + i), ri = vload<NT>(r2 + i);
      if (it != 0)
        po = vload<NT>(p2 + i), wo = vload<NT>(w2 + i);
    }
    if (DZ)
    {
      di.x = dtab[dc & 0xffffu];
      di.y = dtab[dc >> 16];
      zi.x = di.x * ri.x; // z = D^-1 r as the last iteration (or k_init_residual) formed it
      zi.y = di.y * ri.y;
    }
    dbl2 pn = zi, wn = si, zn;
    if (it != 0)
    {
      pn.x = b * po.x + zi.x;
      pn.y = b * po.y + zi.y;
      wn.x = b * wo.x + si.x;
      wn.y = b * wo.y + si.y;
    }
    xi.x = a * pn.x + xi.x;
    xi.y = a * pn.y + xi.y;
    ri.x = -a * wn.x + ri.x;
    ri.y = -a * wn.y + ri.y;
    zn.x = di.x * ri.x;
    zn.y = di.y * ri.y;
    vstore<NT>(pn, p2 + i);
    vstore<NT>(wn, w2 + i);
    vstore<NT>(xi, x2 + i);
    vstore<NT>(ri, r2 + i);
    if (CHEB)
    {
      dbl2 dn;
      dn.x = zn.x / theta;
      dn.y = zn.y / theta;
      reinterpret_cast<dbl2*>(chg)[i] = zn;
      reinterpret_cast<dbl2*>(chd)[i] = dn;
      z2[i] = dn;
    }
    else
      z2[i] = zn;
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0)
    one(n - 1);
}

__device__ inline double reduce_parts(const double* __restrict__ parts, int np, double* sh)
{
  double s = 0;
  for (int i = threadIdx.x; i < np; i += blockDim.x)
    s += parts[i];
  return block_reduce_sum(s, sh);
}

__global__ __launch_bounds__(VB) void k_sqnorm(const double* __restrict__ v, int64_t n, double* __restrict__ parts)
{
  __shared__ double sh[VB / 64];
  double s = 0;
  for (int64_t i = blockIdx.x * (int64_t)VB + threadIdx.x; i < n; i += (int64_t)gridDim.x * VB)
    s += v[i] * v[i];
  const double t = block_reduce_sum(s, sh);
  if (threadIdx.x == 0)
    parts[blockIdx.x] = t;
}
__global__ __launch_bounds__(1024) void k_reduce_plain(const double* __restrict__ parts, int np, double* out)
{
  __shared__ double sh[16];
  const double s = reduce_parts(parts, np, sh);
  if (threadIdx.x == 0)
    out[0] = s;
}

// the polling events of one solve: destroyed on every exit path
template <int N>
struct EventRing
{
  hipEvent_t ev[N] = {};
  int created = 0;
  hipError_t create()
  {
    for (; created < N; ++created)
    {
      hipError_t e = hipEventCreateWithFlags(&ev[created], hipEventDisableTiming);
      if (e != hipSuccess)
        return e;
    }
    return hipSuccess;
  }
  ~EventRing()
  {
    for (int i = 0; i < created; ++i)
      (void)hipEventDestroy(ev[i]);
  }
  hipEvent_t& operator[](int i) { return ev[i]; }
};

// The inverse diagonal (ctx->dinv, n entries) as 16-bit codes into a table of its distinct values, for the kernels that take
// DinvCodes: dzc.codes stays null when there are more than DZ_MAX values.  Synchronises the stream once (the count).
static int dinv_codes_build(zzz_ctx* ctx, int64_t n, DinvCodes& dzc)
{
  hipStream_t s = ctx->stream;
    const int64_t npad = (n + 1) & ~(int64_t)1;
    ZZZ_HIP(ctx, ctx->dd_table.reserve((size_t)1 << DD_BITS));
    ZZZ_HIP(ctx, ctx->dd_slot.reserve((size_t)1 << DD_BITS));
    ZZZ_HIP(ctx, ctx->dd_dict.reserve((size_t)DZ_MAX));
    ZZZ_HIP(ctx, ctx->dd_codes.reserve((size_t)npad));
    ZZZ_HIP(ctx, ctx->dd_info.reserve(8));
    ZZZ_HIP(ctx, hipMemsetAsync(ctx->dd_info.p, 0, 8 * sizeof(int32_t), s));
    ZZZ_HIP(ctx, hipMemsetAsync(ctx->dd_table.p, 0xff, sizeof(unsigned long long) << DD_BITS, s));
    const unsigned gd = (unsigned)std::min<int64_t>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(k_dd_insert, dim3(gd), dim3(256), 0, s, ctx->dinv.p, n, ctx->dd_table.p, ctx->dd_info.p, DZ_MAX);
    hipLaunchKernelGGL(k_dd_number, dim3(1), dim3(1024), 0, s, ctx->dd_table.p, ctx->dd_slot.p, ctx->dd_dict.p, ctx->dd_info.p);
    hipLaunchKernelGGL(k_dd_encode, dim3(gd), dim3(256), 0, s, ctx->dinv.p, n, npad, ctx->dd_table.p, ctx->dd_slot.p, ctx->dd_codes.p,
                       ctx->dd_info.p);
    int32_t h[4] = {0, 1, 0, 0};
    ZZZ_HIP(ctx, hipMemcpyAsync(h, ctx->dd_info.p, sizeof(h), hipMemcpyDeviceToHost, s));
    ZZZ_HIP(ctx, hipStreamSynchronize(s));
    if (!h[1] && h[2] > 0 && h[2] <= DZ_MAX)
      dzc = DinvCodes{reinterpret_cast<const uint32_t*>(ctx->dd_codes.p), ctx->dd_dict.p, ctx->r.p, h[2]};
    return ZZZ_OK;
}

// bytes one CG iteration touches (operator + `nvec` vectors) against the Infinity Cache
static bool loop_exceeds_cache(zzz_ctx* ctx, int nvec)
{
  const double op = sellp_active(ctx) ? (double)sellp_stream_bytes(ctx) : 10.0 * (double)ctx->nnz;
  return op + 8.0 * nvec * (double)ctx->nloc() > 200.0e6;
}

static int vgrid(int64_t n)
{
  // at least 8 entries per thread: every workgroup starts by summing the producer's per-workgroup partials, so for
  // small vectors fewer, longer workgroups are faster (1.25 M rows: 57.8 -> 52.9 us per iteration with 610 instead of
  // 2048 workgroups, 0.5 M rows 40.8 -> 37.0 us; 16 per thread the same, 32 slower); large vectors keep 8 per CU
#ifdef ZZZ_EXPERIMENTS
  static const int per = [] {
    const char* e = getenv("ZZZ_VGRID_PER"); // measurement knob (entries per thread), tools build only
    const int v = e ? atoi(e) : 0;
    return v >= 1 && v <= 64 ? v : 8;
  }();
#else
  constexpr int per = 8;
#endif
  int64_t g = (n + VB * per - 1) / (VB * per);
  if (g > VGRID_MAX)
    g = VGRID_MAX;
  if (g < 1)
    g = 1;
  return (int)g;
}

// sum of squares over the owned entries, all-reduced over ranks when a communicator is attached
int vec_norm_local(zzz_ctx* ctx, const double* v, int64_t n, double* out)
{
  const int g = vgrid(n);
  hipLaunchKernelGGL(k_sqnorm, dim3(g), dim3(VB), 0, ctx->stream, v, n, ctx->part_b.p);
  if (ctx->comm)
  {
    int rc = comm_reduce_allreduce(ctx, nullptr, ctx->part_b.p, nullptr, nullptr, g, 1, ctx->red.p);
    if (rc)
      return rc;
  }
  else
    hipLaunchKernelGGL(k_reduce_plain, dim3(1), dim3(1024), 0, ctx->stream, ctx->part_b.p, g, ctx->red.p);
  ZZZ_HIP(ctx, hipGetLastError());
  double s = 0;
  ZZZ_HIP(ctx, hipMemcpyAsync(&s, ctx->red.p, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  ZZZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  *out = std::sqrt(s);
  return comm_p2p_check(ctx);
}

static int cg_solve_single_reduction(zzz_ctx* ctx, const zzz_solver_opts* o, int* iters, double* rnorm);

// How the solve ended, in KSPConvergedReason's numbering [EXT: petscksp.h]: the reference's solver_function returns
// solver.solve()'s iteration count whatever the reason (src/poisson_problem.cpp:172-178) and the driver prints its summary
// and timings all the same, so a diverged solve is NOT an error here either -- unless the caller asks for PETSc's
// -ksp_error_if_not_converged (zzz_solver_opts.error_if_not_converged).
static int finish_reason(zzz_ctx* ctx, const zzz_solver_opts* o, const CgState& fin, int its)
{
  int reason;
  if (fin.converged == 1)
    reason = (o->variant == ZZZ_CG_PETSC && fin.dp < o->atol) ? 3 /* KSP_CONVERGED_ATOL */ : 2 /* KSP_CONVERGED_RTOL */;
  else if (fin.converged == 2)
    reason = -9; // KSP_DIVERGED_NANORINF
  else if (fin.converged == 3)
    reason = -4; // KSP_DIVERGED_DTOL
  else
    reason = -3; // KSP_DIVERGED_ITS (linalg::cg, src/cg.h:58-83, simply returns kmax)
  ctx->last_reason = reason;
  if (o->error_if_not_converged && reason < 0)
  {
    if (reason == -9)
      return fail(ctx, ZZZ_ERR_DIVERGED, "KSP_DIVERGED_NANORINF: CG broke down, non-finite scalar at iteration %d", its);
    if (reason == -4)
      return fail(ctx, ZZZ_ERR_DIVERGED, "KSP_DIVERGED_DTOL: norm %g >= divtol x initial norm %g at iteration %d", fin.dp,
                  fin.dp0, its);
    return fail(ctx, ZZZ_ERR_DIVERGED, "KSP_DIVERGED_ITS: %d iterations without reaching the tolerance (norm %g)", its, fin.dp);
  }
  return ZZZ_OK;
}


static int cg_solve_chebyshev(zzz_ctx* ctx, const zzz_solver_opts* o, int* iters, double* rnorm);

int cg_solve(zzz_ctx* ctx, const zzz_solver_opts* o, int* iters, double* rnorm)
{
  ctx->last_pc_bound = 0.0;
  if (o->pc == ZZZ_PC_CHEBYSHEV_JACOBI && !o->single_reduction)
    return cg_solve_chebyshev(ctx, o, iters, rnorm);
  if (o->single_reduction)
    return cg_solve_single_reduction(ctx, o, iters, rnorm);
  const int64_t n = ctx->n_owned * ctx->bs; // owned scalar rows
  const int max_it = o->max_it;
  CgParams P{o->variant, o->pc, o->norm, o->rtol, o->atol, o->dtol > 0.0 ? o->dtol : 1.0e4};
  const bool multi = ctx->comm != nullptr;
  const int g = vgrid(n);
  hipStream_t s = ctx->stream;
  const bool nt = loop_exceeds_cache(ctx, 6);
  // (kernels by load policy and by whether z is recomputed from the coded inverse diagonal: chosen where dz is known)
  auto pick_update_p = [&](bool dzf) { return dzf ? (nt ? k_update_p<true, true> : k_update_p<false, true>) : (nt ? k_update_p<true, false> : k_update_p<false, false>); };
  auto pick_update_xr = [&](bool dzf) { return dzf ? (nt ? k_update_xr<true, true> : k_update_xr<false, true>) : (nt ? k_update_xr<true, false> : k_update_xr<false, false>); };
  // A/B knob ZZZ_CG_FUSED=2: two kernels per iteration (product fused with the direction update, zzz_sellp.hip).
  // Bit-identical, but MEASURED slower wherever it was tried: at the 8-GPU per-rank size (1.25 M rows, loop resident
  // in the Infinity Cache) the fused kernel takes 39-41 us against 14.7 + 19.6 us for k_update_p + product (the
  // second gather and the row updates lengthen every wavefront's dependent chain, and at that size the product is
  // latency-bound: ~2.4 slices per wavefront); for HBM-sized loops it saves 4 % of the bytes at best.  Off by default.
  int fused_mode = 0;
#ifdef ZZZ_EXPERIMENTS
  if (const char* e = getenv("ZZZ_CG_FUSED"))
    fused_mode = atoi(e);
#endif
  const bool fused = o->op == ZZZ_OP_CSR && fused_mode == 2 && sellp_active(ctx) && ctx->sp_win_max == 0; // (its kernel gathers from memory)
  ctx->last_solve_fused = fused;
  // multi-GPU: the scalar all-reduce rides in the tail of the product launch when that launch is the operator stream's
  const bool fold_product = multi && !fused && o->op == ZZZ_OP_CSR && sellp_active(ctx);
  if (fused)
    ZZZ_HIP(ctx, ctx->p_alt.alloc((size_t)ctx->nloc()));
  double* pbuf[2] = {ctx->p.p, fused ? ctx->p_alt.p : ctx->p.p};

  ZZZ_HIP(ctx, ctx->beta_hist.reserve((size_t)max_it + 2));
  ZZZ_HIP(ctx, ctx->dp_hist.reserve((size_t)max_it + 2));
  ZZZ_HIP(ctx, ctx->alpha_hist.reserve((size_t)max_it + 2));
  ZZZ_HIP(ctx, hipMemsetAsync(ctx->state.p, 0, sizeof(CgState), s));
  ZZZ_HIP(ctx, hipMemsetAsync(ctx->p.p, 0, sizeof(double) * ctx->p.n, s));

  // PCSetUp(PCJACOBI): inverse diagonal (inside `ZZZ Solve`, as KSPSetUp is in the reference)
  if (o->op == ZZZ_OP_CSR)
    hipLaunchKernelGGL(k_extract_dinv, dim3(g), dim3(VB), 0, s, ctx->rowptr.p, ctx->cols.p, ctx->vals.p, ctx->dinv.p, n,
                       o->pc == ZZZ_PC_JACOBI ? 1 : 0);
  else if (o->pc == ZZZ_PC_JACOBI)
  {
    // the operator is never assembled: its diagonal from the element matrices (zzz_matfree.hip), inverted in place
    if (int rc = launch_matfree_diagonal(ctx, ctx->dinv.p))
      return rc;
    hipLaunchKernelGGL(k_invert, dim3(g), dim3(VB), 0, s, ctx->dinv.p, n);
  }
  else
    hipLaunchKernelGGL(k_extract_dinv, dim3(g), dim3(VB), 0, s, (const rp_t*)nullptr, (const int32_t*)nullptr,
                       (const double*)nullptr, ctx->dinv.p, n, 0);

  // Jacobi's inverse diagonal as 16-bit codes and no z vector (DinvCodes, k_update_p): KSPCG + PCJACOBI, a loop too large for
  // the Infinity Cache (where bytes are what the vector kernels wait for: the criterion of their load policy;
  // ZZZ_CG_DINV_CODES: 0 never, 2 at any size), at most 2 048 values
  DinvCodes dzc{nullptr, nullptr, nullptr, 0};
  if (o->variant == ZZZ_CG_PETSC && o->pc == ZZZ_PC_JACOBI && !fused && ctx->cg_dinv_codes != 0 && (ctx->cg_dinv_codes == 2 || nt))
    if (int rc = dinv_codes_build(ctx, n, dzc))
      return rc;
  ctx->last_solve_dinv_codes = dzc.codes ? dzc.ndict : 0;
  const bool dz = dzc.codes != nullptr;
  auto kern_update_p = pick_update_p(dz);
  auto kern_update_xr = pick_update_xr(dz);

  auto apply = [&](double* x, double* y, double* parts, int* np) -> int {
    // partitioned CSR operator: halo of x overlapped with the interior tiles
    if (multi && o->op == ZZZ_OP_CSR && ctx->overlap && ctx->have_tile_split)
      return launch_spmv_overlapped(ctx, x, y, parts, np);
    if (multi)
    {
      int rc = comm_halo_forward(ctx, x);
      if (rc)
        return rc;
    }
    if (o->op == ZZZ_OP_CSR)
      return launch_spmv(ctx, x, y, parts, np);
    return launch_matfree_action(ctx, x, y, parts, np);
  };

  // initial residual
  const double* w0 = nullptr;
  if (o->variant == ZZZ_CG_CGH)
  {
    int rc = apply(ctx->u.p, ctx->w.p, nullptr, nullptr); // action(x, y), src/cg.h:46
    if (rc)
      return rc;
    w0 = ctx->w.p;
  }
  else
    ZZZ_HIP(ctx, hipMemsetAsync(ctx->u.p, 0, sizeof(double) * ctx->u.n, s)); // KSP zero initial guess
  double* pa = ctx->part_b.p;
  double* pb = ctx->part_b.p + VGRID_MAX;
  hipLaunchKernelGGL(k_init_residual, dim3(g), dim3(VB), 0, s, ctx->b.p, w0, ctx->dinv.p, ctx->r.p, ctx->z.p, n, P.norm,
                     pa, pb);
  // where the consumer kernels find <r,z>, the test norm and <p,w>: the producers' partials, or the
  // single all-reduced value when a communicator is attached
  const double *rz_src = pa, *nn_src = pb, *pw_src = ctx->part_a.p;
  int n_rz = g;
  if (multi)
  {
    rz_src = ctx->red.p;
    nn_src = ctx->red.p + 1;
    pw_src = ctx->red.p + 2;
    n_rz = 1;
  }
  const int* stop_flag = reinterpret_cast<const int*>(ctx->state.p); // CgState::converged
  auto allreduce_beta = [&]() -> int {
    if (!multi)
      return ZZZ_OK;
    return comm_reduce_allreduce(ctx, stop_flag, pa, pb, nullptr, g, 2, ctx->red.p);
  };
  {
    int rc = allreduce_beta();
    if (rc)
      return rc;
  }

  // profiling events around the SpMV launches
  const int max_prof = o->profile ? 512 : 0;
  if ((int)ctx->ev.size() < 2 * max_prof)
  {
    size_t old = ctx->ev.size();
    ctx->ev.resize(2 * max_prof);
    for (size_t i = old; i < ctx->ev.size(); ++i)
      ZZZ_HIP(ctx, hipEventCreate(&ctx->ev[i]));
  }
  int nprof = 0;
  ctx->prof_halo_n = 0;
  ctx->prof_halo_wait_ms = 0.0;

  // host polling: copy the state every CHECK iterations, look at it NSLOT-1 batches later
  constexpr int CHECK = 8, NSLOT = 4;
  EventRing<NSLOT> chk_ev;
  ZZZ_HIP(ctx, chk_ev.create());
  int nchk = 0;
  bool stop = false;

  int it = 0;
  for (; it < max_it && !stop; ++it)
  {
    // convergence test of iteration `it` and the new search direction
    if (!fused)
      hipLaunchKernelGGL(kern_update_p, dim3(g), dim3(VB), 0, s, ctx->state.p, ctx->beta_hist.p, ctx->dp_hist.p,
                         ctx->alpha_hist.p, it, P, rz_src, nn_src, n_rz, ctx->z.p, ctx->p.p, ctx->u.p, n, 1, dzc);
    int np = 0;
    const bool timed = nprof < max_prof && it % PROF_STRIDE == 0;
    ctx->prof_now = timed;
    if (timed)
      (void)hipEventRecord(ctx->ev[2 * nprof], s);
    bool folded = false;
    if (fold_product && comm_tail_args(ctx, ctx->tail, 1, ctx->red.p + 2))
    {
      ctx->tail_armed = true; // <p,w> is all-reduced in the tail of the product launch
      ctx->tail_used = false;
    }
    {
#ifdef ZZZ_EXPERIMENTS
      int rc = fused ? launch_sellp_dir(ctx, ctx->z.p, pbuf[it & 1], pbuf[(it + 1) & 1], ctx->u.p, ctx->w.p, ctx->part_a.p, &np,
                                        it, P, rz_src, nn_src, n_rz, multi && ctx->overlap)
                     : apply(ctx->p.p, ctx->w.p, ctx->part_a.p, &np);
#else
      int rc = apply(ctx->p.p, ctx->w.p, ctx->part_a.p, &np);
#endif
      folded = ctx->tail_used;
      ctx->tail_armed = ctx->tail_used = false;
      if (rc)
        return rc;
    }
    ctx->prof_now = false;
    if (timed)
    {
      (void)hipEventRecord(ctx->ev[2 * nprof + 1], s);
      ++nprof;
    }
    if (multi)
    {
      if (!folded)
      {
        int rc = comm_reduce_allreduce(ctx, stop_flag, ctx->part_a.p, nullptr, nullptr, np, 1, ctx->red.p + 2);
        if (rc)
          return rc;
      }
      np = 1;
    }
    TailArgs Txr;
    const bool folded_xr = multi && comm_tail_args(ctx, Txr, 2, ctx->red.p);
    if (folded_xr)
    {
      Txr.expected = g;
      Txr.base = 0;
    }
    hipLaunchKernelGGL(kern_update_xr, dim3(g), dim3(VB), 0, s, ctx->state.p, ctx->beta_hist.p, ctx->alpha_hist.p, it, pw_src,
                       np, ctx->w.p, ctx->dinv.p, ctx->r.p, ctx->z.p, n, P.norm, pa, pb, P.variant, Txr, dzc);
    if (!folded_xr)
    {
      int rc = allreduce_beta();
      if (rc)
        return rc;
    }
    if ((it + 1) % CHECK == 0)
    {
      const int slot = nchk % NSLOT;
      if (nchk >= NSLOT - 1)
      {
        const int old = (nchk - (NSLOT - 1)) % NSLOT;
        ZZZ_HIP(ctx, hipEventSynchronize(chk_ev[old]));
        if (ctx->h_state[old].converged)
          stop = true;
      }
      ZZZ_HIP(ctx, hipMemcpyAsync(&ctx->h_state[slot], ctx->state.p, sizeof(CgState), hipMemcpyDeviceToHost, s));
      ZZZ_HIP(ctx, hipEventRecord(chk_ev[slot], s));
      ++nchk;
    }
  }
  // the test of the last completed iteration (it == max_it when the loop ran out) and its pending
  // solution update; no new direction
  hipLaunchKernelGGL(kern_update_p, dim3(g), dim3(VB), 0, s, ctx->state.p, ctx->beta_hist.p, ctx->dp_hist.p,
                     ctx->alpha_hist.p, it, P, rz_src, nn_src, n_rz, ctx->z.p, pbuf[it & 1], ctx->u.p, n, 0, dzc);
  ZZZ_HIP(ctx, hipGetLastError());
  CgState fin;
  ZZZ_HIP(ctx, hipMemcpyAsync(&fin, ctx->state.p, sizeof(CgState), hipMemcpyDeviceToHost, s));
  ZZZ_HIP(ctx, hipStreamSynchronize(s));
  if (int rc = comm_p2p_check(ctx))
    return rc;

  const int its = fin.converged ? fin.iters : max_it;
  ctx->last_iters = its;
  if (iters)
    *iters = its;
  if (rnorm)
  {
    rnorm[0] = fin.dp;
    rnorm[1] = fin.dp0;
  }
  ctx->history.resize((size_t)its + 1);
  ZZZ_HIP(ctx, hipMemcpy(ctx->history.data(), ctx->dp_hist.p, sizeof(double) * ((size_t)its + 1), hipMemcpyDeviceToHost));

  ctx->prof_spmv_ms = 0.0;
  ctx->prof_spmv_n = 0;
  const int used = std::min(nprof, (its + PROF_STRIDE - 1) / PROF_STRIDE); // launches past convergence return at once: not counted
  for (int i = 0; i < used; ++i)
  {
    float ms = 0;
    if (hipEventElapsedTime(&ms, ctx->ev[2 * i], ctx->ev[2 * i + 1]) == hipSuccess)
    {
      ctx->prof_spmv_ms += ms;
      ctx->prof_spmv_n++;
    }
  }
  if (ctx->prof_spmv_n)
    ctx->prof_spmv_ms /= (double)ctx->prof_spmv_n;
  {
    int cnt = 0;
    for (int i = 0; i < ctx->prof_halo_n; ++i)
    {
      float ms = 0;
      if (hipEventElapsedTime(&ms, ctx->ev_halo[(size_t)(2 * i)], ctx->ev_halo[(size_t)(2 * i + 1)]) == hipSuccess)
      {
        ctx->prof_halo_wait_ms += ms;
        ++cnt;
      }
    }
    if (cnt)
      ctx->prof_halo_wait_ms /= cnt;
    (void)hipGetLastError();
  }
  return finish_reason(ctx, o, fin, its);
}

// ---- KSPCG with the Chebyshev-Jacobi polynomial preconditioner (ZZZ_PC_CHEBYSHEV_JACOBI; oracle: zo_pcg_cheb) -------
// z = p_k(D^-1 A) D^-1 r by k steps of the Chebyshev iteration for D^-1 A on [hi / ratio, hi], hi = Gershgorin's bound:
// no reduction inside the application, k more products per CG iteration, roughly k + 1 times fewer CG iterations and
// all-reduces.  The rest of the iteration is the classical loop's: k_update_p (test, x and p), the product, then k_cheb_xr
// (k_update_xr's r -= alpha w with the polynomial's first term) and one product + k_cheb_step per further term; the last
// k_cheb_step sums <r,z> and the norm.
__global__ __launch_bounds__(VB) void k_row_abs_max(const rp_t* __restrict__ rowptr, const double* __restrict__ vals,
                                                    const double* __restrict__ dinv, int64_t n, double* __restrict__ parts)
{
  __shared__ double sh[VB / 64];
  double m = 0.0;
  for (int64_t r = blockIdx.x * (int64_t)VB + threadIdx.x; r < n; r += (int64_t)gridDim.x * VB)
  {
    double sum = 0.0;
    for (int64_t k = rowptr[r]; k < rowptr[r + 1]; ++k)
      sum += fabs(vals[k]);
    m = fmax(m, sum * fabs(dinv[r]));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1)
    m = fmax(m, __shfl_down(m, o, 64));
  if ((threadIdx.x & 63) == 0)
    sh[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0)
  {
    for (int i = 1; i < VB / 64; ++i)
      m = fmax(m, sh[i]);
    parts[blockIdx.x] = m;
  }
}
// g = D^-1 r; d = g / theta; z = d (the polynomial's first term: z_1 = 0 + d_0)
__global__ __launch_bounds__(VB) void k_cheb_init(const int* __restrict__ stop, const double* __restrict__ r,
                                                  const double* __restrict__ dinv, double theta, double* __restrict__ gv,
                                                  double* __restrict__ d, double* __restrict__ z, int64_t n)
{
  if (*stop)
    return;
  for (int64_t i = blockIdx.x * (int64_t)VB + threadIdx.x; i < n; i += (int64_t)gridDim.x * VB)
  {
    const double gi = dinv[i] * r[i];
    const double di = gi / theta;
    gv[i] = gi;
    d[i] = di;
    z[i] = di;
  }
}
// k_update_xr and k_cheb_init in one pass: alpha = beta / <p,w>; r -= alpha w; g = D^-1 r; d = g / theta; z = d
__global__ __launch_bounds__(VB) void k_cheb_xr(CgState* __restrict__ st, const double* __restrict__ beta_hist,
                                                double* __restrict__ alpha_hist, int it, const double* __restrict__ pw_parts,
                                                int npw, const double* __restrict__ w, const double* __restrict__ dinv,
                                                double theta, double* __restrict__ r, double* __restrict__ gv,
                                                double* __restrict__ d, double* __restrict__ z, int64_t n)
{
  const int f0 = st->converged;
  const double beta_it = beta_hist[it];
  const double pw1 = pw_parts[0];
  if (block_flag(f0))
    return;
  __shared__ double sh[VB / 64];
  const double pw = npw == 1 ? pw1 : reduce_parts_bcast(pw_parts, npw, sh);
  const double alpha = beta_it / pw;
  if (!isfinite(alpha)) // KSP_DIVERGED_NANORINF, as in k_update_xr
  {
    if (blockIdx.x == 0 && threadIdx.x == 0)
    {
      st->iters = it;
      st->converged = 2;
      __atomic_store_n(&st->conv_it1, it + 1, __ATOMIC_RELAXED);
    }
    return;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0)
    alpha_hist[it] = alpha;
  for (int64_t i = blockIdx.x * (int64_t)VB + threadIdx.x; i < n; i += (int64_t)gridDim.x * VB)
  {
    const double ri = -alpha * w[i] + r[i];
    const double gi = dinv[i] * ri;
    const double di = gi / theta;
    r[i] = ri;
    gv[i] = gi;
    d[i] = di;
    z[i] = di;
  }
}
// one term of the polynomial, w = A d_(j-1) given: g -= D^-1 w; d_j = c1 d_(j-1) + c2 g; z += d_j.  The last term leaves
// g and d alone and sums the partials of <r,z> and of the test norm (the arrays k_update_p reads).
__global__ __launch_bounds__(VB) void k_cheb_step(const int* __restrict__ stop, const double* __restrict__ w,
                                                  const double* __restrict__ dinv, double c1, double c2, int last,
                                                  double* __restrict__ gv, double* __restrict__ d, double* __restrict__ z,
                                                  const double* __restrict__ r, int norm, double* __restrict__ pa,
                                                  double* __restrict__ pb, int64_t n)
{
  if (*stop)
    return;
  __shared__ double sh[VB / 64];
  double sa = 0, sb = 0;
  for (int64_t i = blockIdx.x * (int64_t)VB + threadIdx.x; i < n; i += (int64_t)gridDim.x * VB)
  {
    const double gi = -1.0 * (dinv[i] * w[i]) + gv[i];
    const double dn = c1 * d[i] + c2 * gi; // (built with -ffp-contract=off: rounded as written, here, in the product's epilogue and in the oracle)
    const double zi = z[i] + dn;
    z[i] = zi;
    if (!last)
    {
      gv[i] = gi;
      d[i] = dn;
    }
    else
    {
      const double ri = r[i];
      sa += ri * zi;
      sb += (norm == ZZZ_NORM_UNPRECONDITIONED) ? ri * ri : zi * zi;
    }
  }
  if (last)
  {
    const double ta = block_reduce_sum(sa, sh);
    const double tb = block_reduce_sum(sb, sh);
    if (threadIdx.x == 0)
    {
      pa[blockIdx.x] = ta;
      pb[blockIdx.x] = tb;
    }
  }
}
// partials of <r,z> and of the test norm (the arrays k_update_p reads)
__global__ __launch_bounds__(VB) void k_dots_rz(const int* __restrict__ stop, const double* __restrict__ r,
                                                const double* __restrict__ z, int64_t n, int norm, double* __restrict__ pa,
                                                double* __restrict__ pb)
{
  if (*stop)
    return;
  __shared__ double sh[VB / 64];
  double sa = 0, sb = 0;
  for (int64_t i = blockIdx.x * (int64_t)VB + threadIdx.x; i < n; i += (int64_t)gridDim.x * VB)
  {
    const double ri = r[i], zi = z[i];
    sa += ri * zi;
    sb += (norm == ZZZ_NORM_UNPRECONDITIONED) ? ri * ri : zi * zi;
  }
  const double ta = block_reduce_sum(sa, sh);
  const double tb = block_reduce_sum(sb, sh);
  if (threadIdx.x == 0)
  {
    pa[blockIdx.x] = ta;
    pb[blockIdx.x] = tb;
  }
}

int comm_allgather_max(zzz_ctx* ctx, double* v); // zzz_comm.hip: maximum of one double over the ranks (set-up exchange)
int comm_allgather_double(zzz_ctx* ctx, double v, double* all);
int comm_rank_of(const zzz_ctx* ctx, int* nranks);

// the noise vector of the spectrum estimate (oracle: zo_noise): a fixed hash of the GLOBAL row number in the caller's
// numbering, so that neither the library's internal order nor the partition changes the estimate
__global__ __launch_bounds__(VB) void k_noise(const int32_t* __restrict__ perm, int bs, int64_t offset, int64_t n,
                                              double* __restrict__ v)
{
  for (int64_t i = blockIdx.x * (int64_t)VB + threadIdx.x; i < n; i += (int64_t)gridDim.x * VB)
  {
    int64_t row = i;
    if (perm)
      row = (int64_t)perm[i / bs] * bs + i % bs;
    unsigned int h = (unsigned int)((unsigned long long)(offset + row) * 2654435761ull + 12345ull);
    h ^= h >> 16;
    h *= 0x45d9f3bu;
    h ^= h >> 16;
    v[i] = (double)h / 4294967296.0 - 0.5;
  }
}

// largest eigenvalue of the k x k symmetric tridiagonal (diagonal t, off-diagonal e): bisection on the Sturm count
static double tridiag_lmax(int k, const double* t, const double* e)
{
  double lo = t[0], hi = t[0];
  for (int j = 0; j < k; ++j)
  {
    const double rad = (j > 0 ? std::fabs(e[j - 1]) : 0.0) + (j + 1 < k ? std::fabs(e[j]) : 0.0);
    lo = std::min(lo, t[j] - rad);
    hi = std::max(hi, t[j] + rad);
  }
  for (int itb = 0; itb < 200 && hi - lo > 4.0e-16 * std::fabs(hi); ++itb)
  {
    const double mid = 0.5 * (lo + hi);
    int above = 0;
    double q = 1.0;
    for (int j = 0; j < k; ++j)
    {
      const double off2 = j > 0 ? e[j - 1] * e[j - 1] : 0.0;
      q = (t[j] - mid) - (j > 0 ? off2 / q : 0.0);
      if (q == 0.0)
        q = 1.0e-300;
      if (q > 0.0)
        ++above;
    }
    if (above > 0)
      lo = mid;
    else
      hi = mid;
  }
  return 0.5 * (lo + hi);
}

// PETSc's -ksp_chebyshev_esteig: `its` iterations of Jacobi-PCG (the classical loop above, communicator and all) on the
// noise vector; the Lanczos tridiagonal of its step lengths and <r,z> values gives the largest Ritz value of D^-1 A
// (from below: 0.97-0.98 of the eigenvalue after 10 iterations).  0 when it could not be formed.
static int chebyshev_esteig(zzz_ctx* ctx, const zzz_solver_opts* o, int its, double* ritz)
{
  *ritz = 0.0;
  its = std::min(its, 64);
  const int64_t n = ctx->n_owned * ctx->bs;
  int nranks = 1;
  const int me = comm_rank_of(ctx, &nranks);
  std::vector<double> sizes((size_t)nranks);
  if (int rc = comm_allgather_double(ctx, (double)n, sizes.data()))
    return rc;
  int64_t offset = 0;
  for (int r = 0; r < me; ++r)
    offset += (int64_t)sizes[(size_t)r];
  ZZZ_HIP(ctx, ctx->cheb_noise.alloc(ctx->b.n));
  ZZZ_HIP(ctx, hipMemsetAsync(ctx->cheb_noise.p, 0, sizeof(double) * ctx->cheb_noise.n, ctx->stream));
  hipLaunchKernelGGL(k_noise, dim3(vgrid(n)), dim3(VB), 0, ctx->stream, ctx->renumbered ? ctx->perm.p : (const int32_t*)nullptr,
                     ctx->bs, offset, n, ctx->cheb_noise.p);
  zzz_solver_opts o2 = *o;
  o2.pc = ZZZ_PC_JACOBI;
  o2.norm = ZZZ_NORM_PRECONDITIONED;
  o2.max_it = its;
  o2.rtol = 0.0;
  o2.atol = 0.0;
  o2.dtol = 1.0e300;
  o2.profile = 0;
  o2.single_reduction = 0;
  o2.error_if_not_converged = 0;
  auto swap_b = [&]() {
    std::swap(ctx->b.p, ctx->cheb_noise.p);
    std::swap(ctx->b.n, ctx->cheb_noise.n);
    std::swap(ctx->b.cap, ctx->cheb_noise.cap);
  };
  swap_b();
  int ran = 0;
  double rn[2];
  const int rc = cg_solve(ctx, &o2, &ran, rn);
  swap_b();
  if (rc)
    return rc;
  if (ran < 2)
    return ZZZ_OK;
  std::vector<double> alpha((size_t)ran), rho((size_t)ran + 1);
  ZZZ_HIP(ctx, hipMemcpy(alpha.data(), ctx->alpha_hist.p, sizeof(double) * (size_t)ran, hipMemcpyDeviceToHost));
  ZZZ_HIP(ctx, hipMemcpy(rho.data(), ctx->beta_hist.p, sizeof(double) * ((size_t)ran + 1), hipMemcpyDeviceToHost));
  std::vector<double> t((size_t)ran), e((size_t)ran);
  for (int j = 0; j < ran; ++j)
  {
    if (!(alpha[(size_t)j] > 0.0) || !std::isfinite(alpha[(size_t)j]) || !(rho[(size_t)j] > 0.0) || !std::isfinite(rho[(size_t)j]))
      return ZZZ_OK;
    t[(size_t)j] = 1.0 / alpha[(size_t)j] + (j > 0 ? (rho[(size_t)j] / rho[(size_t)j - 1]) / alpha[(size_t)j - 1] : 0.0);
    if (j + 1 < ran)
      e[(size_t)j] = std::sqrt(rho[(size_t)j + 1] / rho[(size_t)j]) / alpha[(size_t)j];
  }
  for (int j = 0; j + 1 < ran; ++j)
    if (!std::isfinite(e[(size_t)j]))
      return ZZZ_OK;
  *ritz = tridiag_lmax(ran, t.data(), e.data());
  return ZZZ_OK;
}

// What both forms of the Chebyshev-Jacobi solve share: the polynomial's constants, its work vectors, the choice between
// terms as product epilogues and terms as launches of their own.
struct ChebPlan
{
  int degree = 3;
  double hi = 0.0, theta = 0.0, delta = 0.0, sigma = 0.0;
  double *g = nullptr, *d = nullptr, *d2 = nullptr; // residual of the polynomial's recurrence; its direction, two buffers
  bool fused = false, split = false;
};

// Spectrum bound (Gershgorin, tightened by the Lanczos estimate), constants, buffers.  Runs the estimate's short Jacobi
// solve through the classical loop: call it BEFORE the caller initialises its own solve.
static int chebyshev_setup(zzz_ctx* ctx, const zzz_solver_opts* o, ChebPlan& C)
{
  const int64_t n = ctx->n_owned * ctx->bs;
  const bool multi = ctx->comm != nullptr;
  const int g = vgrid(n);
  hipStream_t s = ctx->stream;
  C.degree = o->pc_degree > 0 ? o->pc_degree : 3;
  const double ratio = o->pc_ratio > 1.0 ? o->pc_ratio : 60.0;
  hipLaunchKernelGGL(k_extract_dinv, dim3(g), dim3(VB), 0, s, ctx->rowptr.p, ctx->cols.p, ctx->vals.p, ctx->dinv.p, n, 1);
  const int est_its = o->pc_esteig_its == 0 ? 10 : o->pc_esteig_its;
  double hi = 0.0;
  // the bound belongs to the matrix, not to the solve: kept until the values change (every rank sees the same sequence of
  // assemblies and uploads, so all of them take or skip the estimate's collectives together)
  const bool cached = ctx->cheb_version == ctx->mat_version && ctx->cheb_est_its == est_its && ctx->cheb_hi > 0.0;
  if (cached)
    hi = ctx->cheb_hi;
  else
  {
  // spectrum bound: Gershgorin's for D^-1 A, maximum over the ranks
  hipLaunchKernelGGL(k_row_abs_max, dim3(g), dim3(VB), 0, s, ctx->rowptr.p, ctx->vals.p, ctx->dinv.p, n, ctx->part_b.p);
  std::vector<double> hp((size_t)g);
  ZZZ_HIP(ctx, hipMemcpyAsync(hp.data(), ctx->part_b.p, sizeof(double) * (size_t)g, hipMemcpyDeviceToHost, s));
  ZZZ_HIP(ctx, hipStreamSynchronize(s));
  for (double v : hp)
    hi = std::max(hi, v);
  if (multi)
    if (int rc = comm_allgather_max(ctx, &hi))
      return rc;
  if (!(hi > 0.0) || !std::isfinite(hi))
    hi = 1.0;
  // ... tightened by the Lanczos estimate where that is lower (Gershgorin's bound is exact for P1 Laplacians, 2, and up to
  // 2.7 x too high for P2 / P3 / elasticity, which costs 1.6 x the products); safety factor 1.1 as in PETSc
  if (est_its > 0)
  {
    double ritz = 0.0;
    if (int rc = chebyshev_esteig(ctx, o, est_its, &ritz))
      return rc;
    if (ritz > 0.0 && std::isfinite(ritz) && 1.1 * ritz < hi)
      hi = 1.1 * ritz;
  }
  ctx->cheb_hi = hi;
  ctx->cheb_version = ctx->mat_version;
  ctx->cheb_est_its = est_its;
  }
  const double lo = hi / ratio;
  C.hi = hi;
  C.theta = 0.5 * (hi + lo);
  C.delta = 0.5 * (hi - lo);
  C.sigma = C.theta / C.delta;
  // On the operator stream a term is the EPILOGUE of its product (ChebEpi, zzz_sellp.hip): w = A d never travels through
  // memory and the term costs no launch of its own; d then alternates between two buffers (other lanes still gather the
  // old one).  A/B knob ZZZ_CHEB_FUSED=0: product and k_cheb_step as two launches (the tile kernel's form) -- same bits.
  C.fused = sellp_active(ctx);
  if (const char* e = getenv("ZZZ_CHEB_FUSED"))
    C.fused = C.fused && atoi(e) != 0;
  C.split = multi && ctx->overlap && ctx->have_tile_split;
  if (C.split && !ctx->have_group_split)
    C.fused = false;
  ZZZ_HIP(ctx, ctx->cheb_d.alloc((size_t)ctx->nloc())); // ghost entries: the product gathers it
  ZZZ_HIP(ctx, ctx->cheb_d2.alloc((size_t)ctx->nloc()));
  ZZZ_HIP(ctx, ctx->cheb_g.alloc((size_t)ctx->nloc()));
  C.d = ctx->cheb_d.p;
  C.d2 = ctx->cheb_d2.p;
  C.g = ctx->cheb_g.p;
  ZZZ_HIP(ctx, hipMemsetAsync(C.d, 0, sizeof(double) * (size_t)ctx->nloc(), s));
  ZZZ_HIP(ctx, hipMemsetAsync(C.d2, 0, sizeof(double) * (size_t)ctx->nloc(), s));
  return ZZZ_OK;
}

// The terms after the first of z = p_k(D^-1 A) D^-1 r (g, d and z hold the first).  dots: the LAST term also sums the
// partials of <r,z> and of the test norm -- into the product's partial arrays at strides 1 and 2 (*np_last of them)
// when the term is an epilogue, into pa / pb (vgrid of them) otherwise.
static int chebyshev_terms(zzz_ctx* ctx, const ChebPlan& C, int norm, bool dots, double* pa, double* pb, int* np_last)
{
  const int64_t n = ctx->n_owned * ctx->bs;
  const bool multi = ctx->comm != nullptr;
  const int g = vgrid(n);
  hipStream_t s = ctx->stream;
  const int* stop_flag = reinterpret_cast<const int*>(ctx->state.p);
  const int nn_is_rr = norm == ZZZ_NORM_UNPRECONDITIONED ? 1 : 0;
  double rho = 1.0 / C.sigma;
  double *dcur = C.d, *dalt = C.d2;
  for (int st = 1; st < C.degree; ++st)
  {
    const bool last = dots && st + 1 == C.degree;
    const double rhon = 1.0 / (2.0 * C.sigma - rho);
    const double c1 = rhon * rho, c2 = 2.0 * rhon / C.delta;
    rho = rhon;
    if (C.fused)
    {
      ChebEpi E;
      E.dinv = ctx->dinv.p;
      E.g = C.g;
      E.z = ctx->z.p;
      E.r = ctx->r.p;
      E.c1 = c1;
      E.c2 = c2;
      double* parts = last ? ctx->part_a.p : nullptr;
      int rc;
      if (C.split)
        rc = launch_sellp_overlapped(ctx, dcur, dalt, parts, last ? np_last : nullptr, nullptr, nn_is_rr, &E);
      else
      {
        if (multi)
          if (int rh = comm_halo_forward(ctx, dcur))
            return rh;
        rc = launch_sellp(ctx, dcur, dalt, parts, last ? np_last : nullptr, nullptr, nn_is_rr, &E);
      }
      if (rc)
        return rc;
      std::swap(dcur, dalt);
      continue;
    }
    // terms as launches of their own: the product's output goes to the second direction buffer (free in this form; the
    // CG vector w is the single-reduction form's A p and must survive)
    int rc;
    if (C.split)
      rc = launch_spmv_overlapped(ctx, dcur, C.d2, nullptr, nullptr);
    else
    {
      if (multi)
        if (int rh = comm_halo_forward(ctx, dcur))
          return rh;
      rc = launch_spmv(ctx, dcur, C.d2, nullptr, nullptr);
    }
    if (rc)
      return rc;
    hipLaunchKernelGGL(k_cheb_step, dim3(g), dim3(VB), 0, s, stop_flag, C.d2, ctx->dinv.p, c1, c2, last ? 1 : 0, C.g, dcur,
                       ctx->z.p, ctx->r.p, norm, pa, pb, n);
  }
  return ZZZ_OK;
}

static int cg_solve_chebyshev(zzz_ctx* ctx, const zzz_solver_opts* o, int* iters, double* rnorm)
{
  const int64_t n = ctx->n_owned * ctx->bs;
  const int max_it = o->max_it;
  // the scalar logic is KSPCG's with a preconditioner: k_update_p / k_update_xr take pc = Jacobi semantics for alpha, beta
  CgParams P{o->variant, ZZZ_PC_JACOBI, o->norm, o->rtol, o->atol, o->dtol > 0.0 ? o->dtol : 1.0e4};
  const bool multi = ctx->comm != nullptr;
  const int g = vgrid(n);
  hipStream_t s = ctx->stream;
  ChebPlan C;
  if (int rc = chebyshev_setup(ctx, o, C))
    return rc;
  const int degree = C.degree;
  const double hi = C.hi, theta = C.theta;
  const bool fused = C.fused;
  double *chd = C.d, *chg = C.g;
  ctx->last_solve_fused = false;
  ZZZ_HIP(ctx, ctx->beta_hist.reserve((size_t)max_it + 2));
  ZZZ_HIP(ctx, ctx->dp_hist.reserve((size_t)max_it + 2));
  ZZZ_HIP(ctx, ctx->alpha_hist.reserve((size_t)max_it + 2));
  ZZZ_HIP(ctx, hipMemsetAsync(ctx->state.p, 0, sizeof(CgState), s));
  ZZZ_HIP(ctx, hipMemsetAsync(ctx->p.p, 0, sizeof(double) * ctx->p.n, s));
  ZZZ_HIP(ctx, hipMemsetAsync(ctx->u.p, 0, sizeof(double) * ctx->u.n, s)); // KSP zero initial guess
  hipLaunchKernelGGL(k_extract_dinv, dim3(g), dim3(VB), 0, s, ctx->rowptr.p, ctx->cols.p, ctx->vals.p, ctx->dinv.p, n, 1);

  const int* stop_flag = reinterpret_cast<const int*>(ctx->state.p);
  auto apply = [&](double* x, double* y, double* parts, int* np) -> int {
    if (multi && ctx->overlap && ctx->have_tile_split)
      return launch_spmv_overlapped(ctx, x, y, parts, np);
    if (multi)
      if (int rc = comm_halo_forward(ctx, x))
        return rc;
    return launch_spmv(ctx, x, y, parts, np);
  };
  double* pa = ctx->part_b.p;
  double* pb = ctx->part_b.p + VGRID_MAX;
  int np_last = g;
  // the polynomial's terms after the first, then the partials of <r,z> and the norm (all-reduced when a communicator
  // is attached)
  auto polynomial = [&]() -> int {
    if (int rc = chebyshev_terms(ctx, C, P.norm, true, pa, pb, &np_last))
      return rc;
    if (degree == 1)
      hipLaunchKernelGGL(k_dots_rz, dim3(g), dim3(VB), 0, s, stop_flag, ctx->r.p, ctx->z.p, n, P.norm, pa, pb);
    if (multi)
    {
      if (fused && degree > 1)
        return comm_reduce_allreduce(ctx, stop_flag, ctx->part_a.p + SPMV_PSTRIDE, ctx->part_a.p + 2 * SPMV_PSTRIDE, nullptr,
                                     np_last, 2, ctx->red.p);
      return comm_reduce_allreduce(ctx, stop_flag, pa, pb, nullptr, g, 2, ctx->red.p);
    }
    return ZZZ_OK;
  };
  // r = b
  hipLaunchKernelGGL(k_init_residual, dim3(g), dim3(VB), 0, s, ctx->b.p, (const double*)nullptr, ctx->dinv.p, ctx->r.p, ctx->z.p, n,
                     P.norm, pa, pb);
  hipLaunchKernelGGL(k_cheb_init, dim3(g), dim3(VB), 0, s, stop_flag, ctx->r.p, ctx->dinv.p, theta, chg, chd, ctx->z.p, n);
  if (int rc = polynomial())
    return rc;
  const double *rz_src = pa, *nn_src = pb, *pw_src = ctx->part_a.p;
  int n_rz = g;
  if (fused && degree > 1)
  {
    // the last term's partials sit behind the product's own (strides 1 and 2 of its partial array)
    rz_src = ctx->part_a.p + SPMV_PSTRIDE;
    nn_src = ctx->part_a.p + 2 * SPMV_PSTRIDE;
    n_rz = np_last;
  }
  if (multi)
  {
    rz_src = ctx->red.p;
    nn_src = ctx->red.p + 1;
    pw_src = ctx->red.p + 2;
    n_rz = 1;
  }
  const int max_prof = o->profile ? 512 : 0;
  if ((int)ctx->ev.size() < 2 * max_prof)
  {
    size_t old = ctx->ev.size();
    ctx->ev.resize(2 * max_prof);
    for (size_t i = old; i < ctx->ev.size(); ++i)
      ZZZ_HIP(ctx, hipEventCreate(&ctx->ev[i]));
  }
  int nprof = 0;
  ctx->prof_halo_n = 0;
  ctx->prof_halo_wait_ms = 0.0;
  constexpr int CHECK = 8, NSLOT = 4;
  EventRing<NSLOT> chk_ev;
  ZZZ_HIP(ctx, chk_ev.create());
  int nchk = 0;
  bool stop = false;
  int it = 0;
  for (; it < max_it && !stop; ++it)
  {
    hipLaunchKernelGGL(k_update_p<false>, dim3(g), dim3(VB), 0, s, ctx->state.p, ctx->beta_hist.p, ctx->dp_hist.p, ctx->alpha_hist.p,
                       it, P, rz_src, nn_src, n_rz, ctx->z.p, ctx->p.p, ctx->u.p, n, 1);
    int np = 0;
    const bool timed = nprof < max_prof && it % PROF_STRIDE == 0;
    ctx->prof_now = timed;
    if (timed)
      (void)hipEventRecord(ctx->ev[2 * nprof], s);
    {
      int rc = apply(ctx->p.p, ctx->w.p, ctx->part_a.p, &np);
      ctx->prof_now = false;
      if (rc)
        return rc;
    }
    if (timed)
    {
      (void)hipEventRecord(ctx->ev[2 * nprof + 1], s);
      ++nprof;
    }
    if (multi)
    {
      if (int rc = comm_reduce_allreduce(ctx, stop_flag, ctx->part_a.p, nullptr, nullptr, np, 1, ctx->red.p + 2))
        return rc;
      np = 1;
    }
    hipLaunchKernelGGL(k_cheb_xr, dim3(g), dim3(VB), 0, s, ctx->state.p, ctx->beta_hist.p, ctx->alpha_hist.p, it, pw_src, np,
                       ctx->w.p, ctx->dinv.p, theta, ctx->r.p, chg, chd, ctx->z.p, n);
    if (int rc = polynomial())
      return rc;
    if ((it + 1) % CHECK == 0)
    {
      const int slot = nchk % NSLOT;
      if (nchk >= NSLOT - 1)
      {
        const int old = (nchk - (NSLOT - 1)) % NSLOT;
        ZZZ_HIP(ctx, hipEventSynchronize(chk_ev[old]));
        if (ctx->h_state[old].converged)
          stop = true;
      }
      ZZZ_HIP(ctx, hipMemcpyAsync(&ctx->h_state[slot], ctx->state.p, sizeof(CgState), hipMemcpyDeviceToHost, s));
      ZZZ_HIP(ctx, hipEventRecord(chk_ev[slot], s));
      ++nchk;
    }
  }
  hipLaunchKernelGGL(k_update_p<false>, dim3(g), dim3(VB), 0, s, ctx->state.p, ctx->beta_hist.p, ctx->dp_hist.p, ctx->alpha_hist.p, it,
                     P, rz_src, nn_src, n_rz, ctx->z.p, ctx->p.p, ctx->u.p, n, 0);
  ZZZ_HIP(ctx, hipGetLastError());
  CgState fin;
  ZZZ_HIP(ctx, hipMemcpyAsync(&fin, ctx->state.p, sizeof(CgState), hipMemcpyDeviceToHost, s));
  ZZZ_HIP(ctx, hipStreamSynchronize(s));
  if (int rc = comm_p2p_check(ctx))
    return rc;
  const int its = fin.converged ? fin.iters : max_it;
  ctx->last_iters = its;
  if (iters)
    *iters = its;
  if (rnorm)
  {
    rnorm[0] = fin.dp;
    rnorm[1] = fin.dp0;
  }
  ctx->history.resize((size_t)its + 1);
  ZZZ_HIP(ctx, hipMemcpy(ctx->history.data(), ctx->dp_hist.p, sizeof(double) * ((size_t)its + 1), hipMemcpyDeviceToHost));
  ctx->prof_spmv_ms = 0.0;
  ctx->prof_spmv_n = 0;
  const int used = std::min(nprof, (its + PROF_STRIDE - 1) / PROF_STRIDE);
  for (int i = 0; i < used; ++i)
  {
    float ms = 0;
    if (hipEventElapsedTime(&ms, ctx->ev[2 * i], ctx->ev[2 * i + 1]) == hipSuccess)
    {
      ctx->prof_spmv_ms += ms;
      ctx->prof_spmv_n++;
    }
  }
  if (ctx->prof_spmv_n)
    ctx->prof_spmv_ms /= (double)ctx->prof_spmv_n;
  ctx->last_pc_bound = hi;
  return finish_reason(ctx, o, fin, its);
}

// -ksp_cg_single_reduction: see k_sr_update.  Two kernels and one reduction point per iteration.
static int cg_solve_single_reduction(zzz_ctx* ctx, const zzz_solver_opts* o, int* iters, double* rnorm)
{
  const int64_t n = ctx->n_owned * ctx->bs;
  const int max_it = o->max_it;
  CgParams P{o->variant, o->pc, o->norm, o->rtol, o->atol, o->dtol > 0.0 ? o->dtol : 1.0e4};
  const bool multi = ctx->comm != nullptr;
  const int g = vgrid(n);
  hipStream_t s = ctx->stream;
  const int nn_is_rr = o->norm == ZZZ_NORM_UNPRECONDITIONED ? 1 : 0;
  // Chebyshev-Jacobi in this form: ONE reduction point per k products (the classical form has two) and k + 1 launches
  const bool cheb = o->pc == ZZZ_PC_CHEBYSHEV_JACOBI;
  ChebPlan C;
  if (cheb)
    if (int rc = chebyshev_setup(ctx, o, C))
      return rc;
  const bool ntv = loop_exceeds_cache(ctx, 8);
  ctx->last_solve_fused = false;

  ZZZ_HIP(ctx, ctx->beta_hist.reserve((size_t)max_it + 2));
  ZZZ_HIP(ctx, ctx->dp_hist.reserve((size_t)max_it + 2));
  ZZZ_HIP(ctx, ctx->dpi_hist.reserve((size_t)max_it + 2));
  ZZZ_HIP(ctx, ctx->sr_s.alloc((size_t)ctx->nloc()));
  ZZZ_HIP(ctx, hipMemsetAsync(ctx->state.p, 0, sizeof(CgState), s));
  ZZZ_HIP(ctx, hipMemsetAsync(ctx->p.p, 0, sizeof(double) * ctx->p.n, s));
  ZZZ_HIP(ctx, hipMemsetAsync(ctx->w.p, 0, sizeof(double) * ctx->w.n, s));
  ZZZ_HIP(ctx, hipMemsetAsync(ctx->z.p, 0, sizeof(double) * ctx->z.n, s)); // ghost entries of z are exchanged
  ZZZ_HIP(ctx, hipMemsetAsync(ctx->u.p, 0, sizeof(double) * ctx->u.n, s)); // KSP zero initial guess
  hipLaunchKernelGGL(k_extract_dinv, dim3(g), dim3(VB), 0, s, ctx->rowptr.p, ctx->cols.p, ctx->vals.p, ctx->dinv.p, n,
                     o->pc != ZZZ_PC_NONE ? 1 : 0);
  // Jacobi's inverse diagonal as 16-bit codes, z of the last iteration recomputed from r (k_sr_update<.., DZ>): under the
  // rule of the classical form (a loop too large for the Infinity Cache; ZZZ_CG_DINV_CODES: 0 never, 2 at any size)
  DinvCodes dzc{nullptr, nullptr, nullptr, 0};
  if (!cheb && o->pc == ZZZ_PC_JACOBI && ctx->cg_dinv_codes != 0 && (ctx->cg_dinv_codes == 2 || ntv))
    if (int rc = dinv_codes_build(ctx, n, dzc))
      return rc;
  ctx->last_solve_dinv_codes = dzc.codes ? dzc.ndict : 0;
  const bool dz = dzc.codes != nullptr;
  auto kern_sr_update = cheb ? (ntv ? k_sr_update<true, true> : k_sr_update<false, true>)
                             : (dz ? (ntv ? k_sr_update<true, false, true> : k_sr_update<false, false, true>)
                                   : (ntv ? k_sr_update<true> : k_sr_update<false>));
  // r = b, z = D^-1 r (the partials of this kernel are not used: the SpMV below leaves all three)
  hipLaunchKernelGGL(k_init_residual, dim3(g), dim3(VB), 0, s, ctx->b.p, (const double*)nullptr, ctx->dinv.p, ctx->r.p,
                     ctx->z.p, n, P.norm, ctx->part_b.p, ctx->part_b.p + VGRID_MAX);
  // ... or z = p_k(D^-1 A) D^-1 r: first term, then the others (none of them sums anything: the product s = A z does)
  auto polynomial = [&](bool first) -> int {
    if (!cheb)
      return ZZZ_OK;
    if (first)
      hipLaunchKernelGGL(k_cheb_init, dim3(g), dim3(VB), 0, s, reinterpret_cast<const int*>(ctx->state.p), ctx->r.p, ctx->dinv.p,
                         C.theta, C.g, C.d, ctx->z.p, n);
    return chebyshev_terms(ctx, C, P.norm, false, ctx->part_b.p, ctx->part_b.p + VGRID_MAX, nullptr);
  };
  if (int rc = polynomial(true))
    return rc;

  double* parts = ctx->part_a.p; // <z,s> | <r,z> | norm, SPMV_PSTRIDE apart
  const double *zs_src = parts, *rz_src = parts + SPMV_PSTRIDE, *nn_src = parts + 2 * SPMV_PSTRIDE;
  if (multi)
  {
    rz_src = ctx->red.p;
    nn_src = ctx->red.p + 1;
    zs_src = ctx->red.p + 2;
  }
  int np = 0;
  // s = A z with the three partial dot products; then (multi) one all-reduce of three doubles
  const bool fold_product = multi && sellp_active(ctx);
  auto apply = [&]() -> int {
    int rc;
    bool folded = false;
    if (fold_product && comm_tail_args(ctx, ctx->tail, 3, ctx->red.p))
    {
      ctx->tail_armed = true; // (<r,z>, norm, <z,s>) all-reduced in the tail of the product launch
      ctx->tail_used = false;
    }
    if (multi && ctx->overlap && ctx->have_tile_split)
      rc = launch_spmv_overlapped(ctx, ctx->z.p, ctx->sr_s.p, parts, &np, ctx->r.p, nn_is_rr);
    else
    {
      if (multi)
      {
        rc = comm_halo_forward(ctx, ctx->z.p);
        if (rc)
          return rc;
      }
      rc = launch_spmv(ctx, ctx->z.p, ctx->sr_s.p, parts, &np, ctx->r.p, nn_is_rr);
    }
    folded = ctx->tail_used;
    ctx->tail_armed = ctx->tail_used = false;
    if (rc)
      return rc;
    if (multi)
    {
      if (!folded)
      {
        rc = comm_reduce_allreduce(ctx, reinterpret_cast<const int*>(ctx->state.p), parts + SPMV_PSTRIDE,
                                   parts + 2 * SPMV_PSTRIDE, parts, np, 3, ctx->red.p);
        if (rc)
          return rc;
      }
      np = 1;
    }
    return ZZZ_OK;
  };

  const int max_prof = o->profile ? 512 : 0;
  if ((int)ctx->ev.size() < 2 * max_prof)
  {
    size_t old = ctx->ev.size();
    ctx->ev.resize(2 * max_prof);
    for (size_t i = old; i < ctx->ev.size(); ++i)
      ZZZ_HIP(ctx, hipEventCreate(&ctx->ev[i]));
  }
  int nprof = 0;
  ctx->prof_halo_n = 0;
  ctx->prof_halo_wait_ms = 0.0;
  {
    int rc = apply();
    if (rc)
      return rc;
  }
  constexpr int CHECK = 8, NSLOT = 4;
  EventRing<NSLOT> chk_ev;
  ZZZ_HIP(ctx, chk_ev.create());
  int nchk = 0;
  bool stop = false;
  int it = 0;
  for (; it < max_it && !stop; ++it)
  {
    hipLaunchKernelGGL(kern_sr_update, dim3(g), dim3(VB), 0, s, ctx->state.p, ctx->beta_hist.p, ctx->dpi_hist.p, ctx->dp_hist.p,
                       it, P, rz_src, nn_src, zs_src, np, ctx->dinv.p, ctx->sr_s.p, ctx->z.p, ctx->p.p, ctx->w.p, ctx->u.p,
                       ctx->r.p, n, 0, C.theta, C.g, C.d, dzc);
    if (int rc = polynomial(false))
      return rc;
    const bool timed = nprof < max_prof && it % PROF_STRIDE == 0;
    ctx->prof_now = timed;
    if (timed)
      (void)hipEventRecord(ctx->ev[2 * nprof], s);
    {
      int rc = apply();
      if (rc)
        return rc;
    }
    ctx->prof_now = false;
    if (timed)
    {
      (void)hipEventRecord(ctx->ev[2 * nprof + 1], s);
      ++nprof;
    }
    if ((it + 1) % CHECK == 0)
    {
      const int slot = nchk % NSLOT;
      if (nchk >= NSLOT - 1)
      {
        const int old = (nchk - (NSLOT - 1)) % NSLOT;
        ZZZ_HIP(ctx, hipEventSynchronize(chk_ev[old]));
        if (ctx->h_state[old].converged)
          stop = true;
      }
      ZZZ_HIP(ctx, hipMemcpyAsync(&ctx->h_state[slot], ctx->state.p, sizeof(CgState), hipMemcpyDeviceToHost, s));
      ZZZ_HIP(ctx, hipEventRecord(chk_ev[slot], s));
      ++nchk;
    }
  }
  // convergence test of the last completed iteration: scalars only
  hipLaunchKernelGGL(kern_sr_update, dim3(1), dim3(VB), 0, s, ctx->state.p, ctx->beta_hist.p, ctx->dpi_hist.p, ctx->dp_hist.p, it,
                     P, rz_src, nn_src, zs_src, np, ctx->dinv.p, ctx->sr_s.p, ctx->z.p, ctx->p.p, ctx->w.p, ctx->u.p, ctx->r.p,
                     n, 1, C.theta, C.g, C.d, dzc);
  ZZZ_HIP(ctx, hipGetLastError());
  CgState fin;
  ZZZ_HIP(ctx, hipMemcpyAsync(&fin, ctx->state.p, sizeof(CgState), hipMemcpyDeviceToHost, s));
  ZZZ_HIP(ctx, hipStreamSynchronize(s));
  if (int rc = comm_p2p_check(ctx))
    return rc;
  const int its = fin.converged ? fin.iters : max_it;
  ctx->last_iters = its;
  if (iters)
    *iters = its;
  if (rnorm)
  {
    rnorm[0] = fin.dp;
    rnorm[1] = fin.dp0;
  }
  ctx->history.resize((size_t)its + 1);
  ZZZ_HIP(ctx, hipMemcpy(ctx->history.data(), ctx->dp_hist.p, sizeof(double) * ((size_t)its + 1), hipMemcpyDeviceToHost));
  ctx->prof_spmv_ms = 0.0;
  ctx->prof_spmv_n = 0;
  const int used = std::min(nprof, (its + PROF_STRIDE - 1) / PROF_STRIDE);
  for (int i = 0; i < used; ++i)
  {
    float ms = 0;
    if (hipEventElapsedTime(&ms, ctx->ev[2 * i], ctx->ev[2 * i + 1]) == hipSuccess)
    {
      ctx->prof_spmv_ms += ms;
      ctx->prof_spmv_n++;
    }
  }
  if (ctx->prof_spmv_n)
    ctx->prof_spmv_ms /= (double)ctx->prof_spmv_n;
  {
    int cnt = 0;
    for (int i = 0; i < ctx->prof_halo_n; ++i)
    {
      float ms = 0;
      if (hipEventElapsedTime(&ms, ctx->ev_halo[(size_t)(2 * i)], ctx->ev_halo[(size_t)(2 * i + 1)]) == hipSuccess)
      {
        ctx->prof_halo_wait_ms += ms;
        ++cnt;
      }
    }
    if (cnt)
      ctx->prof_halo_wait_ms /= cnt;
    (void)hipGetLastError();
  }
  if (cheb)
    ctx->last_pc_bound = C.hi;
  return finish_reason(ctx, o, fin, its);
}
ZZZ_PRELOAD_TU(cg)
} // namespace zzz

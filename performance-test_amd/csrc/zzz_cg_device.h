// Device pieces of the CG loop shared by zzz_cg.hip and the fused product + direction kernel of zzz_sellp.hip.
#pragma once
#include "zzz_device.h"
#include "zzz_internal.h"

namespace zzz
{
// one workgroup-wide sum of parts[0..np) (fixed order); result in every thread
__device__ inline double reduce_parts_bcast(const double* __restrict__ parts, int np, double* sh)
{
  double s = 0;
  for (int i = threadIdx.x; i < np; i += blockDim.x)
    s += parts[i];
  const double t = block_reduce_sum(s, sh);
  __shared__ double bc;
  __syncthreads();
  if (threadIdx.x == 0)
    bc = t;
  __syncthreads();
  return bc;
}


// The same for up to three partial arrays in ONE pass and one barrier pair (same tree per array as reduce_parts_bcast:
// strided per-thread sums, shuffles inside a wavefront, the per-wavefront sums added in order -- here by every thread
// from LDS instead of by thread 0 plus a broadcast).  pc may be null.
__device__ inline void reduce_parts3_bcast(const double* __restrict__ pa, const double* __restrict__ pb,
                                           const double* __restrict__ pc, int np, double& ra, double& rb, double& rc)
{
  __shared__ double sh[3 * 16];
  double s0 = 0, s1 = 0, s2 = 0;
  for (int i = threadIdx.x; i < np; i += blockDim.x)
  {
    s0 += pa[i];
    s1 += pb[i];
    if (pc)
      s2 += pc[i];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1)
  {
    s0 += __shfl_down(s0, o, 64);
    s1 += __shfl_down(s1, o, 64);
    s2 += __shfl_down(s2, o, 64);
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if (lane == 0)
  {
    sh[wv] = s0;
    sh[nw + wv] = s1;
    sh[2 * nw + wv] = s2;
  }
  __syncthreads();
  ra = rb = rc = 0.0;
  for (int i = 0; i < nw; ++i)
  {
    ra += sh[i];
    rb += sh[nw + i];
    rc += sh[2 * nw + i];
  }
}

// The scalar logic at the head of iteration `it` (convergence test of the completed iterations and the new
// direction's coefficient), identical in every workgroup; workgroup 0 records it.  pa/pb: partials (or the single
// all-reduced values) of <r,z> and of the test norm.  Returns false when an EARLIER launch has stopped the solve
// (this one must do nothing).  sh: >= blockDim.x/64 doubles of LDS.
struct DirScalars
{
  double rz, bprev;
  int conv;
};
__device__ inline bool cg_direction_scalars(CgState* __restrict__ st, double* __restrict__ beta_hist,
                                            double* __restrict__ dp_hist, int it, const CgParams& P,
                                            const double* __restrict__ pa, const double* __restrict__ pb, int np, double* sh,
                                            DirScalars& S)
{
  // every input of the scalar logic is requested up-front (uniform loads, one round trip): the stop word, the state,
  // last iteration's <r,z>, and -- communicator attached, np == 1 -- the two all-reduced sums themselves
  const int c = __atomic_load_n(&st->conv_it1, __ATOMIC_RELAXED);
  const double dp0_st = st->dp0, ttol_st = st->ttol;
  const double bprev_h = (it == 0) ? 1.0 : beta_hist[it - 1];
  double rz = pa[0], nn = pb[0], unused;
  // stopped by an EARLIER launch (c - 1 < it): that word was written before this launch began, so every wavefront
  // reads the same value and the branch is uniform without a broadcast
  if (c != 0 && c - 1 < it)
    return false;
  if (np != 1)
    reduce_parts3_bcast(pa, pb, nullptr, np, rz, nn, unused);
  // scalar logic, identical in every workgroup; workgroup 0 records it
  double dp, dp0 = dp0_st, ttol = ttol_st;
  int conv = 0;
  if (P.variant == ZZZ_CG_CGH)
  {
    // src/cg.h:53-55,74-79: rnorm = <r,r>; break when rnorm/rnorm0 < rtol^2 (strict), no test at k = 0
    dp = rz;
    if (it == 0)
    {
      dp0 = rz;
      ttol = P.rtol * P.rtol;
    }
    else if (rz / dp0 < P.rtol * P.rtol)
      conv = 1;
  }
  else
  {
    dp = (P.norm == ZZZ_NORM_NATURAL) ? sqrt(fabs(rz)) : sqrt(nn);
    if (it == 0)
    {
      dp0 = dp;
      ttol = fmax(P.rtol * dp, P.atol);
    }
    if (!isfinite(dp))
      conv = 2;
    else if (dp <= ttol) // KSPConvergedDefault
      conv = 1;
    else if (dp >= P.dtol * dp0) // ... KSP_DIVERGED_DTOL
      conv = 3;
  }
  const double bprev = bprev_h;
  if (blockIdx.x == 0 && threadIdx.x == 0)
  {
    beta_hist[it] = rz;
    dp_hist[it] = dp;
    st->dp = dp;
    if (it == 0)
    {
      st->dp0 = dp0;
      st->ttol = ttol;
    }
    if (conv)
    {
      st->iters = it;
      st->converged = conv; // the other workgroups reach the same verdict from the same partials
      __atomic_store_n(&st->conv_it1, it + 1, __ATOMIC_RELAXED);
    }
  }
  S.rz = rz;
  S.bprev = bprev;
  S.conv = conv;
  return true;
}
} // namespace zzz

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
  {
    __shared__ int flag;
    if (threadIdx.x == 0)
    {
      const int c = __atomic_load_n(&st->conv_it1, __ATOMIC_RELAXED);
      flag = c != 0 && c - 1 < it;
    }
    __syncthreads();
    if (flag)
      return false;
  }
  const double rz = reduce_parts_bcast(pa, np, sh);
  const double nn = reduce_parts_bcast(pb, np, sh);
  // scalar logic, identical in every workgroup; workgroup 0 records it
  double dp, dp0 = st->dp0, ttol = st->ttol;
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
  const double bprev = (it == 0) ? 1.0 : beta_hist[it - 1];
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

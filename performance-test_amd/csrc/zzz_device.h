// Device helpers shared by the kernel files.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace zzz
{
// Sum over the workgroup; the result is valid in thread 0.  Fixed tree => reproducible.
__device__ inline double block_reduce_sum(double v, double* sh /* >= blockDim.x/64 doubles of LDS */)
{
#pragma unroll
  for (int o = 32; o > 0; o >>= 1)
    v += __shfl_down(v, o, 64);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0)
    sh[wv] = v;
  __syncthreads();
  double s = 0;
  if (threadIdx.x == 0)
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i)
      s += sh[i];
  return s;
}
} // namespace zzz

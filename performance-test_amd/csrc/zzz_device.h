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

// XCD-aware work mapping for one-shot grids: workgroups are dealt round-robin over the 8 XCDs
// (blockIdx % 8 shares an XCD, MI355X_MICROARCH.md "Workgroup dispatch"), so give XCD x the x-th
// contiguous eighth of the `n` work items: neighbouring items then share that XCD's L2.  Launch
// with xcd_grid(n) workgroups; returns -1 for the few surplus ones.  Placement only changes speed.
__host__ __device__ inline int64_t xcd_grid(int64_t n) { return (n + 7) / 8 * 8; }
__device__ inline int64_t xcd_item(int64_t n)
{
  const int xcd = blockIdx.x & 7;
  const int64_t lo = n * xcd / 8, hi = n * (xcd + 1) / 8;
  const int64_t t = lo + (blockIdx.x >> 3);
  return t < hi ? t : -1;
}
// The same for persistent (grid-stride) kernels: item of step i for this workgroup, -1 when its XCD's eighth is
// exhausted.  Launch with a multiple of 8 workgroups.
__device__ inline int64_t xcd_stride_item(int64_t n, int i)
{
  const int b = blockIdx.x, nb = gridDim.x;
  const int xcd = b & 7;
  const int64_t lo = n * xcd / 8, hi = n * (xcd + 1) / 8;
  const int wg_in_xcd = b >> 3, n_in_xcd = (nb + 7 - xcd) >> 3;
  const int64_t t = lo + wg_in_xcd + (int64_t)i * n_in_xcd;
  return t < hi ? t : -1;
}
} // namespace zzz

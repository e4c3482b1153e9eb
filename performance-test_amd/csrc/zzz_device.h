// Device helpers shared by the kernel files.
#pragma once
#include <climits>
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

// wave-wide minimum in every lane's hands (a scalar): an inclusive min-scan inside the rows of 16 lanes with DPP
// row shifts, two row broadcasts, then lane 63 read out -- thirteen vector instructions, no LDS crossbar
__device__ inline int wave_min_i(int v)
{
  v = min(v, __builtin_amdgcn_update_dpp(INT_MAX, v, 0x111, 0xf, 0xf, false)); // row_shr:1
  v = min(v, __builtin_amdgcn_update_dpp(INT_MAX, v, 0x112, 0xf, 0xf, false)); // row_shr:2
  v = min(v, __builtin_amdgcn_update_dpp(INT_MAX, v, 0x114, 0xf, 0xf, false)); // row_shr:4
  v = min(v, __builtin_amdgcn_update_dpp(INT_MAX, v, 0x118, 0xf, 0xf, false)); // row_shr:8
  v = min(v, __builtin_amdgcn_update_dpp(INT_MAX, v, 0x142, 0xa, 0xf, false)); // row_bcast:15 into rows 1, 3
  v = min(v, __builtin_amdgcn_update_dpp(INT_MAX, v, 0x143, 0xc, 0xf, false)); // row_bcast:31 into rows 2, 3
  return __builtin_amdgcn_readlane(v, 63);
}
__device__ inline int wave_max_i(int v) { return -wave_min_i(-v); } // |v| < 2^31 - 1 here

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

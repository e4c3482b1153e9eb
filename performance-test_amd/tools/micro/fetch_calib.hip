// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 (verdict item 6): kernels that move a KNOWN number
// of bytes from a buffer larger than every cache, each exactly once, in the access shapes of this library's kernels.  Run under
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- ./fetch_calib
// (and once more with WRITE_SIZE); performance-test_amd/tools/r06/fetch_calib.sh turns the counter rows into bytes-per-unit factors.
//   hipcc -O3 --offload-arch=gfx950 -o fetch_calib fetch_calib.hip
//   k_stream16   : every lane one 16-B load, lanes contiguous (the CG vector kernels, the code streams of the products)
//   k_stream16nt : the same with non-temporal loads
//   k_stream8    : 8-B loads, lanes contiguous
//   k_pair16u    : 16-B loads at 8-B alignment, lane l at byte 16 l + 8 of its run (spmv_one_kernel's x loads)
//   k_rec24      : 24-B records as a 16-B + an 8-B load per lane (spmv_blk3_kernel's x loads)
//   k_write16    : 16-B stores, lanes contiguous (WRITE_SIZE)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double dbl2 __attribute__((ext_vector_type(2)));
typedef double dbl2u __attribute__((ext_vector_type(2), aligned(8)));

__global__ __launch_bounds__(256) void k_stream16(const dbl2* __restrict__ a, size_t n, double* __restrict__ out)
{
  double s = 0.0;
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull)
  {
    const dbl2 v = a[i];
    s += v.x + v.y;
  }
  if (s == 1.2345)
    out[0] = s;
}
__global__ __launch_bounds__(256) void k_stream16nt(const dbl2* __restrict__ a, size_t n, double* __restrict__ out)
{
  double s = 0.0;
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull)
  {
    const dbl2 v = __builtin_nontemporal_load(a + i);
    s += v.x + v.y;
  }
  if (s == 1.2345)
    out[0] = s;
}
__global__ __launch_bounds__(256) void k_stream8(const double* __restrict__ a, size_t n, double* __restrict__ out)
{
  double s = 0.0;
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull)
    s += a[i];
  if (s == 1.2345)
    out[0] = s;
}
// n = doubles; every wavefront reads runs of 128 doubles starting one double into a 129-double stride: 16-B loads, 8-B aligned
__global__ __launch_bounds__(256) void k_pair16u(const double* __restrict__ a, size_t nruns, double* __restrict__ out)
{
  const int lane = threadIdx.x & 63;
  double s = 0.0;
  for (size_t r = blockIdx.x * 4ull + (threadIdx.x >> 6); r < nruns; r += gridDim.x * 4ull)
  {
    const dbl2u v = *reinterpret_cast<const dbl2u*>(a + r * 129 + 1 + 2 * lane);
    s += v.x + v.y;
  }
  if (s == 1.2345)
    out[0] = s;
}
__global__ __launch_bounds__(256) void k_rec24(const double* __restrict__ a, size_t nrec, double* __restrict__ out)
{
  double s = 0.0;
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < nrec; i += gridDim.x * 256ull)
  {
    const dbl2u v = *reinterpret_cast<const dbl2u*>(a + 3 * i);
    s += v.x + v.y + a[3 * i + 2];
  }
  if (s == 1.2345)
    out[0] = s;
}
__global__ __launch_bounds__(256) void k_write16(dbl2* __restrict__ a, size_t n)
{
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull)
    a[i] = dbl2{1.0, 2.0};
}

int main()
{
  const size_t bytes = 2ull << 30; // 2 GiB: eight times the Infinity Cache
  double *a, *out;
  if (hipMalloc(&a, bytes + 4096) != hipSuccess || hipMalloc(&out, 64) != hipSuccess)
    return 1;
  hipMemset(a, 0, bytes + 4096);
  const size_t nd = bytes / 8;
  const dim3 g(256 * 8), b(256);
  for (int rep = 0; rep < 3; ++rep)
  {
    hipLaunchKernelGGL(k_stream16, g, b, 0, 0, (const dbl2*)a, nd / 2, out);
    hipLaunchKernelGGL(k_stream16nt, g, b, 0, 0, (const dbl2*)a, nd / 2, out);
    hipLaunchKernelGGL(k_stream8, g, b, 0, 0, a, nd, out);
    hipLaunchKernelGGL(k_pair16u, g, b, 0, 0, a, nd / 129, out);
    hipLaunchKernelGGL(k_rec24, g, b, 0, 0, a, nd / 3, out);
    hipLaunchKernelGGL(k_write16, g, b, 0, 0, (dbl2*)a, nd / 2);
  }
  hipDeviceSynchronize();
  printf("bytes moved per launch: k_stream16 %zu k_stream16nt %zu k_stream8 %zu k_pair16u %zu k_rec24 %zu k_write16 %zu\n", bytes, bytes, bytes,
         (nd / 129) * 128 * 8, (nd / 3) * 24, bytes);
  return 0;
}

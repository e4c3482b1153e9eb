// Microbenchmark: rocprim::radix_sort_pairs(int, int) on 237 M pairs with 24-bit keys under different onesweep
// configurations (digit width, items per thread).  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 sort_cfg.hip -o sort_cfg
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void k_keys(int* k, int* v, long n, int nb)
{
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
  {
    // connectivity-like keys: cell c touches dofs near c/6 (+ offsets of a 217 x 207 lattice)
    const long c = i / 4;
    const int j = (int)(i & 3);
    const long base = c / 6;
    const int off[4] = {0, 1, 217, 217 * 207};
    long d = base + off[(j + (int)(c % 6)) & 3];
    k[i] = (int)(d % nb);
    v[i] = (int)c;
  }
}

template <class Config>
static int run(const char* name, int* kin, int* kout, int* vin, int* vout, size_t n, int bits)
{
  size_t tb = 0;
  CK((rocprim::radix_sort_pairs<Config>(nullptr, tb, kin, kout, vin, vout, n, 0, bits, 0)));
  void* tmp = nullptr;
  CK(hipMalloc(&tmp, tb));
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  float best = 1e9;
  for (int r = 0; r < 5; ++r)
  {
    hipEventRecord(a, 0);
    CK((rocprim::radix_sort_pairs<Config>(tmp, tb, kin, kout, vin, vout, n, 0, bits, 0)));
    hipEventRecord(b, 0);
    CK(hipEventSynchronize(b));
    float ms;
    hipEventElapsedTime(&ms, a, b);
    if (ms < best)
      best = ms;
  }
  printf("%-40s %8.3f ms  (temp %.0f MB)\n", name, best, tb / 1e6);
  hipFree(tmp);
  return 0;
}

using rocprim::kernel_config;
using rocprim::block_radix_rank_algorithm;
template <int BS, int IPT, int BITS, block_radix_rank_algorithm A>
using Cfg = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                       rocprim::radix_sort_onesweep_config<kernel_config<BS, IPT>, kernel_config<BS, IPT>, BITS, A>>;
template <int HB, int HI>
using CfgH = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                        rocprim::radix_sort_onesweep_config<kernel_config<HB, HI>, kernel_config<1024, 8>, 8,
                                                                            block_radix_rank_algorithm::match>>;

int main()
{
  const size_t n = 237074688;
  const int nb = 10016937;
  int *kin, *kout, *vin, *vout;
  CK(hipMalloc(&kin, n * 4));
  CK(hipMalloc(&kout, n * 4));
  CK(hipMalloc(&vin, n * 4));
  CK(hipMalloc(&vout, n * 4));
  hipLaunchKernelGGL(k_keys, dim3(8192), dim3(256), 0, 0, kin, vin, (long)n, nb);
  CK(hipDeviceSynchronize());
  run<rocprim::default_config>("default (1024x16, 8 bits, match)", kin, kout, vin, vout, n, 24);
  run<Cfg<1024, 8, 8, block_radix_rank_algorithm::match>>("H 1024x8, S 1024x8", kin, kout, vin, vout, n, 24);
  run<CfgH<1024, 16>>("H 1024x16, S 1024x8", kin, kout, vin, vout, n, 24);
  run<CfgH<1024, 24>>("H 1024x24, S 1024x8", kin, kout, vin, vout, n, 24);
  run<CfgH<1024, 32>>("H 1024x32, S 1024x8", kin, kout, vin, vout, n, 24);
  run<CfgH<1024, 12>>("H 1024x12, S 1024x8", kin, kout, vin, vout, n, 24);
  run<CfgH<768, 16>>("H 768x16, S 1024x8", kin, kout, vin, vout, n, 24);
  return 0;
}

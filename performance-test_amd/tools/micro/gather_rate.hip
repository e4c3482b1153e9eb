// Micro-benchmark: what does one vector-memory instruction of the CG product's x gather cost the CU's address / L1 path?
// Every wavefront issues batches of 8 independent loads from a 2 MB (L2-resident) buffer in one of several per-lane address
// patterns; 32 wavefronts per CU keep the path full, so time / (instructions per CU) is the path's cost per instruction.
//   hipcc -O3 --offload-arch=gfx950 -o gather_rate gather_rate.hip && ./gather_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double dbl2 __attribute__((ext_vector_type(2)));

// PAT: 0 dwordx2 contiguous, 64-B aligned run | 1 the run shifted by 8 B | 2 shifted by 56 B | 3 dwordx4 contiguous (1 KiB)
//      4 dwordx2, lane pairs share an address | 5 dword contiguous | 6 dwordx2 stride 16 B | 7 dwordx2 random in 4 KiB
//      8 dwordx4 on lanes 0..31 only (512 B) | 9 dwordx2 one address for all lanes | 10 dwordx2 contiguous, lanes 0..1 only
template <int PAT>
__global__ __launch_bounds__(256) void k_gather(const double* __restrict__ x, long nelem, int iters, double* __restrict__ out)
{
  const int lane = threadIdx.x & 63;
  const unsigned wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const unsigned mask = (unsigned)(nelem / 64 / 4 - 1); // (a power of two: the lower half of the buffer, room for the shifted runs)
  double acc = 0.0;
  for (int it = 0; it < iters; ++it)
  {
    double v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e)
    {
      const unsigned blk = (wave * 131u + (unsigned)it * 17u + e * 3u) & mask;
      const double* b = x + (size_t)blk * 64;
      if (PAT == 0)
        v[e] = b[lane];
      else if (PAT == 1)
        v[e] = b[lane + 1];
      else if (PAT == 2)
        v[e] = b[lane + 7];
      else if (PAT == 3)
      {
        const dbl2 q = reinterpret_cast<const dbl2*>(b)[lane];
        v[e] = q.x + q.y;
      }
      else if (PAT == 4)
        v[e] = b[lane >> 1];
      else if (PAT == 5)
        v[e] = (double)reinterpret_cast<const float*>(b)[lane];
      else if (PAT == 6)
        v[e] = b[2 * lane];
      else if (PAT == 7)
        v[e] = b[(lane * 37 + e * 11) & 511];
      else if (PAT == 8)
      {
        dbl2 q;
        q.x = q.y = 0.0;
        if (lane < 32)
          q = reinterpret_cast<const dbl2*>(b)[lane];
        v[e] = q.x + q.y;
      }
      else if (PAT == 9)
        v[e] = b[0];
      else if (PAT == 10)
        v[e] = lane < 2 ? b[lane * 65] : 0.0;
      else if (PAT == 11)
      {
        dbl2 q;
        __builtin_memcpy(&q, b + 1 + 2 * lane, 16); // 16 B per lane at an address that is 8-B aligned only
        v[e] = q.x + q.y;
      }
      else if (PAT == 12)
      {
        dbl2 q;
        __builtin_memcpy(&q, b + 1 + 2 * (lane & 31) + (lane >> 5) * 4096, 16); // two 512-B runs, 8-B aligned, 32 KB apart
        v[e] = q.x + q.y;
      }
      else
      {
        dbl2 q = reinterpret_cast<const dbl2*>(b + (lane >> 5) * 4096)[lane & 31]; // two 512-B runs, 16-B aligned
        v[e] = q.x + q.y;
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e)
      acc += v[e];
  }
  if (acc == 1.2345)
    out[0] = acc;
}

template <int PAT>
static void run(const char* name, const double* x, long nelem, double* out)
{
  const int grid = 2048, iters = 200;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k_gather<PAT>), dim3(grid), dim3(256), 0, 0, x, nelem, iters, out);
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r)
    hipLaunchKernelGGL((k_gather<PAT>), dim3(grid), dim3(256), 0, 0, x, nelem, iters, out);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double per_cu = 5.0 * grid * 4.0 * iters * 8.0 / 256.0; // instructions per CU
  printf("%-52s %8.2f ns per instruction and CU  (= %5.1f clk at 2.1 GHz)\n", name, ms * 1e6 / per_cu, ms * 1e6 / per_cu * 2.1);
}

int main()
{
  const long nelem = 256 * 1024; // 2 MB
  double *x, *out;
  hipMalloc(&x, nelem * 8);
  hipMemset(x, 0, nelem * 8);
  hipMalloc(&out, 64);
  run<0>("dwordx2, contiguous 512-B run, 64-B aligned", x, nelem, out);
  run<1>("dwordx2, contiguous run shifted by 8 B", x, nelem, out);
  run<2>("dwordx2, contiguous run shifted by 56 B", x, nelem, out);
  run<3>("dwordx4, contiguous 1 KiB", x, nelem, out);
  run<4>("dwordx2, lane pairs share an address (256 B)", x, nelem, out);
  run<5>("dword, contiguous 256 B", x, nelem, out);
  run<6>("dwordx2, stride 16 B (1 KiB span)", x, nelem, out);
  run<7>("dwordx2, random within 4 KiB", x, nelem, out);
  run<8>("dwordx4, lanes 0..31 only (512 B)", x, nelem, out);
  run<9>("dwordx2, one address for every lane", x, nelem, out);
  run<10>("dwordx2, two lanes only", x, nelem, out);
  run<11>("dwordx4, contiguous 1 KiB, 8-B aligned only", x, nelem, out);
  run<12>("dwordx4, two 512-B runs, 8-B aligned only", x, nelem, out);
  run<13>("dwordx4, two 512-B runs, 16-B aligned", x, nelem, out);
  return 0;
}

// Micro-benchmark: the memory pattern of k_sr_update (7 vectors read, 5 written, 16-B accesses) at the 8-GPU per-rank
// size, swept over grid size, entries in flight per thread and workgroup size.  Everything is Infinity-Cache resident
// (12 x 10 MB), as in the CG loop at that size.
//   hipcc -O3 --offload-arch=gfx950 -o vec_sweep vec_sweep.hip && ./vec_sweep [n]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double dbl2 __attribute__((ext_vector_type(2)));

template <int U, int VB>
__global__ __launch_bounds__(VB) void k_upd(const dbl2* __restrict__ d2, const dbl2* __restrict__ s2, dbl2* __restrict__ z2,
                                            dbl2* __restrict__ p2, dbl2* __restrict__ w2, dbl2* __restrict__ x2,
                                            dbl2* __restrict__ r2, long n2, double a, double b)
{
  const long stride = (long)gridDim.x * VB;
  for (long i0 = blockIdx.x * (long)VB + threadIdx.x; i0 < n2; i0 += stride * U)
  {
    dbl2 zi[U], si[U], di[U], xi[U], ri[U], po[U], wo[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
    {
      const long i = i0 + u * stride;
      if (i < n2)
      {
        zi[u] = z2[i]; si[u] = s2[i]; di[u] = d2[i]; xi[u] = x2[i]; ri[u] = r2[i]; po[u] = p2[i]; wo[u] = w2[i];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
    {
      const long i = i0 + u * stride;
      if (i < n2)
      {
        dbl2 pn, wn, zn;
        pn.x = b * po[u].x + zi[u].x; pn.y = b * po[u].y + zi[u].y;
        wn.x = b * wo[u].x + si[u].x; wn.y = b * wo[u].y + si[u].y;
        xi[u].x = a * pn.x + xi[u].x; xi[u].y = a * pn.y + xi[u].y;
        ri[u].x = -a * wn.x + ri[u].x; ri[u].y = -a * wn.y + ri[u].y;
        zn.x = di[u].x * ri[u].x; zn.y = di[u].y * ri[u].y;
        p2[i] = pn; w2[i] = wn; x2[i] = xi[u]; r2[i] = ri[u]; z2[i] = zn;
      }
    }
  }
}

template <int U, int VB>
static float run(int grid, dbl2** v, long n2, int reps)
{
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i)
    hipLaunchKernelGGL((k_upd<U, VB>), dim3(grid), dim3(VB), 0, 0, v[0], v[1], v[2], v[3], v[4], v[5], v[6], n2, 1e-3, 0.5);
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i)
    hipLaunchKernelGGL((k_upd<U, VB>), dim3(grid), dim3(VB), 0, 0, v[0], v[1], v[2], v[3], v[4], v[5], v[6], n2, 1e-3, 0.5);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / reps * 1e3f;
}

int main(int argc, char** argv)
{
  const long n = argc > 1 ? atol(argv[1]) : 1250000;
  const long n2 = n / 2;
  dbl2* v[7];
  for (int i = 0; i < 7; ++i) { hipMalloc(&v[i], n * 8); hipMemset(v[i], 0, n * 8); }
  const int reps = 200;
  printf("n = %ld rows, %.1f MB per launch (time per launch incl. the launch boundary, us)\n", n, 96.0 * n / 1e6);
  const int grids[] = {256, 512, 610, 768, 1024, 1280, 1536, 2048, 2560, 4096};
  printf("%6s %10s %10s %10s %10s %10s %10s\n", "grid", "U1/256", "U2/256", "U4/256", "U1/512", "U2/512", "U1/1024");
  for (int g : grids)
  {
    printf("%6d %10.2f %10.2f %10.2f %10.2f %10.2f %10.2f\n", g, run<1, 256>(g, v, n2, reps), run<2, 256>(g, v, n2, reps),
           run<4, 256>(g, v, n2, reps), run<1, 512>(g / 2 > 0 ? g / 2 : 1, v, n2, reps), run<2, 512>(g / 2 > 0 ? g / 2 : 1, v, n2, reps),
           run<1, 1024>(g / 4 > 0 ? g / 4 : 1, v, n2, reps));
  }
  return 0;
}

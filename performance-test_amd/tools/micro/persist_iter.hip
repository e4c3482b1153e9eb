// Micro-benchmark for the verdict's item 2(i): ONE persistent kernel per batch of CG iterations against the launches it
// would replace, at the sizes where the loop is cache-resident (0.5 M rows = C1, 1.25 M rows = the 8-GPU per-rank share
// of C2).  It moves what a single-reduction Jacobi-CG iteration on the operator stream moves, in the same shapes:
//   phase A ("product"): per row 72 B of stream (16-B loads, one lane per row, 64 rows per wavefront), 7 gathers of x,
//                        s written, three partial sums per workgroup;
//   phase B ("update"):  every workgroup sums all workgroups' partials (the scalar logic), then per row 7 vectors read and
//                        5 written (p, w, x, r, z as in k_sr_update).
// Variant L: two launches per iteration (A, B), back to back on one stream.
// Variant P: one launch per batch; between A and B and between B and the next A a grid barrier -- XCD-hierarchical
//            (a counter per XCD, its last arriver goes to the top counter, releases its XCD through a generation word),
//            release fence before arriving, acquire fence after, every spin bounded.
// The arithmetic is a stand-in (results are not a CG); what is timed is the memory pattern, the launches and the barriers.
//   hipcc -O3 --offload-arch=gfx950 -o persist_iter persist_iter.hip && ./persist_iter
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double dbl2 __attribute__((ext_vector_type(2)));

#define CK(x)                                                                                                          \
  do                                                                                                                   \
  {                                                                                                                    \
    hipError_t e_ = (x);                                                                                               \
    if (e_ != hipSuccess)                                                                                              \
    {                                                                                                                  \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                                                          \
      exit(1);                                                                                                         \
    }                                                                                                                  \
  } while (0)

struct Bar
{
  unsigned* xcnt; // [8 * 32] one counter per XCD, 128 B apart
  unsigned* top;
  unsigned* gen; // [8 * 32]
  int* fail;
};

__device__ inline double block_sum(double v, double* sh)
{
  for (int o = 32; o > 0; o >>= 1)
    v += __shfl_down(v, o, 64);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if (lane == 0)
    sh[wv] = v;
  __syncthreads();
  double t = 0;
  for (int w = 0; w < nw; ++w)
    t += sh[w];
  return t;
}

// phase A for the rows of workgroup `b` of `nb`
__device__ inline void phase_a(const dbl2* __restrict__ stream, const double* __restrict__ z, const double* __restrict__ r,
                               double* __restrict__ s, long n, int b, int nb, double* __restrict__ parts, double* sh)
{
  double rz = 0, zs = 0, nn = 0;
  for (long i = (long)b * blockDim.x + threadIdx.x; i < n; i += (long)nb * blockDim.x)
  {
    // 64 B of "values" + 8 B of "codes" per row, laid out per wavefront as the stream is: 4 x 16 B + 8 B
    const long w0 = (i >> 6) * 64 * 9 / 2; // dbl2 index of the wavefront's block (72 B per row = 4.5 dbl2)
    const int lane = (int)(i & 63);
    dbl2 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      v[j] = stream[w0 + 64 * j + lane];
    const double code = reinterpret_cast<const double*>(stream + w0 + 256)[lane];
    double sum = code * 1e-300;
#pragma unroll
    for (int e = 0; e < 7; ++e)
    {
      long c = i + (e - 3) * (e & 1 ? 1 : 353); // neighbours in the row, a mesh line and a plane away
      c = c < 0 ? 0 : (c >= n ? n - 1 : c);
      sum += ((e & 1) ? v[e >> 1].y : v[e >> 1].x) * z[c];
    }
    s[i] = sum;
    const double zi = z[i], ri = r[i];
    rz += ri * zi;
    zs += zi * sum;
    nn += zi * zi;
  }
  const double a0 = block_sum(rz, sh), a1 = block_sum(zs, sh), a2 = block_sum(nn, sh);
  if (threadIdx.x == 0)
  {
    parts[b] = a0;
    parts[nb + b] = a1;
    parts[2 * nb + b] = a2;
  }
}

__device__ inline void phase_b(const double* __restrict__ parts, int nb, const double* __restrict__ dinv,
                               const double* __restrict__ s, double* __restrict__ z, double* __restrict__ p,
                               double* __restrict__ w, double* __restrict__ x, double* __restrict__ r, long n, int b, double* sh)
{
  double t0 = 0, t1 = 0, t2 = 0;
  for (int k = threadIdx.x; k < nb; k += blockDim.x)
  {
    t0 += parts[k];
    t1 += parts[nb + k];
    t2 += parts[2 * nb + k];
  }
  const double rz = block_sum(t0, sh), zs = block_sum(t1, sh), nn = block_sum(t2, sh);
  const double bb = 1e-3 * rz / (1.0 + nn), aa = 1e-3 * rz / (1.0 + zs * zs);
  const dbl2* d2 = reinterpret_cast<const dbl2*>(dinv);
  const dbl2* s2 = reinterpret_cast<const dbl2*>(s);
  dbl2 *z2 = reinterpret_cast<dbl2*>(z), *p2 = reinterpret_cast<dbl2*>(p), *w2 = reinterpret_cast<dbl2*>(w),
       *x2 = reinterpret_cast<dbl2*>(x), *r2 = reinterpret_cast<dbl2*>(r);
  const long n2 = n >> 1;
  for (long i = (long)b * blockDim.x + threadIdx.x; i < n2; i += (long)gridDim.x * blockDim.x)
  {
    dbl2 zi = z2[i], si = s2[i], di = d2[i], xi = x2[i], ri = r2[i], po = p2[i], wo = w2[i], pn, wn, zn;
    pn.x = bb * po.x + zi.x, pn.y = bb * po.y + zi.y;
    wn.x = bb * wo.x + si.x, wn.y = bb * wo.y + si.y;
    xi.x = aa * pn.x + xi.x, xi.y = aa * pn.y + xi.y;
    ri.x = -aa * wn.x + ri.x, ri.y = -aa * wn.y + ri.y;
    zn.x = di.x * ri.x, zn.y = di.y * ri.y;
    p2[i] = pn, w2[i] = wn, x2[i] = xi, r2[i] = ri, z2[i] = zn;
  }
}

template <int VB>
__global__ __launch_bounds__(VB) void k_a(const dbl2* stream, const double* z, const double* r, double* s, long n, double* parts)
{
  __shared__ double sh[VB / 64];
  phase_a(stream, z, r, s, n, blockIdx.x, gridDim.x, parts, sh);
}
template <int VB>
__global__ __launch_bounds__(VB) void k_b(const double* parts, int nb_a, const double* dinv, const double* s, double* z, double* p,
                                          double* w, double* x, double* r, long n)
{
  __shared__ double sh[VB / 64];
  phase_b(parts, nb_a, dinv, s, z, p, w, x, r, n, blockIdx.x, sh);
}

__device__ inline bool grid_barrier(const Bar B, unsigned epoch, int n_in_xcd, int nxcd)
{
  // the workgroup's stores have reached its XCD's L2 (write-through L1) once every wavefront has waited for them; ONE
  // wavefront per XCD -- the last arriver -- writes that L2 back for the other XCDs (MI355X_MICROARCH.md, barrier-xcd)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __syncthreads();
  __shared__ int bad;
  if (threadIdx.x == 0)
  {
    bad = 0;
    const int xcd = blockIdx.x & 7;
    const unsigned v = __hip_atomic_fetch_add(&B.xcnt[xcd * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    const long long t0 = wall_clock64();
    if (v == epoch * (unsigned)n_in_xcd)
    {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      __hip_atomic_fetch_add(B.top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      while (__hip_atomic_load(B.top, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch * (unsigned)nxcd)
      {
        if (wall_clock64() - t0 > 200000000LL) // 2 s
        {
          bad = 1;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      __hip_atomic_store(&B.gen[xcd * 32], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    else
      while (__hip_atomic_load(&B.gen[xcd * 32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch)
      {
        if (wall_clock64() - t0 > 200000000LL)
        {
          bad = 1;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
    if (bad)
      *B.fail = 1;
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); // other workgroups' stores, not this CU's stale lines
  return bad == 0;
}

template <int VB>
__global__ __launch_bounds__(VB) void k_persist(const dbl2* stream, const double* dinv, double* z, double* s, double* p, double* w,
                                                double* x, double* r, long n, double* parts, Bar B, int iters, unsigned epoch0,
                                                int n_in_xcd, int nxcd)
{
  __shared__ double sh[VB / 64];
  unsigned epoch = epoch0;
  for (int it = 0; it < iters; ++it)
  {
    double* pp = parts + (size_t)(it & 1) * 3 * gridDim.x;
    phase_a(stream, z, r, s, n, blockIdx.x, gridDim.x, pp, sh);
    if (!grid_barrier(B, ++epoch, n_in_xcd, nxcd))
      return;
    phase_b(pp, gridDim.x, dinv, s, z, p, w, x, r, n, blockIdx.x, sh);
    if (!grid_barrier(B, ++epoch, n_in_xcd, nxcd))
      return;
  }
}

template <int VB>
static void run(long n, int wg_per_cu)
{
  int dev = 0;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, dev));
  const int cus = prop.multiProcessorCount;
  double* v[7];
  for (auto& q : v)
  {
    CK(hipMalloc(&q, (n + 64) * sizeof(double)));
    CK(hipMemset(q, 0, (n + 64) * sizeof(double)));
  }
  dbl2* stream;
  const size_t sbytes = (size_t)((n + 63) / 64) * 64 * 72;
  CK(hipMalloc(&stream, sbytes));
  CK(hipMemset(stream, 0, sbytes));
  std::vector<double> ones((size_t)n, 1.0);
  CK(hipMemcpy(v[0], ones.data(), n * sizeof(double), hipMemcpyHostToDevice)); // dinv
  CK(hipMemcpy(v[1], ones.data(), n * sizeof(double), hipMemcpyHostToDevice)); // z
  double* parts;
  CK(hipMalloc(&parts, 2 * 3 * 4096 * sizeof(double)));
  Bar B;
  CK(hipMalloc(&B.xcnt, 8 * 32 * 4));
  CK(hipMalloc(&B.gen, 8 * 32 * 4));
  CK(hipMalloc(&B.top, 4));
  CK(hipMalloc(&B.fail, 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int ITERS = 200;
  // ---- variant L: two launches per iteration (product grid 8 workgroups of 256 per CU as the library's, update grid by rows)
  {
    const int ga = cus * 8, gb = (int)std::min<long>((n / 2 + 255) / 256, (long)cus * 8);
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep)
    {
      CK(hipEventRecord(e0));
      for (int it = 0; it < ITERS; ++it)
      {
        hipLaunchKernelGGL(k_a<256>, dim3(ga), dim3(256), 0, 0, stream, v[1], v[6], v[2], n, parts);
        hipLaunchKernelGGL(k_b<256>, dim3(gb), dim3(256), 0, 0, parts, ga, v[0], v[2], v[1], v[3], v[4], v[5], v[6], n);
      }
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      best = std::min(best, ms);
    }
    printf("n %ld  launches (2 per iteration, %d + %d workgroups of 256): %.2f us per iteration\n", n, ga, gb, 1e3 * best / ITERS);
  }
  // ---- variant P
  {
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_persist<VB>, VB, 0));
    const int per = std::min(wg_per_cu, std::max(1, occ - 1)); // (the occupancy API can be one workgroup per CU high: a margin)
    const int g = cus * per;
    const int nxcd = 8, n_in_xcd = g / 8; // g is a multiple of 8 (256 CUs)
    float best = 1e30f;
    int failed = 0;
    for (int rep = 0; rep < 5 && !failed; ++rep)
    {
      CK(hipMemset(B.xcnt, 0, 8 * 32 * 4));
      CK(hipMemset(B.gen, 0, 8 * 32 * 4));
      CK(hipMemset(B.top, 0, 4));
      CK(hipMemset(B.fail, 0, 4));
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_persist<VB>, dim3(g), dim3(VB), 0, 0, stream, v[0], v[1], v[2], v[3], v[4], v[5], v[6], n, parts, B, ITERS,
                         0u, n_in_xcd, nxcd);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      CK(hipMemcpy(&failed, B.fail, 4, hipMemcpyDeviceToHost));
      best = std::min(best, ms);
    }
    printf("n %ld  persistent (%d workgroups of %d = %d per CU, 2 grid barriers per iteration)%s: %.2f us per iteration\n", n, g, VB,
           per, failed ? " BARRIER TIMED OUT" : "", 1e3 * best / ITERS);
  }
  for (auto q : v)
    CK(hipFree(q));
  CK(hipFree(stream));
  CK(hipFree(parts));
}

int main()
{
  for (long n : {500000L, 1250000L})
  {
    run<256>(n, 1);
    run<256>(n, 2);
    run<256>(n, 4);
    run<512>(n, 2);
    run<1024>(n, 1);
    run<1024>(n, 2);
  }
  return 0;
}

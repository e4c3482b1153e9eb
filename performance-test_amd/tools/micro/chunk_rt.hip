// Round trip of one chunk through emit_chunk / read_chunk of zzz_sellp.hip on synthetic columns (debug aid).
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -I../../csrc -I../../../include -I../../host chunk_rt.hip -o chunk_rt
#include "../../csrc/zzz_sellp.hip"
#include <cstdio>
using namespace zzz;
namespace zzz {
int fail(zzz_ctx*, int c, const char*, ...) { return c; }
int comm_halo_begin(zzz_ctx*, double*) { return 0; }
int comm_halo_end(zzz_ctx*) { return 0; }
int comm_halo_forward(zzz_ctx*, double*) { return 0; }
}
__global__ void k_rt(const int* cols_in, int w, int flags, int nrows, double* svals, uint16_t* c16, int32_t* c32, int32_t* meta,
                     int* cols_out)
{
  const int lane = threadIdx.x;
  double v[8];
  int cl[8];
  for (int e = 0; e < 8; ++e)
  {
    cl[e] = cols_in[lane * 8 + e];
    v[e] = cl[e] == INT_MAX ? 0.0 : 1.0 + e;
  }
  bool gh = false;
  if (flags & 0x10000)
    for (int e = 0; e < 8; ++e)
      v[e] = e == ((flags >> 20) & 7) && cl[e] != INT_MAX ? 1.0 : 0.0;
  emit_chunk(0, w, v, cl, lane, nrows, gh, svals, c16, c32, meta, flags & 0xffff);
  __threadfence();
  dbl2 vv[4];
  int cr[8];
  read_chunk<false, true>(0, w, lane, svals, c16, c32, meta, vv, cr);
  for (int e = 0; e < 8; ++e)
    cols_out[lane * 8 + e] = cr[e];
}
int main()
{
  int h[64 * 8];
  for (int i = 0; i < 64 * 8; ++i)
    h[i] = INT_MAX;
  const int rows[6][8] = {{617, 645, 655, 656, 684, 686, 688, 689}, {617, 646, 647, 654, 656, 685, 687, 689},
                          {646, 647, 654, 655, 684, 686, 687, 688}, {620, 648, 658, 659, 687, 689, 691, 692},
                          {620, 649, 650, 657, 659, 688, 690, 692}, {649, 650, 657, 658, 687, 689, 690, 691}};
  for (int l = 58; l < 64; ++l)
    for (int e = 0; e < 8; ++e)
      h[l * 8 + e] = rows[l - 58][e];
  int *din, *dout, *c32, *meta;
  double* sv;
  uint16_t* c16;
  hipMalloc(&din, sizeof(h));
  hipMalloc(&dout, sizeof(h));
  hipMalloc(&sv, 4096 * 8);
  hipMalloc(&c16, 4096);
  hipMalloc(&c32, 4096 * 4);
  hipMalloc(&meta, 64);
  hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_rt, dim3(1), dim3(64), 0, 0, din, 8, 1 | 4 | (2 << 8), 1323, sv, c16, c32, meta, dout);
  int o[64 * 8], m[8], t[25];
  hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
  hipMemcpy(m, meta, 32, hipMemcpyDeviceToHost);
  hipMemcpy(t, c16, 100, hipMemcpyDeviceToHost);
  printf("meta0 %08x\n", m[0]);
  for (int e = 0; e < 8; ++e)
    printf("T[%d] = %d %d %d\n", e, t[3 * e], t[3 * e + 1], t[3 * e + 2]);
  printf("ph %d\n", t[24]);
  for (int l = 58; l < 64; ++l)
  {
    printf("lane %d:", l);
    for (int e = 0; e < 8; ++e)
      printf(" %d%s", o[l * 8 + e], o[l * 8 + e] == h[l * 8 + e] ? "" : "!");
    printf("\n");
  }
  // the product kernel itself: values select one slot, x[i] = i, so y[lane] = the column that slot decodes to
  int2* desc;
  double *x, *y;
  hipMalloc(&desc, 16);
  hipMalloc(&x, 2000 * 8);
  hipMalloc(&y, 64 * 8);
  double hx[2000];
  for (int i = 0; i < 2000; ++i)
    hx[i] = i;
  hipMemcpy(x, hx, sizeof(hx), hipMemcpyHostToDevice);
  const int2 d0 = make_int2(0, 1 | (8 << 24));
  hipMemcpy(desc, &d0, 8, hipMemcpyHostToDevice);
  for (int e = 0; e < 8; ++e)
  {
    hipLaunchKernelGGL(k_rt, dim3(1), dim3(64), 0, 0, din, 8, 1 | 4 | (2 << 8) | 0x10000 | (e << 20), 1323, sv, c16, c32, meta, dout);
    hipLaunchKernelGGL((spmv_sellp_kernel<false, false, false>), dim3(8), dim3(SP_BLOCK), 0, 0, desc, sv, c16, c32, meta,
                       (const int32_t*)nullptr, x, y, 64, (int64_t)1, (double*)nullptr, (const int*)nullptr,
                       (const int32_t*)nullptr, (int64_t)0, (const double*)nullptr, 0, 0);
    double hy[64];
    hipMemcpy(hy, y, sizeof(hy), hipMemcpyDeviceToHost);
    printf("slot %d:", e);
    for (int l = 58; l < 64; ++l)
      printf(" %g%s", hy[l], hy[l] == h[l * 8 + e] ? "" : "!");
    printf("\n");
  }
  return 0;
}

// Round trip of one 64-row slice (columns from a text file, one row per line) through emit_chunk and the product
// kernel of zzz_sellp.hip: values select entry q, x[i] = i, so y[lane] = the column entry q decodes to (debug aid).
#include "../../csrc/zzz_sellp.hip"
#include <cstdio>
#include <fstream>
#include <sstream>
#include <vector>
using namespace zzz;
namespace zzz {
int fail(zzz_ctx*, int c, const char*, ...) { return c; }
int comm_halo_begin(zzz_ctx*, double*) { return 0; }
int comm_halo_end(zzz_ctx*) { return 0; }
int comm_halo_forward(zzz_ctx*, double*) { return 0; }
}
__global__ void k_emit(const int* cols, const int* cnt, int nch, int wl, int qsel, int flags, int nrows, double* svals,
                       uint16_t* c16, int32_t* c32, int32_t* meta, int maxlen)
{
  const int lane = threadIdx.x;
  bool gh = false;
  for (int j = 0; j < nch; ++j)
  {
    double v[8];
    int cl[8];
    for (int e = 0; e < 8; ++e)
    {
      const int q = 8 * j + e;
      const bool has = q < cnt[lane];
      cl[e] = has ? cols[lane * maxlen + q] : INT_MAX;
      v[e] = has && q == qsel ? 1.0 : 0.0;
    }
    emit_chunk(j, j + 1 < nch ? 8 : wl, v, cl, lane, nrows, gh, svals, c16, c32, meta, flags);
  }
}
int main(int argc, char** argv)
{
  std::ifstream in(argv[1]);
  std::vector<std::vector<int>> rows;
  std::string line;
  while (std::getline(in, line))
  {
    std::istringstream ss(line);
    std::vector<int> r;
    int x;
    while (ss >> x)
      r.push_back(x);
    rows.push_back(r);
  }
  const int first_row = atoi(argv[2]), nrows = atoi(argv[3]);
  int maxlen = 0;
  for (auto& r : rows)
    maxlen = std::max(maxlen, (int)r.size());
  const int nch = (maxlen + 7) / 8, wl = maxlen - 8 * (nch - 1);
  std::vector<int> hc(64 * maxlen, 0), hn(64, 0);
  for (int l = 0; l < 64 && l < (int)rows.size(); ++l)
  {
    hn[l] = (int)rows[l].size();
    for (int q = 0; q < hn[l]; ++q)
      hc[l * maxlen + q] = rows[l][q];
  }
  int *dc, *dn, *c32, *meta;
  double *sv, *x, *y;
  uint16_t* c16;
  int2* desc;
  hipMalloc(&dc, hc.size() * 4);
  hipMalloc(&dn, 256);
  hipMalloc(&sv, (size_t)nch * 4096 + 4096);
  hipMalloc(&c16, (size_t)nch * 1024 + 1024);
  hipMalloc(&c32, (size_t)nch * 2048 + 2048);
  hipMalloc(&meta, (size_t)nch * 32 + 32);
  hipMalloc(&x, 8 * 4000);
  hipMalloc(&y, 8 * 64);
  hipMalloc(&desc, 16);
  hipMemcpy(dc, hc.data(), hc.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dn, hn.data(), 256, hipMemcpyHostToDevice);
  std::vector<double> hx(4000);
  for (int i = 0; i < 4000; ++i)
    hx[i] = i;
  hipMemcpy(x, hx.data(), 8 * 4000, hipMemcpyHostToDevice);
  const int2 d0 = make_int2(0, nch | (wl << 24));
  hipMemcpy(desc, &d0, 8, hipMemcpyHostToDevice);
  const int flags = 1 | 4 | ((first_row % 3) << 8);
  int bad = 0;
  for (int q = 0; q < maxlen; ++q)
  {
    hipLaunchKernelGGL(k_emit, dim3(1), dim3(64), 0, 0, dc, dn, nch, wl, q, flags, nrows, sv, c16, c32, meta, maxlen);
    hipLaunchKernelGGL((spmv_sellp_kernel<false, false, false>), dim3(8), dim3(SP_BLOCK), 0, 0, desc, sv, c16, c32, meta,
                       (const int32_t*)nullptr, x, y, 64, (int64_t)1, (double*)nullptr, (const int*)nullptr,
                       (const int32_t*)nullptr, (int64_t)0, (const double*)nullptr, 0, 0);
    double hy[64];
    hipMemcpy(hy, y, sizeof(hy), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l)
      if (q < hn[l] && hy[l] != hc[l * maxlen + q])
      {
        printf("entry %d lane %d: decoded %g expected %d\n", q, l, hy[l], hc[l * maxlen + q]);
        ++bad;
      }
  }
  std::vector<int> hm(nch * 8);
  hipMemcpy(hm.data(), meta, nch * 32, hipMemcpyDeviceToHost);
  for (int j = 0; j < nch; ++j)
    printf("chunk %d mode %08x\n", j, hm[j * 8]);
  std::vector<int> ht(nch * 256);
  hipMemcpy(ht.data(), c16, nch * 1024, hipMemcpyDeviceToHost);
  for (int j = 0; j < nch; ++j)
    if ((unsigned)hm[j * 8] >= 0xC0000000u)
    {
      printf("chunk %d T:", j);
      for (int i = 0; i < 25; ++i)
        printf(" %d", ht[j * 256 + i]);
      printf("\n");
    }
  printf("nch %d wl %d bad %d\n", nch, wl, bad);
  return 0;
}

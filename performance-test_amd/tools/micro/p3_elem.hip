// Micro-benchmark: the P3 element kernel of the matrix-free action (y_e = sum_q D(q)^T G D(q) u_e, factorised tables of
// element_tables.inc) alone, one cell per lane, inputs and outputs dense in memory, R passes per cell so that the time of
// one pass is a difference -- can the table entries come through the scalar cache instead of LDS broadcast reads?
//   A  two copies in LDS, broadcast reads (what k_mf_action does).  NOT representative here: outside the action kernel the
//      compiler hoists the LDS reads and spills 1 500 vector registers; in k_mf_action the same code takes 168 registers and
//      the element phase 0.085 ms at 1.36 M cells (tools/mf_phases.sh)
//   B  constant memory, the mode loop ROLLED and every entry multiplied (no zero skipped: the mode is a run-time index)
//   C  as B in sections of 20 entries with the next section requested before the current one is used
//   D  constant memory, unrolled, zeros skipped at compile time (the compiler schedules the scalar loads)
// Measured (MI355X, 1 361 886 cells, per pass): B 0.095 ms, C 0.20 ms, D 0.074 ms (941 scalar registers spilled into vector
// lanes) -- none is far enough from the LDS form's 0.085 ms to be worth the kernel; the multiply-adds alone would take 0.03.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -I../../csrc -o p3_elem p3_elem.hip && ./p3_elem [cells]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "element_tables.inc"

constexpr int ND = 20, NQ = 10;
__constant__ double c_tab[600];

constexpr bool nz(int a, int q, int j) { return ZZZ_DTAB_P3[(a * NQ + q) * ND + j] != 0.0; }

__device__ inline void load_cell(const double* __restrict__ u, const double* __restrict__ geom, long ncells, long e, double (&ue)[ND],
                                 double (&G)[6])
{
#pragma unroll
  for (int j = 0; j < ND; ++j)
    ue[j] = u[j * ncells + e];
#pragma unroll
  for (int t = 0; t < 6; ++t)
    G[t] = geom[t * ncells + e];
}

template <int T>
__global__ __launch_bounds__(T, 3) void k_lds(const double* __restrict__ tab, const double* __restrict__ u, const double* __restrict__ geom,
                                              double* __restrict__ y, long ncells, int R)
{
#pragma clang fp contract(fast)
  __shared__ double tab_s[600], tabT_s[600];
  for (int k = threadIdx.x; k < 600; k += T)
  {
    const double v = tab[k];
    const int a = k / (NQ * ND), q = (k / ND) % NQ, j = k % ND;
    tab_s[k] = v;
    tabT_s[(q * ND + j) * 3 + a] = v;
  }
  __syncthreads();
  for (long e = blockIdx.x * (long)T + threadIdx.x; e < ncells; e += (long)gridDim.x * T)
  {
    double ue[ND], G[6], ye[ND];
    load_cell(u, geom, ncells, e, ue, G);
#pragma unroll 1
    for (int rep = 0; rep < R; ++rep)
    {
    if (rep)
    {
#pragma unroll
      for (int j = 0; j < ND; ++j)
        ue[j] += 1e-3 * ye[j];
    }
#pragma unroll
    for (int j = 0; j < ND; ++j)
      ye[j] = 0.0;
#pragma unroll
    for (int q = 0; q < NQ; ++q)
    {
      double g[3];
#pragma unroll
      for (int a = 0; a < 3; ++a)
      {
        double acc = 0.0;
#pragma unroll
        for (int j = 0; j < ND; ++j)
          if (nz(a, q, j))
            acc += tab_s[(a * NQ + q) * ND + j] * ue[j];
        g[a] = acc;
      }
      const double h0 = G[0] * g[0] + G[3] * g[1] + G[4] * g[2];
      const double h1 = G[3] * g[0] + G[1] * g[1] + G[5] * g[2];
      const double h2 = G[4] * g[0] + G[5] * g[1] + G[2] * g[2];
#pragma unroll
      for (int j = 0; j < ND; ++j)
      {
        if (nz(0, q, j))
          ye[j] += tabT_s[(q * ND + j) * 3 + 0] * h0;
        if (nz(1, q, j))
          ye[j] += tabT_s[(q * ND + j) * 3 + 1] * h1;
        if (nz(2, q, j))
          ye[j] += tabT_s[(q * ND + j) * 3 + 2] * h2;
      }
    }
    }
#pragma unroll
    for (int j = 0; j < ND; ++j)
      y[j * ncells + e] = ye[j];
  }
}

// B: rolled mode loop, sections of 20 entries from constant memory
template <int T, bool PREFETCH>
__global__ __launch_bounds__(T, 3) void k_smem(const double* __restrict__ u, const double* __restrict__ geom, double* __restrict__ y,
                                               long ncells, int R)
{
#pragma clang fp contract(fast)
  for (long e = blockIdx.x * (long)T + threadIdx.x; e < ncells; e += (long)gridDim.x * T)
  {
    double ue[ND], G[6], ye[ND];
    load_cell(u, geom, ncells, e, ue, G);
#pragma unroll 1
    for (int rep = 0; rep < R; ++rep)
    {
    if (rep)
    {
#pragma unroll
      for (int j = 0; j < ND; ++j)
        ue[j] += 1e-3 * ye[j];
    }
#pragma unroll
    for (int j = 0; j < ND; ++j)
      ye[j] = 0.0;
    if constexpr (!PREFETCH)
    {
#pragma unroll 1
      for (int q = 0; q < NQ; ++q)
      {
        double g[3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
        {
          const double* __restrict__ row = c_tab + (a * NQ + q) * ND;
          double acc = 0.0;
#pragma unroll
          for (int j = 0; j < ND; ++j)
            acc += row[j] * ue[j];
          g[a] = acc;
        }
        const double h[3] = {G[0] * g[0] + G[3] * g[1] + G[4] * g[2], G[3] * g[0] + G[1] * g[1] + G[5] * g[2],
                             G[4] * g[0] + G[5] * g[1] + G[2] * g[2]};
#pragma unroll
        for (int a = 0; a < 3; ++a)
        {
          const double* __restrict__ row = c_tab + (a * NQ + q) * ND;
#pragma unroll
          for (int j = 0; j < ND; ++j)
            ye[j] += row[j] * h[a];
        }
      }
    }
    else
    {
      // sections s = 0 .. 6 NQ - 1: mode q = s / 6, (s % 6) < 3: forward with direction a = s % 6, else backward with a = s % 6 - 3;
      // the rows of a mode are used in the order a = 0, 1, 2, 0, 1, 2
      double cur[ND], nxt[ND];
#pragma unroll
      for (int j = 0; j < ND; ++j)
        cur[j] = c_tab[j];
      double g[3] = {0, 0, 0}, h[3] = {0, 0, 0};
#pragma unroll 1
      for (int s = 0; s < 6 * NQ; ++s)
      {
        const int q = s / 6, k = s - 6 * q, a = k < 3 ? k : k - 3;
        const int s1 = s + 1 < 6 * NQ ? s + 1 : s;
        const int q1 = s1 / 6, k1 = s1 - 6 * q1, a1 = k1 < 3 ? k1 : k1 - 3;
        const double* __restrict__ row1 = c_tab + (a1 * NQ + q1) * ND;
#pragma unroll
        for (int j = 0; j < ND; ++j)
          nxt[j] = row1[j];
        if (k < 3)
        {
          double acc = 0.0;
#pragma unroll
          for (int j = 0; j < ND; ++j)
            acc += cur[j] * ue[j];
          g[a] = acc;
          if (k == 2)
          {
            h[0] = G[0] * g[0] + G[3] * g[1] + G[4] * g[2];
            h[1] = G[3] * g[0] + G[1] * g[1] + G[5] * g[2];
            h[2] = G[4] * g[0] + G[5] * g[1] + G[2] * g[2];
          }
        }
        else
        {
          const double ha = a == 0 ? h[0] : (a == 1 ? h[1] : h[2]);
#pragma unroll
          for (int j = 0; j < ND; ++j)
            ye[j] += cur[j] * ha;
        }
#pragma unroll
        for (int j = 0; j < ND; ++j)
          cur[j] = nxt[j];
      }
    }
    }
#pragma unroll
    for (int j = 0; j < ND; ++j)
      y[j * ncells + e] = ye[j];
  }
}

// D: unrolled, zeros skipped at compile time, entries from constant memory (the compiler schedules the scalar loads)
template <int T>
__global__ __launch_bounds__(T, 3) void k_smem_unrolled(const double* __restrict__ u, const double* __restrict__ geom,
                                                        double* __restrict__ y, long ncells, int R)
{
#pragma clang fp contract(fast)
  for (long e = blockIdx.x * (long)T + threadIdx.x; e < ncells; e += (long)gridDim.x * T)
  {
    double ue[ND], G[6], ye[ND];
    load_cell(u, geom, ncells, e, ue, G);
#pragma unroll 1
    for (int rep = 0; rep < R; ++rep)
    {
    if (rep)
    {
#pragma unroll
      for (int j = 0; j < ND; ++j)
        ue[j] += 1e-3 * ye[j];
    }
#pragma unroll
    for (int j = 0; j < ND; ++j)
      ye[j] = 0.0;
#pragma unroll
    for (int q = 0; q < NQ; ++q)
    {
      double g[3];
#pragma unroll
      for (int a = 0; a < 3; ++a)
      {
        double acc = 0.0;
#pragma unroll
        for (int j = 0; j < ND; ++j)
          if (nz(a, q, j))
            acc += c_tab[(a * NQ + q) * ND + j] * ue[j];
        g[a] = acc;
      }
      const double h0 = G[0] * g[0] + G[3] * g[1] + G[4] * g[2];
      const double h1 = G[3] * g[0] + G[1] * g[1] + G[5] * g[2];
      const double h2 = G[4] * g[0] + G[5] * g[1] + G[2] * g[2];
#pragma unroll
      for (int j = 0; j < ND; ++j)
      {
        if (nz(0, q, j))
          ye[j] += c_tab[(0 * NQ + q) * ND + j] * h0;
        if (nz(1, q, j))
          ye[j] += c_tab[(1 * NQ + q) * ND + j] * h1;
        if (nz(2, q, j))
          ye[j] += c_tab[(2 * NQ + q) * ND + j] * h2;
      }
    }
    }
#pragma unroll
    for (int j = 0; j < ND; ++j)
      y[j * ncells + e] = ye[j];
  }
}

int main(int argc, char** argv)
{
  const long ncells = argc > 1 ? atol(argv[1]) : 1361886;
  double *u, *geom, *y, *tab;
  hipMalloc(&u, ND * ncells * 8);
  hipMalloc(&geom, 6 * ncells * 8);
  hipMalloc(&y, ND * ncells * 8);
  hipMalloc(&tab, 600 * 8);
  std::vector<double> h(ND * ncells);
  for (size_t i = 0; i < h.size(); ++i)
    h[i] = (double)((i * 2654435761u) % 1000) / 1000.0 - 0.5;
  hipMemcpy(u, h.data(), h.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(geom, h.data(), 6 * ncells * 8, hipMemcpyHostToDevice);
  hipMemcpy(tab, ZZZ_DTAB_P3, 600 * 8, hipMemcpyHostToDevice);
  hipMemcpyToSymbol(HIP_SYMBOL(c_tab), ZZZ_DTAB_P3, 600 * 8);
  constexpr int T = 256;
  const int grid = 256 * 3;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  std::vector<double> ref(ND * ncells), out(ND * ncells);
  auto time = [&](const char* name, auto launch, bool check) {
    for (int i = 0; i < 3; ++i)
      launch();
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i)
      launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(out.data(), y, out.size() * 8, hipMemcpyDeviceToHost);
    double d = 0, m = 0;
    if (check)
      for (size_t i = 0; i < out.size(); ++i)
      {
        d = std::max(d, std::abs(out[i] - ref[i]));
        m = std::max(m, std::abs(ref[i]));
      }
    else
      ref = out;
    printf("%-44s %8.4f ms   max diff %.2e of %.2e\n", name, ms / 20, d, m);
  };
  for (int R : {1, 5})
  {
  printf("R = %d element passes per cell\n", R);
  time("A  LDS tables, broadcast reads", [&] { hipLaunchKernelGGL(k_lds<T>, dim3(grid), dim3(T), 0, 0, tab, u, geom, y, ncells, R); }, false);
  time("B  constant memory, rolled modes", [&] { hipLaunchKernelGGL((k_smem<T, false>), dim3(grid), dim3(T), 0, 0, u, geom, y, ncells, R); }, true);
  time("C  ... next section requested ahead", [&] { hipLaunchKernelGGL((k_smem<T, true>), dim3(grid), dim3(T), 0, 0, u, geom, y, ncells, R); }, true);
  time("D  constant memory, unrolled, zeros skipped", [&] { hipLaunchKernelGGL(k_smem_unrolled<T>, dim3(grid), dim3(T), 0, 0, u, geom, y, ncells, R); }, true);
  }
  return 0;
}

// FE assembly on gfx950 (K1-K5 of SURVEY.md 2c): replaces the FFCx tabulate_tensor kernels plus
// fem::assemble_matrix / assemble_vector / set_diagonal / DirichletBC::set
// (src/poisson_problem.cpp:125-157, src/elasticity_problem.cpp:199-231).
//
// Row-gather formulation: one thread owns one scalar matrix row (vector entry) and walks the cells
// incident to its dof in ascending cell order, evaluating only ITS row of each element tensor.
//   * no atomics, no colouring: every CSR value is written exactly once, by one thread, and the
//     per-entry summation order (ascending cell index) is the serial CPU order => reproducible;
//   * the workgroup's CSR segment (values + column indices) lives in LDS while it is accumulated,
//     so the column search never touches HBM, and it leaves as one contiguous coalesced store;
//   * geometry / dof indices of a cell are re-read by its nd owners, but those reads hit L2 (the
//     rows of a workgroup are neighbours); flops are redundant by nd and free (fp64 VALU, no MFMA:
//     the path is bound by the 8 B/nonzero it must write, not by arithmetic).
// Constrained rows and columns are zeroed as the element tensor is produced, and the diagonal of a
// constrained row is set to 1.0 (fem::set_diagonal) by the thread that owns the row.
#include <algorithm>
#include <climits>
#include <cstring>

#include "zzz_device.h"
#include "zzz_internal.h"

#include <rocprim/rocprim.hpp>

namespace zzz
{
constexpr int ASM_BLOCK = 256;
constexpr int ASM_NNZ = 4096; // LDS: 32 KiB values + 16 KiB columns per workgroup ...
constexpr int ASM_NNZ_P1 = 1920;    // scalar P1: ~126 rows per tile for a 128-thread workgroup (one thread per row)
constexpr int ASM_NNZ_SMALL = 2048; // ... or half of that for scalar P1/P2 (short rows): twice the workgroups per CU
                                    // (measured: Poisson P1 6.6 -> 5.6 ms, P2 6.4 -> 4.5 ms; elasticity and P3 lose)
constexpr int ASM_ORD_CAP = 512; // block dofs of a tile that the high-order kernel sorts by cell count
constexpr int ASM_RP_CAP = 255;  // scalar rows of a tile whose offsets the position-based kernel keeps in LDS

struct Geom
{
  double adet;    // |det J|
  double K[3][3]; // J^-1: K[al][a] = dX_al/dx_a
};

__device__ inline void load_cell(const double* __restrict__ x, const int4 v, double p[4][3])
{
  const int vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int k = 0; k < 4; ++k)
  {
    const double* q = x + 3 * (int64_t)vv[k];
    p[k][0] = q[0];
    p[k][1] = q[1];
    p[k][2] = q[2];
  }
}

// P1 through the dof-addressed coordinates (zzz_ctx::xq): dd = the cell's four block dofs
__device__ inline void load_cell_q(const double* __restrict__ xq, const int4 dd, double p[4][3]) { load_cell(xq, dd, p); }
// the same for a row that knows its own vertex (local index li, coordinates own[]): three gathers instead of four.  The
// loads are predicated per lane; where the lanes of a wavefront agree on li (structured feeds) the fourth is not issued.
__device__ inline void load_cell_q3(const double* __restrict__ xq, const int4 dd, int li, const double own[3], double p[4][3])
{
  const int vv[4] = {dd.x, dd.y, dd.z, dd.w};
#pragma unroll
  for (int k = 0; k < 4; ++k)
  {
    if (k != li)
    {
      const double* q = xq + 3 * (int64_t)vv[k];
      p[k][0] = q[0];
      p[k][1] = q[1];
      p[k][2] = q[2];
    }
    else
    {
      p[k][0] = own[0];
      p[k][1] = own[1];
      p[k][2] = own[2];
    }
  }
}

__device__ inline void geometry(const double p[4][3], Geom& G)
{
  double J[3][3];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int al = 0; al < 3; ++al)
      J[a][al] = p[al + 1][a] - p[0][a];
  const double c00 = J[1][1] * J[2][2] - J[1][2] * J[2][1];
  const double c01 = J[1][0] * J[2][2] - J[1][2] * J[2][0];
  const double c02 = J[1][0] * J[2][1] - J[1][1] * J[2][0];
  const double det = J[0][0] * c00 - J[0][1] * c01 + J[0][2] * c02;
  const double id = 1.0 / det;
  G.adet = fabs(det);
  G.K[0][0] = c00 * id;
  G.K[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) * id;
  G.K[0][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) * id;
  G.K[1][0] = -c01 * id;
  G.K[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) * id;
  G.K[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) * id;
  G.K[2][0] = c02 * id;
  G.K[2][1] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) * id;
  G.K[2][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) * id;
}

// physical gradients of the four P1 hat functions: g[i][a] = sum_al K[al][a] * dphi_i/dX_al
__device__ inline void p1_grads(const Geom& G, double g[4][3])
{
#pragma unroll
  for (int a = 0; a < 3; ++a)
  {
    g[1][a] = G.K[0][a];
    g[2][a] = G.K[1][a];
    g[3][a] = G.K[2][a];
    g[0][a] = -(G.K[0][a] + G.K[1][a] + G.K[2][a]);
  }
}

__device__ inline double sel3(double a0, double a1, double a2, int c) { return c == 0 ? a0 : (c == 1 ? a1 : a2); }

__device__ inline int find_pos(const int32_t* __restrict__ c, int len, int32_t col)
{
  int lo = 0, hi = len - 1;
  while (lo < hi)
  {
    const int mid = (lo + hi) >> 1;
    if (c[mid] < col)
      lo = mid + 1;
    else
      hi = mid;
  }
  return lo;
}

// The positions of four columns in a row's sorted columns c[0 .. len): four lower-bound searches side by side, without
// branches -- pos = number of entries below the column, found bit by bit from the top.  (One after the other, as loops
// around find_pos, the four searches of a P1 cell were twenty dependent LDS round trips per (row, cell) pair, half of
// asm_matrix_p1's time; side by side they are five.)
__device__ inline void find_pos4(const int32_t* __restrict__ c, int len, const int (&col)[4], int (&pos)[4])
{
  // byte offsets from c: q = 4 pos; an entry is read at q + 4 step - 4 whether or not it lies in the row (the arrays this is
  // called on are 32 entries longer than the rows they hold) and counts only if it does
  const char* __restrict__ cb = reinterpret_cast<const char*>(c);
  const int end = 4 * len;
  int q[4] = {0, 0, 0, 0};
  int top = 16;
  while (top * 2 <= len) // (not taken on a P1 lattice: 27 columns at most)
    top *= 2;
  for (; top > 16; top >>= 1)
  {
#pragma unroll
    for (int j = 0; j < 4; ++j)
    {
      const int t = q[j] + 4 * top;
      const int v = t <= end ? *reinterpret_cast<const int32_t*>(cb + t - 4) : INT_MAX;
      q[j] = v < col[j] ? t : q[j];
    }
  }
  const bool short_rows = __all(len < 16); // (a P1 lattice of Kuhn simplices: 15 columns per row)
#pragma unroll
  for (int step = 64; step >= 4; step >>= 1) // 16 entries, 8, ... 1
  {
    if (step == 64 && short_rows)
      continue;
    int v[4], t[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
    {
      t[j] = q[j] + step;
      v[j] = *reinterpret_cast<const int32_t*>(cb + q[j] + (step - 4));
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      q[j] = ((int)(t[j] <= end) & (int)(v[j] < col[j])) ? t[j] : q[j];
  }
#pragma unroll
  for (int j = 0; j < 4; ++j)
    pos[j] = q[j] >> 2;
}

// Adjacency transposed in slices of 64 rows: entry a of row (64 s + lane) sits at off[s] + 64 a + lane
// (cell index, -1 = padding) with the dof's local index in that cell beside it.  A wavefront owns a
// slice, so "the a-th cell of my row" is one dense 256-B read instead of 64 reads 96 B apart.
__global__ void k_adjT_slice_len(const int32_t* __restrict__ adj_off, int64_t nb, int64_t nslices, int32_t* __restrict__ slen,
                                 unsigned long long* __restrict__ total)
{
  unsigned long long mine = 0;
  for (int64_t s = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; s <= nslices; s += (int64_t)gridDim.x * blockDim.x)
  {
    int m = 0;
    if (s < nslices)
      for (int64_t r = s * 64; r < min(nb, s * 64 + 64); ++r)
        m = max(m, adj_off[r + 1] - adj_off[r]);
    slen[s] = m * 64;
    mine += (unsigned long long)m * 64ull;
  }
  if (mine)
    atomicAdd(total, mine); // 64-bit size check before the 32-bit scan (one atomic per thread that has slices)
}

__global__ __launch_bounds__(256) void k_adjT_fill(const int32_t* __restrict__ adj_off, const int32_t* __restrict__ adj_cells,
                                                   const int32_t* __restrict__ cell_dofs, int nd, int64_t nb, int64_t nslices,
                                                   const int32_t* __restrict__ off, int32_t* __restrict__ cellT,
                                                   uint8_t* __restrict__ liT)
{
  const int lane = threadIdx.x & 63;
  for (int64_t s = blockIdx.x * 4 + (threadIdx.x >> 6); s < nslices; s += (int64_t)gridDim.x * 4)
  {
    const int o = off[s], len = (off[s + 1] - o) >> 6;
    const int64_t r = s * 64 + lane;
    const int a0 = r < nb ? adj_off[r] : 0, n = r < nb ? adj_off[r + 1] - a0 : 0;
    for (int a = 0; a < len; ++a)
    {
      int32_t c = -1;
      int li = 0;
      if (a < n)
      {
        c = adj_cells[a0 + a];
        const int32_t* cd = cell_dofs + (int64_t)nd * c;
        for (int j = 0; j < nd; ++j)
          if (cd[j] == (int32_t)r)
            li = j;
      }
      cellT[o + a * 64 + lane] = c;
      liT[o + a * 64 + lane] = (uint8_t)li;
    }
  }
}

// iteration over the cells of block dof i through the transposed adjacency
struct AdjIter
{
  const int32_t* cp;
  const uint8_t* lp;
  int len;
  __device__ AdjIter(const int32_t* __restrict__ off, const int32_t* __restrict__ cellT, const uint8_t* __restrict__ liT, int i)
  {
    const int sl = i >> 6, ln = i & 63;
    const int o = off[sl];
    len = (off[sl + 1] - o) >> 6;
    cp = cellT + o + ln;
    lp = liT + o + ln;
  }
  __device__ int cell(int a) const { return cp[a * 64]; } // -1 once the list of this row is exhausted
  __device__ int li(int a) const { return lp[a * 64]; }
};

// ---- matrix, P1, BS = 1 (Poisson a1, src/Poisson.py:31) or 3 (Elasticity a1, src/Elasticity.py:39)
template <int BS, int NNZ, int BLK, int PROBE = 0> // PROBE: timing ablations of the tools build (wrong results)
__global__ __launch_bounds__(BLK) void asm_matrix_p1(const double* __restrict__ xq,
                                                           const int32_t* __restrict__ cell_dofs,
                                                           const int32_t* __restrict__ adjT_off,
                                                           const int32_t* __restrict__ adjT_cells,
                                                           const uint8_t* __restrict__ adj_li,
                                                           const uint8_t* __restrict__ bc,
                                                           const rp_t* __restrict__ rowptr,
                                                           const int32_t* __restrict__ cols, double* __restrict__ vals,
                                                           const int32_t* __restrict__ tiles, int64_t ntiles)
{
  __shared__ double vals_s[NNZ];
  __shared__ int32_t cols_s[NNZ + 32]; // (+ 32: find_pos4 reads past a row's end)
  // (Handing the tiles out in the Morton order of their middle vertex instead of row order -- so that the rows that
  // visit a cell would be in flight on one XCD at about the same time -- was measured in round 4: a tile is a stick of
  // ~126 rows along a mesh line, and the fabric-side fetch counter went UP, 5.1 -> 9.3 GB here and 4.6 -> 12.1 GB in the
  // vector kernel, at 1-6 % more time.  Row order stays.)
  const int64_t tile = xcd_item(ntiles);
  if (tile < 0)
    return;
  const int d0 = tiles[tile], d1 = tiles[tile + 1];
  const int row0 = d0 * BS, row1 = d1 * BS;
  const int64_t s = rowptr[row0];
  const int e = (int)(rowptr[row1] - s); // entries of the tile's CSR segment (fits LDS)
  for (int k = threadIdx.x; k < e; k += BLK)
  {
    cols_s[k] = cols[s + k];
    vals_s[k] = 0.0;
  }
  __syncthreads();
  for (int r = row0 + (int)threadIdx.x; r < row1; r += BLK)
  {
    const int i = r / BS, c = r % BS;
    const int a0 = (int)(rowptr[r] - s), len = (int)(rowptr[r + 1] - rowptr[r]);
    const bool bcr = bc[r] != 0;
    constexpr double Ey = 1.0e6, nu = 0.3; // src/Elasticity.py:12-15
    constexpr double mu = Ey / (2.0 * (1.0 + nu));
    constexpr double lmbda = Ey * nu / ((1.0 + nu) * (1.0 - 2.0 * nu));
    const AdjIter adj(adjT_off, adjT_cells, adj_li, i);
    // Three-stage software pipeline over the row's cells.  A cell needs a chain of three dependent loads (adjacency
    // -> connectivity -> coordinates and BC flags); each link is issued one iteration ahead of the next: while cell a
    // is evaluated the coordinates of a+1, the connectivity of a+2 and the adjacency entry a+3 are in flight, so an
    // iteration waits for one memory round trip, not three (the workgroup's CSR segment in LDS leaves 3-4 wavefronts
    // per SIMD: occupancy alone does not hide the chain).
    // (profiles/r05_pmc_asm_c2.json, 10 M dofs: 296 VALU + 44 SALU + 26 LDS + 17 VMEM instructions per (row, cell) pair,
    // VALU busy 78 % of the kernel's cycles, L2 hit rate 80 %: the kernel is bound by the instructions it issues -- with
    // every gather, the search and the geometry removed (ZZZ_ASM_PROBE=15 of the tools build) it still takes 1.55 of its
    // 2.65 ms: three wavefronts per SIMD, 24 dependent round trips per row.)
    // The walk has no branches: every lane runs the slice's `alen` iterations (the transposed adjacency pads shorter
    // rows with -1), a padding entry loads and evaluates cell 0 and only its update of the row is masked; the
    // coordinate stages alternate between two register sets (DA, DB: no copies of 25 registers per cell).
    struct Conn
    {
      int cell, li;
      int4 dd;
    };
    struct Data
    {
      double p[4][3];
      uint8_t bcj[4 * BS];
    };
    const int alen = adj.len;
    auto adj_at = [&](int a, int& cell, int& li) {
      const int ac = min(a, alen - 1);
      const int cc = adj.cell(ac);
      cell = a < alen ? cc : -1;
      li = adj.li(ac);
    };
    auto conn_at = [&](int cell, int li, Conn& K) {
      K.cell = cell;
      K.li = li;
      K.dd = *reinterpret_cast<const int4*>(cell_dofs + 4 * (int64_t)max(cell, 0));
    };
    auto data_at = [&](const Conn& K, Data& D) {
      if (PROBE & 2) // no coordinate gathers
      {
#pragma unroll
        for (int k = 0; k < 4; ++k)
        {
          D.p[k][0] = (double)(K.dd.x + k * k);
          D.p[k][1] = (double)(K.dd.y - k);
          D.p[k][2] = (double)(K.dd.z + 3 * k);
        }
      }
      else
        load_cell_q(xq, K.dd, D.p); // (skipping the row's own vertex, as asm_vector_p1 does, made this kernel 3 % slower)
      const int dj[4] = {K.dd.x, K.dd.y, K.dd.z, K.dd.w};
      if (PROBE & 4) // no flag gathers
      {
#pragma unroll
        for (int j = 0; j < 4 * BS; ++j)
          D.bcj[j] = 0;
        return;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int d = 0; d < BS; ++d)
          D.bcj[j * BS + d] = bc[dj[j] * BS + d];
    };
    Conn K0, K1;
    Data DA, DB;
    int c2, l2;
    auto iter = [&](int a, const Data& D0, Data& D1) {
      Conn K2;
      int c3, l3;
      data_at(K1, D1);
      conn_at(c2, l2, K2);
      adj_at(a + 3, c3, l3);
      const int li = K0.li;
      const int dofs[4] = {K0.dd.x, K0.dd.y, K0.dd.z, K0.dd.w};
      double g[4][3];
      Geom G;
      if (PROBE & 8) // no geometry
      {
        G.adet = D0.p[0][0];
#pragma unroll
        for (int k = 0; k < 9; ++k)
          G.K[k / 3][k % 3] = D0.p[1 + k / 4][k % 3];
      }
      else
        geometry(D0.p, G);
      p1_grads(G, g);
      const double w = G.adet / 6.0; // reference volume

      double gi[3] = {0, 0, 0};
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (k == li)
        {
          gi[0] = g[k][0];
          gi[1] = g[k][1];
          gi[2] = g[k][2];
        }
      int pos4[4];
      {
        const int col4[4] = {dofs[0] * BS, dofs[1] * BS, dofs[2] * BS, dofs[3] * BS};
        if (PROBE & 1) // no column search
        {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            pos4[j] = (col4[j] + j) & 7;
        }
        else
          find_pos4(cols_s + a0, len, col4, pos4);
      }
      double val[4][BS];
#pragma unroll
      for (int j = 0; j < 4; ++j)
      {
        const double gg = gi[0] * g[j][0] + gi[1] * g[j][1] + gi[2] * g[j][2];
        if (BS == 1)
          val[j][0] = (bcr || D0.bcj[j]) ? 0.0 : w * gg;
        else
        {
          const double gic = sel3(gi[0], gi[1], gi[2], c), gjc = sel3(g[j][0], g[j][1], g[j][2], c);
#pragma unroll
          for (int d = 0; d < BS; ++d)
          {
            // mu (delta_cd g_i.g_j + d_d phi_i d_c phi_j) + lambda d_c phi_i d_d phi_j
            const double v = w * (mu * ((c == d ? gg : 0.0) + gi[d] * gjc) + lmbda * gic * g[j][d]);
            val[j][d] = (bcr || D0.bcj[j * BS + d]) ? 0.0 : v;
          }
        }
      }
      if (K0.cell >= 0) // the cell's four (block) columns are distinct entries of the row: read all, add, write all
      {
        double acc[4][BS];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int d = 0; d < BS; ++d)
            acc[j][d] = vals_s[a0 + pos4[j] + d];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int d = 0; d < BS; ++d)
            vals_s[a0 + pos4[j] + d] = acc[j][d] + val[j][d];
      }
      K0 = K1;
      K1 = K2;
      c2 = c3;
      l2 = l3;
    };
    if (alen > 0)
    {
      {
        int ca, la;
        adj_at(0, ca, la);
        conn_at(ca, la, K0);
        data_at(K0, DA);
        adj_at(1, ca, la);
        conn_at(ca, la, K1);
        adj_at(2, c2, l2);
      }
      int a = 0;
      for (; a + 1 < alen; a += 2)
      {
        iter(a, DA, DB);
        iter(a + 1, DB, DA);
      }
      if (a < alen)
        iter(a, DA, DB);
    }
    if (bcr) // fem::set_diagonal: 1.0 on constrained rows
      vals_s[a0 + find_pos(cols_s + a0, len, r)] = 1.0;
  }
  __syncthreads();
  for (int k = threadIdx.x; k < e; k += BLK)
    vals[s + k] = vals_s[k];
}

// ---- matrix, P1, block size 3 (Elasticity a1, src/Elasticity.py:39), ONE THREAD PER NODE (round 6)
// asm_matrix_p1<3> gives every scalar row a thread: the three rows of a node walk the same cells and each derives the cell's
// geometry, searches the same four block columns and fetches the same connectivity, coordinates and flags -- and a tile of 4 096
// nonzeros holds 91 rows, so 165 of the workgroup's 256 lanes idle (3.85 ms at 1.33 M nodes, C4).  Here a lane owns a NODE: per
// (node, cell) pair one geometry, one search of the node's BLOCK columns (15 per node in LDS instead of 135 scalar columns),
// then the 3 x 4 x 3 entries of the pair, the same expression per entry as before (values bit-identical to asm_matrix_p1<3>).  A
// tile is ~63 nodes = 8 704 nonzeros (68 KiB of LDS: two one-wavefront workgroups per CU); the software pipeline over the
// node's cells (adjacency -> connectivity -> coordinates and flags, one link ahead each) is the one of asm_matrix_p1.
// The three rows of a node must have the same columns (they do: the pattern is built on block dofs).
constexpr int ASM_NNZ_NODE3 = 8704;
template <int NNZ>
__global__ __launch_bounds__(64) void asm_matrix_p1_node3(const double* __restrict__ xq, const int32_t* __restrict__ cell_dofs,
                                                          const int32_t* __restrict__ adjT_off, const int32_t* __restrict__ adjT_cells,
                                                          const uint8_t* __restrict__ adj_li, const uint8_t* __restrict__ bc,
                                                          const rp_t* __restrict__ rowptr, const int32_t* __restrict__ cols,
                                                          double* __restrict__ vals, const int32_t* __restrict__ tiles, int64_t ntiles)
{
  constexpr int BS = 3, BLK = 64;
  __shared__ double vals_s[NNZ];
  __shared__ int32_t bcols_s[NNZ / 9 + 1 + 32]; // block columns of the tile's nodes (+ 32: find_pos4 reads past a row's end)
  const int64_t tile = xcd_item(ntiles);
  if (tile < 0)
    return;
  const int d0 = tiles[tile], d1 = tiles[tile + 1];
  const int64_t s = rowptr[(int64_t)d0 * BS];
  const int e = (int)(rowptr[(int64_t)d1 * BS] - s); // entries of the tile's CSR segment (fits LDS)
  for (int k = threadIdx.x; k < e; k += BLK)
    vals_s[k] = 0.0;
  // block columns: entry q of the tile's block rows = column / 3 of the first scalar row's entry 3 q'
  for (int i = d0 + (int)threadIdx.x; i < d1; i += BLK)
  {
    const int64_t a = rowptr[(int64_t)i * BS];
    const int blen = (int)(rowptr[(int64_t)i * BS + 1] - a) / BS;
    const int b0 = (int)((a - s) / 9);
    for (int k = 0; k < blen; ++k)
      bcols_s[b0 + k] = cols[a + 3 * k] / BS;
  }
  __syncthreads();
  constexpr double Ey = 1.0e6, nu = 0.3; // src/Elasticity.py:12-15
  constexpr double mu = Ey / (2.0 * (1.0 + nu));
  constexpr double lmbda = Ey * nu / ((1.0 + nu) * (1.0 - 2.0 * nu));
  for (int i = d0 + (int)threadIdx.x; i < d1; i += BLK)
  {
    const int64_t ra = rowptr[(int64_t)i * BS];
    const int a0 = (int)(ra - s), len = (int)(rowptr[(int64_t)i * BS + 1] - ra), blen = len / BS;
    const int b0 = a0 / 9;
    bool bcr[BS];
#pragma unroll
    for (int c = 0; c < BS; ++c)
      bcr[c] = bc[(int64_t)i * BS + c] != 0;
    const AdjIter adj(adjT_off, adjT_cells, adj_li, i);
    struct Conn
    {
      int cell, li;
      int4 dd;
    };
    struct Data
    {
      double p[4][3];
      uint8_t bcj[4 * BS];
    };
    const int alen = adj.len;
    auto adj_at = [&](int a, int& cell, int& li) {
      const int ac = min(a, alen - 1);
      const int cc = adj.cell(ac);
      cell = a < alen ? cc : -1;
      li = adj.li(ac);
    };
    auto conn_at = [&](int cell, int li, Conn& K) {
      K.cell = cell;
      K.li = li;
      K.dd = *reinterpret_cast<const int4*>(cell_dofs + 4 * (int64_t)max(cell, 0));
    };
    auto data_at = [&](const Conn& K, Data& D) {
      load_cell_q(xq, K.dd, D.p);
      const int dj[4] = {K.dd.x, K.dd.y, K.dd.z, K.dd.w};
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int d = 0; d < BS; ++d)
          D.bcj[j * BS + d] = bc[dj[j] * BS + d];
    };
    Conn K0, K1;
    Data DA, DB;
    int c2, l2;
    auto iter = [&](int a, const Data& D0, Data& D1) {
      Conn K2;
      int c3, l3;
      data_at(K1, D1);
      conn_at(c2, l2, K2);
      adj_at(a + 3, c3, l3);
      const int li = K0.li;
      const int dofs[4] = {K0.dd.x, K0.dd.y, K0.dd.z, K0.dd.w};
      double g[4][3];
      Geom G;
      geometry(D0.p, G);
      p1_grads(G, g);
      const double w = G.adet / 6.0; // reference volume
      double gi[3] = {0, 0, 0};
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (k == li)
        {
          gi[0] = g[k][0];
          gi[1] = g[k][1];
          gi[2] = g[k][2];
        }
      int pos4[4];
      find_pos4(bcols_s + b0, blen, dofs, pos4);
      if (K0.cell >= 0) // the cell's four block columns are distinct entries of the block row: read all 36, add, write all
      {                 // (one LDS round trip per pair instead of twelve dependent ones: a lane is alone on its SIMD here)
        double acc[4][BS][BS];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int c = 0; c < BS; ++c)
#pragma unroll
            for (int d = 0; d < BS; ++d)
              acc[j][c][d] = vals_s[a0 + c * len + BS * pos4[j] + d];
#pragma unroll
        for (int j = 0; j < 4; ++j)
        {
          const double gg = gi[0] * g[j][0] + gi[1] * g[j][1] + gi[2] * g[j][2];
#pragma unroll
          for (int c = 0; c < BS; ++c)
          {
            const double gic = gi[c], gjc = g[j][c];
#pragma unroll
            for (int d = 0; d < BS; ++d)
            {
              // mu (delta_cd g_i.g_j + d_d phi_i d_c phi_j) + lambda d_c phi_i d_d phi_j
              const double v = w * (mu * ((c == d ? gg : 0.0) + gi[d] * gjc) + lmbda * gic * g[j][d]);
              acc[j][c][d] += (bcr[c] || D0.bcj[j * BS + d]) ? 0.0 : v;
            }
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int c = 0; c < BS; ++c)
#pragma unroll
            for (int d = 0; d < BS; ++d)
              vals_s[a0 + c * len + BS * pos4[j] + d] = acc[j][c][d];
      }
      K0 = K1;
      K1 = K2;
      c2 = c3;
      l2 = l3;
    };
    if (alen > 0)
    {
      {
        int ca, la;
        adj_at(0, ca, la);
        conn_at(ca, la, K0);
        data_at(K0, DA);
        adj_at(1, ca, la);
        conn_at(ca, la, K1);
        adj_at(2, c2, l2);
      }
      int a = 0;
      for (; a + 1 < alen; a += 2)
      {
        iter(a, DA, DB);
        iter(a + 1, DB, DA);
      }
      if (a < alen)
        iter(a, DA, DB);
    }
#pragma unroll
    for (int c = 0; c < BS; ++c)
      if (bcr[c]) // fem::set_diagonal: 1.0 on constrained rows
        vals_s[a0 + c * len + BS * find_pos(bcols_s + b0, blen, i) + c] = 1.0;
  }
  __syncthreads();
  for (int k = threadIdx.x; k < e; k += BLK)
    vals[s + k] = vals_s[k];
}

// ---- vector, P1: Poisson L1 = f v dx + g v ds (src/Poisson.py:32), Elasticity L1 = f.v dx (:40)
// Round 5: two passes. What a (row, cell) pair of the cell term needs from its cell is the same for the cell's four rows:
// |det J| and the sum of the coefficient over the cell's vertices (per component).  k_cell_load_p1 evaluates both once per
// cell (dense, one thread per cell: connectivity, four vertices, four coefficients, one determinant) into a record of
// 1 + BS doubles; the row walk then fetches ONE record per pair -- a chain of two dependent loads (adjacency -> record)
// at 40 registers instead of three (adjacency -> connectivity -> coordinates and coefficients) at 129, no geometry in
// the walk.  The same operations on the same operands in the same order as the one-pass kernel of rounds 1-4 (the sum
// over the vertices was formed in vertex order there too): b is bit-identical.  10 M dofs: 1.85 -> 0.88 + 0.74 ms for the two
// passes; elasticity at 1 M nodes 0.45 -> 0.12 + 0.13 ms.
// The exterior-facet term (cells with a boundary facet: few) is evaluated in the walk as before.
// The record's |det J| carries the cell's "has a boundary facet" bit in its SIGN (|det J| >= 0: the walk takes the absolute value
// and looks the facet mask up only where the sign is set -- one load less per (row, cell) pair).
// (Measured and dropped: vertices as 32-B records {x, y, z, f}, two aligned 16-B loads each, instead of 24-B coordinates plus
// the coefficient -- the pass took the same 0.88-0.90 ms at 10 M dofs: it is bound by what it moves, connectivity in,
// records out and the vertex array once per simplex type of the type-major cell order, not by its 14 loads per cell.)
template <int BS>
__global__ __launch_bounds__(256) void k_cell_load_p1(const double* __restrict__ xq, const int32_t* __restrict__ cell_dofs,
                                                      const double* __restrict__ f, const uint8_t* __restrict__ facet_mask,
                                                      int64_t ncells, double* __restrict__ rec)
{
  // U cells per thread and round, a workgroup's 256 U cells side by side: the U connectivity records are requested together,
  // then the U x 4 vertices and coefficients (two dependent round trips per U cells, not per cell)
  constexpr int U = BS == 1 ? 4 : 1; // (block size 3: twelve coefficient loads per cell already; 0.116 ms at 1, 0.132 at 4)
  auto block = [&](int64_t blk) {
    const int64_t c0 = blk * (256ll * U) + threadIdx.x;
    int4 dd[U];
    unsigned fm[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
    {
      const int64_t c = min(c0 + 256ll * u, ncells - 1);
      dd[u] = *reinterpret_cast<const int4*>(cell_dofs + 4 * c);
      fm[u] = (BS == 1) ? facet_mask[c] : 0u;
    }
    double p[U][4][3], fl[U][4][BS];
#pragma unroll
    for (int u = 0; u < U; ++u)
    {
      const int dj[4] = {dd[u].x, dd[u].y, dd[u].z, dd[u].w};
      load_cell_q(xq, dd[u], p[u]);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int d = 0; d < BS; ++d)
          fl[u][j][d] = f[(int64_t)dj[j] * BS + d];
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
    {
      const int64_t c = c0 + 256ll * u;
      if (c >= ncells)
        break;
      Geom G;
      geometry(p[u], G);
      double* __restrict__ o = rec + c * (BS == 1 ? 2 : 4);
      double out[1 + BS];
      out[0] = (BS == 1 && fm[u]) ? -G.adet : G.adet;
#pragma unroll
      for (int d = 0; d < BS; ++d)
      {
        double fs = 0.0;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          fs += fl[u][j][d];
        out[1 + d] = fs;
      }
      *reinterpret_cast<double2*>(o) = make_double2(out[0], out[1]);
      if (BS != 1)
        *reinterpret_cast<double2*>(o + 2) = make_double2(out[2], out[BS]);
    }
  };
  // The blocks of 256 U cells are walked SIX-WAY INTERLEAVED: block q of the walk is block (q mod 6) nb / 6 + q / 6 of the cell
  // array.  The structured feeds number cells simplex type by simplex type: in array order the pass streamed the vertex array
  // once per type (0.88 ms at 10 M dofs); interleaved, the six types of one region are in flight together and their vertices
  // come out of the caches (0.74 ms).  Any other cell order is just walked in another order.  (A region's six blocks pinned to
  // ONE XCD -- regions x, x + 8, ... on XCD x -- took 0.82 ms: neighbouring regions share vertices too.)
  const int64_t nb = (ncells + 256 * U - 1) / (256 * U), nb6 = nb / 6;
  for (int64_t q = blockIdx.x; q < nb; q += gridDim.x)
    block(q < 6 * nb6 ? (q % 6) * nb6 + q / 6 : q);
}

template <int BS>
__global__ __launch_bounds__(ASM_BLOCK) void asm_vector_p1(const double* __restrict__ xq,
                                                           const int32_t* __restrict__ cell_dofs,
                                                           const int32_t* __restrict__ adjT_off,
                                                           const int32_t* __restrict__ adjT_cells,
                                                           const uint8_t* __restrict__ adj_li,
                                                           const uint8_t* __restrict__ bc,
                                                           const uint8_t* __restrict__ facet_mask,
                                                           const double* __restrict__ f, const double* __restrict__ gc,
                                                           const double* __restrict__ rec, double* __restrict__ b, int64_t nrows)
{
  const int64_t blk = xcd_item((nrows + ASM_BLOCK - 1) / ASM_BLOCK);
  const int64_t r = blk * (int64_t)ASM_BLOCK + threadIdx.x;
  if (blk < 0 || r >= nrows)
    return;
  const int i = (int)(r / BS), c = (int)(r % BS);
  double sum = 0.0;
  const AdjIter adj(adjT_off, adjT_cells, adj_li, i);
  const double own_f = f[r];
  const int alen = adj.len;
  // the walk: the adjacency entry of cell a + 2 and the record (and facet mask) of cell a + 1 are in flight while cell a is
  // added; every lane runs the slice's alen iterations (padding entries: cell 0's record, not added)
  struct Rec
  {
    int cell;
    unsigned mask;
    double adet, fs;
  };
  auto adj_at = [&](int a) -> int {
    const int cc = adj.cell(min(a, alen - 1));
    return a < alen ? cc : -1;
  };
  auto rec_at = [&](int cell, Rec& R) {
    R.cell = cell;
    const int64_t cz = max(cell, 0);
    if (BS == 1)
    {
      const double2 q = *reinterpret_cast<const double2*>(rec + 2 * cz);
      R.adet = fabs(q.x);
      R.fs = q.y;
      R.mask = __builtin_signbit(q.x) ? 1u : 0u; // the cell has a boundary facet: which, is looked up where it is used
    }
    else
    {
      R.adet = rec[4 * cz];
      R.fs = rec[4 * cz + 1 + c];
      R.mask = 0u;
    }
  };
  if (alen > 0)
  {
    Rec R0, R1;
    int c2;
    rec_at(adj_at(0), R0);
    int c1 = adj_at(1);
    for (int a = 0; a < alen; ++a)
    {
      rec_at(c1, R1);
      c2 = adj_at(a + 2);
      const bool live = R0.cell >= 0;
      const double term = R0.adet * ((R0.fs + own_f) / 120.0); // |detJ| * sum_j (1+delta_ij)/120 f_j
      if (live)
        sum += term;
      if (BS == 1)
      {
        if (live && R0.mask)
        {
          // a cell with boundary facets (few): its facet mask, vertices and the boundary coefficient, here
          const unsigned m = facet_mask[R0.cell];
          const int li = adj.li(a);
          const int4 dd = *reinterpret_cast<const int4*>(cell_dofs + 4 * (int64_t)R0.cell);
          const int dofs[4] = {dd.x, dd.y, dd.z, dd.w};
          double p[4][3];
          load_cell_q(xq, dd, p);
          double gl[4], gi = 0.0;
#pragma unroll
          for (int j = 0; j < 4; ++j)
          {
            gl[j] = gc[dofs[j]];
            if (j == li)
              gi = gl[j];
          }
#pragma unroll
          for (int lf = 0; lf < 4; ++lf)
            if (((m >> lf) & 1u) && lf != li)
            {
              // facet lf = the three vertices other than lf
              const int q0 = lf == 0 ? 1 : 0, q1 = lf <= 1 ? 2 : 1, q2 = lf == 3 ? 2 : 3;
              double e1[3], e2[3];
#pragma unroll
              for (int k = 0; k < 3; ++k)
              {
                e1[k] = p[q1][k] - p[q0][k];
                e2[k] = p[q2][k] - p[q0][k];
              }
              const double cx = e1[1] * e2[2] - e1[2] * e2[1], cy = e1[2] * e2[0] - e1[0] * e2[2],
                           cz = e1[0] * e2[1] - e1[1] * e2[0];
              const double scale = sqrt(cx * cx + cy * cy + cz * cz); // 2 * area
              const double gs = gl[q0] + gl[q1] + gl[q2];
              sum += scale * ((gs + gi) / 24.0); // 2 area * sum_j (1+delta_ij)/24 g_j
            }
        }
      }
      R0 = R1;
      c1 = c2;
    }
  }
  b[r] = bc[r] ? 0.0 : sum; // bc->set(b), u0 == 0
}


// ---- generic P2/P3 kernels: the element tensor row is a contraction of per-cell geometry with the
// reference tensors of element_tables.inc, staged in LDS next to the workgroup's CSR segment.
// LPR lanes share one matrix row and split the columns j of each incident cell between them; a
// cell's columns are distinct, the lanes of a row sit in one wavefront and walk the cells in
// lockstep, so the per-entry summation order is still the serial ascending-cell order, no atomics.
#include "element_tables.inc"

template <int ND, int BS, int LPR>
__global__ __launch_bounds__(ASM_BLOCK) void asm_matrix_pk(const double* __restrict__ x,
                                                           const int32_t* __restrict__ cell_verts,
                                                           const int32_t* __restrict__ cell_dofs,
                                                           const int32_t* __restrict__ adjT_off,
                                                           const int32_t* __restrict__ adjT_cells,
                                                           const uint8_t* __restrict__ adj_li,
                                                           const int32_t* __restrict__ adj_off,
                                                           const uint8_t* __restrict__ bc,
                                                           const rp_t* __restrict__ rowptr,
                                                           const int32_t* __restrict__ cols, double* __restrict__ vals,
                                                           const int32_t* __restrict__ tiles, int64_t ntiles,
                                                           const double* __restrict__ tab, int cap)
{
  constexpr int NT = (BS == 1) ? 6 : 9;
  constexpr int NN = ND * ND;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  double* vals_s = reinterpret_cast<double*>(lds_raw);
  double* T_s = vals_s + cap; // cap = nonzeros a tile may hold (asm_tile_nnz)
  int32_t* cols_s = reinterpret_cast<int32_t*>(T_s + NT * NN);
  for (int k = threadIdx.x; k < NT * NN; k += ASM_BLOCK)
  {
    if (BS == 1)
    {
      // symmetrised pieces: G is symmetric, so S^{ab} + S^{ba} is all a < b needs
      const int t = k / NN, ij = k % NN;
      const int a = t < 3 ? t : (t == 3 ? 0 : (t == 4 ? 0 : 1)), b = t < 3 ? t : (t == 3 ? 1 : 2);
      T_s[k] = (t < 3) ? tab[(a * 3 + a) * NN + ij] : tab[(a * 3 + b) * NN + ij] + tab[(b * 3 + a) * NN + ij];
    }
    else
      T_s[k] = tab[k];
  }
  const int64_t tile = xcd_item(ntiles);
  if (tile < 0)
    return;
  const int d0 = tiles[tile], d1 = tiles[tile + 1];
  const int row0 = d0 * BS, row1 = d1 * BS;
  const int64_t s = rowptr[row0];
  const int e = (int)(rowptr[row1] - s); // entries of the tile's CSR segment (fits LDS)
  for (int k = threadIdx.x; k < e; k += ASM_BLOCK)
  {
    cols_s[k] = cols[s + k];
    vals_s[k] = 0.0;
  }
  // Block dofs of the tile in order of descending cell count (counting sort in LDS): with the point-major
  // numbering a tile mixes vertex rows (24 cells) with face rows (2 cells), and a wavefront takes as long as
  // its heaviest row.  Which lane group computes a row changes nothing in the result.
  __shared__ int ord_s[ASM_ORD_CAP];
  __shared__ int hist_s[34];
  const int nt = d1 - d0;
  const bool sorted = nt <= ASM_ORD_CAP;
  if (sorted)
  {
    if (threadIdx.x < 34)
      hist_s[threadIdx.x] = 0;
    __syncthreads();
    for (int k = threadIdx.x; k < nt; k += ASM_BLOCK)
      atomicAdd(&hist_s[32 - min(adj_off[d0 + k + 1] - adj_off[d0 + k], 32)], 1);
    __syncthreads();
    if (threadIdx.x == 0)
    {
      int acc = 0;
      for (int b = 0; b < 33; ++b)
      {
        const int h = hist_s[b];
        hist_s[b] = acc;
        acc += h;
      }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < nt; k += ASM_BLOCK)
      ord_s[atomicAdd(&hist_s[32 - min(adj_off[d0 + k + 1] - adj_off[d0 + k], 32)], 1)] = k;
  }
  __syncthreads();
  const int lane = (int)threadIdx.x % LPR;
  for (int q = (int)threadIdx.x / LPR; q < nt * BS; q += ASM_BLOCK / LPR)
  {
    const int i = d0 + (sorted ? ord_s[q / BS] : q / BS), c = q % BS;
    const int r = i * BS + c;
    const int a0 = (int)(rowptr[r] - s), len = (int)(rowptr[r + 1] - rowptr[r]);
    const bool bcr = bc[r] != 0;
    constexpr double Ey = 1.0e6, nu = 0.3; // src/Elasticity.py:12-15
    constexpr double mu = Ey / (2.0 * (1.0 + nu));
    constexpr double lmbda = Ey * nu / ((1.0 + nu) * (1.0 - 2.0 * nu));
    const AdjIter adj(adjT_off, adjT_cells, adj_li, i);
    // Three-stage software pipeline over the row's cells, as in asm_matrix_p1: the chain adjacency -> connectivity
    // (vertices + this lane's columns) -> coordinates and BC flags is issued one link per iteration ahead.
    constexpr int JM = (ND + LPR - 1) / LPR; // columns of a cell this lane handles
    struct Conn
    {
      int cell, li;
      int4 v;
      int dj[JM];
    };
    struct Data
    {
      double p[4][3];
      uint8_t bcj[JM * BS];
    };
    auto adj_at = [&](int a, int& cell, int& li) {
      const bool in = a < adj.len;
      cell = in ? adj.cell(a) : -1;
      li = in ? adj.li(a) : 0;
    };
    auto conn_at = [&](int cell, int li, Conn& K) {
      K.cell = cell;
      K.li = li;
      if (cell < 0)
        return;
      K.v = *reinterpret_cast<const int4*>(cell_verts + 4 * (int64_t)cell);
      const int32_t* __restrict__ cd = cell_dofs + (int64_t)ND * cell;
#pragma unroll
      for (int q = 0; q < JM; ++q)
        K.dj[q] = lane + q * LPR < ND ? cd[lane + q * LPR] : 0;
    };
    auto data_at = [&](const Conn& K, Data& D) {
      if (K.cell < 0)
        return;
      load_cell(x, K.v, D.p);
#pragma unroll
      for (int q = 0; q < JM; ++q)
#pragma unroll
        for (int d = 0; d < BS; ++d)
          D.bcj[q * BS + d] = bc[K.dj[q] * BS + d];
    };
    Conn K0, K1;
    Data D0;
    int c2, l2;
    {
      int ca, la;
      adj_at(0, ca, la);
      conn_at(ca, la, K0);
      data_at(K0, D0);
      adj_at(1, ca, la);
      conn_at(ca, la, K1);
      adj_at(2, c2, l2);
    }
    for (int a = 0; K0.cell >= 0; ++a)
    {
      Data D1;
      Conn K2;
      int c3, l3;
      data_at(K1, D1);
      conn_at(c2, l2, K2);
      adj_at(a + 3, c3, l3);
      const int li = K0.li;
      Geom G;
      geometry(D0.p, G);
      const double* Tl = T_s + li * ND;
      if (BS == 1)
      {
        // |detJ| (K K^T): 00 11 22 01 02 12
        double GG[6];
        GG[0] = G.adet * (G.K[0][0] * G.K[0][0] + G.K[0][1] * G.K[0][1] + G.K[0][2] * G.K[0][2]);
        GG[1] = G.adet * (G.K[1][0] * G.K[1][0] + G.K[1][1] * G.K[1][1] + G.K[1][2] * G.K[1][2]);
        GG[2] = G.adet * (G.K[2][0] * G.K[2][0] + G.K[2][1] * G.K[2][1] + G.K[2][2] * G.K[2][2]);
        GG[3] = G.adet * (G.K[0][0] * G.K[1][0] + G.K[0][1] * G.K[1][1] + G.K[0][2] * G.K[1][2]);
        GG[4] = G.adet * (G.K[0][0] * G.K[2][0] + G.K[0][1] * G.K[2][1] + G.K[0][2] * G.K[2][2]);
        GG[5] = G.adet * (G.K[1][0] * G.K[2][0] + G.K[1][1] * G.K[2][1] + G.K[1][2] * G.K[2][2]);
#pragma unroll
        for (int q = 0; q < JM; ++q)
        {
          const int j = lane + q * LPR;
          if (j >= ND)
            break;
          const int dj = K0.dj[q];
          const int pos = find_pos(cols_s + a0, len, dj);
          double val = 0.0;
#pragma unroll
          for (int t = 0; t < 6; ++t)
            val += GG[t] * Tl[t * NN + j];
          if (bcr || D0.bcj[q])
            val = 0.0;
          vals_s[a0 + pos] += val;
        }
      }
      else
      {
#pragma unroll
        for (int q = 0; q < JM; ++q)
        {
          const int j = lane + q * LPR;
          if (j >= ND)
            break;
          const int dj = K0.dj[q];
          const int pos = find_pos(cols_s + a0, len, dj * 3);
          // D[cc][d] = |detJ| sum_{al,be} K[al][cc] K[be][d] S^[al][be]_{li,j} = int d_cc phi_i d_d phi_j
          double D[3][3];
          {
            double tmp[3][3];
#pragma unroll
            for (int al = 0; al < 3; ++al)
#pragma unroll
              for (int d = 0; d < 3; ++d)
                tmp[al][d] = Tl[(al * 3 + 0) * NN + j] * G.K[0][d] + Tl[(al * 3 + 1) * NN + j] * G.K[1][d]
                             + Tl[(al * 3 + 2) * NN + j] * G.K[2][d];
#pragma unroll
            for (int cc = 0; cc < 3; ++cc)
#pragma unroll
              for (int d = 0; d < 3; ++d)
                D[cc][d] = G.adet * (G.K[0][cc] * tmp[0][d] + G.K[1][cc] * tmp[1][d] + G.K[2][cc] * tmp[2][d]);
          }
          const double tr = D[0][0] + D[1][1] + D[2][2];
#pragma unroll
          for (int d = 0; d < 3; ++d)
          {
            const double Dcd = sel3(D[0][d], D[1][d], D[2][d], c), Ddc = sel3(D[d][0], D[d][1], D[d][2], c);
            double val = mu * ((c == d ? tr : 0.0) + Ddc) + lmbda * Dcd;
            if (bcr || D0.bcj[q * BS + d])
              val = 0.0;
            vals_s[a0 + pos + d] += val;
          }
        }
      }
      K0 = K1;
      D0 = D1;
      K1 = K2;
      c2 = c3;
      l2 = l3;
    }
    if (bcr && lane == 0) // fem::set_diagonal
      vals_s[a0 + find_pos(cols_s + a0, len, r)] = 1.0;
  }
  __syncthreads();
  for (int k = threadIdx.x; k < e; k += ASM_BLOCK)
    vals[s + k] = vals_s[k];
}

template <int ND, int BS>
__global__ __launch_bounds__(ASM_BLOCK) void asm_vector_pk(const double* __restrict__ x,
                                                           const int32_t* __restrict__ cell_verts,
                                                           const int32_t* __restrict__ cell_dofs,
                                                           const int32_t* __restrict__ adjT_off,
                                                           const int32_t* __restrict__ adjT_cells,
                                                           const uint8_t* __restrict__ adj_li,
                                                           const uint8_t* __restrict__ bc,
                                                           const uint8_t* __restrict__ facet_mask,
                                                           const double* __restrict__ f, const double* __restrict__ gc,
                                                           double* __restrict__ b, int64_t nrows,
                                                           const double* __restrict__ tab)
{
  constexpr int NN = ND * ND;
  __shared__ double M_s[NN];
  __shared__ double F_s[BS == 1 ? 4 * NN : 1];
  for (int k = threadIdx.x; k < NN; k += ASM_BLOCK)
    M_s[k] = tab[9 * NN + k];
  if (BS == 1)
    for (int k = threadIdx.x; k < 4 * NN; k += ASM_BLOCK)
      F_s[k] = tab[10 * NN + k];
  __syncthreads();
  const int64_t blk = xcd_item((nrows + ASM_BLOCK - 1) / ASM_BLOCK);
  const int64_t r = blk * (int64_t)ASM_BLOCK + threadIdx.x;
  if (blk < 0 || r >= nrows)
    return;
  const int i = (int)(r / BS), c = (int)(r % BS);
  double sum = 0.0;
  const AdjIter adj(adjT_off, adjT_cells, adj_li, i);
  for (int a = 0; a < adj.len; ++a)
  {
    const int cell = adj.cell(a);
    if (cell < 0)
      break;
    const int li = adj.li(a);
    const int4 v = *reinterpret_cast<const int4*>(cell_verts + 4 * (int64_t)cell);
    const int32_t* __restrict__ cd = cell_dofs + (int64_t)ND * cell;
    double p[4][3];
    Geom G;
    load_cell(x, v, p);
    geometry(p, G);
    // (the cell's dofs, then the coefficient at them, as two rounds of loads side by side; as a loop over j it was ND dependent
    // pairs of round trips)
    int32_t cdv[ND];
#pragma unroll
    for (int j = 0; j < ND; ++j)
      cdv[j] = cd[j];
    double fv[ND];
#pragma unroll
    for (int j = 0; j < ND; ++j)
      fv[j] = f[(int64_t)cdv[j] * BS + c];
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < ND; ++j)
      acc += M_s[li * ND + j] * fv[j];
    sum += G.adet * acc;
    if (BS == 1)
    {
      const unsigned m = facet_mask[cell];
      if (m)
      {
#pragma unroll
        for (int lf = 0; lf < 4; ++lf)
          if ((m >> lf) & 1u)
          {
            const int q0 = lf == 0 ? 1 : 0, q1 = lf <= 1 ? 2 : 1, q2 = lf == 3 ? 2 : 3;
            double e1[3], e2[3];
#pragma unroll
            for (int k = 0; k < 3; ++k)
            {
              e1[k] = p[q1][k] - p[q0][k];
              e2[k] = p[q2][k] - p[q0][k];
            }
            const double cx = e1[1] * e2[2] - e1[2] * e2[1], cy = e1[2] * e2[0] - e1[0] * e2[2],
                         cz = e1[0] * e2[1] - e1[1] * e2[0];
            const double scale = sqrt(cx * cx + cy * cy + cz * cz);
            double fa = 0.0;
            for (int j = 0; j < ND; ++j)
              fa += F_s[(lf * ND + li) * ND + j] * gc[cdv[j]]; // rows of dofs off the facet are zero
            sum += scale * fa;
          }
      }
    }
  }
  b[r] = bc[r] ? 0.0 : sum;
}

// P1: a cell's four block dofs are its four vertices in some numbering.  When the dofmap numbers them as the mesh
// numbers its vertices (any feed of this repository) the coordinate array serves as it is; otherwise the coordinates
// are copied once into dof order.  Either way the P1 assembly kernels read ONE 16-B connectivity record per cell and
// gather coordinates by dof -- the second record was half of their connectivity traffic.  Runs when the dofmap is
// uploaded (function-space data, as tabulate_dof_coordinates is in the reference), not inside the assembly timers.
__global__ void k_conn_differs(const int32_t* __restrict__ a, const int32_t* __restrict__ b, int64_t n, int* __restrict__ flag)
{
  bool d = false;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    d |= a[i] != b[i];
  if (__any(d) && (threadIdx.x & 63) == 0)
    *flag = 1;
}

__global__ void k_dof_coords(const double* __restrict__ x, const int32_t* __restrict__ cv, const int32_t* __restrict__ cd,
                             int64_t n, double* __restrict__ xd)
{
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
  {
    const double* q = x + 3 * (int64_t)cv[i];
    double* o = xd + 3 * (int64_t)cd[i]; // every writer of a dof stores the same three values
    o[0] = q[0];
    o[1] = q[1];
    o[2] = q[2];
  }
}

int ensure_p1_coords(zzz_ctx* ctx)
{
  if (ctx->xq_valid || ctx->order != 1 || ctx->ncells == 0 || !ctx->cell_dofs.p || !ctx->cell_verts.p)
    return ZZZ_OK;
  const int64_t n = 4 * ctx->ncells, nblock = ctx->n_owned + ctx->n_ghost;
  const int g = (int)std::min<int64_t>((n + 255) / 256, 8192);
  DevBuf<int> flag;
  ZZZ_HIP(ctx, flag.alloc(1));
  ZZZ_HIP(ctx, hipMemsetAsync(flag.p, 0, sizeof(int), ctx->stream));
  hipLaunchKernelGGL(k_conn_differs, dim3(g), dim3(256), 0, ctx->stream, ctx->cell_verts.p, ctx->cell_dofs.p, n, flag.p);
  int differs = 0;
  ZZZ_HIP(ctx, hipMemcpyAsync(&differs, flag.p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  ZZZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (!differs && ctx->nverts >= nblock)
    ctx->xq = ctx->x.p;
  else
  {
    ZZZ_HIP(ctx, ctx->xdof.alloc((size_t)(3 * nblock)));
    ZZZ_HIP(ctx, hipMemsetAsync(ctx->xdof.p, 0, sizeof(double) * 3 * (size_t)nblock, ctx->stream));
    hipLaunchKernelGGL(k_dof_coords, dim3(g), dim3(256), 0, ctx->stream, ctx->x.p, ctx->cell_verts.p, ctx->cell_dofs.p, n,
                       ctx->xdof.p);
    ZZZ_HIP(ctx, hipGetLastError());
    ctx->xq = ctx->xdof.p;
  }
  ctx->xq_valid = true;
  return ZZZ_OK;
}

int ensure_tables(zzz_ctx* ctx)
{
  if (ctx->tables_order == ctx->order)
    return ZZZ_OK;
  const double* src = ctx->order == 1 ? ZZZ_TAB_P1 : (ctx->order == 2 ? ZZZ_TAB_P2 : ZZZ_TAB_P3);
  const size_t n = (size_t)14 * ctx->nd * ctx->nd;
  ZZZ_HIP(ctx, ctx->tables.alloc(n));
  ZZZ_HIP(ctx, hipMemcpy(ctx->tables.p, src, n * sizeof(double), hipMemcpyHostToDevice));
  ctx->tables_order = ctx->order;
  return ZZZ_OK;
}

int asm_tile_nnz(const zzz_ctx* ctx)
{
  if (ctx->bs == 1 && ctx->order == 1)
    return ASM_NNZ_P1;
  if (ctx->bs == 3 && ctx->order == 1 && ctx->asm_node3)
    return ASM_NNZ_NODE3; // (one thread per node: asm_matrix_p1_node3)
#ifdef ZZZ_EXPERIMENTS
  if (const char* e = getenv("ZZZ_ASM_CAP")) // measurement knob, tools build only
    if (atoi(e) >= 1024 && atoi(e) <= 8192)
      return atoi(e);
#endif
  if (ctx->have_asm_pos)
  {
    // position-based kernel (asm_matrix_pk_pos: no columns in LDS, persistent workgroups): smaller tiles = more
    // workgroups per CU.  Measured (tools/ab_asm.sh): Poisson P3 6.2 M dofs 2.80 ms at 2048 against 4.79 at 4096 and 3.30
    // at 1536; elasticity P2 2 M 2.23 ms at 3072 (2.45 at 4096, 3.79 at 2048); elasticity P3 keeps 4096 (a block row of a
    // vertex has ~2000 entries)
    if (ctx->bs == 1)
      return ASM_NNZ_SMALL;
    return ctx->order == 2 ? 3072 : ASM_NNZ;
  }
  return (ctx->bs == 1 && ctx->order == 2) ? ASM_NNZ_SMALL : ASM_NNZ;
}

// Geometry factors of every cell, once per assembly (one thread per cell, dense): BS = 1: |detJ| (K K^T), the six
// numbers 00 11 22 01 02 12 that Poisson's element tensor is contracted with; BS = 3: |detJ| and K (ten numbers).
// The row-gather kernel below then fetches 48 (80) bytes per (row, cell) pair -- all lanes of a row the same address,
// neighbouring rows mostly the same cell -- instead of the cell's vertices, their coordinates and ~150 flops.
template <int BS>
__global__ void k_cell_geom(const double* __restrict__ x, const int32_t* __restrict__ cell_verts, int64_t ncells,
                            double* __restrict__ out)
{
  constexpr int NG = BS == 1 ? 6 : 10;
  for (int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; c < ncells; c += (int64_t)gridDim.x * blockDim.x)
  {
    double p[4][3];
    load_cell(x, *reinterpret_cast<const int4*>(cell_verts + 4 * c), p);
    Geom G;
    geometry(p, G);
    double* o = out + c * NG;
    if (BS == 1)
    {
      o[0] = G.adet * (G.K[0][0] * G.K[0][0] + G.K[0][1] * G.K[0][1] + G.K[0][2] * G.K[0][2]);
      o[1] = G.adet * (G.K[1][0] * G.K[1][0] + G.K[1][1] * G.K[1][1] + G.K[1][2] * G.K[1][2]);
      o[2] = G.adet * (G.K[2][0] * G.K[2][0] + G.K[2][1] * G.K[2][1] + G.K[2][2] * G.K[2][2]);
      o[3] = G.adet * (G.K[0][0] * G.K[1][0] + G.K[0][1] * G.K[1][1] + G.K[0][2] * G.K[1][2]);
      o[4] = G.adet * (G.K[0][0] * G.K[2][0] + G.K[0][1] * G.K[2][1] + G.K[0][2] * G.K[2][2]);
      o[5] = G.adet * (G.K[1][0] * G.K[2][0] + G.K[1][1] * G.K[2][1] + G.K[1][2] * G.K[2][2]);
    }
    else
    {
      o[0] = G.adet;
#pragma unroll
      for (int al = 0; al < 3; ++al)
#pragma unroll
        for (int d = 0; d < 3; ++d)
          o[1 + 3 * al + d] = G.K[al][d];
    }
  }
}

// The same assembly with (i) the POSITIONS of the element-matrix entries handed over by the pattern build
// (zzz_ctx::asm_pos: for the a-th cell of block row i and local column j, the rank of dof j among the row's sorted
// columns -- a by-product of the sort that found those columns) and (ii) the cells' geometry factors evaluated once
// (k_cell_geom).  No column search (it was 7-8 dependent LDS reads per entry, 545 M entries per assembly of the
// 6.2 M-dof P3 problem), no columns in LDS (a third of the tile's footprint), no connectivity, vertex or coordinate
// read in the row walk: a (row, cell) pair is its adjacency entry, 2 B per column and one 48-B geometry record.
// Constrained COLUMNS are zeroed in one dense pass over the tile's CSR segment at the end (columns and flags read
// once per nonzero, not once per contribution); constrained rows contribute zeros and get their diagonal
// (fem::set_diagonal) after that pass.  Same sums in the same order as asm_matrix_pk.
template <int ND, int BS, int LPR>
__global__ __launch_bounds__(ASM_BLOCK) void asm_matrix_pk_pos(const double* __restrict__ geom,
                                                               const int32_t* __restrict__ adjT_off,
                                                               const int32_t* __restrict__ adjT_cells,
                                                               const uint8_t* __restrict__ adj_li,
                                                               const int32_t* __restrict__ adj_off,
                                                               const uint16_t* __restrict__ pos,
                                                               const uint8_t* __restrict__ bc,
                                                               const rp_t* __restrict__ rowptr,
                                                               const int32_t* __restrict__ cols, double* __restrict__ vals,
                                                               const int32_t* __restrict__ tiles, int64_t ntiles,
                                                               const double* __restrict__ tab, int cap,
                                                               int32_t* __restrict__ rownnz, const int64_t* __restrict__ crow,
                                                               double* __restrict__ cvals, int32_t* __restrict__ ccols)
{
  constexpr int NT = (BS == 1) ? 6 : 9;
  constexpr int NN = ND * ND;
  constexpr int NG = BS == 1 ? 6 : 10;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  double* vals_s = reinterpret_cast<double*>(lds_raw);
  double* T_s = vals_s + cap; // cap = nonzeros a tile may hold (asm_tile_nnz)
  for (int k = threadIdx.x; k < NT * NN; k += ASM_BLOCK)
  {
    if (BS == 1)
    {
      // symmetrised pieces: G is symmetric, so S^{ab} + S^{ba} is all a < b needs
      const int t = k / NN, ij = k % NN;
      const int a = t < 3 ? t : (t == 3 ? 0 : (t == 4 ? 0 : 1)), b = t < 3 ? t : (t == 3 ? 1 : 2);
      T_s[k] = (t < 3) ? tab[(a * 3 + a) * NN + ij] : tab[(a * 3 + b) * NN + ij] + tab[(b * 3 + a) * NN + ij];
    }
    else
      T_s[k] = tab[k];
  }
  // persistent workgroups: the reference tensors above are staged once per workgroup, not once per tile (at 2048
  // nonzeros per tile the 6.2 M-dof P3 matrix has 146 k tiles: 2.8 GB of table reads otherwise); XCD x walks the x-th
  // eighth of the tiles
  const int xcd = blockIdx.x & 7, wg_in_xcd = blockIdx.x >> 3, n_in_xcd = (gridDim.x + 7 - xcd) >> 3;
  const int64_t t_lo = ntiles * xcd / 8, t_hi = ntiles * (xcd + 1) / 8;
  for (int64_t tile = t_lo + wg_in_xcd; tile < t_hi; tile += n_in_xcd)
  {
  const int d0 = tiles[tile], d1 = tiles[tile + 1];
  const int row0 = d0 * BS, row1 = d1 * BS;
  const int64_t s = rowptr[row0];
  const int e = (int)(rowptr[row1] - s); // entries of the tile's CSR segment (fits LDS)
  for (int k = threadIdx.x; k < e; k += ASM_BLOCK)
    vals_s[k] = 0.0;
  // block dofs of the tile in order of descending cell count, as in asm_matrix_pk
  __shared__ int ord_s[ASM_ORD_CAP];
  __shared__ int hist_s[34];
  __shared__ int rp_s[ASM_RP_CAP + 1]; // the tile's row offsets relative to s: every later phase reads them here
  const int nt = d1 - d0;
  const bool rp_in_lds = nt * BS <= ASM_RP_CAP;
  if (rp_in_lds)
    for (int k = threadIdx.x; k <= nt * BS; k += ASM_BLOCK)
      rp_s[k] = (int)(rowptr[row0 + k] - s);
  auto row_begin = [&](int q) { return rp_in_lds ? rp_s[q] : (int)(rowptr[row0 + q] - s); }; // q = scalar row - row0
  const bool sorted = nt <= ASM_ORD_CAP;
  if (sorted)
  {
    if (threadIdx.x < 34)
      hist_s[threadIdx.x] = 0;
    __syncthreads();
    for (int k = threadIdx.x; k < nt; k += ASM_BLOCK)
      atomicAdd(&hist_s[32 - min(adj_off[d0 + k + 1] - adj_off[d0 + k], 32)], 1);
    __syncthreads();
    if (threadIdx.x == 0)
    {
      int acc = 0;
      for (int b = 0; b < 33; ++b)
      {
        const int h = hist_s[b];
        hist_s[b] = acc;
        acc += h;
      }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < nt; k += ASM_BLOCK)
      ord_s[atomicAdd(&hist_s[32 - min(adj_off[d0 + k + 1] - adj_off[d0 + k], 32)], 1)] = k;
  }
  __syncthreads();
  const int lane = (int)threadIdx.x % LPR;
  constexpr int JM = (ND + LPR - 1) / LPR; // columns of a cell this lane handles
  constexpr double Ey = 1.0e6, nu = 0.3; // src/Elasticity.py:12-15
  constexpr double mu = Ey / (2.0 * (1.0 + nu));
  constexpr double lmbda = Ey * nu / ((1.0 + nu) * (1.0 - 2.0 * nu));
  for (int q = (int)threadIdx.x / LPR; q < nt * BS; q += ASM_BLOCK / LPR)
  {
    const int i = d0 + (sorted ? ord_s[q / BS] : q / BS), c = q % BS;
    const int r = i * BS + c;
    const int a0 = row_begin(r - row0);
    const bool bcr = bc[r] != 0;
    const AdjIter adj(adjT_off, adjT_cells, adj_li, i);
    const uint16_t* __restrict__ prow = pos + (int64_t)adj_off[i] * ND; // positions of the row's (cell, column) pairs
    // software pipeline over the row's cells: the adjacency entry of cell a + 2 and the geometry record and positions of
    // cell a + 1 are in flight while cell a is evaluated
    struct Item
    {
      int cell, li;
      double g[NG];
      int pj[JM];
    };
    // (Every load below is UNCONDITIONAL, at a clamped index: `in ? p[a] : -1` and `if (cell < 0) return` compiled to one branch
    // per load with the wait for the load inside it -- a row's adjacency entry, geometry record and five positions were eight
    // round trips one after the other per (row, cell) pair; found in the disassembly in the last third of round 6.  Beyond a
    // row's last cell the walk loads cell 0 / the row's last position and uses neither.)
    const int alast = max(adj.len - 1, 0), npairs = max(adj_off[i + 1] - adj_off[i], 1);
    auto adj_at = [&](int a, int& cell, int& li) {
      const int ac = min(a, alast);
      const int cc = adj.cell(ac), ll = adj.li(ac);
      cell = a < adj.len ? cc : -1;
      li = a < adj.len ? ll : 0;
    };
    auto item_at = [&](int a, int cell, int li, Item& K) {
      K.cell = cell;
      K.li = li;
      const double* __restrict__ gp = geom + (int64_t)max(cell, 0) * NG;
#pragma unroll
      for (int t = 0; t < NG; ++t)
        K.g[t] = gp[t];
      const uint16_t* __restrict__ pp = prow + min(a, npairs - 1) * ND;
#pragma unroll
      for (int qq = 0; qq < JM; ++qq)
        K.pj[qq] = pp[min(lane + qq * LPR, ND - 1)];
    };
    Item K0;
    int c1, l1;
    K0.cell = -1;
    if (adj.len > 0) // (a slice without cells has no adjacency entries to read)
    {
      int ca, la;
      adj_at(0, ca, la);
      item_at(0, ca, la, K0);
      adj_at(1, c1, l1);
    }
    for (int a = 0; K0.cell >= 0; ++a)
    {
      Item K1;
      int c2, l2;
      item_at(a + 1, c1, l1, K1);
      adj_at(a + 2, c2, l2);
      const double* Tl = T_s + K0.li * ND;
      if (BS == 1)
      {
#pragma unroll
        for (int qq = 0; qq < JM; ++qq)
        {
          const int j = lane + qq * LPR;
          if (j >= ND)
            break;
          double val = 0.0;
#pragma unroll
          for (int t = 0; t < 6; ++t)
            val += K0.g[t] * Tl[t * NN + j];
          if (bcr)
            val = 0.0;
          vals_s[a0 + K0.pj[qq]] += val;
        }
      }
      else
      {
        const double adet = K0.g[0];
        const double* Kf = K0.g + 1; // K[al][d] at Kf[3 al + d]
#pragma unroll
        for (int qq = 0; qq < JM; ++qq)
        {
          const int j = lane + qq * LPR;
          if (j >= ND)
            break;
          // D[cc][d] = |detJ| sum_{al,be} K[al][cc] K[be][d] S^[al][be]_{li,j} = int d_cc phi_i d_d phi_j
          double D[3][3];
          {
            double tmp[3][3];
#pragma unroll
            for (int al = 0; al < 3; ++al)
#pragma unroll
              for (int d = 0; d < 3; ++d)
                tmp[al][d] = Tl[(al * 3 + 0) * NN + j] * Kf[0 + d] + Tl[(al * 3 + 1) * NN + j] * Kf[3 + d]
                             + Tl[(al * 3 + 2) * NN + j] * Kf[6 + d];
#pragma unroll
            for (int cc = 0; cc < 3; ++cc)
#pragma unroll
              for (int d = 0; d < 3; ++d)
                D[cc][d] = adet * (Kf[0 + cc] * tmp[0][d] + Kf[3 + cc] * tmp[1][d] + Kf[6 + cc] * tmp[2][d]);
          }
          const double tr = D[0][0] + D[1][1] + D[2][2];
#pragma unroll
          for (int d = 0; d < 3; ++d)
          {
            const double Dcd = sel3(D[0][d], D[1][d], D[2][d], c), Ddc = sel3(D[d][0], D[d][1], D[d][2], c);
            double val = mu * ((c == d ? tr : 0.0) + Ddc) + lmbda * Dcd;
            if (bcr)
              val = 0.0;
            vals_s[a0 + K0.pj[qq] * 3 + d] += val;
          }
        }
      }
      K0 = K1;
      c1 = c2;
      l1 = l2;
    }
  }
  __syncthreads();
  // constrained columns: one dense pass over the tile's entries (column index and flag once per nonzero)
  for (int k = threadIdx.x; k < e; k += ASM_BLOCK)
    if (bc[cols[s + k]])
      vals_s[k] = 0.0;
  __syncthreads();
  // fem::set_diagonal on constrained rows (their diagonal was zeroed with the rest of the constrained columns)
  for (int q = threadIdx.x; q < nt * BS; q += ASM_BLOCK)
  {
    const int r = row0 + q;
    if (bc[r])
    {
      const int b0 = row_begin(q);
      vals_s[b0 + find_pos(cols + s + b0, row_begin(q + 1) - b0, r)] = 1.0;
    }
  }
  __syncthreads();
  for (int k = threadIdx.x; k < e; k += ASM_BLOCK)
    vals[s + k] = vals_s[k];
  // the operator stream of the CG product keeps the non-zero entries only: how many each row has is counted here, where
  // the finished values still sit in LDS (the packer's own dense count sweep re-read all of them: 4.5 ms at 2.4 G entries)
  // ... and, with `crow` (capacity-based row starts, zzz_sellp.hip: sellp_capacity_rows), the kept entries themselves go
  // into the packer's compacted copy from here: its own compaction sweep re-read values and columns of the whole matrix
  // (2.2 ms at 6.2 M P3 dofs, 17 ms at 49.8 M)
  if (rownnz)
    for (int q = (int)threadIdx.x >> 3; q < nt * BS; q += ASM_BLOCK / 8) // eight lanes per row
    {
      const int b0 = row_begin(q), len = row_begin(q + 1) - b0;
      const int sub = threadIdx.x & 7, grp = (threadIdx.x & 63) >> 3;
      const int64_t cb = crow ? crow[row0 + q] : 0;
      int n = 0;
      for (int k0 = 0; k0 < len; k0 += 8)
      {
        const int k = k0 + sub;
        const double v = k < len ? vals_s[b0 + k] : 0.0;
        const bool keep = k < len && v != 0.0;
        const unsigned m8 = (unsigned)(__ballot(keep) >> (8 * grp)) & 0xffu;
        if (crow && keep)
        {
          const int at = n + __popc(m8 & ((1u << sub) - 1u));
          cvals[cb + at] = v;
          ccols[cb + at] = cols[s + b0 + k];
        }
        n += __popc(m8);
      }
      if (sub == 0)
        rownnz[row0 + q] = n;
    }
  __syncthreads(); // the next tile reuses vals_s, ord_s, hist_s
  }
}

template <int ND, int BS, int LPR>
static int launch_matrix_pk_pos(zzz_ctx* ctx)
{
  constexpr int NT = (BS == 1) ? 6 : 9;
  constexpr int NG = BS == 1 ? 6 : 10;
  const int cap = asm_tile_nnz(ctx);
  const size_t lds = (size_t)cap * 8 + (size_t)NT * ND * ND * 8;
  auto kern = asm_matrix_pk_pos<ND, BS, LPR>;
  const unsigned bit = (16u << ((ND == 10 ? 0 : 2) + (BS == 1 ? 0 : 1))) << (LPR == 8 ? 0 : (LPR == 4 ? 4 : 8));
  if (!(ctx->lds_attr_set & bit))
  {
    ZZZ_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    ctx->lds_attr_set |= bit;
  }
  ZZZ_HIP(ctx, ctx->cell_geom.alloc((size_t)(ctx->ncells * NG)));
  hipLaunchKernelGGL(k_cell_geom<BS>, dim3((unsigned)std::min<int64_t>((ctx->ncells + 255) / 256, 16384)), dim3(256), 0, ctx->stream,
                     ctx->x.p, ctx->cell_verts.p, ctx->ncells, ctx->cell_geom.p);
  int per_cu = (int)(160 * 1024 / (lds + 3400)); // workgroups a CU holds (LDS-bound; ord_s, rp_s, hist_s on top)
  per_cu = per_cu < 1 ? 1 : (per_cu > 8 ? 8 : per_cu);
  const int64_t grid = std::min<int64_t>(xcd_grid(ctx->n_asm_tiles), 256 * (int64_t)per_cu);
  // the non-zero count per row, for the packer of the operator stream (zzz_sellp.hip: sp_build_sorted)
  int32_t* rownnz = nullptr;
  if (ctx->sellp_mode != 0 && ctx->sellp_drop && ctx->sp_rownnz.alloc((size_t)ctx->nrows + 1) == hipSuccess)
    rownnz = ctx->sp_rownnz.p;
  (void)hipGetLastError();
  // long rows (the packer's synchronous path): the compacted copy of the kept entries is written from here as well
  const int64_t* crow = nullptr;
#ifdef ZZZ_EXPERIMENTS
  const bool compact_here = !getenv("ZZZ_ASM_NO_COMPACT"); // A/B knob (tools build)
#else
  const bool compact_here = true;
#endif
  if (rownnz && ctx->nnz >= 16 * ctx->nrows && ctx->nnz + 8 * ctx->nrows < ((int64_t)1 << 40) && compact_here)
  {
    if (int rc = sellp_capacity_rows(ctx))
      return rc;
    crow = ctx->sp_crow.p;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(ASM_BLOCK), lds, ctx->stream, ctx->cell_geom.p,
                     ctx->adjT_off.p, ctx->adjT_cells.p, ctx->adj_li.p, ctx->adj_off.p, ctx->asm_pos.p, ctx->bc.p,
                     ctx->rowptr.p, ctx->cols.p, ctx->vals.p, ctx->asm_tile.p, ctx->n_asm_tiles, ctx->tables.p, cap, rownnz, crow,
                     ctx->sp_cvals.p, ctx->sp_ccols.p);
  ctx->sp_rownnz_fresh = rownnz != nullptr;
  ctx->sp_compact_fresh = crow != nullptr;
  return ZZZ_OK;
}

template <int ND, int BS, int LPR>
static int launch_matrix_pk(zzz_ctx* ctx)
{
  constexpr int NT = (BS == 1) ? 6 : 9;
  const int cap = asm_tile_nnz(ctx);
  const size_t lds = (size_t)cap * 8 + (size_t)NT * ND * ND * 8 + (size_t)cap * 4;
  auto kern = asm_matrix_pk<ND, BS, LPR>;
  // the attribute belongs to (kernel, device): set it once per context, outside the hot path
  const unsigned bit = 1u << ((ND == 10 ? 0 : 2) + (BS == 1 ? 0 : 1));
  if (!(ctx->lds_attr_set & bit))
  {
    ZZZ_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    ctx->lds_attr_set |= bit;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)xcd_grid(ctx->n_asm_tiles)), dim3(ASM_BLOCK), lds, ctx->stream, ctx->x.p,
                     ctx->cell_verts.p, ctx->cell_dofs.p, ctx->adjT_off.p, ctx->adjT_cells.p, ctx->adj_li.p, ctx->adj_off.p, ctx->bc.p,
                     ctx->rowptr.p, ctx->cols.p, ctx->vals.p, ctx->asm_tile.p, ctx->n_asm_tiles, ctx->tables.p, cap);
  return ZZZ_OK;
}

template <int ND, int BS>
static void launch_vector_pk(zzz_ctx* ctx, int64_t nrows)
{
  const dim3 grid((unsigned)xcd_grid((nrows + ASM_BLOCK - 1) / ASM_BLOCK)), block(ASM_BLOCK);
  hipLaunchKernelGGL((asm_vector_pk<ND, BS>), grid, block, 0, ctx->stream, ctx->x.p, ctx->cell_verts.p, ctx->cell_dofs.p,
                     ctx->adjT_off.p, ctx->adjT_cells.p, ctx->adj_li.p, ctx->bc.p, ctx->facet_mask.p, ctx->coeff[0].p,
                     BS == 1 ? ctx->coeff[1].p : (const double*)nullptr, ctx->b.p, nrows, ctx->tables.p);
}

int launch_assemble_matrix(zzz_ctx* ctx, int form)
{
  ctx->sp_rownnz_fresh = ctx->sp_compact_fresh = false;
  ++ctx->mat_version;
  const int bs = form == ZZZ_FORM_ELASTICITY ? 3 : 1;
  if (bs != ctx->bs)
    return fail(ctx, ZZZ_ERR_ARG, "form %d needs block size %d, dofmap has %d", form, bs, ctx->bs);
  const dim3 grid((unsigned)xcd_grid(ctx->n_asm_tiles)), block(ASM_BLOCK);
  if (int rc = ensure_p1_coords(ctx))
    return rc;
  if (ctx->order == 1)
  {
    if (bs == 1)
    {
#ifdef ZZZ_EXPERIMENTS
      const int probe = getenv("ZZZ_ASM_PROBE") ? atoi(getenv("ZZZ_ASM_PROBE")) & 15 : 0; // timing ablations (wrong results)
#define ZZZ_P1_PROBE(P)                                                                                                        \
  if (probe == P)                                                                                                              \
  {                                                                                                                            \
    hipLaunchKernelGGL((asm_matrix_p1<1, ASM_NNZ_P1, 128, P>), grid, dim3(128), 0, ctx->stream, ctx->xq, ctx->cell_dofs.p,     \
                       ctx->adjT_off.p, ctx->adjT_cells.p, ctx->adj_li.p, ctx->bc.p, ctx->rowptr.p, ctx->cols.p, ctx->vals.p,  \
                       ctx->asm_tile.p, ctx->n_asm_tiles);                                                                     \
    return ZZZ_OK;                                                                                                             \
  }
      ZZZ_P1_PROBE(1) ZZZ_P1_PROBE(6) ZZZ_P1_PROBE(8) ZZZ_P1_PROBE(15)
#undef ZZZ_P1_PROBE
#endif
      // one thread per row: a 1920-nonzero tile holds ~126 rows of 15, so 128 threads leave no lane idle
      hipLaunchKernelGGL((asm_matrix_p1<1, ASM_NNZ_P1, 128>), grid, dim3(128), 0, ctx->stream, ctx->xq, ctx->cell_dofs.p,
                         ctx->adjT_off.p, ctx->adjT_cells.p, ctx->adj_li.p, ctx->bc.p, ctx->rowptr.p, ctx->cols.p, ctx->vals.p,
                         ctx->asm_tile.p, ctx->n_asm_tiles);
    }
    else if (ctx->asm_node3)
      hipLaunchKernelGGL((asm_matrix_p1_node3<ASM_NNZ_NODE3>), grid, dim3(64), 0, ctx->stream, ctx->xq, ctx->cell_dofs.p,
                         ctx->adjT_off.p, ctx->adjT_cells.p, ctx->adj_li.p, ctx->bc.p, ctx->rowptr.p, ctx->cols.p, ctx->vals.p,
                         ctx->asm_tile.p, ctx->n_asm_tiles);
    else
      hipLaunchKernelGGL((asm_matrix_p1<3, ASM_NNZ, ASM_BLOCK>), grid, block, 0, ctx->stream, ctx->xq, ctx->cell_dofs.p,
                         ctx->adjT_off.p, ctx->adjT_cells.p, ctx->adj_li.p, ctx->bc.p, ctx->rowptr.p, ctx->cols.p, ctx->vals.p,
                         ctx->asm_tile.p, ctx->n_asm_tiles);
  }
  else
  {
    int rc = ensure_tables(ctx);
    if (rc)
      return rc;
    if (ctx->have_asm_pos) // positions from the pattern build: no column search (asm_matrix_pk_pos)
    {
#ifdef ZZZ_EXPERIMENTS
      const char* e = getenv("ZZZ_ASM_LPR"); // measurement knob (lanes per row), tools build only
      const int lpr = e ? atoi(e) : 0;
      if (ctx->order == 2)
        rc = bs == 1 ? (lpr == 2 ? launch_matrix_pk_pos<10, 1, 2>(ctx) : launch_matrix_pk_pos<10, 1, 4>(ctx))
                     : launch_matrix_pk_pos<10, 3, 4>(ctx);
      else
        rc = bs == 1 ? (lpr == 8 ? launch_matrix_pk_pos<20, 1, 8>(ctx) : lpr == 2 ? launch_matrix_pk_pos<20, 1, 2>(ctx)
                                                                        : launch_matrix_pk_pos<20, 1, 4>(ctx))
                     : (lpr == 4 ? launch_matrix_pk_pos<20, 3, 4>(ctx) : launch_matrix_pk_pos<20, 3, 8>(ctx));
#else
      // lanes per row: Poisson four (P3: five columns each, no idle lane: 2.80 against 2.92 ms), elasticity P3 eight
      if (ctx->order == 2)
        rc = bs == 1 ? launch_matrix_pk_pos<10, 1, 4>(ctx) : launch_matrix_pk_pos<10, 3, 4>(ctx);
      else
        rc = bs == 1 ? launch_matrix_pk_pos<20, 1, 4>(ctx) : launch_matrix_pk_pos<20, 3, 8>(ctx);
#endif
    }
    else if (ctx->order == 2)
      rc = bs == 1 ? launch_matrix_pk<10, 1, 4>(ctx) : launch_matrix_pk<10, 3, 4>(ctx);
    else
      rc = bs == 1 ? launch_matrix_pk<20, 1, 8>(ctx) : launch_matrix_pk<20, 3, 8>(ctx);
    if (rc)
      return rc;
  }
  ZZZ_HIP(ctx, hipGetLastError());
  return ZZZ_OK;
}

int launch_assemble_vector(zzz_ctx* ctx, int form)
{
  const int bs = form == ZZZ_FORM_ELASTICITY ? 3 : 1;
  if (bs != ctx->bs)
    return fail(ctx, ZZZ_ERR_ARG, "form %d needs block size %d, dofmap has %d", form, bs, ctx->bs);
  const int64_t nrows = ctx->n_owned * bs;
  const dim3 grid((unsigned)xcd_grid((nrows + ASM_BLOCK - 1) / ASM_BLOCK)), block(ASM_BLOCK);
  if (int rc = ensure_p1_coords(ctx))
    return rc;
  if (ctx->order == 1)
  {
    // the cells' records first (|det J| and the coefficient's vertex sum: k_cell_load_p1), then the row walk
    ZZZ_HIP(ctx, ctx->cell_geom.alloc((size_t)(ctx->ncells * (bs == 1 ? 2 : 4))));
    const int cper = bs == 1 ? 1024 : 256; // cells per workgroup and round (k_cell_load_p1: U)
    // (a partition without cells: no cell pass -- a grid of 0 workgroups is an invalid configuration and the kernel clamps
    // cell numbers to ncells - 1; the row walk below then meets empty adjacency lists only)
    const dim3 cgrid((unsigned)std::max<int64_t>(1, std::min<int64_t>((ctx->ncells + cper - 1) / cper, 16384)));
    if (bs == 1)
    {
      if (ctx->ncells > 0)
        hipLaunchKernelGGL(k_cell_load_p1<1>, cgrid, dim3(256), 0, ctx->stream, ctx->xq, ctx->cell_dofs.p, ctx->coeff[0].p,
                           ctx->facet_mask.p, ctx->ncells, ctx->cell_geom.p);
      hipLaunchKernelGGL(asm_vector_p1<1>, grid, block, 0, ctx->stream, ctx->xq, ctx->cell_dofs.p, ctx->adjT_off.p, ctx->adjT_cells.p,
                         ctx->adj_li.p, ctx->bc.p, ctx->facet_mask.p, ctx->coeff[0].p, ctx->coeff[1].p, ctx->cell_geom.p, ctx->b.p,
                         nrows);
    }
    else
    {
      if (ctx->ncells > 0)
        hipLaunchKernelGGL(k_cell_load_p1<3>, cgrid, dim3(256), 0, ctx->stream, ctx->xq, ctx->cell_dofs.p, ctx->coeff[0].p,
                           (const uint8_t*)nullptr, ctx->ncells, ctx->cell_geom.p);
      hipLaunchKernelGGL(asm_vector_p1<3>, grid, block, 0, ctx->stream, ctx->xq, ctx->cell_dofs.p, ctx->adjT_off.p, ctx->adjT_cells.p,
                         ctx->adj_li.p, ctx->bc.p, ctx->facet_mask.p, ctx->coeff[0].p, (const double*)nullptr, ctx->cell_geom.p,
                         ctx->b.p, nrows);
    }
  }
  else
  {
    int rc = ensure_tables(ctx);
    if (rc)
      return rc;
    if (ctx->order == 2)
      bs == 1 ? launch_vector_pk<10, 1>(ctx, nrows) : launch_vector_pk<10, 3>(ctx, nrows);
    else
      bs == 1 ? launch_vector_pk<20, 1>(ctx, nrows) : launch_vector_pk<20, 3>(ctx, nrows);
  }
  ZZZ_HIP(ctx, hipGetLastError());
  return ZZZ_OK;
}

// ---- matrix-free operator of cgpoisson: y = action(a, un = x), then y[bc] = 0
// (form M of src/Poisson.py:33, applied as in src/cgpoisson_problem.cpp:193-230: assemble_vector of M
// with un = x, bc->set(y, 0)).  No element matrix is stored.  Two dense passes:
//   1. matfree_cell  : one thread per cell computes w_c = Ae_c u_c once (geometry once per cell, not
//                      once per row as in a row-gather walk) and stores it component-major
//                      (w[i][cell]); with the feed's simplex-type-major cell numbering neighbouring
//                      lanes hold neighbouring sub-cubes, so connectivity, coordinates and u are read
//                      as dense streams;
//   2. matfree_gather: one thread per owned row adds the w entries of its cells in ascending cell
//                      order (fixed order => reproducible), zeroes constrained rows, leaves the
//                      <x, y> partials for the CG dot product.
// Traffic: ~(16 + 4 nd) B connectivity + 8 nd B of w written per cell, 13 B per (row, cell) incidence
// read back -- ~6 GB at 10 M P1 dofs against 40 GB through L2 for the single-pass walk.
template <int ND>
__global__ __launch_bounds__(ASM_BLOCK) void matfree_cell(const double* __restrict__ x,
                                                          const int32_t* __restrict__ cell_verts,
                                                          const int32_t* __restrict__ cell_dofs, int64_t ncells,
                                                          const double* __restrict__ u, double* __restrict__ w,
                                                          const double* __restrict__ tab,
                                                          const int* __restrict__ stop_flag)
{
  if (stop_flag && *stop_flag)
    return;
  constexpr int NN = ND * ND;
  __shared__ double T_s[ND == 4 ? 1 : 6 * NN];
  if (ND != 4)
  {
    for (int k = threadIdx.x; k < 6 * NN; k += ASM_BLOCK)
    {
      const int t = k / NN, ij = k % NN;
      const int a = t < 3 ? t : (t == 3 ? 0 : (t == 4 ? 0 : 1)), b = t < 3 ? t : (t == 3 ? 1 : 2);
      T_s[k] = (t < 3) ? tab[(a * 3 + a) * NN + ij] : tab[(a * 3 + b) * NN + ij] + tab[(b * 3 + a) * NN + ij];
    }
    __syncthreads();
  }
  for (int64_t c = blockIdx.x * (int64_t)ASM_BLOCK + threadIdx.x; c < ncells; c += (int64_t)gridDim.x * ASM_BLOCK)
  {
    const int4 v = *reinterpret_cast<const int4*>(cell_verts + 4 * c);
    const int32_t* __restrict__ cd = cell_dofs + (int64_t)ND * c;
    double p[4][3];
    Geom G;
    load_cell(x, v, p);
    geometry(p, G);
    if (ND == 4)
    {
      double g[4][3];
      p1_grads(G, g);
      const int4 dd = *reinterpret_cast<const int4*>(cd);
      const double ul[4] = {u[dd.x], u[dd.y], u[dd.z], u[dd.w]};
      double gu[3] = {0, 0, 0};
#pragma unroll
      for (int j = 0; j < 4; ++j)
      {
        gu[0] += g[j][0] * ul[j];
        gu[1] += g[j][1] * ul[j];
        gu[2] += g[j][2] * ul[j];
      }
      const double s = G.adet / 6.0;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        w[(int64_t)i * ncells + c] = s * (g[i][0] * gu[0] + g[i][1] * gu[1] + g[i][2] * gu[2]);
    }
    else
    {
      double GG[6];
      GG[0] = G.adet * (G.K[0][0] * G.K[0][0] + G.K[0][1] * G.K[0][1] + G.K[0][2] * G.K[0][2]);
      GG[1] = G.adet * (G.K[1][0] * G.K[1][0] + G.K[1][1] * G.K[1][1] + G.K[1][2] * G.K[1][2]);
      GG[2] = G.adet * (G.K[2][0] * G.K[2][0] + G.K[2][1] * G.K[2][1] + G.K[2][2] * G.K[2][2]);
      GG[3] = G.adet * (G.K[0][0] * G.K[1][0] + G.K[0][1] * G.K[1][1] + G.K[0][2] * G.K[1][2]);
      GG[4] = G.adet * (G.K[0][0] * G.K[2][0] + G.K[0][1] * G.K[2][1] + G.K[0][2] * G.K[2][2]);
      GG[5] = G.adet * (G.K[1][0] * G.K[2][0] + G.K[1][1] * G.K[2][1] + G.K[1][2] * G.K[2][2]);
      double ul[ND];
#pragma unroll
      for (int j = 0; j < ND; ++j)
        ul[j] = u[cd[j]];
#pragma unroll
      for (int i = 0; i < ND; ++i)
      {
        // (sum_t GG[t] T[t][i][:]) . u ; the table reads are wave-uniform (LDS broadcast)
        double acc = 0.0;
#pragma unroll
        for (int j = 0; j < ND; ++j)
        {
          double aij = 0.0;
#pragma unroll
          for (int t = 0; t < 6; ++t)
            aij += GG[t] * T_s[t * NN + i * ND + j];
          acc += aij * ul[j];
        }
        w[(int64_t)i * ncells + c] = acc;
      }
    }
  }
}

__global__ __launch_bounds__(ASM_BLOCK) void matfree_gather(const int32_t* __restrict__ off, const int32_t* __restrict__ cellT,
                                                            const uint8_t* __restrict__ liT, const uint8_t* __restrict__ bc,
                                                            const double* __restrict__ w, int64_t ncells,
                                                            const double* __restrict__ u, double* __restrict__ y,
                                                            int64_t nrows, int64_t nslices, double* __restrict__ partials,
                                                            const int* __restrict__ stop_flag)
{
  if (stop_flag && *stop_flag)
    return;
  __shared__ double red[ASM_BLOCK / 64];
  const int lane = threadIdx.x & 63;
  double dot = 0.0;
  for (int64_t s = blockIdx.x * 4 + (threadIdx.x >> 6); s < nslices; s += (int64_t)gridDim.x * 4)
  {
    const int o = off[s], len = (off[s + 1] - o) >> 6;
    const int64_t r = s * 64 + lane;
    const int32_t* __restrict__ cp = cellT + o + lane;
    const uint8_t* __restrict__ lp = liT + o + lane;
    double sum = 0.0;
    for (int a = 0; a < len; a += 8)
    {
      int32_t c[8];
      int li[8];
      double wv[8];
#pragma unroll
      for (int k = 0; k < 8; ++k)
      {
        const int aa = min(a + k, len - 1) * 64;
        c[k] = cp[aa];
        li[k] = lp[aa];
      }
#pragma unroll
      for (int k = 0; k < 8; ++k)
        wv[k] = c[k] >= 0 ? w[(int64_t)li[k] * ncells + c[k]] : 0.0;
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (a + k < len && c[k] >= 0)
          sum += wv[k]; // ascending cell order, like the serial assembly loop
    }
    if (r < nrows)
    {
      if (bc[r])
        sum = 0.0; // bc->set(y.array(), std::nullopt, 0.0), src/cgpoisson_problem.cpp:207
      y[r] = sum;
      dot += sum * u[r];
    }
  }
  if (partials)
  {
    const double t = block_reduce_sum(dot, red);
    if (threadIdx.x == 0)
      partials[blockIdx.x] = t;
  }
}

// slice-transposed adjacency with local indices: built once per pattern, read by every row-gather kernel
// slice offsets of the transposed adjacency and its (unfilled) arrays
int build_adjT_offsets(zzz_ctx* ctx)
{
  const int64_t nrows = ctx->n_owned;
  const int64_t nsl = (nrows + 63) / 64;
  ctx->have_adj_li = false;
  {
    DevBuf<int32_t> slen;
    DevBuf<unsigned char> tmp;
    ZZZ_HIP(ctx, slen.alloc((size_t)nsl + 1));
    ZZZ_HIP(ctx, ctx->adjT_off.alloc((size_t)nsl + 1));
    int g0 = (int)((nsl + 256) / 256);
    if (g0 > 4096)
      g0 = 4096;
    DevBuf<unsigned long long> tot64;
    ZZZ_HIP(ctx, tot64.alloc(1));
    ZZZ_HIP(ctx, hipMemsetAsync(tot64.p, 0, sizeof(unsigned long long), ctx->stream));
    hipLaunchKernelGGL(k_adjT_slice_len, dim3(g0), dim3(256), 0, ctx->stream, ctx->adj_off.p, nrows, nsl, slen.p, tot64.p);
    unsigned long long h64 = 0;
    ZZZ_HIP(ctx, hipMemcpyAsync(&h64, tot64.p, sizeof(h64), hipMemcpyDeviceToHost, ctx->stream));
    ZZZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (h64 > (unsigned long long)(INT32_MAX - 4096))
      return fail(ctx, ZZZ_ERR_LIMIT, "transposed adjacency (%llu entries) exceeds int32: use more parts", h64);
    size_t tb = 0;
    ZZZ_HIP(ctx, rocprim::exclusive_scan(nullptr, tb, slen.p, ctx->adjT_off.p, 0, (size_t)nsl + 1, rocprim::plus<int32_t>(),
                                         ctx->stream));
    ZZZ_HIP(ctx, tmp.alloc(tb));
    ZZZ_HIP(ctx, rocprim::exclusive_scan(tmp.p, tb, slen.p, ctx->adjT_off.p, 0, (size_t)nsl + 1, rocprim::plus<int32_t>(),
                                         ctx->stream));
    int32_t total = 0;
    ZZZ_HIP(ctx, hipMemcpyAsync(&total, ctx->adjT_off.p + nsl, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    ZZZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (total < 0)
      return fail(ctx, ZZZ_ERR_LIMIT, "transposed adjacency exceeds int32");
    ZZZ_HIP(ctx, ctx->adjT_cells.alloc((size_t)total + 64));
    ZZZ_HIP(ctx, ctx->adj_li.alloc((size_t)total + 64));
  }
  return ZZZ_OK;
}

int build_adjT(zzz_ctx* ctx)
{
  if (ctx->have_adj_li) // the P1 pattern kernel has written it already
    return ZZZ_OK;
  int rc = build_adjT_offsets(ctx);
  if (rc)
    return rc;
  const int64_t nrows = ctx->n_owned;
  const int64_t nsl = (nrows + 63) / 64;
  {
    int g1 = (int)((nsl + 3) / 4);
    if (g1 > 8192)
      g1 = 8192;
    hipLaunchKernelGGL(k_adjT_fill, dim3(g1), dim3(256), 0, ctx->stream, ctx->adj_off.p, ctx->adj_cells.p, ctx->cell_dofs.p,
                       ctx->nd, nrows, nsl, ctx->adjT_off.p, ctx->adjT_cells.p, ctx->adj_li.p);
    ZZZ_HIP(ctx, hipGetLastError());
    ZZZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->have_adj_li = true;
  }
  return ZZZ_OK;
}

// The operator of cgpoisson: the one-pass cell-block kernel of zzz_matfree.hip on its plan (built on first use, or by
// zzz_matfree_setup); ZZZ_MF_LEGACY=1 keeps the two-pass form below (A/B, and the parity test between the two).
int launch_matfree_action(zzz_ctx* ctx, const double* u, double* y, double* partials, int* npartials)
{
  const char* e = getenv("ZZZ_MF_LEGACY");
  if (e && atoi(e) != 0)
    return launch_matfree_legacy(ctx, u, y, partials, npartials);
  if (!ctx->mf.valid && !ctx->mf.failed)
  {
    const int rc = mf_plan_build(ctx);
    if (rc == ZZZ_ERR_LIMIT && ctx->have_pattern && ctx->have_adj_li)
      ctx->mf.failed = true; // a mesh the plan cannot hold (a dof in more than 254 cells of one step): the two-pass form serves it
    else if (rc)
      return rc;
  }
  if (ctx->mf.failed)
    return launch_matfree_legacy(ctx, u, y, partials, npartials);
  return mf_action(ctx, u, y, partials, npartials);
}

// diag(A) without the matrix (1.0 on constrained rows): the cell-block pass with the element matrices' diagonals, for
// Jacobi on the matrix-free operator.  Needs the plan (no two-pass form of it exists).
int launch_matfree_diagonal(zzz_ctx* ctx, double* d)
{
  if (!ctx->mf.valid)
  {
    if (ctx->mf.failed)
      return fail(ctx, ZZZ_ERR_LIMIT, "the matrix-free diagonal needs the cell-block plan, which this mesh does not fit");
    if (int rc = mf_plan_build(ctx))
      return rc;
  }
  return mf_diagonal(ctx, d);
}

int launch_matfree_legacy(zzz_ctx* ctx, const double* u, double* y, double* partials, int* npartials)
{
  if (ctx->bs != 1)
    return fail(ctx, ZZZ_ERR_ARG, "the matrix-free operator exists for the Poisson form M only (src/Poisson.py:33)");
  if (!ctx->have_pattern)
    return fail(ctx, ZZZ_ERR_ARG, "matrix-free operator needs zzz_csr_pattern_build (dof->cell adjacency)");
  int rc = ensure_tables(ctx);
  if (rc)
    return rc;
  const int64_t nrows = ctx->n_owned, nc = ctx->ncells;
  const int64_t nsl = (nrows + 63) / 64;
  if (!ctx->have_adj_li)
    return fail(ctx, ZZZ_ERR_ARG, "transposed adjacency missing (zzz_csr_pattern_build not run)");
  if (ctx->cell_w.n < (size_t)(nc * ctx->nd))
    ZZZ_HIP(ctx, ctx->cell_w.alloc((size_t)(nc * ctx->nd)));
  const int* stop = partials ? reinterpret_cast<const int*>(ctx->state.p) : nullptr;
  int64_t gc = (nc + ASM_BLOCK - 1) / ASM_BLOCK;
  if (gc > 4096)
    gc = 4096;
#define ZZZ_MFC(ND_)                                                                                                    \
  hipLaunchKernelGGL(matfree_cell<ND_>, dim3((unsigned)gc), dim3(ASM_BLOCK), 0, ctx->stream, ctx->x.p, ctx->cell_verts.p,  \
                     ctx->cell_dofs.p, nc, u, ctx->cell_w.p, ctx->tables.p, stop)
  if (ctx->order == 1)
    ZZZ_MFC(4);
  else if (ctx->order == 2)
    ZZZ_MFC(10);
  else
    ZZZ_MFC(20);
#undef ZZZ_MFC
  int64_t g = (nsl + 3) / 4;
  if (g > 2048)
    g = 2048;
  hipLaunchKernelGGL(matfree_gather, dim3((unsigned)g), dim3(ASM_BLOCK), 0, ctx->stream, ctx->adjT_off.p, ctx->adjT_cells.p,
                     ctx->adj_li.p, ctx->bc.p, ctx->cell_w.p, nc, u, y, nrows, nsl, partials, stop);
  if (npartials)
    *npartials = (int)g;
  ZZZ_HIP(ctx, hipGetLastError());
  return ZZZ_OK;
}
ZZZ_PRELOAD_TU(assemble)
} // namespace zzz

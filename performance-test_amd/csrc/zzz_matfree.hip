// Matrix-free operator of --problem_type cgpoisson on gfx950:  y = action(a, un = x), then y[bc] = 0
// (form M of src/Poisson.py:33, applied as in src/cgpoisson_problem.cpp:193-230: y = 0, assemble_vector(y, M) with
// un = x, bc->set(y, 0)); it is the operator linalg::cg iterates on (src/cg.h:46,62, called at
// src/cgpoisson_problem.cpp:233).  No element matrix is stored and nothing per cell travels through HBM twice.
//
// ONE pass, cell blocks in LDS:
//   * the cells are cut into blocks of `nc` that are contiguous in the Morton order of their centroids (any mesh: no
//     lattice is assumed); a block knows the dofs it touches (`dof_ids`: interior to the block | shared with other
//     blocks | ghost) and every cell carries 16-bit indices into that list;
//   * a workgroup owns a block: it stages u (P1: and the vertex coordinates) of the block's dofs in LDS, then walks the
//     block in steps of one cell per lane.  The cells of a step are dealt out so that neighbours in space sit in
//     different steps;
//   * P1: geometry from the staged coordinates (cofactors, one division), y_e = c (c^T u) / (6 |det J|);
//     P2/P3: FACTORISED stiffness.  d_a phi_j lies in P_(k-1), so with an orthonormal basis psi_q of P_(k-1)
//       S^ab_ij = sum_q D_a[q][i] D_b[q][j]   =>   y_e = sum_a D_a^T ( sum_b G_ab (D_b u_e) ),   G = |detJ| K K^T
//     (ZZZ_DTAB_P2/P3 of element_tables.inc, 2 x 3 nq nd + 9 nq multiply-adds per cell with the tables' zeros skipped
//     at compile time: 872 for P3 instead of 2 400 for the six nd x nd tensors); G per cell is read, not recomputed;
//   * the element vector is added into the block's y in LDS in ROUNDS: the plan gives every (cell, local dof) incidence
//     its rank among the incidences of the same dof in the same step; round r adds the incidences of rank r, so no two
//     lanes meet on an address, nothing is atomic and the order of additions is fixed: reproducible bit for bit;
//   * dofs interior to the block leave as y directly; dofs shared with other blocks leave as partial sums and
//     `k_mf_finish` adds each one's partials in ascending block order.
// The partial sums of <x, y> for the CG's dot product ride along (src/cg.h:65).
#include <algorithm>
#include <climits>
#include <cstdlib>
#include <cstring>

#include "zzz_device.h"
#include "zzz_internal.h"

#include <rocprim/rocprim.hpp>

#include "element_tables.inc"

namespace zzz
{
namespace
{
constexpr int MF_HDR = 8;

// ---- plan build ------------------------------------------------------------------------------------------------------
__device__ inline uint32_t spread10(uint32_t v)
{
  v &= 0x3ffu;
  v = (v | (v << 16)) & 0x030000ffu;
  v = (v | (v << 8)) & 0x0300f00fu;
  v = (v | (v << 4)) & 0x030c30c3u;
  v = (v | (v << 2)) & 0x09249249u;
  return v;
}

__global__ __launch_bounds__(256) void k_mf_bbox(const double* __restrict__ x, int64_t nverts, double* __restrict__ out)
{
  __shared__ double sh[6][4];
  double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
  for (int64_t v = blockIdx.x * 256ll + threadIdx.x; v < nverts; v += gridDim.x * 256ll)
    for (int a = 0; a < 3; ++a)
    {
      const double c = x[3 * v + a];
      lo[a] = fmin(lo[a], c);
      hi[a] = fmax(hi[a], c);
    }
  for (int a = 0; a < 3; ++a)
    for (int o = 32; o > 0; o >>= 1)
    {
      lo[a] = fmin(lo[a], __shfl_down(lo[a], o, 64));
      hi[a] = fmax(hi[a], __shfl_down(hi[a], o, 64));
    }
  if ((threadIdx.x & 63) == 0)
    for (int a = 0; a < 3; ++a)
    {
      sh[a][threadIdx.x >> 6] = lo[a];
      sh[3 + a][threadIdx.x >> 6] = hi[a];
    }
  __syncthreads();
  if (threadIdx.x < 6)
  {
    double v = sh[threadIdx.x][0];
    for (int i = 1; i < 4; ++i)
      v = threadIdx.x < 3 ? fmin(v, sh[threadIdx.x][i]) : fmax(v, sh[threadIdx.x][i]);
    out[blockIdx.x * 6 + threadIdx.x] = v;
  }
}

struct Box
{
  double lo[3], scale[3];
};

__global__ __launch_bounds__(256) void k_mf_cell_keys(const double* __restrict__ x, const int32_t* __restrict__ cell_verts,
                                                      int64_t ncells, Box B, uint32_t* __restrict__ key,
                                                      int32_t* __restrict__ val)
{
  for (int64_t c = blockIdx.x * 256ll + threadIdx.x; c < ncells; c += gridDim.x * 256ll)
  {
    const int4 v = *reinterpret_cast<const int4*>(cell_verts + 4 * c);
    uint32_t q[3];
    for (int a = 0; a < 3; ++a)
    {
      const double m = 0.25 * (x[3ll * v.x + a] + x[3ll * v.y + a] + x[3ll * v.z + a] + x[3ll * v.w + a]);
      const double t = (m - B.lo[a]) * B.scale[a];
      q[a] = (uint32_t)min(1023, max(0, (int)t));
    }
    key[c] = spread10(q[0]) | (spread10(q[1]) << 1) | (spread10(q[2]) << 2);
    val[c] = (int32_t)c;
  }
}

// Which step of its block a cell runs in.  The element vectors of a step are added into the block's y in rounds (one per
// incidence of the most-visited dof of the step), so the steps should spread the cells around every dof evenly.  One
// wavefront per block walks the block's cells in 64 interleaved Morton sequences (lane l takes cells l * nbatch, l * nbatch
// + 1, ...: the 64 cells of a batch lie far apart, consecutive batches are neighbours): every lane prices the steps for its
// cell -- max over the cell's dofs of the cells a step already holds there, from byte counters per (dof, step) in an LDS hash
// table keyed by the global dof -- and takes the cheapest step that has room (ties: rotated by the lane, so that the lanes of a
// batch, which cannot see each other's choice, spread over the steps: 4.9 -> 4.5 rounds at P1); the lanes that picked
// one step get consecutive places in it in lane order, those beyond its capacity pick again.  P1, 8 steps: 4.5 rounds per
// step against 6.4 for dealing the cells out in turn (4.0 when one cell at a time chooses); P3, 3 steps: 9.5 against 11.3
// (8.3).  Deterministic: picks depend on the
// counters at the start of the batch and on lane order only (which hash slot a dof gets does not matter).  A first version
// walked the cells one by one (36-57 ms at P1 10 M dofs: a chain of dependent LDS reads per cell); this one takes 3.8 ms.
// More than 16 steps: dealt out in turn.
__global__ __launch_bounds__(64) void k_mf_assign(const int32_t* __restrict__ sorted, const int32_t* __restrict__ cell_dofs, int nd,
                                                  int64_t ncells, int nc, int T, int nsb, int cw, int H, int W, int64_t nblocks,
                                                  int32_t* __restrict__ mf_cell)
{
  extern __shared__ __align__(16) unsigned char as_lds[];
  int32_t* const keys = reinterpret_cast<int32_t*>(as_lds);  // [H] global dof or -1
  int32_t* const fill = keys + H;                             // [16] cells per step
  uint32_t* const cntw = reinterpret_cast<uint32_t*>(fill + 16); // [H][cw] words = 4 cw byte counters per slot (cw 1, 2 or 4)
  const int lane = threadIdx.x;
  for (int64_t b = blockIdx.x; b < nblocks; b += gridDim.x)
  {
    const int ncb = (int)min((int64_t)nc, ncells - b * nc);
    for (int i = lane; i < nc; i += 64)
      mf_cell[b * nc + i] = -1;
    if (nsb > 16)
    {
      for (int kk = lane; kk < ncb; kk += 64)
        mf_cell[b * nc + (kk % nsb) * T + kk / nsb] = sorted[b * nc + kk];
      continue;
    }
    for (int i = lane; i < H; i += 64)
      keys[i] = -1;
    for (int i = lane; i < H * cw; i += 64)
      cntw[i] = 0;
    if (lane < 16)
      fill[lane] = 0;
    __syncthreads();
    const int nbatch = (ncb + W - 1) / W;
    for (int t = 0; t < nbatch; ++t)
    {
      const int kk = lane * nbatch + t;
      const bool have = lane < W && kk < ncb;
      const int32_t cell = have ? sorted[b * nc + kk] : -1;
      // hash slots of the cell's dofs and the price of every step
      int hs[20];
      int cost[16];
#pragma unroll
      for (int sI = 0; sI < 16; ++sI)
        cost[sI] = 0;
#pragma unroll
      for (int j = 0; j < 20; ++j)
      {
        int found = -1;
        if (have && j < nd)
        {
          const int32_t g = cell_dofs[(int64_t)cell * nd + j];
          int h = (int)(((uint32_t)g * 2654435761u) >> 7) & (H - 1);
          for (int probe = 0; probe < H; ++probe)
          {
            const int32_t old = atomicCAS(&keys[h], -1, g);
            if (old == -1 || old == g)
            {
              found = h;
              break;
            }
            h = (h + 1) & (H - 1);
          }
        }
        hs[j] = found; // -1: table full (a block far beyond the LDS budget: the plan is retried smaller anyway)
      }
      __syncthreads(); // (counters as they stand at the start of the batch)
#pragma unroll
      for (int j = 0; j < 20; ++j)
        if (hs[j] >= 0)
        {
          const uint32_t* const w = cntw + cw * hs[j];
          const uint32_t ww[4] = {w[0], cw > 1 ? w[1] : 0u, cw > 2 ? w[2] : 0u, cw > 2 ? w[3] : 0u};
#pragma unroll
          for (int sI = 0; sI < 16; ++sI)
            cost[sI] = max(cost[sI], (int)((ww[sI >> 2] >> (8 * (sI & 3))) & 255u));
        }
      int pick = -1, pos = 0;
      bool placed = !have;
      for (int guard = 0; guard < 64 && __ballot(!placed) != 0ull; ++guard)
      {
        // cheapest step with room (ties: rotated by the lane)
        int best = INT_MAX;
        if (!placed)
        {
#pragma unroll
          for (int sI = 0; sI < 16; ++sI)
            if (sI < nsb && fill[sI] < T)
              best = min(best, (cost[sI] << 12) | (((sI + lane) % nsb) << 6) | sI);
        }
        const int want = (!placed && best != INT_MAX) ? (best & 63) : -1;
        __syncthreads();
        for (int sI = 0; sI < nsb; ++sI)
        {
          const unsigned long long m = __ballot(want == sI);
          if (m == 0ull)
            continue;
          const int room = T - fill[sI];
          const int rank = __popcll(m & ((1ull << lane) - 1ull));
          if (want == sI && rank < room)
          {
            placed = true;
            pick = sI;
            pos = fill[sI] + rank;
          }
          __syncthreads();
          if (lane == 0)
            fill[sI] += min(room, (int)__popcll(m));
          __syncthreads();
        }
      }
      if (have && pick >= 0)
      {
        mf_cell[b * nc + pick * T + pos] = cell;
#pragma unroll
        for (int j = 0; j < 20; ++j)
          if (hs[j] >= 0)
            atomicAdd(&cntw[cw * hs[j] + (pick >> 2)], 1u << (8 * (pick & 3))); // (a byte counter: at most nc / T... < 256 cells)
      }
      __syncthreads();
    }
    __syncthreads();
  }
}

// one key per incidence: the dof (the block is the segment the incidence sits in: nc * nd incidences per block); value =
// position in the block * nd + local index.  Cells that are not there (the last block's tail): key 0xffffffff, behind
// everything else of their segment.
__global__ __launch_bounds__(256) void k_mf_pairs(const int32_t* __restrict__ mf_cell, const int32_t* __restrict__ cell_dofs,
                                                  int nd, int nc, int64_t total, uint32_t* __restrict__ key,
                                                  uint32_t* __restrict__ val)
{
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total * nd; i += gridDim.x * 256ll)
  {
    const int64_t pe = i / nd;
    const int li = (int)(i - pe * nd);
    const int64_t b = pe / nc;
    const int e = (int)(pe - b * nc);
    const int32_t c = mf_cell[pe];
    key[i] = c < 0 ? 0xffffffffu : (uint32_t)cell_dofs[(int64_t)c * nd + li];
    val[i] = (uint32_t)(e * nd + li);
  }
}

__global__ __launch_bounds__(256) void k_mf_heads(const uint32_t* __restrict__ key, int64_t n, int seg, int32_t* __restrict__ flag)
{
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += gridDim.x * 256ll)
    flag[i] = (key[i] != 0xffffffffu && (i % seg == 0 || key[i] != key[i - 1])) ? 1 : 0;
}

// at the head of every run: the unique's key and where its run starts; how many blocks hold each dof
__global__ __launch_bounds__(256) void k_mf_uniques(const uint32_t* __restrict__ key, const int32_t* __restrict__ flag,
                                                    const int32_t* __restrict__ uidx, int64_t n, int seg, uint64_t* __restrict__ ukey,
                                                    int32_t* __restrict__ run_start, int32_t* __restrict__ nblk_of)
{
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += gridDim.x * 256ll)
    if (flag[i])
    {
      const int32_t j = uidx[i];
      ukey[j] = ((uint64_t)(i / seg) << 32) | key[i];
      run_start[j] = (int32_t)i;
      atomicAdd(&nblk_of[key[i]], 1);
    }
}

// class of every unique (0 interior to its block, 1 shared, 2 ghost), the sort key that groups them, counts per block
__global__ __launch_bounds__(256) void k_mf_classify(const uint64_t* __restrict__ ukey, int64_t nu,
                                                     const int32_t* __restrict__ nblk_of, int64_t n_owned,
                                                     uint64_t* __restrict__ key2, int32_t* __restrict__ val2,
                                                     int32_t* __restrict__ cnt3)
{
  for (int64_t j0 = blockIdx.x * 256ll + (threadIdx.x & ~63); j0 < nu; j0 += gridDim.x * 256ll)
  {
    const int64_t j = j0 + (threadIdx.x & 63);
    uint64_t b = ~0ull;
    int c = -1;
    if (j < nu)
    {
      const uint64_t k = ukey[j];
      const uint32_t g = (uint32_t)k;
      b = k >> 32;
      c = (int64_t)g >= n_owned ? 2 : (nblk_of[g] > 1 ? 1 : 0);
      key2[j] = (b << 34) | ((uint64_t)c << 32) | g;
      val2[j] = (int32_t)j;
    }
    // the uniques of a block sit side by side: a wavefront counts for its first lane's block together (three counters per
    // block took the atomics of thousands of uniques one at a time), the few lanes of the next block on their own
    const uint64_t b0 = ((uint64_t)__builtin_amdgcn_readfirstlane((int)(b >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)b);
    for (int k = 0; k < 3; ++k)
    {
      const unsigned long long m = __ballot(b == b0 && c == k);
      if (m && (threadIdx.x & 63) == __ffsll((long long)m) - 1)
        atomicAdd(&cnt3[b0 * 3 + k], (int)__popcll(m));
    }
    if (c >= 0 && b != b0)
      atomicAdd(&cnt3[b * 3 + c], 1);
  }
}

__global__ __launch_bounds__(256) void k_mf_block_sizes(const int32_t* __restrict__ cnt3, int64_t nb, int32_t* __restrict__ nloc,
                                                        int32_t* __restrict__ nsh)
{
  for (int64_t b = blockIdx.x * 256ll + threadIdx.x; b <= nb; b += gridDim.x * 256ll)
  {
    nloc[b] = b < nb ? cnt3[3 * b] + cnt3[3 * b + 1] + cnt3[3 * b + 2] : 0;
    nsh[b] = b < nb ? cnt3[3 * b + 1] : 0;
  }
}

__global__ __launch_bounds__(256) void k_mf_headers(const int32_t* __restrict__ cnt3, const int32_t* __restrict__ dof_off,
                                                    const int32_t* __restrict__ part_off, int64_t nb, int32_t* __restrict__ hdr,
                                                    int32_t* __restrict__ nloc_max)
{
  int m = 0;
  for (int64_t b = blockIdx.x * 256ll + threadIdx.x; b < nb; b += gridDim.x * 256ll)
  {
    const int nl = cnt3[3 * b] + cnt3[3 * b + 1] + cnt3[3 * b + 2];
    hdr[MF_HDR * b + 0] = dof_off[b];
    hdr[MF_HDR * b + 1] = nl;
    hdr[MF_HDR * b + 2] = cnt3[3 * b];
    hdr[MF_HDR * b + 3] = cnt3[3 * b + 1];
    hdr[MF_HDR * b + 4] = part_off[b];
    hdr[MF_HDR * b + 5] = hdr[MF_HDR * b + 6] = hdr[MF_HDR * b + 7] = 0;
    m = max(m, nl);
  }
  m = wave_max_i(m);
  if ((threadIdx.x & 63) == 0 && m)
    atomicMax(nloc_max, m);
}

// the block-local lists in their final order (sorted key2): global dof, Dirichlet flag, coordinates; local index of every
// unique; (dof, slot) of the shared ones
__global__ __launch_bounds__(256) void k_mf_lists(const uint64_t* __restrict__ key2s, const int32_t* __restrict__ val2s, int64_t nu,
                                                  const int32_t* __restrict__ hdr, const uint8_t* __restrict__ bc,
                                                  const double* __restrict__ xq, int32_t* __restrict__ dof_ids,
                                                  uint8_t* __restrict__ dof_flag, double* __restrict__ xyz,
                                                  int32_t* __restrict__ loc_of, uint32_t* __restrict__ sh_key,
                                                  int32_t* __restrict__ sh_val)
{
  for (int64_t q = blockIdx.x * 256ll + threadIdx.x; q < nu; q += gridDim.x * 256ll)
  {
    const uint64_t k = key2s[q];
    const uint32_t g = (uint32_t)k;
    const int c = (int)((k >> 32) & 3);
    const int64_t b = (int64_t)(k >> 34);
    const int32_t* h = hdr + MF_HDR * b;
    const int loc = (int)(q - h[0]);
    dof_ids[q] = (int32_t)g;
    dof_flag[q] = bc[g];
    if (xyz)
    {
      xyz[3 * q + 0] = xq[3ll * g + 0];
      xyz[3 * q + 1] = xq[3ll * g + 1];
      xyz[3 * q + 2] = xq[3ll * g + 2];
    }
    loc_of[val2s[q]] = loc;
    sh_key[q] = c == 1 ? g : 0xffffffffu;
    sh_val[q] = c == 1 ? h[4] + (loc - h[2]) : -1;
  }
}

// per incidence (in the order of the sorted pairs): its 16-bit local index and its rank among the incidences of the same
// dof in the same step (the run of a (block, dof) key holds them by ascending position).  One workgroup per block: the
// block's words are put together in LDS and leave as whole words (the sorted order scatters the 2-byte and 1-byte items
// over the block's words: written to memory one by one they cost 7 ms at P1 10 M dofs, this way 2).
__global__ __launch_bounds__(1024) void k_mf_cells(const uint32_t* __restrict__ key, const uint32_t* __restrict__ val,
                                                  const int32_t* __restrict__ uidx, const int32_t* __restrict__ flag, int64_t nb,
                                                  int seg, const int32_t* __restrict__ run_start, const int32_t* __restrict__ loc_of, int nd,
                                                  int nc, int T, int ndw, int nrw, uint32_t* __restrict__ idxw,
                                                  uint32_t* __restrict__ rnkw, int32_t* __restrict__ err)
{
  extern __shared__ uint32_t cells_lds[]; // [ndw * nc] index words, [nrw * nc] rank words
  uint32_t* const iw = cells_lds;
  uint32_t* const rw = cells_lds + ndw * nc;
  uint16_t* const idx16 = reinterpret_cast<uint16_t*>(iw);
  uint8_t* const rnk8 = reinterpret_cast<uint8_t*>(rw);
  for (int64_t b = blockIdx.x; b < nb; b += gridDim.x)
  {
    for (int k = threadIdx.x; k < ndw * nc; k += 1024)
      iw[k] = 0;
    for (int k = threadIdx.x; k < nrw * nc; k += 1024)
      rw[k] = 0xffffffffu; // (a rank byte of 0xff: no cell)
    __syncthreads();
    for (int t = threadIdx.x; t < seg; t += 1024)
    {
      const int64_t i = b * seg + t;
      if (key[i] == 0xffffffffu)
        continue;
      const int32_t j = uidx[i] - (flag[i] ? 0 : 1); // uidx is the exclusive scan of the head flags
      const uint32_t v = val[i];
      const int e = (int)(v / (uint32_t)nd), li = (int)(v - (uint32_t)e * nd);
      const int s = e / T;
      // (positions ascend inside a run, so the incidences of one step sit side by side: walk back to the step's first)
      int rank = 0;
      for (int64_t p = i - 1, p0 = run_start[j]; p >= p0 && (int)(val[p] / (uint32_t)nd) / T == s; --p)
        ++rank;
      if (rank > 254)
      {
        *err = 1;
        rank = 254;
      }
      idx16[((li >> 1) * nc + e) * 2 + (li & 1)] = (uint16_t)loc_of[j];
      rnk8[((li >> 2) * nc + e) * 4 + (li & 3)] = (uint8_t)rank;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < ndw * nc; k += 1024)
      idxw[b * ndw * nc + k] = iw[k];
    for (int k = threadIdx.x; k < nrw * nc; k += 1024)
      rnkw[b * nrw * nc + k] = rw[k];
    __syncthreads();
  }
}
// rounds local dof i needs in every (block, step): 1 + the largest rank among the step's cells
__global__ __launch_bounds__(256) void k_mf_rmax(const uint32_t* __restrict__ rnkw, int64_t nb, int nc, int T, int nsb, int nrw,
                                                 uint8_t* __restrict__ rmax)
{
  // one wavefront per (block, step, word of four local dofs); a rank byte of 0xff = no cell
  const int lane = threadIdx.x & 63;
  const int64_t items = nb * nsb * nrw;
  for (int64_t it = blockIdx.x * 4ll + (threadIdx.x >> 6); it < items; it += gridDim.x * 4ll)
  {
    const int w = (int)(it % nrw);
    const int64_t bs_ = it / nrw;
    const int s = (int)(bs_ % nsb);
    const int64_t b = bs_ / nsb;
    const uint32_t* src = rnkw + (b * nrw + w) * nc + (int64_t)s * T;
    int m[4] = {0, 0, 0, 0};
    for (int t = lane; t < T; t += 64)
    {
      const uint32_t r = src[t];
      for (int k = 0; k < 4; ++k)
      {
        const int rb = (int)((r >> (8 * k)) & 255u);
        m[k] = max(m[k], rb == 255 ? 0 : rb + 1);
      }
    }
    for (int k = 0; k < 4; ++k)
      m[k] = wave_max_i(m[k]);
    if (lane == 0)
      reinterpret_cast<uint32_t*>(rmax)[(b * nsb + s) * nrw + w] =
          (uint32_t)m[0] | ((uint32_t)m[1] << 8) | ((uint32_t)m[2] << 16) | ((uint32_t)m[3] << 24);
  }
}

__global__ __launch_bounds__(256) void k_mf_sh_heads(const uint32_t* __restrict__ key, int64_t n, int32_t* __restrict__ flag)
{
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += gridDim.x * 256ll)
    flag[i] = (i == 0 || key[i] != key[i - 1]) ? 1 : 0;
}
// the partial sums of a shared dof sit side by side (ascending block): position p of the sorted (dof, old slot) pairs is
// where the block that owns old slot `val[p]` stores that sum
__global__ __launch_bounds__(256) void k_mf_sh_fill(const uint32_t* __restrict__ key, const int32_t* __restrict__ val,
                                                    const int32_t* __restrict__ flag, const int32_t* __restrict__ uidx, int64_t n,
                                                    const uint8_t* __restrict__ bc, int32_t* __restrict__ sh_dof,
                                                    int32_t* __restrict__ sh_off, uint8_t* __restrict__ sh_flag,
                                                    int32_t* __restrict__ pslot, int64_t nshared)
{
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i <= n; i += gridDim.x * 256ll)
  {
    if (i == n)
    {
      sh_off[nshared] = (int32_t)n;
      continue;
    }
    pslot[val[i]] = (int32_t)i;
    if (flag[i])
    {
      sh_dof[uidx[i]] = (int32_t)key[i];
      sh_off[uidx[i]] = (int32_t)i;
      sh_flag[uidx[i]] = bc[key[i]];
    }
  }
}

// P3: the global dofs of every cell in plan order, component-major inside the block (cells without a cell: dof 0)
__global__ __launch_bounds__(256) void k_mf_gid(const int32_t* __restrict__ mf_cell, const int32_t* __restrict__ cell_dofs, int nd,
                                                int nc, int64_t total, int32_t* __restrict__ gid)
{
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total * nd; i += gridDim.x * 256ll)
  {
    const int64_t pe = i / nd;
    const int j = (int)(i - pe * nd);
    const int64_t b = pe / nc;
    const int e = (int)(pe - b * nc);
    const int32_t c = mf_cell[pe];
    gid[(b * nd + j) * nc + e] = c < 0 ? 0 : cell_dofs[(int64_t)c * nd + j];
  }
}

// P2/P3: |detJ| K K^T of every cell, in plan order, component-major inside the block
__global__ __launch_bounds__(256) void k_mf_geom(const double* __restrict__ x, const int32_t* __restrict__ cell_verts,
                                                 const int32_t* __restrict__ mf_cell, int nc, int64_t total,
                                                 double* __restrict__ geom)
{
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += gridDim.x * 256ll)
  {
    const int64_t b = i / nc;
    const int e = (int)(i - b * nc);
    const int32_t c = mf_cell[i];
    double G[6] = {0, 0, 0, 0, 0, 0};
    if (c >= 0)
    {
      const int4 v = *reinterpret_cast<const int4*>(cell_verts + 4ll * c);
      const int vv[4] = {v.x, v.y, v.z, v.w};
      double p[4][3];
      for (int k = 0; k < 4; ++k)
        for (int a = 0; a < 3; ++a)
          p[k][a] = x[3ll * vv[k] + a];
      double J[3][3];
      for (int a = 0; a < 3; ++a)
        for (int al = 0; al < 3; ++al)
          J[a][al] = p[al + 1][a] - p[0][a];
      // K = J^-1 = C / det, K[al][a] = dX_al / dx_a
      double C[3][3];
      C[0][0] = J[1][1] * J[2][2] - J[1][2] * J[2][1];
      C[0][1] = J[0][2] * J[2][1] - J[0][1] * J[2][2];
      C[0][2] = J[0][1] * J[1][2] - J[0][2] * J[1][1];
      C[1][0] = J[1][2] * J[2][0] - J[1][0] * J[2][2];
      C[1][1] = J[0][0] * J[2][2] - J[0][2] * J[2][0];
      C[1][2] = J[0][2] * J[1][0] - J[0][0] * J[1][2];
      C[2][0] = J[1][0] * J[2][1] - J[1][1] * J[2][0];
      C[2][1] = J[0][1] * J[2][0] - J[0][0] * J[2][1];
      C[2][2] = J[0][0] * J[1][1] - J[0][1] * J[1][0];
      const double det = J[0][0] * C[0][0] + J[0][1] * C[1][0] + J[0][2] * C[2][0];
      const double sc = 1.0 / fabs(det); // |det| K K^T = C C^T / |det|
      const int pa[6] = {0, 1, 2, 0, 0, 1}, pb[6] = {0, 1, 2, 1, 2, 2};
      for (int t = 0; t < 6; ++t)
        G[t] = (C[pa[t]][0] * C[pb[t]][0] + C[pa[t]][1] * C[pb[t]][1] + C[pa[t]][2] * C[pb[t]][2]) * sc;
    }
    for (int t = 0; t < 6; ++t)
      geom[(b * 6 + t) * nc + e] = G[t];
  }
}

// ---- the action ------------------------------------------------------------------------------------------------------
struct MfArgs
{
  const int32_t* hdr;
  const int32_t* dof_ids;
  const uint8_t* dof_flag;
  const double* xyz;
  const uint32_t* idxw;
  const uint32_t* rnkw;
  const uint8_t* rmax;
  const double* geom;
  const double* dtab;
  const int32_t* pslot;
  const int32_t* gid;
  double* ypart;
  const double* u;
  double* y;
  double* partials;
  const int* stop;
  int64_t nblocks;
  int nc, nsb, nloc_cap;
  int dbg; // ZZZ_EXPERIMENTS builds only (ZZZ_MF_DEBUG): phases switched off for timing, results wrong
};
#ifdef ZZZ_EXPERIMENTS
#define MF_DBG(bit) (A.dbg & (bit))
#else
#define MF_DBG(bit) 0
#endif

// The factorised tables: constexpr copies decide at compile time which entries are zero; the values are staged in LDS
// by every (persistent) workgroup and reach the multiply-adds as broadcast reads.  (As literals they occupied ~200
// vector registers of every lane; as scalar loads from constant memory the compiler hoisted them all and spilled 865
// scalar registers; with the loads chained section by section through empty asm statements -- rows of the table as
// scalar operands, no LDS traffic for them -- the kernel still spilled 410 scalar registers into vector lanes and was
// 3 % faster at P3 6.2 M dofs, 0.281 against 0.291 ms, 1 % at P2: measured in round 4, not kept.)
template <int ND>
struct MfTab;
template <>
struct MfTab<10>
{
  static constexpr int NQ = 4;
  static constexpr bool nz(int a, int q, int j) { return ZZZ_DTAB_P2[(a * 4 + q) * 10 + j] != 0.0; }
};
template <>
struct MfTab<20>
{
  static constexpr int NQ = 10;
  static constexpr bool nz(int a, int q, int j) { return ZZZ_DTAB_P3[(a * 10 + q) * 20 + j] != 0.0; }
};

// y_e = sum_q sum_a D_a[q][:]^T h_a(q),  h(q) = G g(q),  g_a(q) = D_a[q][:] . u_e -- mode q by mode q, so that only u_e,
// y_e and six scalars are live.  A table entry is READ TWICE, once for each of its uses, the second time from a second
// copy of the table laid out for that use ([q][j][a]; the compiler cannot tell that the two are equal): kept in
// registers between the uses, the ~39 entries of a mode cost 78 vector registers, 220 in all, two wavefronts per SIMD.  Multiply-adds are fused here (the library is otherwise built with
// -ffp-contract=off): the action is compared with the oracle to a tolerance, not bit for bit.
template <int ND>
__device__ inline void mf_element_pk(const double* __restrict__ tab, const double* __restrict__ tabT, const double (&ue)[ND],
                                     const double (&G)[6], double (&ye)[ND])
{
#pragma clang fp contract(fast)
  constexpr int NQ = MfTab<ND>::NQ;
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
        if (MfTab<ND>::nz(a, q, j))
          acc += tab[(a * NQ + q) * ND + j] * ue[j];
      g[a] = acc;
    }
    const double h0 = G[0] * g[0] + G[3] * g[1] + G[4] * g[2];
    const double h1 = G[3] * g[0] + G[1] * g[1] + G[5] * g[2];
    const double h2 = G[4] * g[0] + G[5] * g[1] + G[2] * g[2];
#pragma unroll
    for (int j = 0; j < ND; ++j)
    {
      if (MfTab<ND>::nz(0, q, j))
        ye[j] += tabT[(q * ND + j) * 3 + 0] * h0;
      if (MfTab<ND>::nz(1, q, j))
        ye[j] += tabT[(q * ND + j) * 3 + 1] * h1;
      if (MfTab<ND>::nz(2, q, j))
        ye[j] += tabT[(q * ND + j) * 3 + 2] * h2;
    }
  }
}

// the element matrix's diagonal: S_jj = sum_q d(q, j)^T G d(q, j), d_a(q, j) = D_a[q][j]
template <int ND>
__device__ inline void mf_element_diag_pk(const double* __restrict__ tabT, const double (&G)[6], double (&ye)[ND])
{
#pragma clang fp contract(fast)
  constexpr int NQ = MfTab<ND>::NQ;
#pragma unroll
  for (int j = 0; j < ND; ++j)
  {
    double acc = 0.0;
#pragma unroll
    for (int q = 0; q < NQ; ++q)
    {
      if (!(MfTab<ND>::nz(0, q, j) || MfTab<ND>::nz(1, q, j) || MfTab<ND>::nz(2, q, j)))
        continue;
      const double d0 = tabT[(q * ND + j) * 3 + 0], d1 = tabT[(q * ND + j) * 3 + 1], d2 = tabT[(q * ND + j) * 3 + 2];
      acc += d0 * (G[0] * d0 + G[3] * d1 + G[4] * d2) + d1 * (G[3] * d0 + G[1] * d1 + G[5] * d2)
             + d2 * (G[4] * d0 + G[5] * d1 + G[2] * d2);
    }
    ye[j] = acc;
  }
}

// DIAG: the same pass with the element matrix's diagonal in place of the element vector: y = diag(A) in the action's own
// summation order (1.0 on constrained rows, fem::set_diagonal) -- what Jacobi needs when the operator is never assembled
template <int ND, int T, bool DIAG>
__global__ __launch_bounds__(T, (ND == 20 ? 3 : 1)) void k_mf_action(const MfArgs A)
{
  if (A.stop && *A.stop)
    return;
  constexpr int NDW = ND / 2, NRW = (ND + 3) / 4;
  extern __shared__ __align__(32) unsigned char mf_lds[];
  // P1: xy[nloc_cap] = {x, y}, zu[nloc_cap] = {z, u} (two arrays of 16-B entries: a 16-lane group of a ds_read_b128 meets
  // 16 bank positions, not the 8 of 32-B records), then ys[nloc_cap]; P2: us[nloc_cap] then ys[nloc_cap]; P3 (GU): ys only,
  // the cells gather u from memory through their global dof numbers (16 B per dof of LDS less: three workgroups per CU)
  constexpr bool GU = ND == 20;
  double* const rec = reinterpret_cast<double*>(mf_lds);
  double* const zu = rec + 2 * (size_t)A.nloc_cap;
  double* const ys = rec + (size_t)(ND == 4 ? 4 : (GU ? 0 : 1)) * A.nloc_cap;
  __shared__ double red[T / 64];
  constexpr int NTAB = ND == 4 ? 1 : 3 * ND * (ND == 10 ? 4 : 10);
  __shared__ double tab_s[NTAB], tabT_s[NTAB];
  const int tid = threadIdx.x;
  if (ND != 4)
    for (int k = tid; k < NTAB; k += T) // (the first block's barrier below orders this)
    {
      constexpr int NQ = ND == 10 ? 4 : 10;
      const double v = A.dtab[k]; // [a][q][j]
      const int a = k / (NQ * ND), q = (k / ND) % NQ, j = k % ND;
      tab_s[k] = v;
      tabT_s[(q * ND + j) * 3 + a] = v;
    }
  double dot = 0.0;
  for (int step = 0;; ++step)
  {
    const int64_t b = xcd_stride_item(A.nblocks, step);
    if (b < 0)
      break;
    const int32_t* __restrict__ h = A.hdr + MF_HDR * b;
    const int dof_off = h[0], nloc = h[1], n_int = h[2], n_sh = h[3], part_off = h[4];
    // stage the block's u (and coordinates), clear its y
    for (int d = tid; d < nloc; d += T)
    {
      const int32_t g = (GU || DIAG || MF_DBG(4)) ? 0 : A.dof_ids[dof_off + d];
      const double uv = (GU || DIAG || MF_DBG(4)) ? 1.0 : A.u[g];
      if (ND == 4)
      {
        const double* __restrict__ q = A.xyz + 3ll * (dof_off + d);
        *reinterpret_cast<double2*>(rec + 2 * d) = make_double2(q[0], q[1]);
        *reinterpret_cast<double2*>(zu + 2 * d) = make_double2(q[2], uv);
      }
      else if (!GU)
        rec[d] = uv;
      ys[d] = 0.0;
    }
    for (int s = 0; s < A.nsb; ++s)
    {
      const int e = s * T + tid;
      uint32_t iw[NDW], rw[NRW];
#pragma unroll
      for (int w = 0; w < NDW; ++w)
        iw[w] = A.idxw[(b * NDW + w) * A.nc + e];
#pragma unroll
      for (int w = 0; w < NRW; ++w)
        rw[w] = A.rnkw[(b * NRW + w) * A.nc + e];
      double ye[ND];
      if (s == 0)
      {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads(); // the staged values are in LDS (the loads above are in flight meanwhile)
      }
      if constexpr (ND == 4)
      {
#pragma clang fp contract(fast)
        const int i0 = iw[0] & 0xffff, i1 = iw[0] >> 16, i2 = iw[1] & 0xffff, i3 = iw[1] >> 16;
        const double2 a0 = *reinterpret_cast<const double2*>(rec + 2 * i0), b0 = *reinterpret_cast<const double2*>(zu + 2 * i0);
        const double2 a1 = *reinterpret_cast<const double2*>(rec + 2 * i1), b1 = *reinterpret_cast<const double2*>(zu + 2 * i1);
        const double2 a2 = *reinterpret_cast<const double2*>(rec + 2 * i2), b2 = *reinterpret_cast<const double2*>(zu + 2 * i2);
        const double2 a3 = *reinterpret_cast<const double2*>(rec + 2 * i3), b3 = *reinterpret_cast<const double2*>(zu + 2 * i3);
        const double4 p0 = make_double4(a0.x, a0.y, b0.x, b0.y), p1 = make_double4(a1.x, a1.y, b1.x, b1.y);
        const double4 p2 = make_double4(a2.x, a2.y, b2.x, b2.y), p3 = make_double4(a3.x, a3.y, b3.x, b3.y);
        // J[a][al] = p_(al+1)[a] - p_0[a]; C = cofactors: K = J^-1 = C / det, grad phi_(al+1) = C[al][:] / det
        const double J00 = p1.x - p0.x, J01 = p2.x - p0.x, J02 = p3.x - p0.x;
        const double J10 = p1.y - p0.y, J11 = p2.y - p0.y, J12 = p3.y - p0.y;
        const double J20 = p1.z - p0.z, J21 = p2.z - p0.z, J22 = p3.z - p0.z;
        const double C00 = J11 * J22 - J12 * J21, C01 = J02 * J21 - J01 * J22, C02 = J01 * J12 - J02 * J11;
        const double C10 = J12 * J20 - J10 * J22, C11 = J00 * J22 - J02 * J20, C12 = J02 * J10 - J00 * J12;
        const double C20 = J10 * J21 - J11 * J20, C21 = J01 * J20 - J00 * J21, C22 = J00 * J11 - J01 * J10;
        const double det = J00 * C00 + J01 * C10 + J02 * C20;
        const double d1 = p1.w - p0.w, d2 = p2.w - p0.w, d3 = p3.w - p0.w;
        const double sc = 1.0 / (6.0 * fabs(det));
        if constexpr (DIAG)
        {
          const double s0 = C00 + C10 + C20, s1 = C01 + C11 + C21, s2 = C02 + C12 + C22;
          ye[0] = (s0 * s0 + s1 * s1 + s2 * s2) * sc;
          ye[1] = (C00 * C00 + C01 * C01 + C02 * C02) * sc;
          ye[2] = (C10 * C10 + C11 * C11 + C12 * C12) * sc;
          ye[3] = (C20 * C20 + C21 * C21 + C22 * C22) * sc;
        }
        else
        {
        const double t0 = (C00 * d1 + C10 * d2 + C20 * d3) * sc;
        const double t1 = (C01 * d1 + C11 * d2 + C21 * d3) * sc;
        const double t2 = (C02 * d1 + C12 * d2 + C22 * d3) * sc;
        ye[1] = C00 * t0 + C01 * t1 + C02 * t2;
        ye[2] = C10 * t0 + C11 * t1 + C12 * t2;
        ye[3] = C20 * t0 + C21 * t1 + C22 * t2;
        ye[0] = -(ye[1] + ye[2] + ye[3]);
        }
      }
      else
      {
        double G[6], ue[ND];
#pragma unroll
        for (int t = 0; t < 6; ++t)
          G[t] = A.geom[(b * 6 + t) * A.nc + e];
        if constexpr (DIAG)
          mf_element_diag_pk<ND>(tabT_s, G, ye);
        else if constexpr (GU)
        {
#pragma unroll
          for (int j = 0; j < ND; ++j)
            ue[j] = A.u[A.gid[(b * ND + j) * A.nc + e]];
        }
        else
        {
#pragma unroll
          for (int j = 0; j < ND; ++j)
            ue[j] = rec[(iw[j >> 1] >> (16 * (j & 1))) & 0xffff];
        }
        if constexpr (DIAG)
          ;
        else if (MF_DBG(1))
        {
#pragma unroll
          for (int j = 0; j < ND; ++j)
            ye[j] = ue[j] * G[j % 6];
        }
        else
          mf_element_pk<ND>(tab_s, tabT_s, ue, G, ye);
      }
      // rounds: the incidences of rank r of this step are added in round r (distinct addresses inside a round).  The
      // additions are LDS atomics WITHOUT return (ds_add_f64: nothing to wait for inside a round; a read-add-write chain
      // per incidence cost a round trip each: P3 3.3x, P1 2x the kernel); no two of a round meet on an address, so their
      // order is immaterial and the sum of a dof is formed in round order: reproducible.  The local dofs of a cell come
      // in classes of falling multiplicity (vertices, edges, faces): the late rounds look at the vertices only.
      const uint32_t* __restrict__ rm = reinterpret_cast<const uint32_t*>(A.rmax) + (b * A.nsb + s) * NRW;
      int Rc[3] = {0, 0, 0}; // rounds the classes [0, 4), [4, NE), [NE, ND) need
      constexpr int NE = ND == 20 ? 16 : ND;
#pragma unroll
      for (int w = 0; w < NRW; ++w)
      {
        const uint32_t m = __builtin_amdgcn_readfirstlane(rm[w]);
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (4 * w + k < ND)
          {
            const int c = 4 * w + k < 4 ? 0 : (4 * w + k < NE ? 1 : 2);
            Rc[c] = max(Rc[c], (int)((m >> (8 * k)) & 255u));
          }
      }
      const int R2 = Rc[2], R1 = max(R2, Rc[1]), R0 = MF_DBG(2) ? 1 : (MF_DBG(32) ? 0 : max(R1, Rc[0]));
      if (MF_DBG(32)) // timing probe: one unconditional addition per local dof, a barrier before each (wrong results)
      {
#pragma unroll
        for (int j = 0; j < ND; ++j)
        {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __syncthreads();
          const int i = (iw[j >> 1] >> (16 * (j & 1))) & 0xffff;
          __hip_atomic_fetch_add(&ys[i], ye[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      }
      for (int r = 0; r < R0; ++r)
      {
        // hipcc (ROCm 7.2) emitted this loop's s_barrier WITHOUT a wait for the LDS store of the round before (seen in
        // the ISA; one dof in 5 x 10^5 lost an addition every few launches): the wait is spelled out
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();
        const int jend = r < R2 ? ND : (r < R1 ? NE : 4); // uniform
#pragma unroll
        for (int j = 0; j < ND; ++j)
        {
          if (j < 4 || (j < NE ? jend > 4 : jend > NE))
            if ((int)((rw[j >> 2] >> (8 * (j & 3))) & 255u) == r)
            {
              const int i = (iw[j >> 1] >> (16 * (j & 1))) & 0xffff;
              __hip_atomic_fetch_add(&ys[i], ye[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    // dofs interior to the block: final values; shared ones: this block's partial sum
    for (int d = tid; d < (MF_DBG(8) ? 0 : n_int); d += T)
    {
      const int32_t g = A.dof_ids[dof_off + d];
      // bc->set(y.array(), std::nullopt, 0.0), src/cgpoisson_problem.cpp:207; the diagonal: 1.0 there
      const double v = A.dof_flag[dof_off + d] ? (DIAG ? 1.0 : 0.0) : ys[d];
      A.y[g] = v;
      if constexpr (!DIAG)
        dot += v * (ND == 4 ? zu[2 * d + 1] : (GU ? A.u[g] : rec[d]));
    }
    for (int d = tid; d < n_sh; d += T)
      A.ypart[A.pslot[part_off + d]] = ys[n_int + d];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
  }
  if (A.partials)
  {
    const double t = block_reduce_sum(dot, red);
    if (tid == 0)
      A.partials[blockIdx.x] = t;
  }
}

// the dofs shared between blocks: their partial sums (side by side, ascending block) added in that order
__global__ __launch_bounds__(256) void k_mf_finish(const int32_t* __restrict__ sh_dof, const int32_t* __restrict__ sh_off,
                                                   const uint8_t* __restrict__ sh_flag, int64_t nshared,
                                                   const double* __restrict__ ypart, const double* __restrict__ u,
                                                   double* __restrict__ y, double* __restrict__ partials,
                                                   const int* __restrict__ stop, double fixed_value)
{
  if (stop && *stop)
    return;
  __shared__ double red[4];
  double dot = 0.0;
  for (int64_t k = blockIdx.x * 256ll + threadIdx.x; k < nshared; k += gridDim.x * 256ll)
  {
    const int32_t g = sh_dof[k];
    const int p0 = sh_off[k], p1 = sh_off[k + 1];
    const double ug = u ? u[g] : 0.0;
    const bool fixed = sh_flag[k] != 0;
    double s = ypart[p0];
    for (int p = p0 + 1; p < p1; ++p)
      s += ypart[p];
    if (fixed)
      s = fixed_value; // bc->set(y.array(), std::nullopt, 0.0), src/cgpoisson_problem.cpp:207 (the diagonal: 1.0)
    y[g] = s;
    dot += s * ug;
  }
  if (partials)
  {
    const double t = block_reduce_sum(dot, red);
    if (threadIdx.x == 0)
      partials[blockIdx.x] = t;
  }
}

struct SegOffset
{
  unsigned seg;
  __host__ __device__ unsigned operator()(unsigned k) const { return k * seg; }
};

int grid_for(int64_t n)
{
  int64_t g = (n + 255) / 256;
  return (int)std::min<int64_t>(std::max<int64_t>(g, 1), 8192);
}

template <typename K, typename V>
int sort_pairs(zzz_ctx* ctx, DevBuf<K>& kin, DevBuf<K>& kout, DevBuf<V>& vin, DevBuf<V>& vout, size_t n, unsigned end_bit)
{
  size_t tb = 0;
  ZZZ_HIP(ctx, rocprim::radix_sort_pairs(nullptr, tb, kin.p, kout.p, vin.p, vout.p, n, 0u, end_bit, ctx->stream));
  DevBuf<unsigned char> tmp;
  ZZZ_HIP(ctx, tmp.alloc(tb));
  ZZZ_HIP(ctx, rocprim::radix_sort_pairs(tmp.p, tb, kin.p, kout.p, vin.p, vout.p, n, 0u, end_bit, ctx->stream));
  ZZZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return ZZZ_OK;
}

int scan_excl(zzz_ctx* ctx, const int32_t* in, int32_t* out, size_t n)
{
  size_t tb = 0;
  ZZZ_HIP(ctx, rocprim::exclusive_scan(nullptr, tb, in, out, 0, n, rocprim::plus<int32_t>(), ctx->stream));
  DevBuf<unsigned char> tmp;
  ZZZ_HIP(ctx, tmp.alloc(tb));
  ZZZ_HIP(ctx, rocprim::exclusive_scan(tmp.p, tb, in, out, 0, n, rocprim::plus<int32_t>(), ctx->stream));
  ZZZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return ZZZ_OK;
}

unsigned bits_for(uint64_t v)
{
  unsigned b = 1;
  while (b < 64 && (v >> b))
    ++b;
  return b;
}

int lds_bytes(int nd, int nloc_cap) { return (nd == 4 ? 40 : (nd == 20 ? 8 : 16)) * nloc_cap; }

// one attempt with blocks of nc cells; *retry: some block touches more dofs than LDS holds
int plan_attempt(zzz_ctx* ctx, int nc, int T, int nloc_limit, bool* retry)
{
  MfPlan& M = ctx->mf;
  hipStream_t s = ctx->stream;
  const int nd = ctx->nd;
  const int64_t ncells = ctx->ncells;
  const int nsb = nc / T;
  const int64_t nb = (ncells + nc - 1) / nc, total = nb * nc;
  *retry = false;
  if (total * nd > (int64_t)INT32_MAX - 1024 || nb >= (1ll << 27))
    return fail(ctx, ZZZ_ERR_LIMIT, "matrix-free plan: %lld incidences exceed int32: use more parts", (long long)(total * nd));
  M.nd = nd;
  M.nc = nc;
  M.threads = T;
  M.nsb = nsb;
  M.ndw = nd / 2;
  M.nrw = (nd + 3) / 4;
  M.nblocks = nb;

  // 1. cells in the Morton order of their centroids, dealt into execution order
  DevBuf<int32_t> sorted_cells;
  {
    DevBuf<double> bb;
    const int g = 256;
    ZZZ_HIP(ctx, bb.alloc(6 * g));
    hipLaunchKernelGGL(k_mf_bbox, dim3(g), dim3(256), 0, s, ctx->x.p, ctx->nverts, bb.p);
    std::vector<double> hb(6 * g);
    ZZZ_HIP(ctx, hipMemcpyAsync(hb.data(), bb.p, hb.size() * sizeof(double), hipMemcpyDeviceToHost, s));
    ZZZ_HIP(ctx, hipStreamSynchronize(s));
    Box B;
    for (int a = 0; a < 3; ++a)
    {
      double lo = 1e300, hi = -1e300;
      for (int i = 0; i < g; ++i)
      {
        lo = std::min(lo, hb[6 * i + a]);
        hi = std::max(hi, hb[6 * i + 3 + a]);
      }
      B.lo[a] = lo;
      B.scale[a] = hi > lo ? 1024.0 / (hi - lo) : 0.0;
    }
    // one resolution for the three axes (the longest extent gets the 1024 bins): Morton cells are cubes in space
    const double sc = std::min({B.scale[0] > 0 ? B.scale[0] : 1e300, B.scale[1] > 0 ? B.scale[1] : 1e300,
                                B.scale[2] > 0 ? B.scale[2] : 1e300});
    for (int a = 0; a < 3; ++a)
      B.scale[a] = sc < 1e300 ? sc : 0.0;
    DevBuf<uint32_t> k0, k1;
    DevBuf<int32_t> v0;
    ZZZ_HIP(ctx, k0.alloc((size_t)ncells));
    ZZZ_HIP(ctx, k1.alloc((size_t)ncells));
    ZZZ_HIP(ctx, v0.alloc((size_t)ncells));
    ZZZ_HIP(ctx, sorted_cells.alloc((size_t)ncells));
    hipLaunchKernelGGL(k_mf_cell_keys, dim3(grid_for(ncells)), dim3(256), 0, s, ctx->x.p, ctx->cell_verts.p, ncells, B, k0.p, v0.p);
    if (int rc = sort_pairs(ctx, k0, k1, v0, sorted_cells, (size_t)ncells, 30))
      return rc;
  }
  ZZZ_HIP(ctx, M.mf_cell.alloc((size_t)total));
  {
    // hash table of the block's dofs: twice the dofs a block may touch, as far as LDS goes (a table that fills up only
    // costs the assignment some of its quality: dofs without a slot are not priced)
    const int cw = nsb <= 4 ? 1 : (nsb <= 8 ? 2 : 4); // words of byte counters per slot
    const int per_slot = 4 + 4 * cw, fixed = 16 * 4 + 16;
    int H = 64;
    while (H < 2 * std::min<int64_t>(nloc_limit, (int64_t)nc * nd) && 2 * H * per_slot + fixed <= 144 * 1024)
      H <<= 1;
    const int lds = H * per_slot + fixed;
    if (lds > 48 * 1024)
      ZZZ_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_mf_assign), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const int per_cu = std::max(1, std::min(16, (160 * 1024) / (lds + 256)));
    int W = 64;
#ifdef ZZZ_EXPERIMENTS
    if (const char* e = getenv("ZZZ_MF_ASSIGN_W"))
      W = std::max(1, std::min(64, atoi(e)));
#endif
    hipLaunchKernelGGL(k_mf_assign, dim3((unsigned)std::min<int64_t>(nb, 256ll * per_cu)), dim3(64), lds, s, sorted_cells.p,
                       ctx->cell_dofs.p, nd, ncells, nc, T, nsb, cw, H, W, nb, M.mf_cell.p);
    ZZZ_HIP(ctx, hipGetLastError());
  }
  sorted_cells.release();

  // 2. the incidences of every block sorted by dof (one segment of nc * nd pairs per block: a block-level sort each, the
  // data moves once -- the global sort of (block, dof) keys this replaces took six passes over 64-bit keys); unique keys
  const int64_t np = total * nd;
  const int seg = nc * nd;
  DevBuf<uint32_t> key, val;
  {
    DevBuf<uint32_t> key0, val0;
    ZZZ_HIP(ctx, key0.alloc((size_t)np));
    ZZZ_HIP(ctx, val0.alloc((size_t)np));
    ZZZ_HIP(ctx, key.alloc((size_t)np));
    ZZZ_HIP(ctx, val.alloc((size_t)np));
    hipLaunchKernelGGL(k_mf_pairs, dim3(grid_for(np)), dim3(256), 0, s, M.mf_cell.p, ctx->cell_dofs.p, nd, nc, total, key0.p, val0.p);
    auto begin = rocprim::make_transform_iterator(rocprim::counting_iterator<unsigned>(0), SegOffset{(unsigned)seg});
    auto end = rocprim::make_transform_iterator(rocprim::counting_iterator<unsigned>(1), SegOffset{(unsigned)seg});
    // (one bit above the dofs': the key of a missing cell stays behind every dof)
    const unsigned end_bit = std::min(32u, bits_for((uint64_t)(ctx->n_owned + ctx->n_ghost)) + 1u);
    size_t tb = 0;
    ZZZ_HIP(ctx, rocprim::segmented_radix_sort_pairs(nullptr, tb, key0.p, key.p, val0.p, val.p, (unsigned)np, (unsigned)nb, begin, end,
                                                     0u, end_bit, s));
    DevBuf<unsigned char> tmp;
    ZZZ_HIP(ctx, tmp.alloc(tb));
    ZZZ_HIP(ctx, rocprim::segmented_radix_sort_pairs(tmp.p, tb, key0.p, key.p, val0.p, val.p, (unsigned)np, (unsigned)nb, begin, end,
                                                     0u, end_bit, s));
    ZZZ_HIP(ctx, hipStreamSynchronize(s));
  }
  DevBuf<int32_t> flag, uidx;
  ZZZ_HIP(ctx, flag.alloc((size_t)np + 1));
  ZZZ_HIP(ctx, uidx.alloc((size_t)np + 1));
  hipLaunchKernelGGL(k_mf_heads, dim3(grid_for(np)), dim3(256), 0, s, key.p, np, seg, flag.p);
  ZZZ_HIP(ctx, hipMemsetAsync(flag.p + np, 0, sizeof(int32_t), s));
  if (int rc = scan_excl(ctx, flag.p, uidx.p, (size_t)np + 1))
    return rc;
  int32_t nu32 = 0;
  ZZZ_HIP(ctx, hipMemcpy(&nu32, uidx.p + np, sizeof(int32_t), hipMemcpyDeviceToHost));
  const int64_t nu = nu32;
  const int64_t nloc_all = ctx->n_owned + ctx->n_ghost;
  DevBuf<uint64_t> ukey;
  DevBuf<int32_t> run_start, nblk_of;
  ZZZ_HIP(ctx, ukey.alloc((size_t)nu));
  ZZZ_HIP(ctx, run_start.alloc((size_t)nu + 1));
  ZZZ_HIP(ctx, nblk_of.alloc((size_t)nloc_all));
  ZZZ_HIP(ctx, hipMemsetAsync(nblk_of.p, 0, (size_t)nloc_all * sizeof(int32_t), s));
  hipLaunchKernelGGL(k_mf_uniques, dim3(grid_for(np)), dim3(256), 0, s, key.p, flag.p, uidx.p, np, seg, ukey.p, run_start.p, nblk_of.p);

  // 3. classes, block-local order, headers
  DevBuf<uint64_t> key2, key2s;
  DevBuf<int32_t> val2, val2s, cnt3, nloc_b, nsh_b, dof_off, part_off, nlmax;
  ZZZ_HIP(ctx, key2.alloc((size_t)nu));
  ZZZ_HIP(ctx, key2s.alloc((size_t)nu));
  ZZZ_HIP(ctx, val2.alloc((size_t)nu));
  ZZZ_HIP(ctx, val2s.alloc((size_t)nu));
  ZZZ_HIP(ctx, cnt3.alloc((size_t)nb * 3));
  ZZZ_HIP(ctx, nloc_b.alloc((size_t)nb + 1));
  ZZZ_HIP(ctx, nsh_b.alloc((size_t)nb + 1));
  ZZZ_HIP(ctx, dof_off.alloc((size_t)nb + 1));
  ZZZ_HIP(ctx, part_off.alloc((size_t)nb + 1));
  ZZZ_HIP(ctx, nlmax.alloc(2));
  ZZZ_HIP(ctx, hipMemsetAsync(cnt3.p, 0, (size_t)nb * 3 * sizeof(int32_t), s));
  ZZZ_HIP(ctx, hipMemsetAsync(nlmax.p, 0, 2 * sizeof(int32_t), s));
  hipLaunchKernelGGL(k_mf_classify, dim3(grid_for(nu)), dim3(256), 0, s, ukey.p, nu, nblk_of.p, ctx->n_owned, key2.p, val2.p, cnt3.p);
  hipLaunchKernelGGL(k_mf_block_sizes, dim3(grid_for(nb + 1)), dim3(256), 0, s, cnt3.p, nb, nloc_b.p, nsh_b.p);
  if (int rc = scan_excl(ctx, nloc_b.p, dof_off.p, (size_t)nb + 1))
    return rc;
  if (int rc = scan_excl(ctx, nsh_b.p, part_off.p, (size_t)nb + 1))
    return rc;
  ZZZ_HIP(ctx, M.hdr.alloc((size_t)nb * MF_HDR));
  hipLaunchKernelGGL(k_mf_headers, dim3(grid_for(nb)), dim3(256), 0, s, cnt3.p, dof_off.p, part_off.p, nb, M.hdr.p, nlmax.p);
  int32_t h_nlmax = 0, h_nslots = 0;
  ZZZ_HIP(ctx, hipMemcpyAsync(&h_nlmax, nlmax.p, sizeof(int32_t), hipMemcpyDeviceToHost, s));
  ZZZ_HIP(ctx, hipMemcpyAsync(&h_nslots, part_off.p + nb, sizeof(int32_t), hipMemcpyDeviceToHost, s));
  ZZZ_HIP(ctx, hipStreamSynchronize(s));
  if (h_nlmax > nloc_limit)
  {
    *retry = true;
    return ZZZ_OK;
  }
  M.nloc_max = h_nlmax;
  M.nslots = h_nslots;
  if (int rc = sort_pairs(ctx, key2, key2s, val2, val2s, (size_t)nu, 34 + bits_for((uint64_t)nb)))
    return rc;
  DevBuf<int32_t> loc_of, sh_val, sh_val_s;
  DevBuf<uint32_t> sh_key, sh_key_s;
  ZZZ_HIP(ctx, M.dof_ids.alloc((size_t)nu));
  ZZZ_HIP(ctx, M.dof_flag.alloc((size_t)nu));
  if (nd == 4)
    ZZZ_HIP(ctx, M.xyz.alloc((size_t)nu * 3));
  ZZZ_HIP(ctx, loc_of.alloc((size_t)nu));
  ZZZ_HIP(ctx, sh_key.alloc((size_t)nu));
  ZZZ_HIP(ctx, sh_key_s.alloc((size_t)nu));
  ZZZ_HIP(ctx, sh_val.alloc((size_t)nu));
  ZZZ_HIP(ctx, sh_val_s.alloc((size_t)nu));
  hipLaunchKernelGGL(k_mf_lists, dim3(grid_for(nu)), dim3(256), 0, s, key2s.p, val2s.p, nu, M.hdr.p, ctx->bc.p,
                     nd == 4 ? ctx->xq : (const double*)nullptr, M.dof_ids.p, M.dof_flag.p, nd == 4 ? M.xyz.p : (double*)nullptr,
                     loc_of.p, sh_key.p, sh_val.p);

  // 4. per cell: local indices, ranks; rounds per (block, step, local dof)
  ZZZ_HIP(ctx, M.idxw.alloc((size_t)(nb * M.ndw * nc)));
  ZZZ_HIP(ctx, M.rnkw.alloc((size_t)(nb * M.nrw * nc)));
  ZZZ_HIP(ctx, M.rmax.alloc((size_t)(nb * nsb * M.nrw * 4)));
  {
    const int lds = (M.ndw + M.nrw) * nc * 4;
    if (lds > 48 * 1024)
      ZZZ_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_mf_cells), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL(k_mf_cells, dim3((unsigned)std::min<int64_t>(nb, 256ll * std::max(1, std::min(2, (160 * 1024) / (lds + 256))))), dim3(1024), lds,
                       s, key.p, val.p, uidx.p, flag.p, nb, seg, run_start.p, loc_of.p, nd, nc, T, M.ndw, M.nrw, M.idxw.p, M.rnkw.p,
                       nlmax.p + 1);
  }
  hipLaunchKernelGGL(k_mf_rmax, dim3((unsigned)std::min<int64_t>((nb * nsb * M.nrw + 3) / 4, 8192)), dim3(256), 0, s, M.rnkw.p, nb, nc,
                     T, nsb, M.nrw, M.rmax.p);
  int32_t h_err = 0;
  ZZZ_HIP(ctx, hipMemcpyAsync(&h_err, nlmax.p + 1, sizeof(int32_t), hipMemcpyDeviceToHost, s));
  ZZZ_HIP(ctx, hipStreamSynchronize(s));
#ifdef ZZZ_EXPERIMENTS
  if (getenv("ZZZ_MF_PLAN_STATS"))
  {
    // quality of the step assignment: rounds per (block, step) = the most cells of the step at one dof
    std::vector<uint8_t> hr((size_t)(nb * nsb * M.nrw * 4));
    ZZZ_HIP(ctx, hipMemcpy(hr.data(), M.rmax.p, hr.size(), hipMemcpyDeviceToHost));
    double sum = 0;
    int worst = 0;
    for (int64_t bs_ = 0; bs_ < nb * nsb; ++bs_)
    {
      int m = 0;
      for (int j = 0; j < nd; ++j)
        m = std::max(m, (int)hr[(size_t)bs_ * M.nrw * 4 + j]);
      sum += m;
      worst = std::max(worst, m);
    }
    fprintf(stderr, "[zzz] matrix-free plan: %lld blocks x %d steps, rounds per step: mean %.3f, worst %d\n", (long long)nb, nsb,
            sum / (double)(nb * nsb), worst);
  }
#endif
  if (h_err)
    return fail(ctx, ZZZ_ERR_LIMIT, "matrix-free plan: a dof meets more than 254 cells of one step");

  // 5. the shared dofs and their slots
  if (int rc = sort_pairs(ctx, sh_key, sh_key_s, sh_val, sh_val_s, (size_t)nu, 32))
    return rc;
  M.nshared = 0;
  if (M.nslots > 0)
  {
    DevBuf<int32_t> f2, u2;
    ZZZ_HIP(ctx, f2.alloc((size_t)M.nslots + 1));
    ZZZ_HIP(ctx, u2.alloc((size_t)M.nslots + 1));
    hipLaunchKernelGGL(k_mf_sh_heads, dim3(grid_for(M.nslots)), dim3(256), 0, s, sh_key_s.p, M.nslots, f2.p);
    ZZZ_HIP(ctx, hipMemsetAsync(f2.p + M.nslots, 0, sizeof(int32_t), s));
    if (int rc = scan_excl(ctx, f2.p, u2.p, (size_t)M.nslots + 1))
      return rc;
    int32_t ns = 0;
    ZZZ_HIP(ctx, hipMemcpy(&ns, u2.p + M.nslots, sizeof(int32_t), hipMemcpyDeviceToHost));
    M.nshared = ns;
    ZZZ_HIP(ctx, M.sh_dof.alloc((size_t)ns));
    ZZZ_HIP(ctx, M.sh_off.alloc((size_t)ns + 1));
    ZZZ_HIP(ctx, M.sh_slot.alloc((size_t)M.nslots));
    ZZZ_HIP(ctx, M.sh_flag.alloc((size_t)ns));
    hipLaunchKernelGGL(k_mf_sh_fill, dim3(grid_for(M.nslots + 1)), dim3(256), 0, s, sh_key_s.p, sh_val_s.p, f2.p, u2.p, M.nslots,
                       ctx->bc.p, M.sh_dof.p, M.sh_off.p, M.sh_flag.p, M.sh_slot.p, (int64_t)ns);
    ZZZ_HIP(ctx, hipStreamSynchronize(s));
  }
  ZZZ_HIP(ctx, M.ypart.alloc((size_t)std::max<int64_t>(M.nslots, 1)));

  // 6. P2/P3: geometry factors per cell, the factorised reference tables
  if (nd != 4)
  {
    const double* src = nd == 10 ? ZZZ_DTAB_P2 : ZZZ_DTAB_P3;
    const size_t nt = nd == 10 ? 120 : 600;
    ZZZ_HIP(ctx, M.dtab.alloc(nt));
    ZZZ_HIP(ctx, hipMemcpyAsync(M.dtab.p, src, nt * sizeof(double), hipMemcpyHostToDevice, s));
    if (nd == 20)
    {
      ZZZ_HIP(ctx, M.gid.alloc((size_t)(total * nd)));
      hipLaunchKernelGGL(k_mf_gid, dim3(grid_for(total * nd)), dim3(256), 0, s, M.mf_cell.p, ctx->cell_dofs.p, nd, nc, total, M.gid.p);
    }
    ZZZ_HIP(ctx, M.geom.alloc((size_t)(nb * 6 * nc)));
    hipLaunchKernelGGL(k_mf_geom, dim3(grid_for(total)), dim3(256), 0, s, ctx->x.p, ctx->cell_verts.p, M.mf_cell.p, nc, total, M.geom.p);
  }
  ZZZ_HIP(ctx, hipGetLastError());
  ZZZ_HIP(ctx, hipStreamSynchronize(s));
  // what one action addresses: the per-cell streams, the block lists (ids, flags, coordinates), u gathered and y
  // written per list entry, the partial sums out and back, the shared dofs' finish
  M.bytes_per_action = total * (4ll * M.ndw + 4ll * M.nrw + (nd == 4 ? 0 : 48) + (nd == 20 ? 4 * nd + 8 * nd : 0))
                       + nu * (nd == 20 ? 0 : 4 + 1 + 8 + (nd == 4 ? 24 : 0)) + (nd == 20 ? (nu - M.nslots) * (4 + 1 + 8) : 0)
                       + (nu - M.nslots) * 8 + M.nslots * (8 + 8 + 4) + M.nshared * (4 + 4 + 1 + 8 + 8) + nb * (MF_HDR * 4 + nsb * M.nrw * 4);
  return ZZZ_OK;
}
} // namespace

int mf_plan_build(zzz_ctx* ctx)
{
  MfPlan& M = ctx->mf;
  M.valid = M.failed = false;
  if (ctx->bs != 1)
    return fail(ctx, ZZZ_ERR_ARG, "the matrix-free operator exists for the Poisson form M only (src/Poisson.py:33)");
  if (ctx->order == 0 || ctx->ncells == 0)
    return fail(ctx, ZZZ_ERR_ARG, "matrix-free operator before zzz_dofmap_upload");
  if (int rc = ensure_p1_coords(ctx))
    return rc;
  const int nd = ctx->nd;
  // defaults (measured): workgroup size and cells per block; ZZZ_MF_T / ZZZ_MF_NC override them
  // (tools/mf_sweep.sh, MI355X: P1 10 M dofs 0.70 ms at 2048 x 256 against 0.84 at 4096 x 512 and 1.30 at 8192 x 512 --
  // small blocks share more dofs but keep five workgroups on a CU; P3 6.2 M dofs 0.30 ms at 768 x 256, 0.32 at 512, 0.38 at 1024)
  int T = 256;
  int nc = nd == 4 ? 2048 : (nd == 10 ? 1024 : 768);
  if (const char* e = getenv("ZZZ_MF_T"))
  {
    const int v = atoi(e);
    if (v == 128 || v == 256 || v == 512 || v == 1024)
      T = v;
  }
  if (const char* e = getenv("ZZZ_MF_NC"))
  {
    const int v = atoi(e);
    if (v >= T && v <= 65536 && v % T == 0)
      nc = v;
  }
  nc = std::max(nc, T);
  // (the plan's own kernels put a block's index and rank words together in LDS: 4 (nd / 2 + (nd + 3) / 4) bytes per cell)
  while (nc / 2 >= T && (nc / 2) % T == 0 && (nd / 2 + (nd + 3) / 4) * 4 * nc > 150 * 1024)
    nc /= 2;
  // LDS budget per workgroup: 64 KiB unless ZZZ_MF_LDS_KB says otherwise (160 KiB per CU)
  int lds_kb = nd == 4 ? 40 : 64;
#ifdef ZZZ_EXPERIMENTS
  if (const char* e = getenv("ZZZ_MF_LDS_KB")) // measurement knob, tools build only
    lds_kb = std::min(160, std::max(8, atoi(e)));
#endif
  const int nloc_limit = std::min(65535, lds_kb * 1024 / (nd == 4 ? 40 : (nd == 20 ? 8 : 16)));
  for (;;)
  {
    bool retry = false;
    if (int rc = plan_attempt(ctx, nc, T, nloc_limit, &retry))
      return rc;
    if (!retry)
      break;
    if (nc / 2 < T || (nc / 2) % T)
      return fail(ctx, ZZZ_ERR_LIMIT, "matrix-free plan: a block of %d cells touches more dofs than LDS holds", nc);
    nc /= 2;
  }
  M.valid = true;
  return ZZZ_OK;
}

template <int ND, int T, bool DIAG>
static int mf_launch(zzz_ctx* ctx, const MfArgs& A, int grid, int lds)
{
  static int attr_lds[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; // per device: the limit this instantiation was given
  if (lds > 48 * 1024 && lds > attr_lds[ctx->device & 15])
  {
    ZZZ_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_mf_action<ND, T, DIAG>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    attr_lds[ctx->device & 15] = lds;
  }
  hipLaunchKernelGGL((k_mf_action<ND, T, DIAG>), dim3(grid), dim3(T), lds, ctx->stream, A);
  return ZZZ_OK;
}

static int mf_run(zzz_ctx* ctx, bool diag, const double* u, double* y, double* partials, int* npartials);

int mf_action(zzz_ctx* ctx, const double* u, double* y, double* partials, int* npartials)
{
  return mf_run(ctx, false, u, y, partials, npartials);
}

// y = diag(A) (1.0 on constrained rows), summed in the order of the action
int mf_diagonal(zzz_ctx* ctx, double* y) { return mf_run(ctx, true, nullptr, y, nullptr, nullptr); }

static int mf_run(zzz_ctx* ctx, bool diag, const double* u, double* y, double* partials, int* npartials)
{
  MfPlan& M = ctx->mf;
  if (!M.valid)
    return fail(ctx, ZZZ_ERR_ARG, "matrix-free plan missing");
  const int lds = lds_bytes(M.nd, M.nloc_max);
  // persistent workgroups: as many as fit a CU by LDS and threads, XCD-aware walk over the blocks
  int per_cu = std::min((160 * 1024) / std::max(lds + 512, 1), 2048 / M.threads);
  per_cu = std::max(1, std::min(per_cu, 8));
  int grid = (int)std::min<int64_t>(256ll * per_cu, (M.nblocks + 7) / 8 * 8);
  grid = std::max(8, grid / 8 * 8);
  const int gf = (int)std::min<int64_t>(std::max<int64_t>((M.nshared + 255) / 256, 1), 2048);
  MfArgs A;
  A.hdr = M.hdr.p;
  A.dof_ids = M.dof_ids.p;
  A.dof_flag = M.dof_flag.p;
  A.xyz = M.xyz.p;
  A.idxw = M.idxw.p;
  A.rnkw = M.rnkw.p;
  A.rmax = M.rmax.p;
  A.geom = M.geom.p;
  A.dtab = M.dtab.p;
  A.ypart = M.ypart.p;
  A.pslot = M.sh_slot.p;
  A.gid = M.gid.p;
  A.u = u;
  A.y = y;
  A.partials = partials;
  A.stop = partials ? reinterpret_cast<const int*>(ctx->state.p) : nullptr;
  A.nblocks = M.nblocks;
  A.nc = M.nc;
  A.nsb = M.nsb;
  A.nloc_cap = M.nloc_max;
  A.dbg = 0;
#ifdef ZZZ_EXPERIMENTS
  if (const char* e = getenv("ZZZ_MF_DEBUG"))
    A.dbg = atoi(e);
#endif
  int rc = ZZZ_OK;
#define ZZZ_MF_T(ND_, DG_)                                                                                              \
  (M.threads == 128 ? mf_launch<ND_, 128, DG_>(ctx, A, grid, lds)                                                         \
                    : M.threads == 256 ? mf_launch<ND_, 256, DG_>(ctx, A, grid, lds)                                      \
                                       : M.threads == 512 ? mf_launch<ND_, 512, DG_>(ctx, A, grid, lds)                   \
                                                          : mf_launch<ND_, 1024, DG_>(ctx, A, grid, lds))
  if (M.nd == 4)
    rc = diag ? ZZZ_MF_T(4, true) : ZZZ_MF_T(4, false);
  else if (M.nd == 10)
    rc = diag ? ZZZ_MF_T(10, true) : ZZZ_MF_T(10, false);
  else
    rc = diag ? ZZZ_MF_T(20, true) : ZZZ_MF_T(20, false);
#undef ZZZ_MF_T
  if (rc)
    return rc;
  if (M.nshared > 0 && !(A.dbg & 16))
    hipLaunchKernelGGL(k_mf_finish, dim3(gf), dim3(256), 0, ctx->stream, M.sh_dof.p, M.sh_off.p, M.sh_flag.p, M.nshared, M.ypart.p, u,
                       y, partials ? partials + grid : nullptr, A.stop, diag ? 1.0 : 0.0);
  if (npartials)
    *npartials = grid + (M.nshared > 0 ? gf : 0);
  ZZZ_HIP(ctx, hipGetLastError());
  return ZZZ_OK;
}
ZZZ_PRELOAD_TU(matfree)
} // namespace zzz

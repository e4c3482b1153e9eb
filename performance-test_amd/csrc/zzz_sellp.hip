// The CG operator stream: a sliced-ELL copy of the assembled matrix, built for how gfx950 reads memory.
//
// Replaces PETSc MatMult inside KSPSolve (src/poisson_problem.cpp:177) and the `action` of linalg::cg
// (src/cg.h:62) for matrices whose rows have similar lengths; the CSR tile kernel (zzz_spmv.hip) stays
// the operator for the others.  The CSR arrays remain the matrix of record (zzz_csr_download, Jacobi,
// parity); after every assembly (MatAssemblyEnd) the values are re-packed into this stream:
//
//   * rows in slices of 64 (one wavefront, one lane per row); optionally the rows of a window of
//     SIGMA rows are ordered by length first (SELL-C-sigma), so that rows of very different lengths
//     (P2/P3 vertex / edge / face dofs) do not pad each other;
//   * entries whose assembled value is exactly zero are left out.  On the Kuhn mesh more than half of
//     the P1 Laplacian's pattern is exact zeros (the face- and body-diagonal couplings)
//     which PETSc stores and multiplies; 0 * x adds nothing to a row sum, so y keeps its bits as long
//     as x is finite (PETSc's MAT_IGNORE_ZERO_ENTRIES has the same effect on MatMult);
//   * a slice is a sequence of CHUNKS of 8 entries per row.  A chunk is 4 KiB of values laid out
//     [4][64 lanes][2] (four 16-B loads per lane, each one dense 1-KiB wave read), 1 KiB of 16-bit column
//     codes [64 lanes][8] (ONE 16-B load per lane) and 8 slot bases (scalar loads): the column of
//     (lane, slot e) is base[e] + code.  Lanes are consecutive rows, so the e-th entries of a chunk are
//     (nearly) consecutive columns: the codes are small and the x gather of one wave instruction is
//     (nearly) one dense read.  A chunk whose slot range exceeds 16 bits keeps int32 columns (flag in
//     the sign bit of base[0]) -- the scheme never fails, it only stops paying;
//   * no LDS, no barrier; each row is summed in ascending column order like the scalar CPU loop
//     (mul and add rounded separately), so y is bit-identical to the CSR product.
//
// Padding entries carry the value +0.0 and a valid column.
#include <climits>
#include <cstring>
#include <cstdlib>
#include <vector>

#include "zzz_sellp.h"
#include "zzz_cg_device.h"

#include <rocprim/rocprim.hpp>

namespace zzz
{
// ---- build ------------------------------------------------------------------------------------------
// entries of each row that the stream keeps
// The same counts with dense loads: one wavefront sweeps the CSR range of its 64 rows 64 entries at a time; every lane
// (= row) counts the non-zero entries of the group that fall into its own row from the group's ballot.  k_sp_count below
// has one lane walk one row 8 B at a time: for the long rows of P3 that moved 31 GB through L2 for 2.4 GB of values
// (4.4 ms at 6.2 M dofs).
__global__ __launch_bounds__(256) void k_sp_count_sweep(const rp_t* __restrict__ rowptr, const double* __restrict__ vals,
                                                        int nrows, int64_t nslices, int32_t* __restrict__ rownnz)
{
  const int lane = threadIdx.x & 63;
  for (int64_t s = blockIdx.x * 4 + (threadIdx.x >> 6); s < nslices; s += (int64_t)gridDim.x * 4)
  {
    const int64_t r = s * 64 + lane;
    const int64_t a = r < nrows ? rowptr[r] : 0, b = r < nrows ? rowptr[r + 1] : 0;
    const int64_t S = rowptr[s * 64], E = rowptr[min(s * 64 + 64, (int64_t)nrows)];
    int n = 0;
    for (int64_t g = S; g < E; g += 64)
    {
      const int64_t k = g + lane;
      const unsigned long long m = __ballot(k < E && vals[k] != 0.0);
      // my row's part of [g, g + 64)
      const int lo = (int)min(max(a - g, (int64_t)0), (int64_t)64), hi = (int)min(max(b - g, (int64_t)0), (int64_t)64);
      if (hi > lo)
      {
        const unsigned long long below_hi = hi == 64 ? ~0ull : (1ull << hi) - 1ull;
        n += __popcll(m & below_hi & ~((1ull << lo) - 1ull)); // lo < 64 here
      }
    }
    if (r < nrows)
      rownnz[r] = n;
  }
}

__global__ __launch_bounds__(256) void k_sp_count(const rp_t* __restrict__ rowptr, const double* __restrict__ vals,
                                                  int nrows, int drop, int32_t* __restrict__ rownnz)
{
  for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < nrows; r += (int64_t)gridDim.x * blockDim.x)
  {
    const int64_t a = rowptr[r], b = rowptr[r + 1];
    int n = (int)(b - a);
    if (drop)
    {
      n = 0;
      for (int64_t k = a; k < b; ++k)
        n += vals[k] != 0.0 ? 1 : 0;
    }
    rownnz[r] = n;
  }
}

// natural row order: chunks of slice s = ceil(longest of its 64 rows / 8); entry nslices = 0 (scan sentinel)
__global__ __launch_bounds__(256) void k_sp_slice_len(const int32_t* __restrict__ rownnz, int nrows, int64_t nslices,
                                                      int32_t* __restrict__ nch, uint8_t* __restrict__ wlast)
{
  const int lane = threadIdx.x & 63;
  for (int64_t s = blockIdx.x * 4 + (threadIdx.x >> 6); s <= nslices; s += (int64_t)gridDim.x * 4)
  {
    const int64_t r = s * 64 + lane;
    const int m = wave_max_i((s < nslices && r < nrows) ? rownnz[r] : 0);
    if (lane == 0)
    {
      nch[s] = (m + 7) >> 3;
      if (s < nslices)
        wlast[s] = (uint8_t)(m ? m - 8 * ((m - 1) >> 3) : 8); // entries of the longest row in the last chunk: 1..8
    }
  }
}

// sorted form: one workgroup orders the SP_SIGMA rows of its window by length (descending, ties by row:
// a stable counting rank), writes the row of every (slice, lane) and the slice lengths
__global__ __launch_bounds__(SP_SIGMA) void k_sp_sort(const int32_t* __restrict__ rownnz, int nrows, int64_t nslices,
                                                      int32_t* __restrict__ perm, int32_t* __restrict__ nch,
                                                      uint8_t* __restrict__ wlast)
{
  __shared__ int len[SP_SIGMA];
  __shared__ int srt[SP_SIGMA];
  const int64_t w = blockIdx.x;
  const int t = threadIdx.x;
  const int64_t r = w * SP_SIGMA + t;
  const int mine = r < nrows ? rownnz[r] : -1;
  len[t] = mine;
  __syncthreads();
  int rank = 0;
  for (int j = 0; j < SP_SIGMA; ++j)
  {
    const int lj = len[j];
    rank += (lj > mine || (lj == mine && j < t)) ? 1 : 0;
  }
  srt[rank] = mine;
  const int64_t slot = w * SP_SIGMA + rank;
  if (slot < nslices * 64)
    perm[slot] = r < nrows ? (int32_t)r : -1;
  __syncthreads();
  // slice lengths: the first row of a sorted slice is its longest
  if (t < SP_SIGMA / 64)
  {
    const int64_t s = w * (SP_SIGMA / 64) + t;
    if (s < nslices)
    {
      const int m = max(srt[t * 64], 0);
      nch[s] = (m + 7) >> 3;
      wlast[s] = (uint8_t)(m ? m - 8 * ((m - 1) >> 3) : 8);
    }
  }
  if (w == 0 && t == 0)
    nch[nslices] = 0;
}

// Write chunk c of a slice from the lanes' next eight kept entries (v, cl; cl == INT_MAX: no entry).  Only the first
// w <= 8 slots are in use by any lane (w < 8: the last chunk of a slice): unused value blocks and the unused half of a
// code block are neither written nor ever read, so a narrow chunk costs its used bytes only -- an interior P1 row
// (7 entries) streams 3.5 KiB of values instead of 4.  Codes are as narrow as the chunk's slot ranges allow: 8-bit
// (consecutive rows reach consecutive columns: the usual case), 16-bit, or plain int32 columns.
// meta[c][0] carries the mode: bit 31 int32 columns, bit 30 8-bit codes.  Returns the bytes a product reads.
__device__ inline int emit_chunk(int c, int w, const double (&v_in)[8], int (&cl)[8], int lane, int nrows, bool& gh,
                                 double* __restrict__ svals, uint16_t* __restrict__ c16, int32_t* __restrict__ c32,
                                 int32_t* __restrict__ meta, int tail_codes, int& cls)
{
  // cls: what a product loads per lane for the chunk's columns (SP_CLS_*, zzz_sellp.h; -1: int32 columns)
  const bool affine_ok = (tail_codes & 2) == 0; // knob ZZZ_SELLP_AFFINE=0 sets bit 1
  double v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e)
    v[e] = v_in[e];
  // Aligned slices (flags bit 3: the slice's only chunk, scalar rows in natural order).  A slice that contains the end
  // of a mesh line has a few short rows (boundary vertices) whose entries, placed by rank, fall into other slots than
  // the same columns of their neighbours -- and the whole chunk needs codes.  Placed by COLUMN instead, into the slot
  // where the longest row of the slice has column - row = the same offset, every row fits the affine form
  // column = delta[slot] + lane, with holes (value +0.0) where a row has no such entry.  A row is still summed in
  // ascending column order; a hole adds +0.0 * x.
  bool aligned = false;
  int delta[8];
  if ((tail_codes & 8) && affine_ok)
  {
    int cnt = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e)
      cnt += (e < w && cl[e] != INT_MAX) ? 1 : 0;
    const unsigned long long full = __ballot(cnt == w); // w = the longest row's entries: never empty
    const int ref = __builtin_amdgcn_readfirstlane(__builtin_ctzll(full));
    bool okp = true; // uniform part: every lane's predicted column is a valid one
#pragma unroll
    for (int e = 0; e < 8; ++e)
    {
      delta[e] = e < w ? __builtin_amdgcn_readlane(cl[e], ref) - ref : 0;
      okp &= e >= w || (delta[e] >= 0 && delta[e] + 63 < nrows);
    }
    double nv[8];
    int nc[8], placed = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e)
    {
      nv[e] = 0.0;
      nc[e] = INT_MAX;
      if (e < w)
      {
        const int target = delta[e] + lane;
#pragma unroll
        for (int q = 0; q < 8; ++q)
          if (q < w && cl[q] == target)
          {
            nv[e] = v[q];
            nc[e] = target;
            ++placed;
          }
      }
    }
    aligned = okp && __all(placed == cnt);
    if (aligned)
    {
#pragma unroll
      for (int e = 0; e < 8; ++e)
      {
        v[e] = nv[e];
        cl[e] = nc[e];
      }
    }
  }
  int base[8];
  bool over8 = false, over16 = false, affine = true;
  unsigned has_mask = 0; // bit e: this lane has an entry in slot e
#pragma unroll
  for (int e = 0; e < 8; ++e)
  {
    base[e] = 0;
    if (e < w) // wave-uniform
    {
      const bool has = cl[e] != INT_MAX;
      has_mask |= has ? 1u << e : 0u;
      gh |= has && cl[e] >= nrows;
      int mn = wave_min_i(cl[e]);
      if (mn == INT_MAX)
        mn = 0;
      base[e] = mn;
      affine &= has && cl[e] - mn == lane; // 64 consecutive rows reach 64 consecutive columns
      if (!has)
        cl[e] = mn; // padding: value +0.0, a column some lane reads anyway
      over8 |= cl[e] - mn > 255;
      over16 |= cl[e] - mn > 65535;
    }
    else
      cl[e] = 0;
  }
  bool all_affine = __all(affine) && affine_ok;
  if (aligned)
  {
    all_affine = true;
#pragma unroll
    for (int e = 0; e < 8; ++e)
      base[e] = delta[e];
  }
  // Periodic chunks (block size 3, natural row order; flags bit 2, bits 8-9 = first row mod 3): rows 3 i + k reach
  // columns T[slot][k] + 3 i', i' = i - i0 -- three rows of a vertex share a block-column set, consecutive vertices
  // consecutive block columns.  25 scalars instead of 512 B - 2 KB of codes.
  bool periodic = false;
  int T[8][3];
  int q3 = 0;
  if (!all_affine && (tail_codes & 4))
  {
    const int l = lane + ((tail_codes >> 8) & 3);
    const int q = l / 3, k = l - 3 * q;
    q3 = 3 * q;
    bool ok = true;
#pragma unroll
    for (int e = 0; e < 8; ++e)
    {
#pragma unroll
      for (int kk = 0; kk < 3; ++kk)
        T[e][kk] = 0;
      if (e < w)
      {
        // (padding lanes carry cl == base here: they are free, so only lanes with an entry vote)
        const bool has = ((has_mask >> e) & 1u) != 0;
        const int d = cl[e] - q3;
        // T[e][kk] = the (common) value of d over the lanes of class kk that have an entry; a class without entries takes
        // another class's value: any column a lane reads anyway
        int any_t = INT_MAX;
#pragma unroll
        for (int kk = 0; kk < 3; ++kk)
        {
          T[e][kk] = wave_min_i((has && k == kk) ? d : INT_MAX);
          any_t = min(any_t, T[e][kk]);
        }
#pragma unroll
        for (int kk = 0; kk < 3; ++kk)
          if (T[e][kk] == INT_MAX)
            T[e][kk] = any_t;
        const int mine = k == 0 ? T[e][0] : (k == 1 ? T[e][1] : T[e][2]);
        ok &= !has || d == mine;
        ok &= mine + q3 >= 0 && mine + q3 < nrows; // padding lanes gather too
      }
    }
    periodic = __all(ok);
  }
  const int range = __any(over16) ? 65536 : (__any(over8) ? 256 : 0);
  double* sp = svals + (size_t)c * 512;
#pragma unroll
  for (int j = 0; j < 4; ++j)
  {
    if (2 * j + 1 < w)
    {
      dbl2 q;
      q.x = v[2 * j];
      q.y = v[2 * j + 1];
      reinterpret_cast<dbl2*>(sp + 128 * j)[lane] = q;
    }
    else if (2 * j < w)
      sp[128 * j + lane] = v[2 * j]; // odd width: the last entry alone, 8 B per lane
  }
  int mode = 0, code_bytes;
  if (periodic)
  {
    // the chunk's code block holds the 24 column bases T[slot][row mod 3] and the phase (scalar loads in the product)
    int32_t* tp = reinterpret_cast<int32_t*>(c16 + (size_t)c * 512);
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
      for (int kk = 0; kk < 3; ++kk)
        if (lane == 3 * e + kk)
          tp[3 * e + kk] = T[e][kk];
    if (lane == 24)
      tp[24] = (tail_codes >> 8) & 3;
    mode = (int)0xC0000000;
    code_bytes = 128;
    cls = SP_CLS_NONE;
  }
  else if (all_affine)
  {
    // every slot: column = base + lane.  No codes at all (an interior P1 slice away from the ends of a mesh line:
    // 56 instead of 64 B per row)
    mode = 0x20000000;
    code_bytes = 0;
    cls = SP_CLS_NONE;
  }
  else if (range > 65535)
  {
    int4v q0, q1;
    q0.x = cl[0], q0.y = cl[1], q0.z = cl[2], q0.w = cl[3];
    q1.x = cl[4], q1.y = cl[5], q1.z = cl[6], q1.w = cl[7];
    int4v* cp = reinterpret_cast<int4v*>(c32 + (size_t)c * 512) + 2 * lane;
    cp[0] = q0;
    cp[1] = q1;
    mode = (int)0x80000000;
    code_bytes = 2048;
    cls = -1;
  }
  else if (range > 255)
  {
    uint4v q;
    q.x = (unsigned)(cl[0] - base[0]) | ((unsigned)(cl[1] - base[1]) << 16);
    q.y = (unsigned)(cl[2] - base[2]) | ((unsigned)(cl[3] - base[3]) << 16);
    q.z = (unsigned)(cl[4] - base[4]) | ((unsigned)(cl[5] - base[5]) << 16);
    q.w = (unsigned)(cl[6] - base[6]) | ((unsigned)(cl[7] - base[7]) << 16);
    reinterpret_cast<uint4v*>(c16 + (size_t)c * 512)[lane] = q;
    code_bytes = 1024;
    cls = SP_CLS_C16;
  }
  else
  {
    uint2v q;
    q.x = (unsigned)(cl[0] - base[0]) | ((unsigned)(cl[1] - base[1]) << 8) | ((unsigned)(cl[2] - base[2]) << 16)
          | ((unsigned)(cl[3] - base[3]) << 24);
    q.y = (unsigned)(cl[4] - base[4]) | ((unsigned)(cl[5] - base[5]) << 8) | ((unsigned)(cl[6] - base[6]) << 16)
          | ((unsigned)(cl[7] - base[7]) << 24);
    if (w <= 7 && (tail_codes & 1))
    {
      // the chunk's value block has a free last 512 B: codes there, and the chunk is ONE contiguous 4-KiB read
      reinterpret_cast<uint2v*>(sp + 448)[lane] = q;
      mode = 0x60000000;
      cls = SP_CLS_C8T;
    }
    else
    {
      reinterpret_cast<uint2v*>(c16 + (size_t)c * 512)[lane] = q; // first half of the chunk's code block
      mode = 0x40000000;
      cls = SP_CLS_C8;
    }
    code_bytes = 512;
  }
#pragma unroll
  for (int e = 0; e < 8; ++e)
    if (lane == e)
      meta[(size_t)c * 8 + e] = e == 0 ? (base[e] | mode) : base[e];
  return (w >> 1) * 1024 + (w & 1) * 512 + code_bytes + 32;
}

// One wavefront packs one slice, one lane walking one row (rows too long for the LDS staging of k_sp_pack, and the
// length-sorted form).  desc[s] = {first chunk, chunks | width of the last chunk << 24}.  ghost_flag (or null): does
// the slice reference a column >= nrows?  bytes: the stream bytes a product will read are added up there.
template <bool PERM>
__global__ __launch_bounds__(256) void k_sp_fill(const rp_t* __restrict__ rowptr, const int32_t* __restrict__ cols,
                                                 const double* __restrict__ vals, int nrows, int64_t nslices, int drop,
                                                 const int32_t* __restrict__ perm, const int2* __restrict__ desc,
                                                 double* __restrict__ svals, uint16_t* __restrict__ c16,
                                                 int32_t* __restrict__ c32, int32_t* __restrict__ meta,
                                                 uint8_t* __restrict__ ghost_flag, unsigned long long* __restrict__ bytes, int tail_codes,
                                                 unsigned long long* __restrict__ smode, int* __restrict__ nopipe)
{
  const int lane = threadIdx.x & 63;
  unsigned long long mine = 0;
  for (int64_t s = blockIdx.x * 4 + (threadIdx.x >> 6); s < nslices; s += (int64_t)gridDim.x * 4)
  {
    int r = PERM ? perm[s * 64 + lane] : (int)(s * 64 + lane);
    if (!PERM && r >= nrows)
      r = -1;
    unsigned long long sm = 0; // the slice's mode word (zzz_sellp.h)
    bool sm_bad = false;
    int64_t k = r >= 0 ? rowptr[r] : 0;
    const int64_t end = r >= 0 ? rowptr[r + 1] : 0;
    const int2 ds = desc[s];
    const int c0 = ds.x, nch = ds.y & 0xffffff, wl = ds.y >> 24;
    // periodic chunks (flags bit 2) need the slice's first row mod 3 (bits 8-9); not for permuted rows
    // per-slice flags: bit 3 = the slice's only chunk (flags bit 4 allows the aligned placement); bits 8-9 = first row mod 3
    const int tc = PERM ? (tail_codes & ~(4 | 16)) : (((tail_codes & 4) ? (tail_codes | ((int)((s * 64) % 3) << 8)) : tail_codes) | (((tail_codes & 16) && nch == 1) ? 8 : 0));
    bool gh = false;
    for (int j = 0; j < nch; ++j)
    {
      double v[8];
      int cl[8];
#pragma unroll
      for (int e = 0; e < 8; ++e)
      {
        v[e] = 0.0;
        cl[e] = INT_MAX;
        while (k < end)
        {
          const double t = vals[k];
          const int64_t kk = k++;
          if (!drop || t != 0.0)
          {
            v[e] = t;
            cl[e] = cols[kk];
            break;
          }
        }
      }
      int cls = 0;
      mine += (unsigned long long)emit_chunk(c0 + j, j + 1 < nch ? 8 : wl, v, cl, lane, nrows, gh, svals, c16, c32, meta, tc, cls);
      sm_bad |= cls < 0 || j >= SP_SMODE_CHUNKS;
      if (cls > 0 && j < SP_SMODE_CHUNKS)
        sm |= (unsigned long long)cls << (2 * j);
    }
    if (lane == 0)
    {
      smode[s] = sm;
      if (sm_bad)
        *nopipe = 1;
    }
    if (ghost_flag)
    {
      const unsigned long long m = __ballot(gh);
      if (lane == 0)
        ghost_flag[s] = m != 0ull;
    }
  }
  // one atomic per workgroup (atomics on one address serialise at ~10 ns each)
  __shared__ unsigned long long mine_s[4];
  if (lane == 0)
    mine_s[threadIdx.x >> 6] = mine;
  __syncthreads();
  if (threadIdx.x == 0 && (mine_s[0] | mine_s[1] | mine_s[2] | mine_s[3]))
    atomicAdd(bytes, mine_s[0] + mine_s[1] + mine_s[2] + mine_s[3]);
}

// ---- long rows (P3): pack from a compacted copy.  k_sp_fill above has one lane walk one CSR row entry by entry; for
// rows of 50-200 entries every 8-B access of a lane is its own L2 request (58 GB through L2 for a 3.6-GB job at
// 6.2 M P3 dofs, 7.6 ms).  Instead: (1) k_sp_compact sweeps each slice's CSR range with dense loads and writes the kept
// entries row by row into a copy whose rows start at multiples of 8 entries (crow, from a scan of the padded counts);
// (2) k_sp_fill_c reads a lane's next eight entries as 64 + 32 contiguous, aligned bytes (four 16-B and two 16-B loads).
struct Even2
{
  __host__ __device__ int64_t operator()(int32_t n) const { return ((int64_t)n + 1) & ~(int64_t)1; }
};
struct Pad8
{
  __host__ __device__ int64_t operator()(int32_t n) const { return ((int64_t)n + 7) & ~(int64_t)7; }
};

__global__ __launch_bounds__(256) void k_sp_compact(const rp_t* __restrict__ rowptr, const int32_t* __restrict__ cols,
                                                    const double* __restrict__ vals, int nrows, int64_t nslices, int drop,
                                                    const int64_t* __restrict__ crow, double* __restrict__ cvals,
                                                    int32_t* __restrict__ ccols)
{
  const int lane = threadIdx.x & 63;
  for (int64_t s = blockIdx.x * 4 + (threadIdx.x >> 6); s < nslices; s += (int64_t)gridDim.x * 4)
  {
    const int64_t r = s * 64 + lane, rl = min(r, (int64_t)nrows - 1);
    const int64_t S = rowptr[s * 64], E = rowptr[min(s * 64 + 64, (int64_t)nrows)];
    const int64_t C = crow[s * 64];
    // this lane's row relative to the slice: CSR range [a, b), start in the compacted copy, entries kept so far
    const int a = r < nrows ? (int)(rowptr[r] - S) : (int)(E - S), b = r < nrows ? (int)(rowptr[r + 1] - S) : (int)(E - S);
    const int cst = (int)(crow[rl] - C);
    int kept = 0;
    int rho = 0; // first row that may still have entries at or behind the sweep position (wave-uniform)
    for (int64_t g = S; g < E; g += 64)
    {
      const int64_t k = g + lane;
      const bool in = k < E;
      const double v = in ? vals[k] : 0.0;
      const int32_t c = in ? cols[k] : 0;
      const bool keep = in && (!drop || v != 0.0);
      const unsigned long long m = __ballot(keep);
      const int g0 = (int)(g - S);
      int dest = -1;
      while (rho < 64)
      {
        const int ur = __builtin_amdgcn_readfirstlane(rho);
        const int ar = __builtin_amdgcn_readlane(a, ur), br = __builtin_amdgcn_readlane(b, ur);
        if (ar >= g0 + 64)
          break;
        const int lo = max(ar - g0, 0), hi = min(br - g0, 64);
        if (hi > lo)
        {
          const unsigned long long below_hi = hi == 64 ? ~0ull : (1ull << hi) - 1ull;
          const unsigned long long mask = m & below_hi & ~((1ull << lo) - 1ull);
          const int before = __builtin_amdgcn_readlane(kept, ur), st = __builtin_amdgcn_readlane(cst, ur);
          if (lane >= lo && lane < hi && keep)
            dest = st + before + __popcll(mask & ((1ull << lane) - 1ull));
          if (lane == ur)
            kept = before + __popcll(mask);
        }
        if (br > g0 + 64)
          break; // the row goes on in the next group
        ++rho;
      }
      if (dest >= 0)
      {
        cvals[C + dest] = v;
        ccols[C + dest] = c;
      }
    }
  }
}

// ---- x windows ------------------------------------------------------------------------------------------------
// Rows with many entries (P2 / P3, block size 3) gather x at 30-100 scattered places each; the gathers, not the stream,
// are then what the product waits for (-12 % / -17 % measured with the gathers taken off the memory
// path).  Where the columns a group of four slices (256 rows) reaches form a few contiguous segments that fit LDS, the
// product loads those segments once per group with wide coalesced loads and gathers from LDS.  The stream's column
// codes of such a group are LDS indices: the map column -> index is monotone and a translation inside a segment, so the
// chunk encodings (affine, periodic, 8- / 16-bit) and the ascending-column summation order are what they were.
// (SP_WIN_NSEG, SP_WIN_GAP, SP_WIN_WORDS, SP_WIN_SPAN: zzz_sellp.h)

// One workgroup per group of four slices (256 rows, natural order).  The group's kept columns (its CSR range swept with
// coalesced loads; entries that are exactly zero do not count when the stream drops them) are looked at as a set --
// bitmap over [smallest, largest], gaps of <= SP_WIN_GAP columns filled, runs = segments -- and where they form
// <= SP_WIN_NSEG segments of <= wmax doubles in all the group gets an x window: info[g] = {segments, doubles},
// seg[g][i] = {first column, length}; otherwise info[g] = {0, 0}.  count += window doubles.
__global__ __launch_bounds__(256) void k_sp_windows(const rp_t* __restrict__ rowptr, const int32_t* __restrict__ cols,
                                                    const double* __restrict__ vals, int nrows, int64_t ngroups, int drop,
                                                    int wmax, int2* __restrict__ info, int2* __restrict__ seg,
                                                    unsigned long long* __restrict__ count)
{
  __shared__ unsigned bits[SP_WIN_WORDS];
  __shared__ int win_red[8];
  __shared__ int win_cnt[2][257];
  __shared__ int win_pos[2][SP_WIN_NSEG];
  __shared__ int2 win_sg[SP_WIN_NSEG];
  __shared__ int win_hdr[2];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  unsigned long long win_w = 0;
  for (int64_t g = blockIdx.x; g < ngroups; g += gridDim.x)
  {
    const int r0 = (int)(g * 256), r1 = min(r0 + 256, nrows);
    const int64_t a = rowptr[r0], b = rowptr[r1];
    // bounds from the rows' first and last PATTERN entries (columns ascend within a row): at most a little wider than
    // the kept entries' own, and one sweep of the values instead of two
    int lo = INT_MAX, hi = -1;
    {
      const int r = r0 + (int)threadIdx.x;
      if (r < r1 && rowptr[r + 1] > rowptr[r])
      {
        lo = cols[rowptr[r]];
        hi = cols[rowptr[r + 1] - 1];
      }
    }
    lo = wave_min_i(lo);
    hi = wave_max_i(hi);
    __syncthreads(); // the previous group's shared state is done with
    if (lane == 0)
    {
      win_red[wv] = lo;
      win_red[4 + wv] = hi;
    }
    __syncthreads();
    lo = min(min(win_red[0], win_red[1]), min(win_red[2], win_red[3]));
    hi = max(max(win_red[4], win_red[5]), max(win_red[6], win_red[7]));
    const long long span = (long long)hi - lo + 1;
    bool ok = hi >= lo && span <= SP_WIN_SPAN; // (uniform over the workgroup)
    int nseg = 0, wlen = 0;
    if (ok)
    {
      const int nw = (int)((span + 31) / 32) + 1; // a spare word: the filled bitmap may carry into it
      for (int k = threadIdx.x; k < nw; k += 256)
        bits[k] = 0u;
      __syncthreads();
      for (int64_t k = a + threadIdx.x; k < b; k += 256)
        if (!drop || vals[k] != 0.0)
        {
          const int c = cols[k] - lo;
          atomicOr(&bits[c >> 5], 1u << (c & 31));
        }
      __syncthreads();
      // F = the bitmap with gaps of <= SP_WIN_GAP columns filled; a thread owns a contiguous range of words, so that
      // run starts and run ends come out in ascending order
      auto fword = [&](int k) -> unsigned {
        if (k < 0 || k >= nw)
          return 0u;
        const unsigned long long two = ((unsigned long long)bits[k] << 32) | (k > 0 ? bits[k - 1] : 0u);
        unsigned long long f = 0;
#pragma unroll
        for (int sft = 0; sft <= SP_WIN_GAP; ++sft)
          f |= two << sft;
        return (unsigned)(f >> 32);
      };
      const int per = (nw + 255) / 256;
      const int w0 = min((int)threadIdx.x * per, nw), w1 = min(w0 + per, nw);
      int ns = 0, ne = 0;
      for (int k = w0; k < w1; ++k)
      {
        const unsigned f = fword(k), below = fword(k - 1) >> 31, above = fword(k + 1) & 1u;
        ns += __popc(f & ~((f << 1) | below));
        ne += __popc(f & ~((f >> 1) | (above << 31)));
      }
      win_cnt[0][threadIdx.x] = ns;
      win_cnt[1][threadIdx.x] = ne;
      __syncthreads();
      if (threadIdx.x < 2)
      {
        int acc = 0;
        for (int k = 0; k < 256; ++k)
        {
          const int t = win_cnt[threadIdx.x][k];
          win_cnt[threadIdx.x][k] = acc;
          acc += t;
        }
        win_cnt[threadIdx.x][256] = acc;
      }
      __syncthreads();
      nseg = win_cnt[0][256];
      ok = nseg <= SP_WIN_NSEG && nseg == win_cnt[1][256];
      if (ok)
      {
        int is = win_cnt[0][threadIdx.x], ie = win_cnt[1][threadIdx.x];
        for (int k = w0; k < w1; ++k)
        {
          const unsigned f = fword(k), below = fword(k - 1) >> 31, above = fword(k + 1) & 1u;
          unsigned st = f & ~((f << 1) | below), en = f & ~((f >> 1) | (above << 31));
          while (st)
          {
            win_pos[0][is++] = k * 32 + __builtin_ctz(st);
            st &= st - 1;
          }
          while (en)
          {
            win_pos[1][ie++] = k * 32 + __builtin_ctz(en);
            en &= en - 1;
          }
        }
        __syncthreads();
        if (threadIdx.x == 0)
        {
          int total = 0;
          for (int q = 0; q < nseg; ++q)
          {
            const int a0 = win_pos[0][q];
            int e0 = win_pos[1][q]; // last filled bit: at most SP_WIN_GAP past the run's last column
            if (e0 >= (int)span)
              e0 = (int)span - 1;
            win_sg[q] = make_int2(lo + a0, e0 - a0 + 1);
            total += e0 - a0 + 1;
          }
          win_hdr[0] = total <= wmax ? nseg : 0;
          win_hdr[1] = total;
        }
        __syncthreads();
        nseg = win_hdr[0];
        wlen = win_hdr[1];
        ok = nseg > 0;
      }
    }
    if (ok && (int)threadIdx.x < nseg)
      seg[g * SP_WIN_NSEG + threadIdx.x] = win_sg[threadIdx.x];
    if (threadIdx.x == 0)
    {
      info[g] = ok ? make_int2(nseg, wlen) : make_int2(0, 0);
      if (ok)
        win_w += (unsigned long long)wlen;
    }
  }
  if (threadIdx.x == 0 && win_w)
    atomicAdd(count, win_w);
}

// column -> index into the group's window (segments in ascending order, laid out back to back)
__device__ inline int win_index(const int2* __restrict__ sg, int nseg, int col)
{
  int off = 0, idx = 0;
  for (int i = 0; i < nseg; ++i)
  {
    const int2 q = sg[i];
    if (col >= q.x)
      idx = off + (col - q.x);
    off += q.y;
  }
  return idx;
}

template <bool PERM>
__global__ __launch_bounds__(256) void k_sp_fill_c(const int64_t* __restrict__ crow, const int32_t* __restrict__ rownnz,
                                                   const double* __restrict__ cvals, const int32_t* __restrict__ ccols,
                                                   int nrows, int64_t nslices, const int32_t* __restrict__ perm,
                                                   const int2* __restrict__ desc, double* __restrict__ svals,
                                                   uint16_t* __restrict__ c16, int32_t* __restrict__ c32,
                                                   int32_t* __restrict__ meta, uint8_t* __restrict__ ghost_flag,
                                                   unsigned long long* __restrict__ bytes, int tail_codes,
                                                   unsigned long long* __restrict__ smode, int* __restrict__ nopipe)
{
  const int lane = threadIdx.x & 63;
  unsigned long long mine = 0;
  for (int64_t s = blockIdx.x * 4 + (threadIdx.x >> 6); s < nslices; s += (int64_t)gridDim.x * 4)
  {
    int r = PERM ? perm[s * 64 + lane] : (int)(s * 64 + lane);
    if (!PERM && r >= nrows)
      r = -1;
    unsigned long long sm = 0;
    bool sm_bad = false;
    const int n = r >= 0 ? rownnz[r] : 0;
    const int64_t base = r >= 0 ? crow[r] : 0;
    const int2 ds = desc[s];
    const int c0 = ds.x, nch = ds.y & 0xffffff, wl = ds.y >> 24;
    // per-slice flags: bit 3 = the slice's only chunk (flags bit 4 allows the aligned placement); bits 8-9 = first row mod 3
    const int tc = PERM ? (tail_codes & ~(4 | 16)) : (((tail_codes & 4) ? (tail_codes | ((int)((s * 64) % 3) << 8)) : tail_codes) | (((tail_codes & 16) && nch == 1) ? 8 : 0));
    bool gh = false;
    for (int j = 0; j < nch; ++j)
    {
      double v[8];
      int cl[8];
      const int rem = n - 8 * j; // entries this row still has
      if (rem > 0)
      {
        const dbl2* vp = reinterpret_cast<const dbl2*>(cvals + base + 8 * j);
        const int4v* cp = reinterpret_cast<const int4v*>(ccols + base + 8 * j);
        const dbl2 q0 = vp[0], q1 = vp[1], q2 = vp[2], q3 = vp[3];
        const int4v k0 = cp[0], k1 = cp[1];
        v[0] = q0.x, v[1] = q0.y, v[2] = q1.x, v[3] = q1.y, v[4] = q2.x, v[5] = q2.y, v[6] = q3.x, v[7] = q3.y;
        cl[0] = k0.x, cl[1] = k0.y, cl[2] = k0.z, cl[3] = k0.w, cl[4] = k1.x, cl[5] = k1.y, cl[6] = k1.z, cl[7] = k1.w;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (e >= rem)
        {
          v[e] = 0.0;
          cl[e] = INT_MAX;
        }
      int cls = 0;
      mine += (unsigned long long)emit_chunk(c0 + j, j + 1 < nch ? 8 : wl, v, cl, lane, nrows, gh, svals, c16, c32, meta, tc, cls);
      sm_bad |= cls < 0 || j >= SP_SMODE_CHUNKS;
      if (cls > 0 && j < SP_SMODE_CHUNKS)
        sm |= (unsigned long long)cls << (2 * j);
    }
    if (lane == 0)
    {
      smode[s] = sm;
      if (sm_bad)
        *nopipe = 1;
    }
    if (ghost_flag)
    {
      const unsigned long long m = __ballot(gh);
      if (lane == 0)
        ghost_flag[s] = m != 0ull;
    }
  }
  __shared__ unsigned long long mine_s[4];
  if (lane == 0)
    mine_s[threadIdx.x >> 6] = mine;
  __syncthreads();
  if (threadIdx.x == 0 && (mine_s[0] | mine_s[1] | mine_s[2] | mine_s[3]))
    atomicAdd(bytes, mine_s[0] + mine_s[1] + mine_s[2] + mine_s[3]);
}

// Slice bounds from the pattern alone (once per pattern): the longest CSR range of a slice (LDS staging of
// k_sp_pack) and the number of chunks the natural-order stream can need at most (no zero dropped).
__global__ __launch_bounds__(256) void k_sp_bounds(const rp_t* __restrict__ rowptr, int nrows, int64_t nslices,
                                                   int* __restrict__ out /* [0] max range, [1],[2] chunk bound lo/hi */)
{
  const int lane = threadIdx.x & 63;
  int mr = 0;
  unsigned long long ch = 0;
  for (int64_t s = blockIdx.x * 4 + (threadIdx.x >> 6); s < nslices; s += (int64_t)gridDim.x * 4)
  {
    const int r0 = (int)(s * 64), r = min(r0 + lane, nrows - 1);
    const int len = (r0 + lane < nrows) ? (int)(rowptr[r + 1] - rowptr[r]) : 0;
    const int m = wave_max_i(len);
    mr = max(mr, (int)(rowptr[min(r0 + 64, nrows)] - rowptr[r0]));
    ch += (unsigned long long)((m + 7) >> 3);
  }
  // one pair of atomics per WORKGROUP: returning or not, atomics on one address serialise (~12 ns each; 16 k wavefronts
  // made this kernel 0.39 ms at 10 M rows for 80 MB of row pointers)
  __shared__ int mr_s[4];
  __shared__ unsigned long long ch_s[4];
  if (lane == 0)
  {
    mr_s[threadIdx.x >> 6] = mr;
    ch_s[threadIdx.x >> 6] = ch;
  }
  __syncthreads();
  if (threadIdx.x == 0)
  {
    atomicMax(&out[0], max(max(mr_s[0], mr_s[1]), max(mr_s[2], mr_s[3])));
    atomicAdd(reinterpret_cast<unsigned long long*>(out + 2), ch_s[0] + ch_s[1] + ch_s[2] + ch_s[3]);
  }
}

// One pass from the CSR arrays to the stream, natural row order.  One wavefront per slice:
//   1. sweeps the slice's CSR range with dense loads, keeps the entries that are not exactly zero (all of
//      them when !drop) and parks them, compacted, in LDS; a row's first parked entry is found from the same
//      ballots (no search);
//   2. takes ceil(longest row / 8) chunks from a bump allocator (chunks of concurrently packed slices are
//      neighbours in memory; where a slice lands does not change any result);
//   3. every lane reads its row's entries back from LDS, chunk by chunk, and the chunk is written exactly as
//      k_sp_fill writes it.
// desc[s] = {first chunk, chunks | width of the last chunk << 24}.  ghost_flag as in k_sp_fill.
// WINB: groups with an x window (k_sp_windows) get window indices for columns before the chunks are written.
template <bool WINB>
__global__ __launch_bounds__(256) void k_sp_pack(const rp_t* __restrict__ rowptr, const int32_t* __restrict__ cols,
                                                 const double* __restrict__ vals, int nrows, int64_t nslices, int drop, int cap,
                                                 int* __restrict__ counter, int2* __restrict__ desc,
                                                 double* __restrict__ svals, uint16_t* __restrict__ c16,
                                                 int32_t* __restrict__ c32, int32_t* __restrict__ meta,
                                                 uint8_t* __restrict__ ghost_flag, int tail_codes,
                                                 const int2* __restrict__ win_info, const int2* __restrict__ win_seg,
                                                 unsigned long long* __restrict__ smode)
{
  extern __shared__ __attribute__((aligned(16))) char sp_smem[];

  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nwv = blockDim.x >> 6;
  double* lv = reinterpret_cast<double*>(sp_smem + (size_t)wv * cap * 12);
  int* lc = reinterpret_cast<int*>(sp_smem + (size_t)wv * cap * 12 + (size_t)cap * 8);
  const unsigned long long lt = (1ull << lane) - 1ull;
  __shared__ __attribute__((aligned(16))) int wg_sh[8]; // [0..3] chunks per wavefront, [4] the workgroup's first chunk
                                                          // (32 B: the dynamic region behind it stays 16-B aligned)
  int* wg_nch = wg_sh;
  unsigned long long kept_w = 0, bytes_w = 0;
  // the wavefronts of a workgroup take consecutive slices and walk in step: ONE allocator atomic per workgroup and
  // round (a returning atomic per slice on one address serialises: 156 k of them cost 2 ms at 10 M dofs, and
  // three per slice 5.6 ms)
  for (int64_t s0 = (int64_t)blockIdx.x * nwv; s0 < nslices; s0 += (int64_t)gridDim.x * nwv)
  {
    const int64_t s = s0 + wv;
    const bool live = s < nslices; // wave-uniform
    const int r0 = live ? (int)(s * 64) : 0;
    const int64_t a = live ? rowptr[r0] : 0, b = live ? rowptr[min(r0 + 64, nrows)] : 0;
    const int64_t my_start = live ? rowptr[min(r0 + lane, nrows)] : 0;
    int running = 0, cstart = 0;
    // four groups of 64 entries per round: their eight loads are in flight together (the sweep is a chain of
    // dependent ballots, but the loads depend on nothing)
    for (int64_t g0 = a; g0 < b; g0 += 256)
    {
      double vv[4];
      int cc[4];
#pragma unroll
      for (int u = 0; u < 4; ++u)
      {
        const int64_t k = g0 + 64 * u + lane;
        const bool in = k < b;
        vv[u] = in ? __builtin_nontemporal_load(vals + k) : 0.0;
        cc[u] = in ? __builtin_nontemporal_load(cols + k) : 0;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
      {
        const int64_t g = g0 + 64 * u;
        if (g >= b) // wave-uniform
          break;
        const bool nz = g + lane < b && (!drop || vv[u] != 0.0);
        const unsigned long long m = __ballot(nz);
        if (my_start >= g && my_start < g + 64)
          cstart = running + __popcll(m & ((1ull << (my_start - g)) - 1ull));
        if (nz)
        {
          const int pos = running + __popcll(m & lt);
          lv[pos] = vv[u];
          lc[pos] = cc[u];
        }
        running += __popcll(m);
      }
    }
    if (my_start >= b)
      cstart = running;
    const int nxt = __shfl_down(cstart, 1, 64);
    const int cnt = (lane == 63 ? running : nxt) - cstart;
    const int mlen = wave_max_i(cnt);
    const int nch = (mlen + 7) >> 3, wl = mlen ? mlen - 8 * ((mlen - 1) >> 3) : 8;
    if (lane == 0)
      wg_nch[wv] = live ? nch : 0;
    __syncthreads();
    if (threadIdx.x == 0)
    {
      int tot = 0;
      for (int q = 0; q < nwv; ++q)
        tot += wg_nch[q];
      wg_sh[4] = tot ? atomicAdd(counter, tot) : 0;
    }
    __syncthreads();
    int c0 = wg_sh[4];
    for (int q = 0; q < wv; ++q)
      c0 += wg_nch[q];
    __syncthreads(); // wg_sh is rewritten next round
    bool gh = false;
    int limit = nrows;
    if (WINB)
    {
      // the group's x window (k_sp_windows, before this kernel): its columns become window indices
      const int2 wi = live ? win_info[s >> 2] : make_int2(0, 0);
      if (wi.x > 0 && live)
      {
        const int2* __restrict__ sg = win_seg + (s >> 2) * SP_WIN_NSEG;
        for (int k = lane; k < running; k += 64)
        {
          const int c = lc[k];
          gh |= c >= nrows;
          lc[k] = win_index(sg, wi.x, c);
        }
        limit = wi.y;
      }
    }
    if (!live)
      continue;
    if (lane == 0)
      desc[s] = make_int2(c0, nch | (wl << 24));
    kept_w += (unsigned long long)running;
    const int tc = ((tail_codes & 4) ? (tail_codes | ((int)((s * 64) % 3) << 8)) : tail_codes) | (((tail_codes & 16) && nch == 1) ? 8 : 0);
    unsigned long long sm = 0; // the slice's mode word (zzz_sellp.h)
    bool sm_bad = false;
    for (int j = 0; j < nch; ++j)
    {
      double v[8];
      int cl[8];
#pragma unroll
      for (int e = 0; e < 8; ++e)
      {
        const int q = 8 * j + e;
        const bool has = q < cnt;
        v[e] = has ? lv[cstart + q] : 0.0;
        cl[e] = has ? lc[cstart + q] : INT_MAX;
      }
      bool gh2 = false; // (windowed: the ghost test was made on the columns themselves, above)
      int cls = 0;
      bytes_w += (unsigned long long)emit_chunk(c0 + j, j + 1 < nch ? 8 : wl, v, cl, lane, limit, (WINB && limit != nrows) ? gh2 : gh,
                                                svals, c16, c32, meta, tc, cls);
      sm_bad |= cls < 0 || j >= SP_SMODE_CHUNKS;
      if (cls > 0 && j < SP_SMODE_CHUNKS)
        sm |= (unsigned long long)cls << (2 * j);
    }
    if (lane == 0)
    {
      smode[s] = sm;
      if (sm_bad)
        counter[14] = 1; // a chunk with int32 columns or a slice of more than 32 chunks: the generic product
      if (nch > 1)
        counter[15] = 1; // not a stream of one-chunk slices (spmv_one_kernel serves those)
    }
    if (ghost_flag)
    {
      const unsigned long long m = __ballot(gh);
      if (lane == 0)
        ghost_flag[s] = m != 0ull;
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (lane == 0)
  {
    atomicAdd(reinterpret_cast<unsigned long long*>(counter + 2), kept_w);  // entries kept
    atomicAdd(reinterpret_cast<unsigned long long*>(counter + 8), bytes_w); // stream bytes a product reads
  }

}

// sorted form: {first chunk, chunks} of every slice from the scanned offsets
__global__ void k_sp_desc(const int32_t* __restrict__ off, const uint8_t* __restrict__ wlast, int64_t nslices,
                          int2* __restrict__ desc)
{
  for (int64_t s = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; s < nslices; s += (int64_t)gridDim.x * blockDim.x)
    desc[s] = make_int2(off[s], (off[s + 1] - off[s]) | ((int)wlast[s] << 24));
}

// ---- value dictionary -----------------------------------------------------------------------------------
// The entries of the packed stream as (slice, chunk, lane, slot) with the values the product would load: slots of a
// slice's last chunk beyond its width and lanes without a row are never loaded (and hold anything).
constexpr int SP_DICT_BITS = 18;                        // table of 2^18 slots for at most 65 535 values
constexpr unsigned long long SP_DICT_EMPTY = ~0ull;     // (a NaN pattern no assembled value has; met all the same: no dictionary)
constexpr int SP_DICT_MAX = 65535;
constexpr int SP_DICT_LDS_MAX = SP_DICT_LDS_ENTRIES;

__device__ inline unsigned sp_dict_hash(unsigned long long b)
{
  b ^= b >> 29;
  b *= 0x9E3779B97F4A7C15ull;
  return (unsigned)(b >> (64 - SP_DICT_BITS));
}

// value of entry (chunk c of width w, lane, slot e) in the value blocks: [4][64 lanes][2]; the last entry of an odd width
// sits alone, 8 B per lane (emit_chunk)
__device__ inline unsigned long long sp_value_bits(const double* __restrict__ svals, int64_t c, int w, int lane, int e)
{
  const int64_t at = ((w & 1) && e == w - 1) ? 128 * (e >> 1) + lane : 128 * (e >> 1) + 2 * lane + (e & 1);
  return reinterpret_cast<const unsigned long long*>(svals)[c * 512 + at];
}

// info[0] distinct values so far, info[1] overflow / unusable.  The lanes of a slice mostly hold the same value in a slot:
// one lane per distinct value of the wavefront goes to the table; the table is read past the L1 cache (a line cached as
// empty before another CU's insertion would send every later occurrence of that value to the atomic: 3.4 ms at 1.25 M rows
// instead of 0.05).
template <bool PERM>
__global__ __launch_bounds__(256) void k_sp_dict_insert(const int2* __restrict__ desc, const int32_t* __restrict__ perm,
                                                         const double* __restrict__ svals, int nrows, int64_t nslices,
                                                         unsigned long long* __restrict__ table, int* __restrict__ info, int limit)
{
  const int lane = threadIdx.x & 63;
  for (int64_t s = blockIdx.x * 4ll + (threadIdx.x >> 6); s < nslices; s += gridDim.x * 4ll)
  {
    if (__hip_atomic_load(&info[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
      return; // more distinct values than the dictionary may hold (an unstructured mesh): nothing left to find out
    const int r = PERM ? perm[s * 64 + lane] : (int)(s * 64 + lane);
    const bool row = r >= 0 && r < nrows;
    const int2 ds = desc[s];
    const int c0 = ds.x, nch = ds.y & 0xffffff, wl = ds.y >> 24;
    unsigned long long last = 0ull; // (+0.0 is code 0 without the table)
    for (int j = 0; j < nch; ++j)
    {
      const int w = j + 1 < nch ? 8 : wl;
      for (int e = 0; e < w; ++e)
      {
        const unsigned long long b = row ? sp_value_bits(svals, c0 + j, w, lane, e) : 0ull;
        bool need = b != last && b != 0ull; // (a row repeats its values: the previous one is in the table already)
        last = b;
        unsigned long long todo = __ballot(need);
        while (todo)
        {
          if (__hip_atomic_load(&info[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            return; // (the waves in flight when the limit is met would fill the table up otherwise)
          const int src = __ffsll((long long)todo) - 1;
          const unsigned long long bb = ((unsigned long long)(unsigned)__shfl((int)(b >> 32), src) << 32)
                                        | (unsigned)__shfl((int)(unsigned)b, src);
          if (lane == src)
          {
            if (bb == SP_DICT_EMPTY)
              info[1] = 1;
            else
            {
              unsigned h = sp_dict_hash(bb);
              for (int probe = 0; probe < (1 << SP_DICT_BITS); ++probe)
              {
                const unsigned long long cur = __hip_atomic_load(&table[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (cur == bb)
                  break;
                if (cur == SP_DICT_EMPTY)
                {
                  const unsigned long long old = atomicCAS(&table[h], SP_DICT_EMPTY, bb);
                  if (old == SP_DICT_EMPTY)
                  {
                    if (atomicAdd(&info[0], 1) >= limit - 1)
                      info[1] = 1;
                    break;
                  }
                  if (old == bb)
                    break;
                }
                h = (h + 1) & ((1u << SP_DICT_BITS) - 1);
                if (__hip_atomic_load(&info[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                  break; // too many distinct values: the table may be filling up, stop looking
              }
            }
          }
          need = need && b != bb;
          todo = __ballot(need);
        }
      }
    }
  }
}

// codes: every thread numbers the occupied slots it meets (slot = k * 1024 + thread), threads in order; code 0 = +0.0
__global__ __launch_bounds__(1024) void k_sp_dict_number(const unsigned long long* __restrict__ table, int32_t* __restrict__ slot_code,
                                                         double* __restrict__ dict, int* __restrict__ info, int lds_max, int forced)
{
  __shared__ int wsum[16];
  if (info[1])
    return;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int mine = 0;
  for (int k = threadIdx.x; k < (1 << SP_DICT_BITS); k += 1024)
    mine += table[k] != SP_DICT_EMPTY ? 1 : 0;
  // exclusive scan of `mine` over the 1024 threads
  int incl = mine;
  for (int d = 1; d < 64; d <<= 1)
  {
    const int t = __shfl_up(incl, d);
    if (lane >= d)
      incl += t;
  }
  if (lane == 63)
    wsum[wv] = incl;
  __syncthreads();
  int off = 1; // (code 0 is +0.0)
  for (int q = 0; q < wv; ++q)
    off += wsum[q];
  int code = off + incl - mine;
  for (int k = threadIdx.x; k < (1 << SP_DICT_BITS); k += 1024)
  {
    const unsigned long long b = table[k];
    if (b != SP_DICT_EMPTY)
    {
      slot_code[k] = code;
      if (code <= SP_DICT_MAX)
        dict[code] = __longlong_as_double((long long)b);
      ++code;
    }
  }
  if (threadIdx.x == 1023)
  {
    dict[0] = 0.0;
    info[2] = code; // entries of the dictionary, +0.0 included
    if (code > lds_max && !forced)
      info[1] = 2; // too large for the LDS copy: the stream stays as doubles, the encoding pass has nothing to do
  }
}

// the stream's values as codes, [chunk][lane][8] (16 B per lane and chunk); info[4..5]: bytes the product reads in this form
template <bool PERM>
__global__ __launch_bounds__(256) void k_sp_dict_encode(const int2* __restrict__ desc, const int32_t* __restrict__ perm,
                                                         const double* __restrict__ svals, const int32_t* __restrict__ meta,
                                                         int nrows, int64_t nslices, const unsigned long long* __restrict__ table,
                                                         const int32_t* __restrict__ slot_code, uint16_t* __restrict__ vcode,
                                                         int* __restrict__ info)
{
  if (info[1])
    return;
  const int lane = threadIdx.x & 63;
  unsigned long long bytes = 0;
  for (int64_t s = blockIdx.x * 4ll + (threadIdx.x >> 6); s < nslices; s += gridDim.x * 4ll)
  {
    const int r = PERM ? perm[s * 64 + lane] : (int)(s * 64 + lane);
    const bool row = r >= 0 && r < nrows;
    const int2 ds = desc[s];
    const int c0 = ds.x, nch = ds.y & 0xffffff, wl = ds.y >> 24;
    unsigned long long last = 0ull;
    unsigned last_code = 0;
    for (int j = 0; j < nch; ++j)
    {
      const int w = j + 1 < nch ? 8 : wl;
      unsigned code[8];
#pragma unroll
      for (int e = 0; e < 8; ++e)
      {
        code[e] = 0;
        if (row && e < w)
        {
          const unsigned long long b = sp_value_bits(svals, c0 + j, w, lane, e);
          if (b == 0ull)
            continue;
          if (b != last)
          {
            unsigned h = sp_dict_hash(b);
            while (table[h] != b)
              h = (h + 1) & ((1u << SP_DICT_BITS) - 1);
            last = b;
            last_code = (unsigned)slot_code[h];
          }
          code[e] = last_code;
        }
      }
      uint4v q;
      q.x = code[0] | (code[1] << 16);
      q.y = code[2] | (code[3] << 16);
      q.z = code[4] | (code[5] << 16);
      q.w = code[6] | (code[7] << 16);
      reinterpret_cast<uint4v*>(vcode + (size_t)(c0 + j) * 512)[lane] = q;
      if (lane == 0)
      {
        // what the product reads of this chunk: 1 KiB of value codes, the slot bases, the column codes by the chunk's mode
        const int m0 = meta[(size_t)(c0 + j) * 8];
        unsigned cb = 0;
        if (m0 < 0 && (m0 & 0x40000000))
          cb = 100; // periodic: 25 scalar words
        else if (m0 < 0)
          cb = 2048; // int32 columns
        else if ((m0 & 0x60000000) == 0x20000000)
          cb = 0; // affine
        else if (m0 & 0x40000000)
          cb = 512; // 8-bit codes
        else
          cb = 1024; // 16-bit codes
        bytes += 1024 + 32 + cb;
      }
    }
  }
  if (lane == 0 && bytes)
    atomicAdd(reinterpret_cast<unsigned long long*>(info + 4), bytes);
}

// ---- per-slice value dictionaries (long rows: P3) ----------------------------------------------------------------
// A slice of 64 rows of one entity type holds a few hundred distinct values even where the whole matrix holds thousands
// (P3 at 30^3 sub-cubes: median 296 per slice, all slices below 1 024; 7 400 in the matrix; 8 270 at 61^3).  One wavefront
// per slice: the slice's distinct values into a hash set in LDS (at most 1 023 besides +0.0), numbered as they arrive; then
// every value of the slice as a 16-bit code in the layout of the matrix-wide dictionary's codes ([chunk][lane][8], 16 B per
// lane and chunk), and the table beside it (sd_info[slice] = entries, 0 = this slice stays doubles; the tables back to back in
// sd_vals, slice s at sd_off[s]: a first pass (COUNT) finds the sizes, a scan the offsets -- 1 024 doubles reserved per slice
// were 6.4 GB at 49.8 M rows for 1.8 GB of tables).
// The product copies a slice's table into its wavefront's part of LDS (8 KiB per wavefront: five workgroups per CU).
// Tried: 8-bit codes and tables of 256 (a third of P3's slices qualify: product 0.67 -> 0.54 ms at 6.2 M dofs), tables of 512
// (0.46 ms there, 4.19 ms at 49.8 M dofs), tables of 1 024 (0.47 / 3.68 ms: kept).
// (SD_SLOTS = 2048, SD_MAX = 1024: zzz_sellp.h)
template <bool PERM, bool COUNT>
__global__ __launch_bounds__(128) void k_sp_sd_build(const int2* __restrict__ desc, const int32_t* __restrict__ perm,
                                                      const double* __restrict__ svals, const int32_t* __restrict__ meta,
                                                      int nrows, int64_t nslices, uint16_t* __restrict__ vcode8,
                                                      double* __restrict__ sd_vals, const int64_t* __restrict__ sd_off,
                                                      int32_t* __restrict__ sd_info, unsigned long long* __restrict__ bytes_out)
{
  __shared__ unsigned long long keys_s[2][SD_SLOTS];
  __shared__ uint16_t code_s[2][SD_SLOTS];
  __shared__ int cnt_s[2];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  unsigned long long* const keys = keys_s[wv];
  uint16_t* const codes = code_s[wv];
  unsigned long long bytes = 0;
  for (int64_t s = blockIdx.x * 2ll + wv; s < nslices; s += gridDim.x * 2ll)
  {
    for (int k = lane; k < SD_SLOTS; k += 64)
      keys[k] = ~0ull;
    if (lane == 0)
      cnt_s[wv] = 1; // (entry 0 is +0.0)
    __builtin_amdgcn_wave_barrier();
    const int r = PERM ? perm[s * 64 + lane] : (int)(s * 64 + lane);
    const bool row = r >= 0 && r < nrows;
    const int2 ds = desc[s];
    const int c0 = ds.x, nch = ds.y & 0xffffff, wl = ds.y >> 24;
    if (!COUNT && sd_info[s] == 0) // (the first pass found more values than a table holds: this slice stays doubles)
    {
      if (lane == 0)
      {
        for (int j = 0; j < nch; ++j)
        {
          const int w = j + 1 < nch ? 8 : wl;
          const int m0 = meta[(size_t)(c0 + j) * 8];
          const unsigned cb = (m0 < 0 && (m0 & 0x40000000)) ? 100u : (m0 < 0 ? 2048u : ((m0 & 0x60000000) == 0x20000000 ? 0u : ((m0 & 0x40000000) ? 512u : 1024u)));
          bytes += 32 + cb + (unsigned)((w >> 1) * 1024 + (w & 1) * 512);
        }
        bytes += 4;
        atomicAdd(reinterpret_cast<int*>(bytes_out) + 2, 1); // slices that stay doubles
      }
      continue;
    }
    double* const tab = COUNT ? nullptr : sd_vals + sd_off[s];
    if (!COUNT && lane == 0)
      tab[0] = 0.0;
    unsigned long long last = 0ull;
    for (int j = 0; j < nch; ++j)
    {
      if (cnt_s[wv] > SD_MAX)
        break; // (more values than the table holds: this slice stays doubles)
      const int w = j + 1 < nch ? 8 : wl;
      for (int e = 0; e < w; ++e)
      {
        const unsigned long long b = row ? sp_value_bits(svals, c0 + j, w, lane, e) : 0ull;
        if (b != 0ull && b != last && b != ~0ull)
        {
          unsigned h = sp_dict_hash(b) & (SD_SLOTS - 1);
          for (int probe = 0; probe < SD_SLOTS; ++probe)
          {
            const unsigned long long cur = keys[h];
            if (cur == b)
              break;
            if (cur == ~0ull)
            {
              const unsigned long long old = atomicCAS(&keys[h], ~0ull, b);
              if (old == ~0ull)
              {
                const int c = atomicAdd(&cnt_s[wv], 1);
                codes[h] = (uint16_t)c;
                if (!COUNT && c < SD_MAX)
                  tab[c] = __longlong_as_double((long long)b);
                break;
              }
              if (old == b)
                break;
            }
            h = (h + 1) & (SD_SLOTS - 1);
            if (cnt_s[wv] > SD_MAX)
              break;
          }
        }
        if (b == ~0ull)
          cnt_s[wv] = SD_MAX + 1;
        last = b;
      }
    }
    __builtin_amdgcn_wave_barrier();
    const int n = cnt_s[wv];
    const bool ok = n <= SD_MAX;
    if (COUNT)
    {
      if (lane == 0)
        sd_info[s] = ok ? n : 0;
      __builtin_amdgcn_wave_barrier();
      continue;
    }
    if (ok)
    {
      last = 0ull;
      unsigned last_code = 0;
      for (int j = 0; j < nch; ++j)
      {
        const int w = j + 1 < nch ? 8 : wl;
        unsigned code[8];
#pragma unroll
        for (int e = 0; e < 8; ++e)
        {
          code[e] = 0;
          if (row && e < w)
          {
            const unsigned long long b = sp_value_bits(svals, c0 + j, w, lane, e);
            if (b == 0ull)
              continue;
            if (b != last)
            {
              unsigned h = sp_dict_hash(b) & (SD_SLOTS - 1);
              while (keys[h] != b)
                h = (h + 1) & (SD_SLOTS - 1);
              last = b;
              last_code = codes[h];
            }
            code[e] = last_code;
          }
        }
        uint4v q;
        q.x = code[0] | (code[1] << 16);
        q.y = code[2] | (code[3] << 16);
        q.z = code[4] | (code[5] << 16);
        q.w = code[6] | (code[7] << 16);
        reinterpret_cast<uint4v*>(vcode8 + (size_t)(c0 + j) * 512)[lane] = q;
      }
    }
    if (lane == 0)
    {
      // what the product reads of this slice: the table and per chunk 512 B of codes, or the values as before
      for (int j = 0; j < nch; ++j)
      {
        const int w = j + 1 < nch ? 8 : wl;
        const int m0 = meta[(size_t)(c0 + j) * 8];
        unsigned cb = 0;
        if (m0 < 0 && (m0 & 0x40000000))
          cb = 100;
        else if (m0 < 0)
          cb = 2048;
        else if ((m0 & 0x60000000) == 0x20000000)
          cb = 0;
        else if (m0 & 0x40000000)
          cb = 512;
        else
          cb = 1024;
        bytes += 32 + cb + (ok ? 1024 : (unsigned)((w >> 1) * 1024 + (w & 1) * 512));
      }
      bytes += ok ? (unsigned)n * 8 + 4 : 4;
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (lane == 0 && bytes)
    atomicAdd(bytes_out, bytes);
}

// ---- the product --------------------------------------------------------------------------------------
template <bool NT, typename T>
__device__ inline T sp_load(const T* p)
{
  return NT ? __builtin_nontemporal_load(p) : *p;
}

// Values and columns of chunk c for this lane.  FULL: all eight slots (every chunk but the last of a slice);
// otherwise only the first w: unused value blocks are not loaded (an odd w loads its last entry with one 8-B load).
// Columns by the chunk's mode (meta[c][0]: bit 31 int32, bit 30 8-bit codes, else 16-bit codes on the slot bases).
// DICT: the values are 16-bit codes into the stream's dictionary (vcode: [chunk][lane][8], one 16-B load; code 0 = +0.0,
// which is also what the slots beyond a last chunk's width hold)
template <bool NT, bool FULL, int DICT = 0>
__device__ inline void read_chunk(int c, int w, int lane, const double* __restrict__ svals, const uint16_t* __restrict__ c16,
                                  const int32_t* __restrict__ c32, const int32_t* __restrict__ meta, dbl2 (&v)[4], int (&cl)[8],
                                  const uint16_t* __restrict__ vcode = nullptr, const double* __restrict__ dict = nullptr)
{
  const double* __restrict__ sp = svals + (size_t)c * 512;
  if (DICT == 1 || DICT == 2 || (DICT == 3 && dict != nullptr))
  {
    const uint4v q = sp_load<NT>(reinterpret_cast<const uint4v*>(vcode + (size_t)c * 512) + lane);
    // DICT == 2: `dict` is the workgroup's copy in LDS (a small dictionary: the lanes of a slice mostly hold the same
    // code in a slot -- broadcast reads); DICT == 1: gathers from memory; DICT == 3: the slice's own table in the
    // wavefront's LDS (dict == nullptr: this slice stays doubles, below)
    auto look = [&](unsigned code) -> double { return DICT == 1 ? gather(dict, (int)code) : dict[code]; };
    v[0].x = look(q.x & 0xffffu);
    v[0].y = look(q.x >> 16);
    v[1].x = look(q.y & 0xffffu);
    v[1].y = look(q.y >> 16);
    v[2].x = look(q.z & 0xffffu);
    v[2].y = look(q.z >> 16);
    v[3].x = look(q.w & 0xffffu);
    v[3].y = look(q.w >> 16);
  }
  else
#pragma unroll
  for (int j = 0; j < 4; ++j)
  {
    if (FULL || 2 * j + 1 < w)
      v[j] = sp_load<NT>(reinterpret_cast<const dbl2*>(sp + 128 * j) + lane);
    else if (2 * j < w)
    {
      v[j].x = sp_load<NT>(sp + 128 * j + lane);
      v[j].y = 0.0;
    }
    else
    {
      v[j].x = 0.0;
      v[j].y = 0.0;
    }
  }
  const int32_t* __restrict__ mp = meta + (size_t)c * 8;
  const int m0 = mp[0];
  if (m0 < 0 && (m0 & 0x40000000))
  {
    // periodic chunk (block size 3): column = T[slot][row mod 3] + 3 (row div 3 - first), nothing per lane to load
    const int32_t* __restrict__ tp = reinterpret_cast<const int32_t*>(c16 + (size_t)c * 512);
    // the 25 words are wave-uniform: pinned into scalar registers, the selects below stay register selects.  (Left to
    // itself the compiler folds them into a per-lane ADDRESS select, tp + 3 e + k, behind divergent branches -- and
    // that code decoded slot 0 of the third class wrongly when the chunk sits inside the chunk loop.)
    // (All 25 requested first, pinned after: pinning each word as it is loaded made 25 DEPENDENT scalar round trips of
    // them -- `s_load_dword; s_waitcnt lgkmcnt(0)` 25 times per periodic chunk, 2-3 us of a 3.5-us chunk at C4.)
    int t[25];
#pragma unroll
    for (int i = 0; i < 25; ++i)
      t[i] = tp[i];
#pragma unroll
    for (int i = 0; i < 25; ++i)
      asm volatile("" : "+s"(t[i]));
    const int l = lane + t[24];
    const int q = l / 3, k = l - 3 * q, q3 = 3 * q;
    // T[slot][k] without a select per class: T0 + (k >= 1) (T1 - T0) + (k == 2) (T2 - T1), the brackets as masks.  (The nested
    // selects compiled into divergent control flow, ~15 scalar instructions per slot: 163 scalar instructions per chunk at C4,
    // more than the CU's scalar unit issues in the time the chunk's bytes take.)
    const int k1 = k >= 1 ? -1 : 0, k2 = k == 2 ? -1 : 0;
#pragma unroll
    for (int e = 0; e < 8; ++e)
      cl[e] = (t[3 * e] + q3) + (k1 & (t[3 * e + 1] - t[3 * e])) + (k2 & (t[3 * e + 2] - t[3 * e + 1]));
  }
  else if (m0 < 0)
  {
    const int4v* __restrict__ cp = reinterpret_cast<const int4v*>(c32 + (size_t)c * 512) + 2 * lane;
    const int4v q0 = sp_load<NT>(cp), q1 = sp_load<NT>(cp + 1);
    cl[0] = q0.x, cl[1] = q0.y, cl[2] = q0.z, cl[3] = q0.w;
    cl[4] = q1.x, cl[5] = q1.y, cl[6] = q1.z, cl[7] = q1.w;
  }
  else if ((m0 & 0x60000000) == 0x20000000)
  {
    // affine chunk: column = slot base + lane, nothing to load
    cl[0] = (m0 & 0x1fffffff) + lane;
#pragma unroll
    for (int e = 1; e < 8; ++e)
      cl[e] = mp[e] + lane;
  }
  else if (m0 & 0x40000000)
  {
    const uint2v q = (m0 & 0x20000000) ? sp_load<NT>(reinterpret_cast<const uint2v*>(sp + 448) + lane)
                                       : sp_load<NT>(reinterpret_cast<const uint2v*>(c16 + (size_t)c * 512) + lane);
    cl[0] = (m0 & 0x1fffffff) + (int)(q.x & 0xffu);
    cl[1] = mp[1] + (int)((q.x >> 8) & 0xffu);
    cl[2] = mp[2] + (int)((q.x >> 16) & 0xffu);
    cl[3] = mp[3] + (int)(q.x >> 24);
    cl[4] = mp[4] + (int)(q.y & 0xffu);
    cl[5] = mp[5] + (int)((q.y >> 8) & 0xffu);
    cl[6] = mp[6] + (int)((q.y >> 16) & 0xffu);
    cl[7] = mp[7] + (int)(q.y >> 24);
  }
  else
  {
    const uint4v q = sp_load<NT>(reinterpret_cast<const uint4v*>(c16 + (size_t)c * 512) + lane);
    cl[0] = m0 + (int)(q.x & 0xffffu);
    cl[1] = mp[1] + (int)(q.x >> 16);
    cl[2] = mp[2] + (int)(q.y & 0xffffu);
    cl[3] = mp[3] + (int)(q.y >> 16);
    cl[4] = mp[4] + (int)(q.z & 0xffffu);
    cl[5] = mp[5] + (int)(q.z >> 16);
    cl[6] = mp[6] + (int)(q.w & 0xffffu);
    cl[7] = mp[7] + (int)(q.w >> 16);
  }
}

// sum += the chunk's products in ascending column order, mul and add rounded separately (the scalar CPU loop's bits)
template <bool NT, bool FULL, bool LDS = false, int DICT = 0>
__device__ inline void chunk_product(int c, int w, int lane, const double* __restrict__ svals, const uint16_t* __restrict__ c16,
                                     const int32_t* __restrict__ c32, const int32_t* __restrict__ meta,
                                     const double* __restrict__ x, double& sum, const uint16_t* __restrict__ vcode = nullptr,
                                     const double* __restrict__ dict = nullptr)
{
  dbl2 v[4];
  int cl[8];
  read_chunk<NT, FULL, DICT>(c, w, lane, svals, c16, c32, meta, v, cl, vcode, dict);
  double xv[8];
#pragma unroll
  for (int e = 0; e < 8; ++e)
    xv[e] = (FULL || e < w) ? (LDS ? x[cl[e]] : gather(x, cl[e])) : 0.0; // LDS: x is the group's window, cl its index
#pragma unroll
  for (int e = 0; e < 8; ++e)
    if (FULL || e < w)
      sum += ((e & 1) ? v[e >> 1].y : v[e >> 1].x) * xv[e];
}

template <bool DOT, bool NT, bool PERM, bool CHEB = false, bool WIN = false, int DICT = 0>
__global__ __launch_bounds__(SP_BLOCK, 8) void spmv_sellp_kernel(const int2* __restrict__ desc,
                                                              const double* __restrict__ svals,
                                                              const uint16_t* __restrict__ c16,
                                                              const int32_t* __restrict__ c32,
                                                              const int32_t* __restrict__ meta,
                                                              const int32_t* __restrict__ perm,
                                                              const double* __restrict__ x, double* __restrict__ y,
                                                              int nrows, int64_t nslices, double* __restrict__ partials,
                                                              const int* __restrict__ stop_flag,
                                                              const int32_t* __restrict__ group_list, int64_t nlist,
                                                              const double* __restrict__ rvec, int pstride, int nn_is_rr,
                                                              TailArgs tail, ChebEpi epi, const int2* __restrict__ win_info,
                                                              const int2* __restrict__ win_seg, const uint16_t* __restrict__ vcode,
                                                              const double* __restrict__ dict_g, int dict_n,
                                                              const int32_t* __restrict__ sd_info,
                                                              const int64_t* __restrict__ sd_off)
{
  // dynamic LDS: DICT == 2: the value dictionary (dict_n doubles, rounded up to 2); WIN: the group's x window behind it
  // (launch: sp_win_max doubles)
  extern __shared__ __attribute__((aligned(16))) double sp_lds[];
  double* const xwin = sp_lds + (DICT == 2 ? ((dict_n + 1) & ~1) : 0);
  // DICT == 3: dict_g holds the slices' tables (SD_MAX entries each), sp_lds one table per wavefront
  double* const sdl = sp_lds + (threadIdx.x >> 6) * SD_MAX;
  const double* dict = DICT == 2 ? sp_lds : dict_g;
  // group_list != nullptr: only the listed groups of 4 slices (interior or boundary subset of a partitioned
  // matrix); rvec != nullptr: also the partials of <r,x> and of the test norm (single-reduction CG), as in
  // spmv_tile_kernel
  __shared__ double red[SP_BLOCK / 64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t ngroups = group_list ? nlist : (nslices + 3) / 4; // a workgroup takes 4 consecutive slices
  double dot = 0.0, dot_rx = 0.0, dot_nn = 0.0;
  // A slice is a chain of dependent round trips: (group list ->) descriptor -> values / codes / bases -> gathers.  At the
  // 8-GPU per-rank size a wavefront has two or three slices in all, so the kernel's length is that chain times three;
  // the slice number and descriptor of the NEXT slice are therefore requested (scalar loads) before the current slice's
  // chunks, and those of the first slice before the stop word is looked at.
  auto next_slice = [&](int i, int& s_out, int2& ds_out) -> int { // 1 valid, 0 no slice for this wavefront, -1 done
    const int64_t gi = sp_xcd_item(ngroups, blockIdx.x, gridDim.x, i);
    if (gi < 0)
      return -1;
    const int64_t g = group_list ? group_list[gi] : gi;
    const int s = __builtin_amdgcn_readfirstlane((int)(4 * g + wv));
    s_out = s;
    if (s >= nslices)
      return 0;
    ds_out = desc[s];
    return 1;
  };
  int s_n = 0;
  int2 ds_n = make_int2(0, 0);
  int st_n = next_slice(0, s_n, ds_n);
  if (stop_flag && *stop_flag) // CG already converged: the host is a few iterations ahead
    return;
  if (DICT == 2)
  {
    // the dictionary into LDS: every thread's (at most eight) entries requested together, one round trip (five dependent ones
    // at C2's 1 204 values cost the 8-GPU per-rank product ~2 of its 12 us)
    double t[SP_DICT_LDS_ENTRIES / SP_BLOCK];
#pragma unroll
    for (int i = 0; i < SP_DICT_LDS_ENTRIES / SP_BLOCK; ++i)
    {
      const int k = (int)threadIdx.x + i * SP_BLOCK;
      t[i] = k < dict_n ? dict_g[k] : 0.0;
    }
#pragma unroll
    for (int i = 0; i < SP_DICT_LDS_ENTRIES / SP_BLOCK; ++i)
    {
      const int k = (int)threadIdx.x + i * SP_BLOCK;
      if (k < dict_n)
        sp_lds[k] = t[i];
    }
    __syncthreads();
  }
  for (int i = 0; st_n >= 0; ++i)
  {
    const int st = st_n, s = s_n;
    const int2 ds = ds_n;
    st_n = next_slice(i + 1, s_n, ds_n);
    // WIN: the four wavefronts work on one group; where the group has a window its segments of x are loaded into LDS
    // first (every wavefront takes part, also one without a slice of its own at the end of the matrix)
    int nwin = 0;
    if (WIN)
    {
      const int2 wi = win_info[s >> 2];
      nwin = __builtin_amdgcn_readfirstlane(wi.x);
      if (nwin > 0)
      {
        const int2* __restrict__ sg = win_seg + (int64_t)(s >> 2) * SP_WIN_NSEG;
        __syncthreads(); // the previous group's window is done with
        // Every entry of the window is REQUESTED before any is stored (a loop of load -> store per segment was five to
        // seven dependent round trips per group, a quarter of the group's time at C4): the segments are cut into blocks of
        // 256 entries, block i is thread-uniformly one segment's, thread t takes entry t of each; at most WIN_BLOCKS blocks
        // in registers at a time.
        constexpr int WIN_BLOCKS = 12;
        int q = 0, seg_first = 0, off = 0; // the segment of the current block, the first block of it, where it starts in the window
        int2 sq = sg[0];
        int c0s = __builtin_amdgcn_readfirstlane(sq.x), len = __builtin_amdgcn_readfirstlane(sq.y);
        for (int b0 = 0; q < nwin; b0 += WIN_BLOCKS)
        {
          double t[WIN_BLOCKS];
          int at[WIN_BLOCKS]; // where the entry goes (-1: none)
#pragma unroll
          for (int i = 0; i < WIN_BLOCKS; ++i)
          {
            at[i] = -1;
            t[i] = 0.0;
            while (q < nwin && (b0 + i - seg_first) * SP_BLOCK >= len) // (uniform: on to the segment this block belongs to)
            {
              off += len;
              seg_first = b0 + i;
              ++q;
              if (q < nwin)
              {
                sq = sg[q];
                c0s = __builtin_amdgcn_readfirstlane(sq.x);
                len = __builtin_amdgcn_readfirstlane(sq.y);
              }
            }
            if (q < nwin)
            {
              const int k = (b0 + i - seg_first) * SP_BLOCK + (int)threadIdx.x;
              if (k < len)
              {
                t[i] = x[c0s + k];
                at[i] = off + k;
              }
            }
          }
#pragma unroll
          for (int i = 0; i < WIN_BLOCKS; ++i)
            if (at[i] >= 0)
              xwin[at[i]] = t[i];
        }
        __syncthreads();
      }
    }
    if (st == 0)
      continue;
    if (DICT == 3)
    {
      // the slice's table into the wavefront's LDS (entries 0 .. n - 1; n == 0: the slice's values are doubles)
      const int n_sd = __builtin_amdgcn_readfirstlane(sd_info[s]);
      const int64_t sd_tab = sd_off[s];
      if (n_sd > 0)
      {
        __builtin_amdgcn_wave_barrier(); // (the previous slice's lookups are done)
        // the table requested 512 entries at a time before any of them is stored, 16 B per lane and request (the tables start
        // at even entries) -- not one dependent round trip per 64 entries (five for a median table of 296)
        const dbl2* __restrict__ tsrc = reinterpret_cast<const dbl2*>(dict_g + sd_tab);
        const int n2 = (n_sd + 1) >> 1;
        for (int b0 = 0; b0 * 64 < n2; b0 += 4) // (four requests = 512 entries at a time: registers)
        {
          dbl2 t[4];
#pragma unroll
          for (int i = 0; i < 4; ++i)
            t[i] = lane + 64 * (b0 + i) < n2 ? tsrc[lane + 64 * (b0 + i)] : dbl2{0.0, 0.0};
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (lane + 64 * (b0 + i) < n2)
              reinterpret_cast<dbl2*>(sdl)[lane + 64 * (b0 + i)] = t[i];
        }
        __builtin_amdgcn_wave_barrier();
        dict = sdl;
      }
      else
        dict = nullptr;
    }
    const int c0 = ds.x, nch = ds.y & 0xffffff, wl = ds.y >> 24;
    int r = PERM ? perm[(int64_t)s * 64 + lane] : s * 64 + lane;
    if (!PERM && r >= nrows)
      r = -1;
    const double xr = ((DOT || CHEB) && r >= 0) ? x[r] : 0.0;
    double sum = 0.0;
    if (WIN && nwin > 0)
    {
      for (int j = 0; j + 1 < nch; ++j)
        chunk_product<NT, true, true, DICT>(c0 + j, 8, lane, svals, c16, c32, meta, xwin, sum, vcode, dict);
      if (nch)
      {
        if (wl == 8)
          chunk_product<NT, true, true, DICT>(c0 + nch - 1, 8, lane, svals, c16, c32, meta, xwin, sum, vcode, dict);
        else
          chunk_product<NT, false, true, DICT>(c0 + nch - 1, wl, lane, svals, c16, c32, meta, xwin, sum, vcode, dict);
      }
    }
    else
    {
      for (int j = 0; j + 1 < nch; ++j)
        chunk_product<NT, true, false, DICT>(c0 + j, 8, lane, svals, c16, c32, meta, x, sum, vcode, dict);
      if (nch)
      {
        if (wl == 8)
          chunk_product<NT, true, false, DICT>(c0 + nch - 1, 8, lane, svals, c16, c32, meta, x, sum, vcode, dict);
        else
          chunk_product<NT, false, false, DICT>(c0 + nch - 1, wl, lane, svals, c16, c32, meta, x, sum, vcode, dict);
      }
    }
    if (CHEB)
    {
      // a term of the Chebyshev-Jacobi polynomial (ChebEpi): x is d, y the next d; DOT marks the last term
      if (r >= 0)
      {
        const double gi = -1.0 * (epi.dinv[r] * sum) + epi.g[r];
        const double dn = epi.c1 * xr + epi.c2 * gi;
        const double zi = epi.z[r] + dn;
        epi.z[r] = zi;
        if (DOT)
        {
          const double ri = epi.r[r];
          dot_rx += ri * zi;
          dot_nn += nn_is_rr ? ri * ri : zi * zi;
        }
        else
        {
          epi.g[r] = gi;
          y[r] = dn;
        }
      }
    }
    else if (r >= 0)
    {
      y[r] = sum;
      if (DOT)
      {
        dot += sum * xr;
        if (rvec)
        {
          const double rr_ = rvec[r];
          dot_rx += rr_ * xr;
          dot_nn += nn_is_rr ? rr_ * rr_ : xr * xr;
        }
      }
    }
  }
  if (DOT)
  {
    const double sres = CHEB ? 0.0 : block_reduce_sum(dot, red);
    double s1 = 0.0, s2 = 0.0;
    if (rvec || CHEB)
    {
      s1 = block_reduce_sum(dot_rx, red);
      s2 = block_reduce_sum(dot_nn, red);
    }
    if (!CHEB && tail.parts)
    {
      // multi-GPU: the all-reduce of these sums happens in the tail of this launch (zzz_tail.h); output order
      // (<r,x>, norm, <x,y>) for the single-reduction form, <x,y> alone otherwise
      if (rvec)
        tail_arrive(tail, s1, s2, sres);
      else
        tail_arrive(tail, sres, 0.0, 0.0);
      return;
    }
    if (threadIdx.x == 0)
    {
      if (!CHEB)
        partials[blockIdx.x] = sres;
      if (rvec || CHEB)
      {
        partials[pstride + blockIdx.x] = s1;
        partials[2 * pstride + blockIdx.x] = s2;
      }
    }
  }
}

#ifdef ZZZ_EXPERIMENTS // measurement-only kernels: the tools build (libzzz_hip_exp.so), never the product library
// ---- TIMING PROBE (ZZZ_EXP_WIN=<doubles>): what would an x window in LDS buy? ------------------------------------
// The cost structure of a windowed product without its packer: per group of four slices the workgroup loads <doubles>
// consecutive entries of x into LDS (coalesced 16-B loads), and every gather of the chunk loop reads LDS at a
// pseudo-random index instead of global memory.  The RESULT IS WRONG by construction; only zzz_spmv_time may run it.
template <bool NT>
__global__ __launch_bounds__(SP_BLOCK) void spmv_sellp_win_probe_kernel(const int2* __restrict__ desc,
                                                                    const double* __restrict__ svals,
                                                                    const uint16_t* __restrict__ c16,
                                                                    const int32_t* __restrict__ c32,
                                                                    const int32_t* __restrict__ meta,
                                                                    const double* __restrict__ x, double* __restrict__ y,
                                                                    int nrows, int64_t nslices, double* __restrict__ partials,
                                                                    int wlen, int lds_slots)
{
  extern __shared__ __attribute__((aligned(16))) double win[];
  __shared__ double red[SP_BLOCK / 64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t ngroups = (nslices + 3) / 4;
  double dot = 0.0;
  for (int i = 0;; ++i)
  {
    const int64_t g = sp_xcd_item(ngroups, blockIdx.x, gridDim.x, i);
    if (g < 0)
      break;
    __syncthreads();
    {
      int64_t base = g * 256 - wlen / 2;
      if (base < 0)
        base = 0;
      if (base + wlen > nrows)
        base = nrows > wlen ? nrows - wlen : 0;
      base &= ~(int64_t)1;
      const dbl2* __restrict__ src = reinterpret_cast<const dbl2*>(x + base);
      dbl2* dst = reinterpret_cast<dbl2*>(win);
      for (int k = threadIdx.x; k < wlen / 2; k += SP_BLOCK)
        dst[k] = src[k];
    }
    __syncthreads();
    const int s = __builtin_amdgcn_readfirstlane((int)(4 * g + wv));
    if (s >= nslices)
      continue;
    const int2 ds = desc[s];
    const int c0 = ds.x, nch = ds.y & 0xffffff, wl = ds.y >> 24;
    const int r = s * 64 + lane < nrows ? s * 64 + lane : -1;
    double sum = 0.0;
    for (int j = 0; j < nch; ++j)
    {
      dbl2 v[4];
      int cl[8];
      if (j + 1 < nch || wl == 8)
        read_chunk<NT, true>(c0 + j, 8, lane, svals, c16, c32, meta, v, cl);
      else
        read_chunk<NT, false>(c0 + j, wl, lane, svals, c16, c32, meta, v, cl);
#pragma unroll
      for (int e = 0; e < 8; ++e)
      {
        // the first lds_slots slots of every chunk gather from the window, the others from memory (a PARTIAL window)
        const unsigned idx = (unsigned)cl[e] % (unsigned)wlen;
        const double xv = e < lds_slots ? win[idx] : gather(x, min(cl[e], nrows - 1));
        sum += ((e & 1) ? v[e >> 1].y : v[e >> 1].x) * xv;
      }
    }
    if (r >= 0)
    {
      y[r] = sum;
      dot += sum;
    }
  }
  const double sres = block_reduce_sum(dot, red);
  if (threadIdx.x == 0 && partials)
    partials[blockIdx.x] = sres;
}

// ---- product fused with the direction update -------------------------------------------------------------
// One CG iteration as TWO kernels instead of three: the head of iteration `it` (convergence test, k_update_p of
// zzz_cg.hip) and the product w = A p, with p = z + b p_old formed on the fly where the product gathers it
// (the same two roundings as k_update_p, so p, w and every scalar keep their bits), written once per owned row
// into the OTHER p buffer (the gathers of other workgroups still read p_old), together with the pending
// solution update x += alpha_{it-1} p_old (src/cg.h:68,82).  Ghost entries of p follow the same recurrence from the
// ghost values of z, so the halo exchange of an iteration moves z instead of p.  An A/B variant (ZZZ_CG_FUSED=2):
// measured slower than the three-kernel form at every size tried (cg_solve has the numbers), kept because it
// pins the iteration's arithmetic from a second side -- tests demand identical bits from both forms.
template <bool NT, bool PERM>
__global__ __launch_bounds__(SP_BLOCK, 4) void spmv_sellp_dir_kernel(
    const int2* __restrict__ desc, const double* __restrict__ svals, const uint16_t* __restrict__ c16,
    const int32_t* __restrict__ c32, const int32_t* __restrict__ meta, const int32_t* __restrict__ perm,
    const double* __restrict__ z, const double* __restrict__ p_old, double* __restrict__ p_new, double* __restrict__ xsol,
    double* __restrict__ y, int nrows, int ncols, int64_t nslices, double* __restrict__ partials,
    const int32_t* __restrict__ group_list, int64_t nlist, int ghost_update, CgState* __restrict__ st,
    double* __restrict__ beta_hist, double* __restrict__ dp_hist, const double* __restrict__ alpha_hist, int it, CgParams P,
    const double* __restrict__ pa, const double* __restrict__ pb, int np)
{
  __shared__ double red[SP_BLOCK / 64];
  DirScalars S;
  if (!cg_direction_scalars(st, beta_hist, dp_hist, it, P, pa, pb, np, red, S))
    return;
  const double bcoef = S.rz / S.bprev;
  const double alpha = it > 0 ? alpha_hist[it - 1] : 0.0;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t ngroups = group_list ? nlist : (nslices + 3) / 4;
  double dot = 0.0;
  for (int i = 0;; ++i)
  {
    const int64_t gi = sp_xcd_item(ngroups, blockIdx.x, gridDim.x, i);
    if (gi < 0)
      break;
    const int64_t g = group_list ? group_list[gi] : gi;
    const int s = __builtin_amdgcn_readfirstlane((int)(4 * g + wv));
    if (s >= nslices)
      continue;
    int r = PERM ? perm[(int64_t)s * 64 + lane] : s * 64 + lane;
    if (!PERM && r >= nrows)
      r = -1;
    const double po = r >= 0 ? p_old[r] : 0.0;
    if (r >= 0 && it > 0)
      xsol[r] = alpha * po + xsol[r]; // the previous iteration's solution update, also by the launch that stops
    if (S.conv)
      continue;
    const double pn = r >= 0 ? bcoef * po + z[r] : 0.0;
    const int2 ds = desc[s];
    const int c0 = ds.x, nch = ds.y & 0xffffff, wl = ds.y >> 24;
    double sum = 0.0;
    for (int j = 0; j < nch; ++j)
    {
      const int w = j + 1 < nch ? 8 : wl;
      dbl2 v[4];
      int cl[8];
      read_chunk<NT, false>(c0 + j, w, lane, svals, c16, c32, meta, v, cl);
      double zv[8], pv[8];
#pragma unroll
      for (int e = 0; e < 8; ++e)
      {
        zv[e] = e < w ? gather(z, cl[e]) : 0.0;
        pv[e] = e < w ? gather(p_old, cl[e]) : 0.0;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (e < w)
          sum += ((e & 1) ? v[e >> 1].y : v[e >> 1].x) * (bcoef * pv[e] + zv[e]);
    }
    if (r >= 0)
    {
      p_new[r] = pn;
      y[r] = sum;
      dot += sum * pn;
    }
  }
  if (S.conv)
    return;
  if (ghost_update) // ghost entries of the new direction (the launch that runs behind the halo of z)
    for (int64_t k = nrows + blockIdx.x * (int64_t)SP_BLOCK + threadIdx.x; k < ncols; k += (int64_t)gridDim.x * SP_BLOCK)
      p_new[k] = bcoef * p_old[k] + z[k];
  const double sres = block_reduce_sum(dot, red);
  if (threadIdx.x == 0)
    partials[blockIdx.x] = sres;
}

#endif // ZZZ_EXPERIMENTS

// ---- host side ------------------------------------------------------------------------------------------
static int grid_cap(int64_t items, int per, int cap)
{
  int64_t g = (items + per - 1) / per;
  if (g > cap)
    g = cap;
  if (g < 1)
    g = 1;
  return (int)g;
}

// Time estimates (relative) of one product: bytes over the rate each form was measured to stream at on MI355X
// (tile kernel 3.5-4.1 TB/s of its 10 B per pattern entry; stream in natural row order 4.8-5.4 TB/s, with sorted
// rows 4.4-4.8 TB/s: the x gather is no longer dense).
static double cost_tile(const zzz_ctx* ctx) { return 10.0 * (double)ctx->nnz / 3.8; }
static double cost_stream(int64_t chunks, bool sorted) { return 5152.0 * (double)chunks / (sorted ? 4.5 : 5.0); }

// Chunk storage for `total` chunks.
static int sp_alloc_stream(zzz_ctx* ctx, int64_t total)
{
  const size_t ne = (size_t)total * 512 + 512;
  ZZZ_HIP(ctx, ctx->sp_vals.alloc(ne));
  ZZZ_HIP(ctx, ctx->sp_codes16.alloc(ne));
  ZZZ_HIP(ctx, ctx->sp_codes32.alloc(ne)); // touched only by chunks that need int32 columns
  ZZZ_HIP(ctx, ctx->sp_meta.alloc((size_t)total * 8 + 8));
  ZZZ_HIP(ctx, ctx->sp_smode.alloc((size_t)ctx->nslices + 1)); // the slices' mode words (zzz_sellp.h)
  return ZZZ_OK;
}

// interior / boundary groups of 4 slices for the halo-compute overlap of a partitioned matrix
static int sp_group_split(zzz_ctx* ctx, const uint8_t* gflag)
{
  hipStream_t s = ctx->stream;
  const int64_t nsl = ctx->nslices;
  ctx->n_groups_interior = ctx->n_groups_boundary = 0;
  ctx->have_group_split = false;
  if (!gflag)
    return ZZZ_OK;
  std::vector<uint8_t> h((size_t)nsl);
  ZZZ_HIP(ctx, hipMemcpyAsync(h.data(), gflag, h.size(), hipMemcpyDeviceToHost, s));
  ZZZ_HIP(ctx, hipStreamSynchronize(s));
  const int64_t ng = (nsl + 3) / 4;
  std::vector<int32_t> in, bd;
  for (int64_t g2 = 0; g2 < ng; ++g2)
  {
    bool gh = false;
    for (int64_t q = 4 * g2; q < std::min(nsl, 4 * g2 + 4); ++q)
      gh |= h[(size_t)q] != 0;
    (gh ? bd : in).push_back((int32_t)g2);
  }
  ZZZ_HIP(ctx, ctx->groups_interior.alloc(in.size()));
  ZZZ_HIP(ctx, ctx->groups_boundary.alloc(bd.size()));
  if (!in.empty())
    ZZZ_HIP(ctx, hipMemcpyAsync(ctx->groups_interior.p, in.data(), in.size() * sizeof(int32_t), hipMemcpyHostToDevice, s));
  if (!bd.empty())
    ZZZ_HIP(ctx, hipMemcpyAsync(ctx->groups_boundary.p, bd.data(), bd.size() * sizeof(int32_t), hipMemcpyHostToDevice, s));
  ZZZ_HIP(ctx, hipStreamSynchronize(s));
  ctx->n_groups_interior = (int64_t)in.size();
  ctx->n_groups_boundary = (int64_t)bd.size();
  ctx->have_group_split = true;
  return ZZZ_OK;
}

// Rows ordered by length inside windows (SELL-C-sigma): count, sort, scan, fill -- a synchronous build, used only
// for matrices whose natural-order stream would be padded beyond use.
static int sp_build_sorted(zzz_ctx* ctx, int64_t* total_out, bool sorted = true)
{
  hipStream_t s = ctx->stream;
  const int nrows = (int)ctx->nrows;
  const int64_t nsl = ctx->nslices;
  const int drop = ctx->sellp_drop ? 1 : 0;
  ZZZ_HIP(ctx, ctx->sp_rownnz.alloc((size_t)nrows + 1));
  ZZZ_HIP(ctx, hipMemsetAsync(ctx->sp_rownnz.p + nrows, 0, sizeof(int32_t), s)); // closes the scans over nrows + 1 entries
  ZZZ_HIP(ctx, ctx->sp_nch.alloc((size_t)nsl + 1));
  ZZZ_HIP(ctx, ctx->sp_chunk_off.alloc((size_t)nsl + 1));
  ZZZ_HIP(ctx, ctx->sp_perm.alloc((size_t)nsl * 64));
  ZZZ_HIP(ctx, ctx->sp_wlast.alloc((size_t)nsl + 1));
  const bool counted = drop && ctx->sp_rownnz_fresh; // the matrix assembly has left the counts (asm_matrix_pk_pos)
  ctx->sp_rownnz_fresh = false;
  if (counted)
    ;
  else if (drop && ctx->nnz >= 16 * ctx->nrows) // long rows: dense sweep (short rows: a lane's row is one or two cache lines)
    hipLaunchKernelGGL(k_sp_count_sweep, dim3(grid_cap(nsl, 4, 8192)), dim3(256), 0, s, ctx->rowptr.p, ctx->vals.p, nrows, nsl,
                       ctx->sp_rownnz.p);
  else
    hipLaunchKernelGGL(k_sp_count, dim3(grid_cap(nrows, 256, 16384)), dim3(256), 0, s, ctx->rowptr.p, ctx->vals.p, nrows, drop,
                       ctx->sp_rownnz.p);
  const int64_t nwin = (ctx->nrows + SP_SIGMA - 1) / SP_SIGMA;
  if (sorted)
    hipLaunchKernelGGL(k_sp_sort, dim3((unsigned)nwin), dim3(SP_SIGMA), 0, s, ctx->sp_rownnz.p, nrows, nsl, ctx->sp_perm.p,
                       ctx->sp_nch.p, ctx->sp_wlast.p);
  else // natural row order, rows too long for the LDS staging of k_sp_pack
    hipLaunchKernelGGL(k_sp_slice_len, dim3(grid_cap(nsl + 1, 4, 8192)), dim3(256), 0, s, ctx->sp_rownnz.p, nrows, nsl,
                       ctx->sp_nch.p, ctx->sp_wlast.p);
  size_t tb = 0;
  ZZZ_HIP(ctx, rocprim::exclusive_scan(nullptr, tb, ctx->sp_nch.p, ctx->sp_chunk_off.p, 0, (size_t)nsl + 1,
                                       rocprim::plus<int32_t>(), s));
  ZZZ_HIP(ctx, ctx->scr_tmp.alloc(tb));
  ZZZ_HIP(ctx, rocprim::exclusive_scan(ctx->scr_tmp.p, tb, ctx->sp_nch.p, ctx->sp_chunk_off.p, 0, (size_t)nsl + 1,
                                       rocprim::plus<int32_t>(), s));
  int32_t* tot = reinterpret_cast<int32_t*>(ctx->h_state + 4); // pinned scratch
  ZZZ_HIP(ctx, hipMemcpyAsync(&tot[0], ctx->sp_chunk_off.p + nsl, sizeof(int32_t), hipMemcpyDeviceToHost, s));
  ZZZ_HIP(ctx, hipStreamSynchronize(s));
  *total_out = tot[0];
  return ZZZ_OK;
}

static int sp_fill_sorted(zzz_ctx* ctx, int64_t total, bool sorted = true)
{
  hipStream_t s = ctx->stream;
  const int nrows = (int)ctx->nrows;
  const int64_t nsl = ctx->nslices;
  int rc = sp_alloc_stream(ctx, total);
  if (rc)
    return rc;
  DevBuf<uint8_t> flag;
  uint8_t* gflag = nullptr;
  if (ctx->n_ghost > 0)
  {
    ZZZ_HIP(ctx, flag.alloc((size_t)nsl));
    gflag = flag.p;
  }
  // descriptors first: the fill reads chunk ranges and last-chunk widths from them
  int2* desc = reinterpret_cast<int2*>(ctx->sp_desc.p);
  hipLaunchKernelGGL(k_sp_desc, dim3(grid_cap(nsl, 256, 4096)), dim3(256), 0, s, ctx->sp_chunk_off.p, ctx->sp_wlast.p, nsl, desc);
  unsigned long long* bytes = reinterpret_cast<unsigned long long*>(ctx->sp_counter.p + 8);
  ZZZ_HIP(ctx, hipMemsetAsync(bytes, 0, sizeof(unsigned long long), s));
  int* nopipe = ctx->sp_counter.p + 14; // set by a chunk with int32 columns or a slice of more than 32 chunks
  ZZZ_HIP(ctx, hipMemsetAsync(nopipe, 0, sizeof(int), s));
  if (ctx->nnz >= 16 * ctx->nrows && ctx->nnz + 8 * ctx->nrows < ((int64_t)1 << 40))
  {
    // long rows: through the compacted copy (crow = scan of the kept counts padded to 8)
    const int64_t cap = ctx->nnz + 8 * ctx->nrows;
    const bool compacted = ctx->sp_compact_fresh && ctx->sp_crow_is_cap; // the matrix assembly has written the copy already
    ctx->sp_compact_fresh = false;
    if (!compacted)
    {
      ZZZ_HIP(ctx, ctx->sp_crow.alloc((size_t)nrows + 1));
      ZZZ_HIP(ctx, ctx->sp_cvals.alloc((size_t)cap));
      ZZZ_HIP(ctx, ctx->sp_ccols.alloc((size_t)cap));
      ctx->sp_crow_is_cap = false;
      const auto padded = rocprim::make_transform_iterator(ctx->sp_rownnz.p, Pad8{});
      size_t tb = 0;
      ZZZ_HIP(ctx, rocprim::exclusive_scan(nullptr, tb, padded, ctx->sp_crow.p, (int64_t)0, (size_t)nrows + 1,
                                           rocprim::plus<int64_t>(), s));
      ZZZ_HIP(ctx, ctx->scr_tmp.alloc(tb));
      ZZZ_HIP(ctx, rocprim::exclusive_scan(ctx->scr_tmp.p, tb, padded, ctx->sp_crow.p, (int64_t)0, (size_t)nrows + 1,
                                           rocprim::plus<int64_t>(), s));
      hipLaunchKernelGGL(k_sp_compact, dim3(grid_cap(nsl, 4, 8192)), dim3(256), 0, s, ctx->rowptr.p, ctx->cols.p, ctx->vals.p, nrows,
                         nsl, ctx->sellp_drop ? 1 : 0, ctx->sp_crow.p, ctx->sp_cvals.p, ctx->sp_ccols.p);
    }
    if (sorted)
      hipLaunchKernelGGL(k_sp_fill_c<true>, dim3(grid_cap(nsl, 4, 8192)), dim3(256), 0, s, ctx->sp_crow.p, ctx->sp_rownnz.p,
                         ctx->sp_cvals.p, ctx->sp_ccols.p, nrows, nsl, ctx->sp_perm.p, desc, ctx->sp_vals.p, ctx->sp_codes16.p,
                         ctx->sp_codes32.p, ctx->sp_meta.p, gflag, bytes, ctx->sellp_tail | ((ctx->bs == 3 && ctx->sellp_periodic) ? 4 : 0) | ((ctx->bs == 1 && ctx->sellp_align) ? 16 : 0), ctx->sp_smode.p, nopipe);
    else
      hipLaunchKernelGGL(k_sp_fill_c<false>, dim3(grid_cap(nsl, 4, 8192)), dim3(256), 0, s, ctx->sp_crow.p, ctx->sp_rownnz.p,
                         ctx->sp_cvals.p, ctx->sp_ccols.p, nrows, nsl, (const int32_t*)nullptr, desc, ctx->sp_vals.p,
                         ctx->sp_codes16.p, ctx->sp_codes32.p, ctx->sp_meta.p, gflag, bytes, ctx->sellp_tail | ((ctx->bs == 3 && ctx->sellp_periodic) ? 4 : 0) | ((ctx->bs == 1 && ctx->sellp_align) ? 16 : 0), ctx->sp_smode.p, nopipe);
  }
  else if (sorted)
    hipLaunchKernelGGL(k_sp_fill<true>, dim3(grid_cap(nsl, 4, 16384)), dim3(256), 0, s, ctx->rowptr.p, ctx->cols.p, ctx->vals.p,
                       nrows, nsl, ctx->sellp_drop ? 1 : 0, ctx->sp_perm.p, desc, ctx->sp_vals.p, ctx->sp_codes16.p,
                       ctx->sp_codes32.p, ctx->sp_meta.p, gflag, bytes, ctx->sellp_tail | ((ctx->bs == 3 && ctx->sellp_periodic) ? 4 : 0) | ((ctx->bs == 1 && ctx->sellp_align) ? 16 : 0), ctx->sp_smode.p, nopipe);
  else
    hipLaunchKernelGGL(k_sp_fill<false>, dim3(grid_cap(nsl, 4, 16384)), dim3(256), 0, s, ctx->rowptr.p, ctx->cols.p, ctx->vals.p,
                       nrows, nsl, ctx->sellp_drop ? 1 : 0, (const int32_t*)nullptr, desc, ctx->sp_vals.p, ctx->sp_codes16.p,
                       ctx->sp_codes32.p, ctx->sp_meta.p, gflag, bytes, ctx->sellp_tail | ((ctx->bs == 3 && ctx->sellp_periodic) ? 4 : 0) | ((ctx->bs == 1 && ctx->sellp_align) ? 16 : 0), ctx->sp_smode.p, nopipe);
  ZZZ_HIP(ctx, hipGetLastError());
  unsigned long long hb = 0;
  int hnp = 0;
  ZZZ_HIP(ctx, hipMemcpyAsync(&hb, bytes, sizeof(hb), hipMemcpyDeviceToHost, s));
  ZZZ_HIP(ctx, hipMemcpyAsync(&hnp, nopipe, sizeof(hnp), hipMemcpyDeviceToHost, s));
  ZZZ_HIP(ctx, hipStreamSynchronize(s));
  ctx->sp_bytes = (int64_t)hb;
  ctx->sp_pipe_ok = hnp == 0;
  ctx->sp_one_chunk = false; // (the synchronous builds serve long rows)
  ctx->sp_sorted = sorted;
  ctx->sp_chunks = total;
  return sp_group_split(ctx, gflag);
}

__global__ void k_sp_cap_len(const rp_t* __restrict__ rowptr, int nrows, int64_t* __restrict__ out)
{
  for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r <= nrows; r += (int64_t)gridDim.x * blockDim.x)
    out[r] = r < nrows ? ((rowptr[r + 1] - rowptr[r] + 7) & ~(int64_t)7) : 0;
}

// Row starts of the compacted copy by CAPACITY (every row has room for its whole pattern row, padded to 8): a
// function of the pattern alone, so that the matrix assembly can write the kept entries of a row where the packer
// will look for them without knowing how many the rows before it keep (asm_matrix_pk_pos; k_sp_fill_c reads
// crow[r] and rownnz[r] only).
int sellp_capacity_rows(zzz_ctx* ctx)
{
  if (ctx->sp_crow_is_cap)
    return ZZZ_OK;
  hipStream_t s = ctx->stream;
  const int nrows = (int)ctx->nrows;
  const int64_t cap = ctx->nnz + 8 * ctx->nrows;
  ZZZ_HIP(ctx, ctx->sp_crow.alloc((size_t)nrows + 1));
  ZZZ_HIP(ctx, ctx->sp_cvals.alloc((size_t)cap));
  ZZZ_HIP(ctx, ctx->sp_ccols.alloc((size_t)cap));
  hipLaunchKernelGGL(k_sp_cap_len, dim3(grid_cap((int64_t)nrows + 1, 256, 4096)), dim3(256), 0, s, ctx->rowptr.p, nrows, ctx->sp_crow.p);
  size_t tb = 0;
  ZZZ_HIP(ctx, rocprim::exclusive_scan(nullptr, tb, ctx->sp_crow.p, ctx->sp_crow.p, (int64_t)0, (size_t)nrows + 1,
                                       rocprim::plus<int64_t>(), s));
  ZZZ_HIP(ctx, ctx->scr_tmp.alloc(tb));
  ZZZ_HIP(ctx, rocprim::exclusive_scan(ctx->scr_tmp.p, tb, ctx->sp_crow.p, ctx->sp_crow.p, (int64_t)0, (size_t)nrows + 1,
                                       rocprim::plus<int64_t>(), s));
  ZZZ_HIP(ctx, hipGetLastError());
  ctx->sp_crow_is_cap = true;
  return ZZZ_OK;
}

// Pattern-only bounds of the natural-order stream (once per pattern; one small read-back).
int sellp_pattern_bounds(zzz_ctx* ctx)
{
  ctx->sp_bounds_ok = false;
  ctx->sp_crow_is_cap = false;
  ctx->sp_compact_fresh = ctx->sp_rownnz_fresh = false;
  ctx->have_sell = ctx->sell_current = ctx->sp_pending = false;
  if (ctx->nrows <= 0)
    return ZZZ_OK;
  hipStream_t s = ctx->stream;
  const int64_t nsl = (ctx->nrows + 63) / 64;
  ctx->nslices = nsl;
  ZZZ_HIP(ctx, ctx->sp_counter.alloc(16)); // [0] chunk allocator, [2,3] entries kept, [4..7] pattern bounds, [8,9] stream bytes,
                                           // [10..13] windowed groups / window doubles
  ZZZ_HIP(ctx, hipMemsetAsync(ctx->sp_counter.p, 0, 16 * sizeof(int), s));
  hipLaunchKernelGGL(k_sp_bounds, dim3(grid_cap(nsl, 4, 1024)), dim3(256), 0, s, ctx->rowptr.p, (int)ctx->nrows, nsl,
                     ctx->sp_counter.p + 4);
  int h[4] = {0, 0, 0, 0};
  ZZZ_HIP(ctx, hipMemcpyAsync(h, ctx->sp_counter.p + 4, sizeof(h), hipMemcpyDeviceToHost, s));
  ZZZ_HIP(ctx, hipStreamSynchronize(s));
  ctx->sp_max_range = h[0];
  unsigned long long ch = 0;
  memcpy(&ch, &h[2], sizeof(ch));
  ctx->sp_chunk_bound = (int64_t)ch;
  ctx->sp_bounds_ok = true;
  return ZZZ_OK;
}

// (Re)build the operator stream from the CSR values (MatAssemblyEnd).  The natural-order stream is packed by ONE
// kernel without waiting for the host; how many chunks it took (the allocator's counter) travels to pinned memory
// behind it and is looked at when the first product is launched (sellp_resolve).
int sell_update(zzz_ctx* ctx, bool structure)
{
  (void)structure;
  ctx->have_sell = ctx->sell_current = ctx->sp_pending = false;
  ctx->sp_dict_done = ctx->sp_dict_on = ctx->sp_sd_on = false; // (the values changed: the dictionaries are rebuilt at the stream's first use)
  ctx->sp_win_max = 0; // (set again by the long-row packer when most groups get an x window)
  ctx->sp_win_bytes = 0;
  const bool forced = (ctx->spmv_variant & 8) != 0 && !ctx->spmv_auto;
  if (ctx->sellp_mode == 0 || (!ctx->spmv_auto && !forced) || !ctx->vals.p)
    return ZZZ_OK;
  if (ctx->spmv_lpr_forced >= 0) // ZZZ_SPMV_LPR: an A/B knob of the CSR tile kernel's row phase
    return ZZZ_OK;
  if (ctx->nloc() >= ((int64_t)1 << 29)) // 32-bit byte offsets of the x gather
    return ZZZ_OK;
  if (!ctx->sp_bounds_ok)
  {
    int rc = sellp_pattern_bounds(ctx);
    if (rc)
      return rc;
  }
  hipStream_t s = ctx->stream;
  const int nrows = (int)ctx->nrows;
  const int64_t nsl = ctx->nslices;
  ZZZ_HIP(ctx, ctx->sp_desc.alloc(2 * (size_t)nsl + 2));
  ctx->sp_forced = forced;
  if (ctx->sellp_mode == 3)
  {
    int64_t t1 = 0;
    int rc = sp_build_sorted(ctx, &t1);
    if (!rc)
      rc = sp_fill_sorted(ctx, t1);
    if (rc)
      return rc;
    ctx->have_sell = ctx->sell_current = true;
    return ZZZ_OK;
  }
  // natural order.  Not worth packing when even the pattern bound is hopeless (rows of very different lengths)
  const double full = (double)ctx->nnz + 64.0 * 512.0;
  const bool long_rows = (double)ctx->nnz >= 100.0 * (double)ctx->nrows;
  const bool always = ctx->sellp_mode == 2 || forced;
  if (ctx->sp_chunk_bound >= INT32_MAX)
    return ZZZ_OK;
  size_t lds = (size_t)((ctx->sp_max_range + 63) & ~63) * 12;
  int waves = 4;
  while (waves > 1 && lds * waves > 64 * 1024)
    waves >>= 1;
  // (one wavefront per CU with up to 160 KB of staging was tried for the long rows of P3: the packing got 3 ms
  // faster, but its allocation order made the product 1.3 % slower: a net loss)
  // ZZZ_SELLP_SYNC=1: take the long-row path (count / compact / pack, synchronous) whatever the row lengths (tests)
  const bool lds_fits = lds * waves <= 160 * 1024 - 64 && (waves >= 2 || lds <= 64 * 1024) && !getenv("ZZZ_SELLP_SYNC");
  if (!lds_fits || (!always && (double)ctx->sp_chunk_bound * 512.0 > 2.2 * full))
  {
    // Synchronous builds (count, scan, read-back, fill): rows too long for the LDS staging of the one-pass packer
    // (then each lane streams a long contiguous row anyway), or a pattern whose natural-order stream is hopeless.
    int64_t t0 = -1, t1 = -1;
    int rc = ZZZ_OK;
    if (!lds_fits)
    {
      rc = sp_build_sorted(ctx, &t0, false);
      if (rc)
        return rc;
      if (always || (cost_stream(t0, false) <= cost_tile(ctx) && (double)t0 * 512.0 <= 1.5 * full))
      {
        rc = sp_fill_sorted(ctx, t0, false);
        if (!rc)
          ctx->have_sell = ctx->sell_current = true;
        return rc;
      }
    }
    if (always || !long_rows)
      return ZZZ_OK;
    rc = sp_build_sorted(ctx, &t1);
    if (rc)
      return rc;
    if (cost_stream(t1, true) > cost_tile(ctx))
      return ZZZ_OK;
    rc = sp_fill_sorted(ctx, t1);
    if (rc)
      return rc;
    ctx->have_sell = ctx->sell_current = true;
    return ZZZ_OK;
  }
  int rc = sp_alloc_stream(ctx, ctx->sp_chunk_bound);
  if (rc)
    return rc;
  uint8_t* gflag = nullptr;
  if (ctx->n_ghost > 0)
  {
    ZZZ_HIP(ctx, ctx->sp_gflag.alloc((size_t)nsl));
    gflag = ctx->sp_gflag.p;
  }
  // x windows (k_sp_pack<true>): block size 3 -- 45 entries per row at 15 places of x, which a group of 256 rows
  // shares almost completely (1400 doubles in 3-7 segments); needs the four wavefronts of a workgroup on one group and
  // room for the bitmap beside the parked entries.  ZZZ_SELLP_WIN: doubles of LDS per workgroup of the product
  // (default 2048 = 16 KiB: eight workgroups per CU as before; 0: off)
  // ... and a stream that comes from HBM: where the loop is cache-resident (the 8-GPU per-rank share of C4: 22 M
  // nonzeros) the two barriers and the window load per group cost more than the gathers they replace (product 34.5 ->
  // 37.9 us), so without the knob windows are built for matrices beyond ~300 MB of values only
  const char* win_env = getenv("ZZZ_SELLP_WIN");
  // (at most 6136 doubles: the product's dynamic LDS -- the window AND the value dictionary's copy of up to
  // SP_DICT_LDS_ENTRIES doubles -- plus its static words must stay inside the 64 KiB a launch gets without raising the
  // kernel's limit; a larger window would fail at the first product, after the stream was packed)
  const int win_knob = win_env ? std::min(atoi(win_env), (64 * 1024 - SP_DICT_LDS_ENTRIES * 8 - 64) / 8) : 2048;
  const bool winb = ctx->bs == 3 && win_knob >= 256 && (win_env || (double)ctx->nnz * 8.0 > 300.0e6);
  if (lds * waves > 64 * 1024 && !ctx->sp_lds_attr)
  {
    ZZZ_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_sp_pack<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     160 * 1024 - 64)); // the kernel's 32 B of static LDS count too
    ZZZ_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_sp_pack<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     160 * 1024 - 64));
    ctx->sp_lds_attr = true;
  }
  ZZZ_HIP(ctx, hipMemsetAsync(ctx->sp_counter.p, 0, 4 * sizeof(int), s));
  ZZZ_HIP(ctx, hipMemsetAsync(ctx->sp_counter.p + 8, 0, 8 * sizeof(int), s)); // ([14]: "not for the pipelined product")
  const int cap = (ctx->sp_max_range + 63) & ~63;
  const int tcodes = ctx->sellp_tail | ((ctx->bs == 3 && ctx->sellp_periodic) ? 4 : 0) | ((ctx->bs == 1 && ctx->sellp_align) ? 16 : 0);
  // workgroups of the packer take slices in fours only when they have four wavefronts: the group of a slice is s >> 2
  // either way, and a workgroup of one or two wavefronts starts at a multiple of its size inside the group
  if (winb)
  {
    const int64_t ngroups = (nsl + 3) / 4;
    ZZZ_HIP(ctx, ctx->sp_win_info.alloc(2 * (size_t)ngroups));
    ZZZ_HIP(ctx, ctx->sp_win_seg.alloc(2 * (size_t)ngroups * SP_WIN_NSEG));
    hipLaunchKernelGGL(k_sp_windows, dim3((unsigned)std::min<int64_t>(ngroups, 256 * 8)), dim3(256), 0, s, ctx->rowptr.p, ctx->cols.p,
                       ctx->vals.p, nrows, ngroups, ctx->sellp_drop ? 1 : 0, win_knob, reinterpret_cast<int2*>(ctx->sp_win_info.p),
                       reinterpret_cast<int2*>(ctx->sp_win_seg.p), reinterpret_cast<unsigned long long*>(ctx->sp_counter.p + 10));
    hipLaunchKernelGGL(k_sp_pack<true>, dim3(grid_cap(nsl, waves, 256 * 12)), dim3(64 * waves), lds * waves, s, ctx->rowptr.p,
                       ctx->cols.p, ctx->vals.p, nrows, nsl, ctx->sellp_drop ? 1 : 0, cap, ctx->sp_counter.p,
                       reinterpret_cast<int2*>(ctx->sp_desc.p), ctx->sp_vals.p, ctx->sp_codes16.p, ctx->sp_codes32.p, ctx->sp_meta.p,
                       gflag, tcodes, reinterpret_cast<const int2*>(ctx->sp_win_info.p), reinterpret_cast<const int2*>(ctx->sp_win_seg.p),
                       ctx->sp_smode.p);
    ctx->sp_win_max = win_knob; // the product reads win_info per group; a group without a window gathers from memory
  }
  else
    hipLaunchKernelGGL(k_sp_pack<false>, dim3(grid_cap(nsl, waves, 256 * 12)), dim3(64 * waves), lds * waves, s, ctx->rowptr.p,
                       ctx->cols.p, ctx->vals.p, nrows, nsl, ctx->sellp_drop ? 1 : 0, cap, ctx->sp_counter.p,
                       reinterpret_cast<int2*>(ctx->sp_desc.p), ctx->sp_vals.p, ctx->sp_codes16.p, ctx->sp_codes32.p, ctx->sp_meta.p,
                       gflag, tcodes, (const int2*)nullptr, (const int2*)nullptr, ctx->sp_smode.p);
  ZZZ_HIP(ctx, hipGetLastError());
  if (!ctx->sp_event)
    ZZZ_HIP(ctx, hipEventCreateWithFlags(&ctx->sp_event, hipEventDisableTiming));
  int32_t* tot = reinterpret_cast<int32_t*>(ctx->h_state + 5); // pinned
  ZZZ_HIP(ctx, hipMemcpyAsync(tot, ctx->sp_counter.p, 16 * sizeof(int32_t), hipMemcpyDeviceToHost, s));
  ZZZ_HIP(ctx, hipEventRecord(ctx->sp_event, s));
  ctx->sp_sorted = false;
  ctx->sp_pending = true;
  ctx->have_sell = ctx->sell_current = true; // provisional until sellp_resolve has seen the size
  if (gflag)
    return sellp_resolve(ctx);
  return ZZZ_OK;
}

// The packed stream's size is known: keep it, or fall back (sorted form for long rows, else the CSR tile kernel).
int sellp_resolve(zzz_ctx* ctx)
{
  if (!ctx->sp_pending)
    return ZZZ_OK;
  ctx->sp_pending = false;
  ZZZ_HIP(ctx, hipEventSynchronize(ctx->sp_event));
  const int32_t* hc = reinterpret_cast<int32_t*>(ctx->h_state + 5);
  const int64_t t0 = hc[0];
  unsigned long long kept = 0;
  memcpy(&kept, hc + 2, sizeof(kept));
  ctx->sp_kept = (int64_t)kept;
  unsigned long long hb = 0;
  memcpy(&hb, hc + 8, sizeof(hb));
  ctx->sp_bytes = (int64_t)hb;
  ctx->sp_chunks = t0;
  unsigned long long hw = 0;
  memcpy(&hw, hc + 10, sizeof(hw));
  ctx->sp_win_bytes = ctx->sp_win_max > 0 ? (int64_t)hw * 8 : 0;
  ctx->sp_pipe_ok = hc[14] == 0;
  ctx->sp_one_chunk = hc[14] == 0 && hc[15] == 0;
  const double full = (double)ctx->nnz + 64.0 * 512.0;
  const bool always = ctx->sellp_mode == 2 || ctx->sp_forced;
  // Natural row order unless its padding makes it slower than the alternatives: the length-sorted form (priced only
  // when the natural stream is padded by more than a third: it costs a synchronous build) or the CSR tile kernel.
  const double c_nat = (double)(ctx->sp_bytes + ctx->nslices * 8) / 5.0, c_tile = cost_tile(ctx);
  const bool padded = (double)t0 * 512.0 > 1.33 * (double)ctx->sp_kept + 64.0 * 512.0;
  (void)full;
  if (always || (c_nat <= c_tile && !padded))
    return ctx->n_ghost > 0 ? sp_group_split(ctx, ctx->sp_gflag.p) : ZZZ_OK;
  int64_t t1 = 0;
  int rc = sp_build_sorted(ctx, &t1);
  if (rc)
    return rc;
  const double c_srt = cost_stream(t1, true);
  if (c_nat <= c_tile && c_nat <= c_srt)
    return ctx->n_ghost > 0 ? sp_group_split(ctx, ctx->sp_gflag.p) : ZZZ_OK;
  ctx->have_sell = ctx->sell_current = false;
  if (c_srt > c_tile)
    return ZZZ_OK;
  rc = sp_fill_sorted(ctx, t1);
  if (rc)
    return rc;
  ctx->have_sell = ctx->sell_current = true;
  return ZZZ_OK;
}

// The value dictionary of the finished stream (see zzz_internal.h): distinct values into a hash set, numbered, every
// value of the stream replaced by its code.  Synchronous (once per assembly, at the first use of the stream): 1-2 ms at
// 10 M rows.  More than 65 535 distinct values, or ZZZ_SELLP_DICT=0: the stream keeps being read as values.
static int sp_dict_build(zzz_ctx* ctx)
{
  ctx->sp_dict_done = true;
  ctx->sp_dict_on = false;
  ctx->sp_dict_n = 0;
  // ZZZ_SELLP_DICT: 0 never, 2 always (tests at small sizes), 1: for streams of more than 48 MB of values -- below that the
  // whole loop sits in the Infinity Cache, bytes are not what the product waits for, and building the dictionary (three
  // passes over the stream and a synchronisation, ~0.4 ms at 500 k rows) costs more than a solve gains
  if (!ctx->sellp_dict || ctx->sp_chunks <= 0 || (ctx->sellp_dict == 1 && ctx->sp_bytes < 48ll << 20))
    return ZZZ_OK;
  hipStream_t s = ctx->stream;
  const int64_t nsl = ctx->nslices;
  ZZZ_HIP(ctx, ctx->sp_dict_table.alloc((size_t)1 << SP_DICT_BITS));
  ZZZ_HIP(ctx, ctx->sp_dict_slot.alloc((size_t)1 << SP_DICT_BITS));
  ZZZ_HIP(ctx, ctx->sp_dict.alloc((size_t)SP_DICT_MAX + 1));
  ZZZ_HIP(ctx, ctx->sp_vcode.alloc((size_t)ctx->sp_chunks * 512));
  DevBuf<int32_t>& info = ctx->sp_dict_info;
  ZZZ_HIP(ctx, info.reserve(8));
  ZZZ_HIP(ctx, hipMemsetAsync(info.p, 0, 8 * sizeof(int32_t), s));
  ZZZ_HIP(ctx, hipMemsetAsync(ctx->sp_dict_table.p, 0xff, sizeof(unsigned long long) << SP_DICT_BITS, s));
  const int2* desc = reinterpret_cast<const int2*>(ctx->sp_desc.p);
  const unsigned grid = (unsigned)std::min<int64_t>((nsl + 3) / 4, 256 * 16);
  // (the search stops at the first value beyond what will be used: the LDS copy's capacity, unless the memory form is forced)
  const int limit = ctx->sellp_dict == 2 ? SP_DICT_MAX : SP_DICT_LDS_MAX - 1;
  if (ctx->sp_sorted)
    hipLaunchKernelGGL(k_sp_dict_insert<true>, dim3(grid), dim3(256), 0, s, desc, ctx->sp_perm.p, ctx->sp_vals.p, (int)ctx->nrows, nsl,
                       ctx->sp_dict_table.p, info.p, limit);
  else
    hipLaunchKernelGGL(k_sp_dict_insert<false>, dim3(grid), dim3(256), 0, s, desc, (const int32_t*)nullptr, ctx->sp_vals.p,
                       (int)ctx->nrows, nsl, ctx->sp_dict_table.p, info.p, limit);
  hipLaunchKernelGGL(k_sp_dict_number, dim3(1), dim3(1024), 0, s, ctx->sp_dict_table.p, ctx->sp_dict_slot.p, ctx->sp_dict.p, info.p,
                     SP_DICT_LDS_MAX, ctx->sellp_dict == 2 ? 1 : 0);
  if (ctx->sp_sorted)
    hipLaunchKernelGGL(k_sp_dict_encode<true>, dim3(grid), dim3(256), 0, s, desc, ctx->sp_perm.p, ctx->sp_vals.p, ctx->sp_meta.p,
                       (int)ctx->nrows, nsl, ctx->sp_dict_table.p, ctx->sp_dict_slot.p, ctx->sp_vcode.p, info.p);
  else
    hipLaunchKernelGGL(k_sp_dict_encode<false>, dim3(grid), dim3(256), 0, s, desc, (const int32_t*)nullptr, ctx->sp_vals.p,
                       ctx->sp_meta.p, (int)ctx->nrows, nsl, ctx->sp_dict_table.p, ctx->sp_dict_slot.p, ctx->sp_vcode.p, info.p);
  ZZZ_HIP(ctx, hipGetLastError());
  int32_t h[8];
  ZZZ_HIP(ctx, hipMemcpyAsync(h, info.p, sizeof(h), hipMemcpyDeviceToHost, s));
  ZZZ_HIP(ctx, hipStreamSynchronize(s));
  if (h[1] || h[2] <= 0 || h[2] > SP_DICT_MAX + 1)
    return ZZZ_OK;
  // A dictionary too large for the LDS copy is gathered from memory: 2.3x fewer bytes at P3 6.2 M dofs (8 270 values) and the
  // same 0.61-0.63 ms per product -- the gathers, not the bytes, are what the kernel waits for -- so that form is not used
  // unless ZZZ_SELLP_DICT=2 asks for it (tests)
  if (h[2] > SP_DICT_LDS_MAX && ctx->sellp_dict != 2)
    return ZZZ_OK;
  unsigned long long bytes = 0;
  memcpy(&bytes, h + 4, sizeof(bytes));
  ctx->sp_dict_n = h[2];
  ctx->sp_dict_bytes = (int64_t)bytes + (int64_t)h[2] * 8;
  ctx->sp_dict_on = true;
  return ZZZ_OK;
}

// Per-slice dictionaries (k_sp_sd_build) for streams whose global dictionary does not fit LDS: long rows (P3).  Kept when
// they take the stream below 60 % of its bytes.  ZZZ_SELLP_DICT: 0 none of this, 3 slice dictionaries whenever they apply.
static int sp_sd_build(zzz_ctx* ctx)
{
  ctx->sp_sd_on = ctx->sp_sd_all = false;
  if (!ctx->sellp_dict || ctx->sp_dict_on || ctx->sp_chunks <= 0 || ctx->sp_win_max > 0)
    return ZZZ_OK;
  if (ctx->sellp_dict != 3 && (ctx->sp_bytes < 48ll << 20 || ctx->sp_chunks < 4 * ctx->nslices))
    return ZZZ_OK; // (small streams: bytes do not matter; short rows -- P1: the table would cost as much as it saves)
  hipStream_t s = ctx->stream;
  const int64_t nsl = ctx->nslices;
  ZZZ_HIP(ctx, ctx->sp_vcode8.alloc((size_t)ctx->sp_chunks * 512));
  ZZZ_HIP(ctx, ctx->sp_sd_info.alloc((size_t)nsl + 1));
  ZZZ_HIP(ctx, ctx->sp_sd_off.alloc((size_t)nsl + 1));
  ZZZ_HIP(ctx, ctx->sp_dict_info.reserve(8));
  ZZZ_HIP(ctx, hipMemsetAsync(ctx->sp_dict_info.p, 0, 8 * sizeof(int32_t), s));
  ZZZ_HIP(ctx, hipMemsetAsync(ctx->sp_sd_info.p + nsl, 0, sizeof(int32_t), s)); // closes the scan
  const int2* desc = reinterpret_cast<const int2*>(ctx->sp_desc.p);
  const unsigned grid = (unsigned)std::min<int64_t>((nsl + 1) / 2, 256 * 8);
  unsigned long long* bytes = reinterpret_cast<unsigned long long*>(ctx->sp_dict_info.p + 4);
  // first pass: the tables' sizes; scan: where each starts; second pass: tables and codes
  if (ctx->sp_sorted)
    hipLaunchKernelGGL((k_sp_sd_build<true, true>), dim3(grid), dim3(128), 0, s, desc, ctx->sp_perm.p, ctx->sp_vals.p, ctx->sp_meta.p,
                       (int)ctx->nrows, nsl, (uint16_t*)nullptr, (double*)nullptr, (const int64_t*)nullptr, ctx->sp_sd_info.p, bytes);
  else
    hipLaunchKernelGGL((k_sp_sd_build<false, true>), dim3(grid), dim3(128), 0, s, desc, (const int32_t*)nullptr, ctx->sp_vals.p,
                       ctx->sp_meta.p, (int)ctx->nrows, nsl, (uint16_t*)nullptr, (double*)nullptr, (const int64_t*)nullptr,
                       ctx->sp_sd_info.p, bytes);
  {
    const auto even = rocprim::make_transform_iterator(ctx->sp_sd_info.p, Even2{}); // (tables start at even entries: 16-B aligned)
    size_t tb = 0;
    ZZZ_HIP(ctx, rocprim::exclusive_scan(nullptr, tb, even, ctx->sp_sd_off.p, (int64_t)0, (size_t)nsl + 1, rocprim::plus<int64_t>(), s));
    ZZZ_HIP(ctx, ctx->scr_tmp.alloc(tb));
    ZZZ_HIP(ctx, rocprim::exclusive_scan(ctx->scr_tmp.p, tb, even, ctx->sp_sd_off.p, (int64_t)0, (size_t)nsl + 1, rocprim::plus<int64_t>(), s));
  }
  int64_t total = 0;
  ZZZ_HIP(ctx, hipMemcpyAsync(&total, ctx->sp_sd_off.p + nsl, sizeof(total), hipMemcpyDeviceToHost, s));
  ZZZ_HIP(ctx, hipStreamSynchronize(s));
  ZZZ_HIP(ctx, ctx->sp_sd_vals.alloc((size_t)total + 2));
  if (ctx->sp_sorted)
    hipLaunchKernelGGL((k_sp_sd_build<true, false>), dim3(grid), dim3(128), 0, s, desc, ctx->sp_perm.p, ctx->sp_vals.p, ctx->sp_meta.p,
                       (int)ctx->nrows, nsl, ctx->sp_vcode8.p, ctx->sp_sd_vals.p, ctx->sp_sd_off.p, ctx->sp_sd_info.p, bytes);
  else
    hipLaunchKernelGGL((k_sp_sd_build<false, false>), dim3(grid), dim3(128), 0, s, desc, (const int32_t*)nullptr, ctx->sp_vals.p,
                       ctx->sp_meta.p, (int)ctx->nrows, nsl, ctx->sp_vcode8.p, ctx->sp_sd_vals.p, ctx->sp_sd_off.p, ctx->sp_sd_info.p,
                       bytes);
  ZZZ_HIP(ctx, hipGetLastError());
  int32_t h[8];
  ZZZ_HIP(ctx, hipMemcpyAsync(h, ctx->sp_dict_info.p, sizeof(h), hipMemcpyDeviceToHost, s));
  ZZZ_HIP(ctx, hipStreamSynchronize(s));
  unsigned long long b = 0;
  memcpy(&b, h + 4, sizeof(b));
  if (ctx->sellp_dict != 3 && (double)b > 0.6 * (double)ctx->sp_bytes)
    return ZZZ_OK;
  ctx->sp_sd_bytes = (int64_t)b;
  ctx->sp_sd_on = true;
  ctx->sp_sd_all = h[6] == 0;
  return ZZZ_OK;
}

bool sellp_active(zzz_ctx* ctx)
{
  if (ctx->sp_pending)
    (void)sellp_resolve(ctx);
  if (!ctx->have_sell || !ctx->sell_current)
    return false;
  if (!(ctx->spmv_auto || (ctx->spmv_variant & 8) != 0))
    return false;
  if (!ctx->sp_dict_done)
  {
    // a failed build (a hipMalloc that did not fit) leaves the stream on doubles -- valid -- but must not leave its partial
    // allocations or HIP's sticky last error behind (the next hipGetLastError after a product launch would report it)
    if (sp_dict_build(ctx) != ZZZ_OK)
    {
      ctx->sp_dict_on = false;
      (void)hipGetLastError();
    }
    if (!ctx->sp_dict_on)
    {
      ctx->sp_vcode.release();
      ctx->sp_dict_table.release();
      ctx->sp_dict_slot.release();
    }
    if (sp_sd_build(ctx) != ZZZ_OK)
    {
      ctx->sp_sd_on = ctx->sp_sd_all = false;
      (void)hipGetLastError();
    }
    if (!ctx->sp_sd_on)
    {
      ctx->sp_vcode8.release();
      ctx->sp_sd_vals.release();
      ctx->sp_sd_info.release();
      ctx->sp_sd_off.release();
    }
    ctx->sp_pairs_ok = false;
    if (ctx->sp_one_chunk && ctx->sp_dict_on && ctx->bs == 1 && !ctx->sp_sorted)
      ctx->sp_pairs_ok = sellp_pairs_build(ctx) == ZZZ_OK;
  }
  return true;
}

// bytes one product reads from the stream (values or value codes + dictionary, column codes, bases; int32 chunks are not
// counted separately)
int64_t sellp_stream_bytes(const zzz_ctx* ctx)
{
  return (ctx->sp_sd_on ? ctx->sp_sd_bytes : (ctx->sp_dict_on ? ctx->sp_dict_bytes : ctx->sp_bytes)) + ctx->nslices * 8; // (x windows: sp_win_bytes, reported apart)
}

// Load policy of the product: non-temporal stream loads when one CG iteration (the stream and six vectors) cannot stay in
// the 256 MiB Infinity Cache anyway -- then the stream should not push the vectors' lines out; plain loads when it is
// resident.  (Until the value dictionary the stream alone decided, at 300 MB; a coded stream of 167 MB beside 480 MB of
// vectors reads 4 % faster non-temporally: 0.091 -> 0.087 ms at C2.)
static bool sp_stream_nt(const zzz_ctx* ctx)
{
  return (double)sellp_stream_bytes(ctx) + 48.0 * (double)(ctx->n_owned + ctx->n_ghost) * ctx->bs > 200.0e6;
}

static int sp_grid(const zzz_ctx* ctx, int64_t ngroups, bool sr)
{
  // persistent workgroups: as many as are resident at once (a second round of a grid that is not would run on part of the chip)
  const int pw = sellp_pipe_wgs(ctx, sr);
  int64_t gs = 256 * (pw ? pw : 8); // (1024 or 1536 workgroups at the per-rank size: no faster)
#ifdef ZZZ_EXPERIMENTS
  if (const char* e = getenv("ZZZ_SP_WGS_PER_CU")) // how the product's time depends on the wavefronts in flight
    gs = 256 * std::max(1, std::min(8, atoi(e)));
#endif
  const int64_t need = (ngroups + 7) / 8 * 8;
  if (gs > need)
    gs = need;
  if (gs < 8)
    gs = 8;
  return (int)gs;
}

template <bool DOT>
static void launch_one(zzz_ctx* ctx, int grid, const double* x, double* y, double* partials, const int* stop,
                       const int32_t* group_list, int64_t nlist, const double* rvec, int nn_is_rr,
                       const TailArgs& tail = TailArgs(), const ChebEpi* epi = nullptr)
{
  // load policy by stream size, as for the tile kernel: a stream that stays in the 256 MiB Infinity Cache from
  // one CG iteration to the next is read with plain loads, a larger one with non-temporal loads
  bool nt = sp_stream_nt(ctx);
  if (!ctx->spmv_auto)
    nt = (ctx->spmv_variant & 1) != 0;
  if (!epi && !tail.parts && launch_sellp_pipe(ctx, DOT, nt, grid, x, y, partials, stop, group_list, nlist, rvec, nn_is_rr))
    return;
  const int2* off = reinterpret_cast<const int2*>(ctx->sp_desc.p);
  const int2* winfo = reinterpret_cast<const int2*>(ctx->sp_win_info.p);
  const int2* wseg = reinterpret_cast<const int2*>(ctx->sp_win_seg.p);
#define ZZZ_SP_GO6(NT, PERM, WIN, LDSB, DICT)                                                                          \
  do                                                                                                                   \
  {                                                                                                                    \
    const uint16_t* VC_ = (DICT) == 3 ? ctx->sp_vcode8.p : ctx->sp_vcode.p;         \
    const double* DG_ = (DICT) == 3 ? ctx->sp_sd_vals.p : ctx->sp_dict.p;                                               \
    if (epi)                                                                                                           \
      hipLaunchKernelGGL((spmv_sellp_kernel<DOT, NT, PERM, true, WIN, DICT>), dim3(grid), dim3(SP_BLOCK), LDSB,         \
                         ctx->stream, off, ctx->sp_vals.p, ctx->sp_codes16.p, ctx->sp_codes32.p, ctx->sp_meta.p,           \
                         ctx->sp_perm.p, x, y, (int)ctx->nrows, ctx->nslices, partials, stop, group_list, nlist, rvec,     \
                         SPMV_PSTRIDE, nn_is_rr, TailArgs(), *epi, winfo, wseg, VC_, DG_, ctx->sp_dict_n,                  \
                         ctx->sp_sd_info.p, ctx->sp_sd_off.p);                                                         \
    else                                                                                                               \
      hipLaunchKernelGGL((spmv_sellp_kernel<DOT, NT, PERM, false, WIN, DICT>), dim3(grid), dim3(SP_BLOCK), LDSB,        \
                         ctx->stream, off, ctx->sp_vals.p, ctx->sp_codes16.p, ctx->sp_codes32.p, ctx->sp_meta.p,           \
                         ctx->sp_perm.p, x, y, (int)ctx->nrows, ctx->nslices, partials, stop, group_list, nlist, rvec,     \
                         SPMV_PSTRIDE, nn_is_rr, tail, ChebEpi(), winfo, wseg, VC_, DG_, ctx->sp_dict_n,                   \
                         ctx->sp_sd_info.p, ctx->sp_sd_off.p);                                                         \
  } while (0)
#define ZZZ_SP_GO5(NT, PERM, WIN, LDSB)                                                                                \
  do                                                                                                                   \
  {                                                                                                                    \
    if (ctx->sp_sd_on && !(WIN))                                                                                       \
      ZZZ_SP_GO6(NT, PERM, false, (size_t)4 * SD_MAX * sizeof(double), 3);                                             \
    else if (ctx->sp_dict_on && ctx->sp_dict_n <= SP_DICT_LDS_MAX)                                                     \
      ZZZ_SP_GO6(NT, PERM, WIN, (LDSB) + (size_t)((ctx->sp_dict_n + 1) & ~1) * sizeof(double), 2);                     \
    else if (ctx->sp_dict_on)                                                                                          \
      ZZZ_SP_GO6(NT, PERM, WIN, LDSB, 1);                                                                              \
    else                                                                                                               \
      ZZZ_SP_GO6(NT, PERM, WIN, LDSB, 0);                                                                              \
  } while (0)
#define ZZZ_SP_GO(NT, PERM) ZZZ_SP_GO5(NT, PERM, false, 0)
  if (ctx->sp_win_max > 0 && !ctx->sp_sorted)
  {
    // windowed groups: their codes index the LDS window the kernel loads per group
    const size_t ldsb = (size_t)ctx->sp_win_max * sizeof(double);
    if (nt)
      ZZZ_SP_GO5(true, false, true, ldsb);
    else
      ZZZ_SP_GO5(false, false, true, ldsb);
  }
  else if (ctx->sp_sorted)
  {
    if (nt)
      ZZZ_SP_GO(true, true);
    else
      ZZZ_SP_GO(false, true);
  }
  else
  {
    if (nt)
      ZZZ_SP_GO(true, false);
    else
      ZZZ_SP_GO(false, false);
  }
#undef ZZZ_SP_GO
#undef ZZZ_SP_GO5
#undef ZZZ_SP_GO6
}

int launch_sellp(zzz_ctx* ctx, const double* x, double* y, double* partials, int* npartials, const double* rvec, int nn_is_rr,
                 const ChebEpi* epi)
{
  const int* stop = partials ? reinterpret_cast<const int*>(ctx->state.p) : nullptr; // CgState::converged
  const int gs = sp_grid(ctx, (ctx->nslices + 3) / 4, (partials && rvec));
#ifdef ZZZ_EXPERIMENTS
  const char* e = ctx->timing_only ? getenv("ZZZ_EXP_WIN") : nullptr; // timing probe, wrong results by construction (see
  if (e)                                                               // the kernel): inside zzz_spmv_time only
  {
    const int wlen = atoi(e) & ~1;
    if (wlen >= 256 && wlen <= 8192 && !ctx->sp_sorted && !epi && wlen < ctx->nrows && ctx->sp_win_max == 0)
    {
      const int lds_slots = getenv("ZZZ_EXP_WIN_SLOTS") ? atoi(getenv("ZZZ_EXP_WIN_SLOTS")) : 8;
      const int per_cu = std::max(1, std::min(8, (int)(160 * 1024 / ((size_t)wlen * 8 + 512))));
      const int grid = std::min(gs, 256 * per_cu);
      const bool nt = sp_stream_nt(ctx);
      const int2* off = reinterpret_cast<const int2*>(ctx->sp_desc.p);
      if (nt)
        hipLaunchKernelGGL(spmv_sellp_win_probe_kernel<true>, dim3(grid), dim3(SP_BLOCK), (size_t)wlen * 8, ctx->stream, off,
                           ctx->sp_vals.p, ctx->sp_codes16.p, ctx->sp_codes32.p, ctx->sp_meta.p, x, y, (int)ctx->nrows,
                           ctx->nslices, partials, wlen, lds_slots);
      else
        hipLaunchKernelGGL(spmv_sellp_win_probe_kernel<false>, dim3(grid), dim3(SP_BLOCK), (size_t)wlen * 8, ctx->stream, off,
                           ctx->sp_vals.p, ctx->sp_codes16.p, ctx->sp_codes32.p, ctx->sp_meta.p, x, y, (int)ctx->nrows,
                           ctx->nslices, partials, wlen, lds_slots);
      if (npartials)
        *npartials = grid;
      ZZZ_HIP(ctx, hipGetLastError());
      return ZZZ_OK;
    }
  }
#endif
  if (partials)
  {
    TailArgs T;
    if (ctx->tail_armed)
    {
      T = ctx->tail;
      T.expected = gs;
      T.base = 0;
      ctx->tail_armed = false;
      ctx->tail_used = true;
    }
    launch_one<true>(ctx, gs, x, y, partials, stop, nullptr, 0, rvec, nn_is_rr, T, epi);
    if (npartials)
      *npartials = gs;
  }
  else
    launch_one<false>(ctx, gs, x, y, nullptr, stop, nullptr, 0, nullptr, 0, TailArgs(), epi);
  ZZZ_HIP(ctx, hipGetLastError());
  return ZZZ_OK;
}

// Partitioned matrix: forward halo of x overlapped with the groups that reference no ghost column
// (scheme and the 7-of-8 workgroup slots: launch_spmv_overlapped in zzz_spmv.hip)
int launch_sellp_overlapped(zzz_ctx* ctx, double* x, double* y, double* partials, int* npartials, const double* rvec,
                            int nn_is_rr, const ChebEpi* epi)
{
  const int* stop = partials ? reinterpret_cast<const int*>(ctx->state.p) : nullptr;
  const int64_t gi = ctx->n_groups_interior, gb = ctx->n_groups_boundary;
  int g_in = gi ? sp_grid(ctx, gi, (partials && rvec)) : 0;
  const int pw = sellp_pipe_wgs(ctx, partials && rvec);
  const int room = 256 * ((pw ? pw : 8) - 1); // (one workgroup slot per CU left to the exchange's kernel)
  if (g_in > room && ctx->nneigh > 0)
    g_in = room;
  const int g_bd = gb ? sp_grid(ctx, gb, (partials && rvec)) : 0;
  if (partials && (size_t)(g_in + g_bd) > (size_t)SPMV_PSTRIDE)
    return fail(ctx, ZZZ_ERR_ARG, "partials buffer too small");
  TailArgs Ti, Tb;
  if (partials && ctx->tail_armed)
  {
    // one ticket over both launches: the workgroup that arrives last (in the boundary launch, or in the interior
    // one when no group touches a ghost column) finishes the reduction
    Ti = ctx->tail;
    Ti.expected = g_in + g_bd;
    Ti.base = 0;
    Tb = Ti;
    Tb.base = g_in;
    ctx->tail_armed = false;
    ctx->tail_used = true;
  }
  int rc = comm_halo_begin(ctx, x);
  if (rc)
    return rc;
  if (gi)
  {
    if (partials)
      launch_one<true>(ctx, g_in, x, y, partials, stop, ctx->groups_interior.p, gi, rvec, nn_is_rr, Ti, epi);
    else
      launch_one<false>(ctx, g_in, x, y, nullptr, stop, ctx->groups_interior.p, gi, nullptr, 0, TailArgs(), epi);
  }
  rc = comm_halo_end(ctx);
  if (rc)
    return rc;
  if (gb)
  {
    if (partials)
      launch_one<true>(ctx, g_bd, x, y, partials + g_in, stop, ctx->groups_boundary.p, gb, rvec, nn_is_rr, Tb, epi);
    else
      launch_one<false>(ctx, g_bd, x, y, nullptr, stop, ctx->groups_boundary.p, gb, nullptr, 0, TailArgs(), epi);
  }
  if (npartials)
    *npartials = g_in + g_bd;
  ZZZ_HIP(ctx, hipGetLastError());
  return ZZZ_OK;
}
#ifdef ZZZ_EXPERIMENTS
// the fused product + direction kernel on the whole matrix, or (partitioned matrix) interior groups, halo of z,
// boundary groups.  Partials of <p,w>: interior workgroups first.
int launch_sellp_dir(zzz_ctx* ctx, double* z, const double* p_old, double* p_new, double* xsol, double* y, double* partials,
                     int* npartials, int it, const CgParams& P, const double* pa, const double* pb, int np, bool overlap)
{
  const bool nt = ctx->spmv_auto ? sp_stream_nt(ctx) : (ctx->spmv_variant & 1) != 0;
  const int2* off = reinterpret_cast<const int2*>(ctx->sp_desc.p);
  const int ncols = (int)ctx->nloc();
  auto go = [&](int grid, const int32_t* list, int64_t nlist, double* parts, int ghost) {
#define ZZZ_SPD_GO(NT, PERM)                                                                                           \
  hipLaunchKernelGGL((spmv_sellp_dir_kernel<NT, PERM>), dim3(grid), dim3(SP_BLOCK), 0, ctx->stream, off, ctx->sp_vals.p,  \
                     ctx->sp_codes16.p, ctx->sp_codes32.p, ctx->sp_meta.p, ctx->sp_perm.p, z, p_old, p_new, xsol, y,   \
                     (int)ctx->nrows, ncols, ctx->nslices, parts, list, nlist, ghost, ctx->state.p, ctx->beta_hist.p, \
                     ctx->dp_hist.p, ctx->alpha_hist.p, it, P, pa, pb, np)
    if (ctx->sp_sorted)
    {
      if (nt)
        ZZZ_SPD_GO(true, true);
      else
        ZZZ_SPD_GO(false, true);
    }
    else
    {
      if (nt)
        ZZZ_SPD_GO(true, false);
      else
        ZZZ_SPD_GO(false, false);
    }
#undef ZZZ_SPD_GO
  };
  if (overlap && ctx->have_group_split)
  {
    const int64_t gi = ctx->n_groups_interior, gb = ctx->n_groups_boundary;
    int g_in = gi ? sp_grid(ctx, gi, false) : 0;
    if (g_in > 256 * 7 && ctx->nneigh > 0)
      g_in = 256 * 7; // room for the exchange's kernel beside the persistent workgroups (launch_spmv_overlapped)
    const int g_bd = gb ? sp_grid(ctx, gb, false) : 8;
    if ((size_t)(g_in + g_bd) > (size_t)SPMV_PSTRIDE)
      return fail(ctx, ZZZ_ERR_ARG, "partials buffer too small");
    int rc = comm_halo_begin(ctx, z);
    if (rc)
      return rc;
    if (gi)
      go(g_in, ctx->groups_interior.p, gi, partials, 0);
    rc = comm_halo_end(ctx);
    if (rc)
      return rc;
    go(g_bd, ctx->groups_boundary.p, gb, partials + g_in, 1); // also with no boundary group: the ghost entries of p
    *npartials = g_in + g_bd;
  }
  else
  {
    if (ctx->comm)
    {
      int rc = comm_halo_forward(ctx, z);
      if (rc)
        return rc;
    }
    const int gs = sp_grid(ctx, (ctx->nslices + 3) / 4, false);
    go(gs, nullptr, 0, partials, ctx->n_ghost > 0 ? 1 : 0);
    *npartials = gs;
  }
  ZZZ_HIP(ctx, hipGetLastError());
  return ZZZ_OK;
}
#endif // ZZZ_EXPERIMENTS
} // namespace zzz

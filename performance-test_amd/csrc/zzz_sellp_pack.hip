// The CG operator stream, part 1 of 3: the PACKER (this file), the value dictionaries (zzz_sellp_dict.hip), the product
// (zzz_sellp.hip; streams of one-chunk slices: zzz_sellp_pipe.hip).
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

#ifdef ZZZ_EXPERIMENTS
// EXPERIMENT (ZZZ_SELLP=5, tools build): block size 3 with the rows of a slice all of ONE component -- slice 3 u + k holds
// rows 3 (64 u + lane) + k -- so that a slot's 64 columns are 64 consecutive blocks' (stride 3: 8-bit codes on one base, no
// per-lane decode of a periodic table).  Rides on the sorted form's machinery (perm, synchronous build).
__global__ __launch_bounds__(192) void k_sp_cm(const int32_t* __restrict__ rownnz, int nrows, int64_t nslices,
                                               int32_t* __restrict__ perm, int32_t* __restrict__ nch, uint8_t* __restrict__ wlast)
{
  __shared__ int mx[3];
  const int64_t u = blockIdx.x;
  const int t = threadIdx.x;
  if (t < 3)
    mx[t] = 0;
  __syncthreads();
  const int64_t r = u * 192 + t;
  const int k = t % 3, q = t / 3;
  atomicMax(&mx[k], r < nrows ? rownnz[r] : 0);
  perm[(3 * u + k) * 64 + q] = r < nrows ? (int32_t)r : -1;
  __syncthreads();
  if (t < 3 && 3 * u + t < nslices)
  {
    const int m = mx[t];
    nch[3 * u + t] = (m + 7) >> 3;
    wlast[3 * u + t] = (uint8_t)(m ? m - 8 * ((m - 1) >> 3) : 8);
  }
  if (u == 0 && t == 0)
    nch[nslices] = 0;
}
#endif

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
  const bool affine_ok = (tail_codes & 2) == 0; // knob ZZZ_SELLP_FORMS without bit 0 sets bit 1
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

// ---- host side ------------------------------------------------------------------------------------------
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
#ifdef ZZZ_EXPERIMENTS
  if (sorted && ctx->sellp_mode == 5)
    hipLaunchKernelGGL(k_sp_cm, dim3((unsigned)(nsl / 3)), dim3(192), 0, s, ctx->sp_rownnz.p, nrows, nsl, ctx->sp_perm.p, ctx->sp_nch.p,
                       ctx->sp_wlast.p);
  else
#endif
  if (sorted)
    hipLaunchKernelGGL(k_sp_sort, dim3((unsigned)nwin), dim3(SP_SIGMA), 0, s, ctx->sp_rownnz.p, nrows, nsl, ctx->sp_perm.p,
                       ctx->sp_nch.p, ctx->sp_wlast.p);
  else // natural row order, rows too long for the LDS staging of k_sp_pack
    hipLaunchKernelGGL(k_sp_slice_len, dim3(grid_cap(nsl + 1, 4, 8192)), dim3(256), 0, s, ctx->sp_rownnz.p, nrows, nsl,
                       ctx->sp_nch.p, ctx->sp_wlast.p);
  size_t tb = 0;
  ZZZ_HIP(ctx, rocprim::exclusive_scan(nullptr, tb, ctx->sp_nch.p, ctx->sp_chunk_off.p, 0, (size_t)nsl + 1,
                                       rocprim::plus<int32_t>(), s));
  ZZZ_HIP(ctx, ctx->scr_tmp.grow_keep(tb, ctx->retired));
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
      ZZZ_HIP(ctx, ctx->scr_tmp.grow_keep(tb, ctx->retired));
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
  ZZZ_HIP(ctx, ctx->scr_tmp.grow_keep(tb, ctx->retired));
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
  ctx->bk_on = false;
  ctx->bw_on = false;
  ctx->sp_generic = ctx->sp_special_tried = false;
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
  // The special forms first (block rows for block size 3, block windows for long scalar rows): where one of them serves this
  // matrix's products the generic stream is not packed at all (4.9 ms per assembly at C4, 2 ms at C5's per-GPU share).  Only
  // where no launch can need the generic kernel: automatic mode, no partition (the overlapped launches of a partitioned matrix
  // take their group split from the generic stream).  A launch that asks for the generic kernel after all (the tools build's
  // folded all-reduce) packs it then: sellp_need_generic.
  if (ctx->sellp_mode == 1 && !forced && ctx->sellp_early && ctx->n_ghost == 0 && !ctx->comm)
  {
    ctx->sp_sorted = false;
    const bool serves = sellp_special_build(ctx);
    ctx->sp_special_tried = true;
    if (serves)
    {
      ctx->sp_dict_done = true;
      ctx->sp_dict_n = 0;
      ctx->sp_pairs_ok = ctx->sp_one_chunk = ctx->sp_pipe_ok = false;
      ctx->sp_bytes = ctx->sp_chunks = ctx->sp_kept = 0;
      ctx->have_sell = ctx->sell_current = true;
      return ZZZ_OK;
    }
  }
  return sell_pack_generic(ctx);
}

int sellp_need_generic(zzz_ctx* ctx)
{
  if (ctx->sp_generic)
    return ZZZ_OK;
  int rc = sell_pack_generic(ctx);
  if (!rc)
    rc = sellp_resolve(ctx);
  if (!rc && !(ctx->sp_generic && ctx->have_sell && ctx->sell_current))
    return fail(ctx, ZZZ_ERR_LIMIT, "this launch needs the generic operator stream, which this matrix does not get");
  ctx->have_sell = ctx->sell_current = true;
  return rc;
}

int sell_pack_generic(zzz_ctx* ctx)
{
  const bool forced = (ctx->spmv_variant & 8) != 0 && !ctx->spmv_auto;
  ctx->sp_generic = true;
  ctx->have_sell = ctx->sell_current = false; // (until a stream is packed below)
  hipStream_t s = ctx->stream;
  const int nrows = (int)ctx->nrows;
  const int64_t nsl = ctx->nslices;
#ifdef ZZZ_EXPERIMENTS
  if (ctx->sellp_mode == 5 && ctx->bs == 3) // (experiment: slices of one component, k_sp_cm)
    ctx->nslices = 3 * ((ctx->nrows + 191) / 192);
  else if (ctx->sellp_mode == 5)
    ctx->sellp_mode = 1;
#endif
  ZZZ_HIP(ctx, ctx->sp_desc.alloc(2 * (size_t)ctx->nslices + 2));
  ctx->sp_forced = forced;
  if (ctx->sellp_mode == 3 || ctx->sellp_mode == 5)
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
  // ZZZ_SELLP=4: take the long-row path (count / compact / pack, synchronous) whatever the row lengths (tests)
  const bool lds_fits = lds * waves <= 160 * 1024 - 64 && (waves >= 2 || lds <= 64 * 1024) && !ctx->sellp_long_rows;
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
ZZZ_PRELOAD_TU(sellp_pack)
} // namespace zzz

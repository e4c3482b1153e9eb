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

#include "zzz_device.h"
#include "zzz_internal.h"
#include "zzz_cg_device.h"

#include <rocprim/rocprim.hpp>

namespace zzz
{
typedef double dbl2 __attribute__((ext_vector_type(2)));
typedef unsigned uint4v __attribute__((ext_vector_type(4)));
typedef int int4v __attribute__((ext_vector_type(4)));
constexpr int SP_BLOCK = 256;
constexpr int SP_SIGMA = 512; // sorting window (rows) of the sorted form: one workgroup

// tile index for (workgroup b, step i): XCD x = b % 8 owns items [x*T/8, (x+1)*T/8)   (as in zzz_spmv.hip)
__device__ inline int64_t sp_xcd_item(int64_t n, int b, int nb, int i)
{
  const int xcd = b & 7;
  const int64_t lo = n * xcd / 8, hi = n * (xcd + 1) / 8;
  const int wg_in_xcd = b >> 3, n_in_xcd = (nb + 7 - xcd) >> 3;
  const int64_t t = lo + wg_in_xcd + (int64_t)i * n_in_xcd;
  return t < hi ? t : -1;
}

__device__ inline int wave_min_i(int v)
{
#pragma unroll
  for (int o = 32; o > 0; o >>= 1)
    v = min(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ inline int wave_max_i(int v)
{
#pragma unroll
  for (int o = 32; o > 0; o >>= 1)
    v = max(v, __shfl_xor(v, o, 64));
  return v;
}

// ---- build ------------------------------------------------------------------------------------------
// entries of each row that the stream keeps
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
                                                      int32_t* __restrict__ nch)
{
  const int lane = threadIdx.x & 63;
  for (int64_t s = blockIdx.x * 4 + (threadIdx.x >> 6); s <= nslices; s += (int64_t)gridDim.x * 4)
  {
    const int64_t r = s * 64 + lane;
    const int m = wave_max_i((s < nslices && r < nrows) ? rownnz[r] : 0);
    if (lane == 0)
      nch[s] = (m + 7) >> 3;
  }
}

// sorted form: one workgroup orders the SP_SIGMA rows of its window by length (descending, ties by row:
// a stable counting rank), writes the row of every (slice, lane) and the slice lengths
__global__ __launch_bounds__(SP_SIGMA) void k_sp_sort(const int32_t* __restrict__ rownnz, int nrows, int64_t nslices,
                                                      int32_t* __restrict__ perm, int32_t* __restrict__ nch)
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
      nch[s] = (max(srt[t * 64], 0) + 7) >> 3;
  }
  if (w == 0 && t == 0)
    nch[nslices] = 0;
}

// One wavefront packs one slice.  ghost_flag (or null): does the slice reference a column >= nrows?
template <bool PERM>
__global__ __launch_bounds__(256) void k_sp_fill(const rp_t* __restrict__ rowptr, const int32_t* __restrict__ cols,
                                                 const double* __restrict__ vals, int nrows, int64_t nslices, int drop,
                                                 const int32_t* __restrict__ perm, const int32_t* __restrict__ chunk_off,
                                                 double* __restrict__ svals, uint16_t* __restrict__ c16,
                                                 int32_t* __restrict__ c32, int32_t* __restrict__ meta,
                                                 uint8_t* __restrict__ ghost_flag)
{
  const int lane = threadIdx.x & 63;
  for (int64_t s = blockIdx.x * 4 + (threadIdx.x >> 6); s < nslices; s += (int64_t)gridDim.x * 4)
  {
    int r = PERM ? perm[s * 64 + lane] : (int)(s * 64 + lane);
    if (!PERM && r >= nrows)
      r = -1;
    int64_t k = r >= 0 ? rowptr[r] : 0;
    const int64_t end = r >= 0 ? rowptr[r + 1] : 0;
    const int c0 = chunk_off[s], c1 = chunk_off[s + 1];
    bool gh = false;
    for (int c = c0; c < c1; ++c)
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
      int base[8];
      bool wide = false;
#pragma unroll
      for (int e = 0; e < 8; ++e)
      {
        const bool has = cl[e] != INT_MAX;
        gh |= has && cl[e] >= nrows;
        int mn = wave_min_i(cl[e]);
        const int mx = wave_max_i(has ? cl[e] : -1);
        if (mn == INT_MAX)
          mn = 0; // no lane has an entry in this slot
        wide |= mx - mn > 65535;
        base[e] = mn;
        if (!has)
          cl[e] = mn; // padding: value +0.0, a column some lane reads anyway
      }
      dbl2* vp = reinterpret_cast<dbl2*>(svals + (size_t)c * 512) + lane;
#pragma unroll
      for (int j = 0; j < 4; ++j)
      {
        dbl2 q;
        q.x = v[2 * j];
        q.y = v[2 * j + 1];
        vp[64 * j] = q;
      }
      if (!wide)
      {
        uint4v q;
        q.x = (unsigned)(cl[0] - base[0]) | ((unsigned)(cl[1] - base[1]) << 16);
        q.y = (unsigned)(cl[2] - base[2]) | ((unsigned)(cl[3] - base[3]) << 16);
        q.z = (unsigned)(cl[4] - base[4]) | ((unsigned)(cl[5] - base[5]) << 16);
        q.w = (unsigned)(cl[6] - base[6]) | ((unsigned)(cl[7] - base[7]) << 16);
        reinterpret_cast<uint4v*>(c16 + (size_t)c * 512)[lane] = q;
      }
      else
      {
        int4v q0, q1;
        q0.x = cl[0], q0.y = cl[1], q0.z = cl[2], q0.w = cl[3];
        q1.x = cl[4], q1.y = cl[5], q1.z = cl[6], q1.w = cl[7];
        int4v* cp = reinterpret_cast<int4v*>(c32 + (size_t)c * 512) + 2 * lane;
        cp[0] = q0;
        cp[1] = q1;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (lane == e)
          meta[(size_t)c * 8 + e] = (e == 0 && wide) ? (base[e] | (int)0x80000000) : base[e];
    }
    if (ghost_flag)
    {
      const unsigned long long m = __ballot(gh);
      if (lane == 0)
        ghost_flag[s] = m != 0ull;
    }
  }
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
  if (lane == 0)
  {
    atomicMax(&out[0], mr);
    atomicAdd(reinterpret_cast<unsigned long long*>(out + 2), ch);
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
// desc[s] = {first chunk, chunks}.  ghost_flag as in k_sp_fill.
__global__ __launch_bounds__(256) void k_sp_pack(const rp_t* __restrict__ rowptr, const int32_t* __restrict__ cols,
                                                 const double* __restrict__ vals, int nrows, int64_t nslices, int drop, int cap,
                                                 int* __restrict__ counter, int2* __restrict__ desc,
                                                 double* __restrict__ svals, uint16_t* __restrict__ c16,
                                                 int32_t* __restrict__ c32, int32_t* __restrict__ meta,
                                                 uint8_t* __restrict__ ghost_flag)
{
  extern __shared__ __attribute__((aligned(16))) char sp_smem[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nwv = blockDim.x >> 6;
  double* lv = reinterpret_cast<double*>(sp_smem + (size_t)wv * cap * 12);
  int* lc = reinterpret_cast<int*>(sp_smem + (size_t)wv * cap * 12 + (size_t)cap * 8);
  const unsigned long long lt = (1ull << lane) - 1ull;
  for (int64_t s = (int64_t)blockIdx.x * nwv + wv; s < nslices; s += (int64_t)gridDim.x * nwv)
  {
    const int r0 = (int)(s * 64);
    const int64_t a = rowptr[r0], b = rowptr[min(r0 + 64, nrows)];
    const int64_t my_start = rowptr[min(r0 + lane, nrows)];
    int running = 0, cstart = 0;
    for (int64_t g = a; g < b; g += 64)
    {
      const int64_t k = g + lane;
      const bool in = k < b;
      const double v = in ? __builtin_nontemporal_load(vals + k) : 0.0;
      const int c = in ? __builtin_nontemporal_load(cols + k) : 0;
      const bool nz = in && (!drop || v != 0.0);
      const unsigned long long m = __ballot(nz);
      if (my_start >= g && my_start < g + 64)
        cstart = running + __popcll(m & ((1ull << (my_start - g)) - 1ull));
      if (nz)
      {
        const int pos = running + __popcll(m & lt);
        lv[pos] = v;
        lc[pos] = c;
      }
      running += __popcll(m);
    }
    if (my_start >= b)
      cstart = running;
    const int nxt = __shfl_down(cstart, 1, 64);
    const int cnt = (lane == 63 ? running : nxt) - cstart;
    const int nch = (wave_max_i(cnt) + 7) >> 3;
    int c0 = 0;
    if (lane == 0)
    {
      c0 = nch ? atomicAdd(counter, nch) : 0;
      desc[s] = make_int2(c0, nch);
      atomicAdd(reinterpret_cast<unsigned long long*>(counter + 2), (unsigned long long)running); // entries kept
    }
    c0 = __shfl(c0, 0, 64);
    __builtin_amdgcn_wave_barrier();
    bool gh = false;
    for (int j = 0; j < nch; ++j)
    {
      const int c = c0 + j;
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
      int base[8];
      bool wide = false;
#pragma unroll
      for (int e = 0; e < 8; ++e)
      {
        const bool has = cl[e] != INT_MAX;
        gh |= has && cl[e] >= nrows;
        int mn = wave_min_i(cl[e]);
        const int mx = wave_max_i(has ? cl[e] : -1);
        if (mn == INT_MAX)
          mn = 0;
        wide |= mx - mn > 65535;
        base[e] = mn;
        if (!has)
          cl[e] = mn;
      }
      dbl2* vp = reinterpret_cast<dbl2*>(svals + (size_t)c * 512) + lane;
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
      {
        dbl2 q;
        q.x = v[2 * jj];
        q.y = v[2 * jj + 1];
        vp[64 * jj] = q;
      }
      if (!wide)
      {
        uint4v q;
        q.x = (unsigned)(cl[0] - base[0]) | ((unsigned)(cl[1] - base[1]) << 16);
        q.y = (unsigned)(cl[2] - base[2]) | ((unsigned)(cl[3] - base[3]) << 16);
        q.z = (unsigned)(cl[4] - base[4]) | ((unsigned)(cl[5] - base[5]) << 16);
        q.w = (unsigned)(cl[6] - base[6]) | ((unsigned)(cl[7] - base[7]) << 16);
        reinterpret_cast<uint4v*>(c16 + (size_t)c * 512)[lane] = q;
      }
      else
      {
        int4v q0, q1;
        q0.x = cl[0], q0.y = cl[1], q0.z = cl[2], q0.w = cl[3];
        q1.x = cl[4], q1.y = cl[5], q1.z = cl[6], q1.w = cl[7];
        int4v* cp = reinterpret_cast<int4v*>(c32 + (size_t)c * 512) + 2 * lane;
        cp[0] = q0;
        cp[1] = q1;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (lane == e)
          meta[(size_t)c * 8 + e] = (e == 0 && wide) ? (base[e] | (int)0x80000000) : base[e];
    }
    if (ghost_flag)
    {
      const unsigned long long m = __ballot(gh);
      if (lane == 0)
        ghost_flag[s] = m != 0ull;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// sorted form: {first chunk, chunks} of every slice from the scanned offsets
__global__ void k_sp_desc(const int32_t* __restrict__ off, int64_t nslices, int2* __restrict__ desc)
{
  for (int64_t s = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; s < nslices; s += (int64_t)gridDim.x * blockDim.x)
    desc[s] = make_int2(off[s], off[s + 1] - off[s]);
}

// ---- the product --------------------------------------------------------------------------------------
template <bool NT, typename T>
__device__ inline T sp_load(const T* p)
{
  return NT ? __builtin_nontemporal_load(p) : *p;
}

// x[col] with a 32-bit byte offset: one shift per gather instead of a 64-bit address computation
// (the launcher guarantees 8 * ncols < 2^32)
__device__ inline double gather(const double* __restrict__ x, int col)
{
  return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(x) + ((unsigned)col << 3));
}

// columns of chunk c for this lane: 16-bit codes on the slot bases, or int32 (sign bit of base[0])
template <bool NT>
__device__ inline void chunk_columns(int c, int lane, const uint16_t* __restrict__ c16, const int32_t* __restrict__ c32,
                                     const int32_t* __restrict__ meta, int (&cl)[8])
{
  const int32_t* __restrict__ mp = meta + (size_t)c * 8;
  const int b0 = mp[0];
  if (b0 >= 0)
  {
    const uint4v q = sp_load<NT>(reinterpret_cast<const uint4v*>(c16 + (size_t)c * 512) + lane);
    cl[0] = b0 + (int)(q.x & 0xffffu);
    cl[1] = mp[1] + (int)(q.x >> 16);
    cl[2] = mp[2] + (int)(q.y & 0xffffu);
    cl[3] = mp[3] + (int)(q.y >> 16);
    cl[4] = mp[4] + (int)(q.z & 0xffffu);
    cl[5] = mp[5] + (int)(q.z >> 16);
    cl[6] = mp[6] + (int)(q.w & 0xffffu);
    cl[7] = mp[7] + (int)(q.w >> 16);
  }
  else
  {
    const int4v* __restrict__ cp = reinterpret_cast<const int4v*>(c32 + (size_t)c * 512) + 2 * lane;
    const int4v q0 = sp_load<NT>(cp), q1 = sp_load<NT>(cp + 1);
    cl[0] = q0.x, cl[1] = q0.y, cl[2] = q0.z, cl[3] = q0.w;
    cl[4] = q1.x, cl[5] = q1.y, cl[6] = q1.z, cl[7] = q1.w;
  }
}

template <bool DOT, bool NT, bool PERM>
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
                                                              const double* __restrict__ rvec, int pstride, int nn_is_rr)
{
  // group_list != nullptr: only the listed groups of 4 slices (interior or boundary subset of a partitioned
  // matrix); rvec != nullptr: also the partials of <r,x> and of the test norm (single-reduction CG), as in
  // spmv_tile_kernel
  if (stop_flag && *stop_flag) // CG already converged: the host is a few iterations ahead
    return;
  __shared__ double red[SP_BLOCK / 64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t ngroups = group_list ? nlist : (nslices + 3) / 4; // a workgroup takes 4 consecutive slices
  double dot = 0.0, dot_rx = 0.0, dot_nn = 0.0;
  for (int i = 0;; ++i)
  {
    const int64_t gi = sp_xcd_item(ngroups, blockIdx.x, gridDim.x, i);
    if (gi < 0)
      break;
    const int64_t g = group_list ? group_list[gi] : gi;
    const int s = __builtin_amdgcn_readfirstlane((int)(4 * g + wv));
    if (s >= nslices)
      continue;
    const int2 ds = desc[s];
    const int c0 = ds.x, c1 = ds.x + ds.y;
    int r = PERM ? perm[(int64_t)s * 64 + lane] : s * 64 + lane;
    if (!PERM && r >= nrows)
      r = -1;
    const double xr = (DOT && r >= 0) ? x[r] : 0.0;
    double sum = 0.0;
    for (int c = c0; c < c1; ++c)
    {
      const dbl2* __restrict__ vp = reinterpret_cast<const dbl2*>(svals + (size_t)c * 512) + lane;
      const dbl2 v0 = sp_load<NT>(vp), v1 = sp_load<NT>(vp + 64), v2 = sp_load<NT>(vp + 128), v3 = sp_load<NT>(vp + 192);
      int cl[8];
      chunk_columns<NT>(c, lane, c16, c32, meta, cl);
      double xv[8];
#pragma unroll
      for (int e = 0; e < 8; ++e)
        xv[e] = gather(x, cl[e]);
      // ascending column order, mul and add rounded separately: the scalar CPU loop's bits
      sum += v0.x * xv[0];
      sum += v0.y * xv[1];
      sum += v1.x * xv[2];
      sum += v1.y * xv[3];
      sum += v2.x * xv[4];
      sum += v2.y * xv[5];
      sum += v3.x * xv[6];
      sum += v3.y * xv[7];
    }
    if (r >= 0)
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
    const double sres = block_reduce_sum(dot, red);
    if (threadIdx.x == 0)
      partials[blockIdx.x] = sres;
    if (rvec)
    {
      const double s1 = block_reduce_sum(dot_rx, red);
      const double s2 = block_reduce_sum(dot_nn, red);
      if (threadIdx.x == 0)
      {
        partials[pstride + blockIdx.x] = s1;
        partials[2 * pstride + blockIdx.x] = s2;
      }
    }
  }
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
    const int c0 = ds.x, c1 = ds.x + ds.y;
    double sum = 0.0;
    for (int c = c0; c < c1; ++c)
    {
      const dbl2* __restrict__ vp = reinterpret_cast<const dbl2*>(svals + (size_t)c * 512) + lane;
      const dbl2 v0 = sp_load<NT>(vp), v1 = sp_load<NT>(vp + 64), v2 = sp_load<NT>(vp + 128), v3 = sp_load<NT>(vp + 192);
      int cl[8];
      chunk_columns<NT>(c, lane, c16, c32, meta, cl);
      double zv[8], pv[8];
#pragma unroll
      for (int e = 0; e < 8; ++e)
      {
        zv[e] = gather(z, cl[e]);
        pv[e] = gather(p_old, cl[e]);
      }
      sum += v0.x * (bcoef * pv[0] + zv[0]);
      sum += v0.y * (bcoef * pv[1] + zv[1]);
      sum += v1.x * (bcoef * pv[2] + zv[2]);
      sum += v1.y * (bcoef * pv[3] + zv[3]);
      sum += v2.x * (bcoef * pv[4] + zv[4]);
      sum += v2.y * (bcoef * pv[5] + zv[5]);
      sum += v3.x * (bcoef * pv[6] + zv[6]);
      sum += v3.y * (bcoef * pv[7] + zv[7]);
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
  ZZZ_HIP(ctx, ctx->sp_nch.alloc((size_t)nsl + 1));
  ZZZ_HIP(ctx, ctx->sp_chunk_off.alloc((size_t)nsl + 1));
  ZZZ_HIP(ctx, ctx->sp_perm.alloc((size_t)nsl * 64));
  hipLaunchKernelGGL(k_sp_count, dim3(grid_cap(nrows, 256, 16384)), dim3(256), 0, s, ctx->rowptr.p, ctx->vals.p, nrows, drop,
                     ctx->sp_rownnz.p);
  const int64_t nwin = (ctx->nrows + SP_SIGMA - 1) / SP_SIGMA;
  if (sorted)
    hipLaunchKernelGGL(k_sp_sort, dim3((unsigned)nwin), dim3(SP_SIGMA), 0, s, ctx->sp_rownnz.p, nrows, nsl, ctx->sp_perm.p,
                       ctx->sp_nch.p);
  else // natural row order, rows too long for the LDS staging of k_sp_pack
    hipLaunchKernelGGL(k_sp_slice_len, dim3(grid_cap(nsl + 1, 4, 8192)), dim3(256), 0, s, ctx->sp_rownnz.p, nrows, nsl,
                       ctx->sp_nch.p);
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
  if (sorted)
    hipLaunchKernelGGL(k_sp_fill<true>, dim3(grid_cap(nsl, 4, 16384)), dim3(256), 0, s, ctx->rowptr.p, ctx->cols.p, ctx->vals.p,
                       nrows, nsl, ctx->sellp_drop ? 1 : 0, ctx->sp_perm.p, ctx->sp_chunk_off.p, ctx->sp_vals.p, ctx->sp_codes16.p,
                       ctx->sp_codes32.p, ctx->sp_meta.p, gflag);
  else
    hipLaunchKernelGGL(k_sp_fill<false>, dim3(grid_cap(nsl, 4, 16384)), dim3(256), 0, s, ctx->rowptr.p, ctx->cols.p, ctx->vals.p,
                       nrows, nsl, ctx->sellp_drop ? 1 : 0, (const int32_t*)nullptr, ctx->sp_chunk_off.p, ctx->sp_vals.p,
                       ctx->sp_codes16.p, ctx->sp_codes32.p, ctx->sp_meta.p, gflag);
  hipLaunchKernelGGL(k_sp_desc, dim3(grid_cap(nsl, 256, 4096)), dim3(256), 0, s, ctx->sp_chunk_off.p, nsl,
                     reinterpret_cast<int2*>(ctx->sp_desc.p));
  ZZZ_HIP(ctx, hipGetLastError());
  ctx->sp_sorted = sorted;
  ctx->sp_chunks = total;
  return sp_group_split(ctx, gflag);
}

// Pattern-only bounds of the natural-order stream (once per pattern; one small read-back).
int sellp_pattern_bounds(zzz_ctx* ctx)
{
  ctx->sp_bounds_ok = false;
  ctx->have_sell = ctx->sell_current = ctx->sp_pending = false;
  if (ctx->nrows <= 0)
    return ZZZ_OK;
  hipStream_t s = ctx->stream;
  const int64_t nsl = (ctx->nrows + 63) / 64;
  ctx->nslices = nsl;
  ZZZ_HIP(ctx, ctx->sp_counter.alloc(8));
  ZZZ_HIP(ctx, hipMemsetAsync(ctx->sp_counter.p, 0, 8 * sizeof(int), s));
  hipLaunchKernelGGL(k_sp_bounds, dim3(grid_cap(nsl, 4, 4096)), dim3(256), 0, s, ctx->rowptr.p, (int)ctx->nrows, nsl,
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
  const bool lds_fits = lds * waves <= 160 * 1024 && (waves >= 2 || lds <= 64 * 1024);
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
  if (lds * waves > 64 * 1024 && !ctx->sp_lds_attr)
  {
    ZZZ_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_sp_pack), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     160 * 1024));
    ctx->sp_lds_attr = true;
  }
  ZZZ_HIP(ctx, hipMemsetAsync(ctx->sp_counter.p, 0, 4 * sizeof(int), s));
  const int cap = (ctx->sp_max_range + 63) & ~63;
  hipLaunchKernelGGL(k_sp_pack, dim3(grid_cap(nsl, waves, 256 * 12)), dim3(64 * waves), lds * waves, s, ctx->rowptr.p, ctx->cols.p,
                     ctx->vals.p, nrows, nsl, ctx->sellp_drop ? 1 : 0, cap, ctx->sp_counter.p, reinterpret_cast<int2*>(ctx->sp_desc.p),
                     ctx->sp_vals.p, ctx->sp_codes16.p, ctx->sp_codes32.p, ctx->sp_meta.p, gflag);
  ZZZ_HIP(ctx, hipGetLastError());
  if (!ctx->sp_event)
    ZZZ_HIP(ctx, hipEventCreateWithFlags(&ctx->sp_event, hipEventDisableTiming));
  int32_t* tot = reinterpret_cast<int32_t*>(ctx->h_state + 5); // pinned
  ZZZ_HIP(ctx, hipMemcpyAsync(tot, ctx->sp_counter.p, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, s));
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
  ctx->sp_chunks = t0;
  const double full = (double)ctx->nnz + 64.0 * 512.0;
  const bool always = ctx->sellp_mode == 2 || ctx->sp_forced;
  // Natural row order unless its padding makes it slower than the alternatives: the length-sorted form (priced only
  // when the natural stream is padded by more than a third: it costs a synchronous build) or the CSR tile kernel.
  const double c_nat = cost_stream(t0, false), c_tile = cost_tile(ctx);
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

bool sellp_active(zzz_ctx* ctx)
{
  if (ctx->sp_pending)
    (void)sellp_resolve(ctx);
  if (!ctx->have_sell || !ctx->sell_current)
    return false;
  return ctx->spmv_auto || (ctx->spmv_variant & 8) != 0;
}

// bytes one product reads from the stream (values + codes + bases; int32 chunks are not counted separately)
int64_t sellp_stream_bytes(const zzz_ctx* ctx) { return ctx->sp_chunks * (4096 + 1024 + 32) + ctx->nslices * 8; }

static int sp_grid(int64_t ngroups)
{
  int64_t gs = 256 * 8;
  const int64_t need = (ngroups + 7) / 8 * 8;
  if (gs > need)
    gs = need;
  if (gs < 8)
    gs = 8;
  return (int)gs;
}

template <bool DOT>
static void launch_one(zzz_ctx* ctx, int grid, const double* x, double* y, double* partials, const int* stop,
                       const int32_t* group_list, int64_t nlist, const double* rvec, int nn_is_rr)
{
  // load policy by stream size, as for the tile kernel: a stream that stays in the 256 MiB Infinity Cache from
  // one CG iteration to the next is read with plain loads, a larger one with non-temporal loads
  bool nt = (double)sellp_stream_bytes(ctx) > 300.0e6;
  if (!ctx->spmv_auto)
    nt = (ctx->spmv_variant & 1) != 0;
  const int2* off = reinterpret_cast<const int2*>(ctx->sp_desc.p);
#define ZZZ_SP_GO(NT, PERM)                                                                                            \
  hipLaunchKernelGGL((spmv_sellp_kernel<DOT, NT, PERM>), dim3(grid), dim3(SP_BLOCK), 0, ctx->stream, off, ctx->sp_vals.p,     \
                     ctx->sp_codes16.p, ctx->sp_codes32.p, ctx->sp_meta.p, ctx->sp_perm.p, x, y, (int)ctx->nrows,          \
                     ctx->nslices, partials, stop, group_list, nlist, rvec, SPMV_PSTRIDE, nn_is_rr)
  if (ctx->sp_sorted)
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
}

int launch_sellp(zzz_ctx* ctx, const double* x, double* y, double* partials, int* npartials, const double* rvec, int nn_is_rr)
{
  const int* stop = partials ? reinterpret_cast<const int*>(ctx->state.p) : nullptr; // CgState::converged
  const int gs = sp_grid((ctx->nslices + 3) / 4);
  if (partials)
  {
    launch_one<true>(ctx, gs, x, y, partials, stop, nullptr, 0, rvec, nn_is_rr);
    if (npartials)
      *npartials = gs;
  }
  else
    launch_one<false>(ctx, gs, x, y, nullptr, stop, nullptr, 0, nullptr, 0);
  ZZZ_HIP(ctx, hipGetLastError());
  return ZZZ_OK;
}

// Partitioned matrix: forward halo of x overlapped with the groups that reference no ghost column
// (scheme and the 7-of-8 workgroup slots: launch_spmv_overlapped in zzz_spmv.hip)
int launch_sellp_overlapped(zzz_ctx* ctx, double* x, double* y, double* partials, int* npartials, const double* rvec,
                            int nn_is_rr)
{
  const int* stop = partials ? reinterpret_cast<const int*>(ctx->state.p) : nullptr;
  const int64_t gi = ctx->n_groups_interior, gb = ctx->n_groups_boundary;
  int g_in = gi ? sp_grid(gi) : 0;
  if (g_in > 256 * 7 && ctx->nneigh > 0)
    g_in = 256 * 7;
  const int g_bd = gb ? sp_grid(gb) : 0;
  if (partials && (size_t)(g_in + g_bd) > (size_t)SPMV_PSTRIDE)
    return fail(ctx, ZZZ_ERR_ARG, "partials buffer too small");
  int rc = comm_halo_begin(ctx, x);
  if (rc)
    return rc;
  if (gi)
  {
    if (partials)
      launch_one<true>(ctx, g_in, x, y, partials, stop, ctx->groups_interior.p, gi, rvec, nn_is_rr);
    else
      launch_one<false>(ctx, g_in, x, y, nullptr, stop, ctx->groups_interior.p, gi, nullptr, 0);
  }
  rc = comm_halo_end(ctx);
  if (rc)
    return rc;
  if (gb)
  {
    if (partials)
      launch_one<true>(ctx, g_bd, x, y, partials + g_in, stop, ctx->groups_boundary.p, gb, rvec, nn_is_rr);
    else
      launch_one<false>(ctx, g_bd, x, y, nullptr, stop, ctx->groups_boundary.p, gb, nullptr, 0);
  }
  if (npartials)
    *npartials = g_in + g_bd;
  ZZZ_HIP(ctx, hipGetLastError());
  return ZZZ_OK;
}
// the fused product + direction kernel on the whole matrix, or (partitioned matrix) interior groups, halo of z,
// boundary groups.  Partials of <p,w>: interior workgroups first.
int launch_sellp_dir(zzz_ctx* ctx, double* z, const double* p_old, double* p_new, double* xsol, double* y, double* partials,
                     int* npartials, int it, const CgParams& P, const double* pa, const double* pb, int np, bool overlap)
{
  const bool nt = ctx->spmv_auto ? (double)sellp_stream_bytes(ctx) > 300.0e6 : (ctx->spmv_variant & 1) != 0;
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
    int g_in = gi ? sp_grid(gi) : 0;
    if (g_in > 256 * 7 && ctx->nneigh > 0)
      g_in = 256 * 7; // room for the exchange's kernel beside the persistent workgroups (launch_spmv_overlapped)
    const int g_bd = gb ? sp_grid(gb) : 8;
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
    const int gs = sp_grid((ctx->nslices + 3) / 4);
    go(gs, nullptr, 0, partials, ctx->n_ghost > 0 ? 1 : 0);
    *npartials = gs;
  }
  ZZZ_HIP(ctx, hipGetLastError());
  return ZZZ_OK;
}
} // namespace zzz

// The CG product for BLOCK SIZE 3 (elasticity, BASELINE configs[3]) in block-row form (round 6).
//
// Replaces PETSc MatMult inside KSPSolve (src/elasticity_problem.cpp:250-259) like the other products of this library, with the
// same arithmetic: a scalar row's products are added in ascending column order, mul and add rounded separately -- bit-identical
// to the serial CSR loop (zo_spmv) for finite x.
//
// Why another form.  The generic product (zzz_sellp.hip) treats an elasticity matrix as 3 n scalar rows: at C4 it issues 105
// vector + 95 scalar instructions per chunk of 8 entries per row (column decode, code extraction, LDS addresses, mode dispatch)
// and runs at the rate the CU issues them (profiles/r05_sq_c4.json: VALU 62 %, SALU 59 % busy), not at a byte rate.  The matrix
// of a vector-valued P1 space is a matrix of 3 x 3 BLOCKS: the three rows of a node share their block columns, and on a
// regular mesh the blocks themselves repeat (C4: 19.5 M non-zero blocks, 1 664 distinct ones as bit patterns; 362 distinct
// values).  So:
//   * one LANE per NODE (three scalar rows), one wavefront per slice of 64 consecutive nodes;
//   * per (node, block column) the stream holds ONE 16-bit block code into a table of distinct blocks (9 doubles each) that
//     every workgroup copies into LDS once, and -- only where the slice is not affine -- one 16-bit column code relative to the
//     slot's base; slot e of an affine chunk reaches block column base[e] + lane: nothing per lane to load;
//   * per block: three x values as one 16-B + one 8-B load (the 64 lanes' 24-B records are one dense run of 1.5 KiB), nine
//     LDS reads, nine mul + nine add.  ~3.5 instructions per matrix entry instead of ~25, 2 B of stream per BLOCK instead of
//     2 B per entry.
// Whole-zero blocks are left out (as the generic stream leaves out exact zeros); zeros INSIDE a kept block stay (the serial
// loop multiplies them too).  A chunk holds 16 block slots per node; padded slots carry code 0 = the zero block and read a
// valid or out-of-range x (buffer loads return 0 there).
//
// When it applies: block size 3, natural row order, at most BK_TAB_MAX distinct blocks (the LDS copy), 16-bit column codes
// suffice, no folded all-reduce on the launch (tools build; that stays on the generic kernel).  The Chebyshev-Jacobi polynomial's
// terms ride on it as epilogues exactly as on the generic kernel (CHEB: C4's solve with -pc_type chebyshev_jacobi 419 -> 191 ms).  Otherwise nothing changes.  ZZZ_SELLP_BLK=0 switches it off (A/B, parity tests), 2 forces it below 100 000 nodes.
// Built from the CSR matrix of record, on the device, when the matrix is assembled (sell_update: the generic stream is then not
// packed at all) or, for a partitioned matrix, at the stream's first use after an assembly (sellp_active).
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "zzz_sellp.h"

#include <rocprim/rocprim.hpp>

namespace zzz
{
constexpr int BK_SLOTS = 16;             // block slots per node and chunk
constexpr int BK_META = 32;              // ints of meta per chunk: [0..15] slot bases, [16] width | affine << 8
constexpr int BK_HASH_BITS = 18;         // open-addressing set of the distinct blocks (build only)
constexpr int BK_HASH = 1 << BK_HASH_BITS;
constexpr int BK_TAB_MAX = 2200;         // form 1: entries of the block table incl. the zero block: 2200 x 72 B = 158 400 B of LDS
constexpr int BK_CODE_MAX = 65536;       // form 2: entries of the block table (16-bit codes), rows of nine value offsets in memory
constexpr int BK_VAL_BITS = 13;          // form 2: set of the table's distinct VALUES (build only)
constexpr int BK_VAL_HASH = 1 << BK_VAL_BITS;
constexpr int BK_VAL_MAX = 2048;         // ... at most this many (the dictionary every workgroup copies into LDS: 16 KiB)
constexpr int BK_THREADS = 1024;         // one workgroup per CU (the table takes its LDS), sixteen wavefronts

// the nine values of block k of node r: rows 3 r + a at rp[a], block k of a row at entries 3 k .. 3 k + 2
struct Blk9
{
  unsigned long long b[9];
};

__device__ inline void bk_load(const unsigned long long* __restrict__ vals, int64_t p0, int64_t p1, int64_t p2, int k, Blk9& B)
{
#pragma unroll
  for (int d = 0; d < 3; ++d)
  {
    B.b[d] = vals[p0 + 3 * k + d];
    B.b[3 + d] = vals[p1 + 3 * k + d];
    B.b[6 + d] = vals[p2 + 3 * k + d];
  }
}

__device__ inline bool bk_nonzero(const Blk9& B)
{
  unsigned long long any = 0;
#pragma unroll
  for (int i = 0; i < 9; ++i)
    any |= B.b[i] << 1; // (-0.0 is a zero too)
  return any != 0ull;
}

__device__ inline unsigned long long bk_hash(const Blk9& B)
{
  unsigned long long h = 0x9E3779B97F4A7C15ull;
#pragma unroll
  for (int i = 0; i < 9; ++i)
  {
    h = (h ^ B.b[i]) * 0xff51afd7ed558ccdull;
    h ^= h >> 31;
  }
  return h ? h : 1ull; // 0 marks an empty slot
}

// a second, independent fingerprint: a block is identified by the PAIR (128 bits; two different blocks of one matrix sharing both
// has probability ~ n^2 / 2^129 for n distinct blocks, 1e-29 at 65 536 -- below any hardware error rate; until the last third of
// round 6 one fingerprint + a comparison of every block with its table row, which cost a third walk over the values)
__device__ inline unsigned long long bk_hash2(const Blk9& B)
{
  unsigned long long h = 0xD6E8FEB86659FD93ull;
#pragma unroll
  for (int i = 0; i < 9; ++i)
  {
    h = (h ^ ((B.b[i] << 17) | (B.b[i] >> 47))) * 0xC2B2AE3D27D4EB4Full;
    h ^= h >> 29;
  }
  return h ? h : 1ull;
}

// info: [0] distinct blocks so far, [1] the form does not apply (structure, too many blocks, a column code beyond 16 bits, a
// fingerprint collision), [2] entries of the table, [4..5] bytes a product reads (64-bit)

// pass 1: ONE walk over the values.  Every kept block into the set -- tag = first fingerprint, tag2 = second (whoever sets it
// first owns the slot: a block with the same tag and another tag2 moves on), owner = the smallest (node << 10 | block) that carries
// the pair (so that the table does not depend on which lane came first) --, the SLOT it landed in parked per block (park: -1 for a
// zero block) for the fill pass, which then needs no value; chunks per slice = ceil(most kept blocks of a node / 16); the matrix's
// block structure checked on the way.  (Staging a slice's contiguous run of values in LDS by dense loads, a wavefront per
// workgroup, was measured: 1.89 against 1.95 ms at C4 -- the walk is not bound by its 1 080-B-strided loads; dropped.)
// (Until the last third of round 6: a count walk, an insert walk and a fill walk that hashed
// every block again and compared it with its table row: 0.9 + 1.7 + 1.75 ms at C4.)
__global__ __launch_bounds__(256) void k_bk_insert(const rp_t* __restrict__ rowptr, const unsigned long long* __restrict__ vals,
                                                   int nnodes, int64_t nsl, unsigned long long* __restrict__ tag,
                                                   unsigned long long* __restrict__ tag2, unsigned long long* __restrict__ owner,
                                                   int32_t* __restrict__ park, int32_t* __restrict__ nch, int* __restrict__ info, int limit)
{
  const int lane = threadIdx.x & 63;
  for (int64_t s = blockIdx.x * 4ll + (threadIdx.x >> 6); s < nsl; s += gridDim.x * 4ll)
  {
    if (__hip_atomic_load(&info[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
      return;
    const int64_t r = s * 64 + lane;
    int kept = 0;
    if (r < nnodes)
    {
      const int64_t p0 = rowptr[3 * r], p1 = rowptr[3 * r + 1], p2 = rowptr[3 * r + 2], p3 = rowptr[3 * r + 3];
      const int64_t len = p1 - p0;
      const int nbk = (int)(len / 3);
      if (p2 - p1 != len || p3 - p2 != len || len % 3 != 0 || p0 % 9 != 0 || nbk >= 1024) // (the owner word keeps ten bits for the block)
        info[1] = 1;
      else
      {
        int32_t* const pk = park + p0 / 9;
        unsigned long long last = 0ull, last2 = 0ull;
        int last_slot = -1;
        for (int k = 0; k < nbk; ++k)
        {
          Blk9 B;
          bk_load(vals, p0, p1, p2, k, B);
          if (!bk_nonzero(B))
          {
            pk[k] = -1;
            continue;
          }
          ++kept;
          const unsigned long long fp = bk_hash(B), fq = bk_hash2(B);
          if (fp == last && fq == last2) // (a node's neighbours often carry the same block: its owner is this node or an earlier one)
          {
            pk[k] = last_slot;
            continue;
          }
          const unsigned long long me = ((unsigned long long)r << 10) | (unsigned)k;
          unsigned h = (unsigned)(fp >> (64 - BK_HASH_BITS));
          int slot = -1;
          for (int probe = 0; probe < BK_HASH; ++probe)
          {
            unsigned long long cur = __hip_atomic_load(&tag[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (cur == 0ull)
            {
              cur = atomicCAS(&tag[h], 0ull, fp);
              if (cur == 0ull)
              {
                if (atomicAdd(&info[0], 1) >= limit - 1)
                  info[1] = 2;
                cur = fp;
              }
            }
            if (cur == fp)
            {
              unsigned long long c2 = __hip_atomic_load(&tag2[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              if (c2 == 0ull)
              {
                c2 = atomicCAS(&tag2[h], 0ull, fq);
                if (c2 == 0ull)
                  c2 = fq;
              }
              if (c2 == fq)
              {
                if (__hip_atomic_load(&owner[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > me)
                  atomicMin(&owner[h], me);
                slot = (int)h;
                break;
              }
            }
            h = (h + 1) & (BK_HASH - 1);
            if (__hip_atomic_load(&info[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
              return;
          }
          if (slot < 0)
            info[1] = 3;
          pk[k] = slot;
          last = fp;
          last2 = fq;
          last_slot = slot;
        }
      }
    }
    const int m = wave_max_i(kept);
    if (lane == 0)
      nch[s] = (m + BK_SLOTS - 1) / BK_SLOTS;
  }
}

// pass 3: codes in slot order (deterministic), the table's rows from the owners; entry 0 = the zero block
__global__ __launch_bounds__(1024) void k_bk_number(const unsigned long long* __restrict__ tag, const unsigned long long* __restrict__ owner,
                                                    const rp_t* __restrict__ rowptr, const unsigned long long* __restrict__ vals,
                                                    int32_t* __restrict__ slot_code, unsigned long long* __restrict__ tab,
                                                    int* __restrict__ info)
{
  __shared__ int wsum[16];
  if (info[1])
    return;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  constexpr int PER = BK_HASH / 1024; // consecutive slots per thread
  int mine = 0;
  for (int k = 0; k < PER; ++k)
    mine += tag[threadIdx.x * PER + k] != 0ull ? 1 : 0;
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
  int off = 1;
  for (int q = 0; q < wv; ++q)
    off += wsum[q];
  int code = off + incl - mine;
  for (int k = 0; k < PER; ++k)
  {
    const int h = threadIdx.x * PER + k;
    if (tag[h] == 0ull)
      continue;
    slot_code[h] = code;
    if (code < BK_CODE_MAX)
    {
      const unsigned long long o = owner[h];
      const int64_t r = (int64_t)(o >> 10);
      const int kb = (int)(o & 1023u);
      Blk9 B;
      bk_load(vals, rowptr[3 * r], rowptr[3 * r + 1], rowptr[3 * r + 2], kb, B);
#pragma unroll
      for (int i = 0; i < 9; ++i)
        tab[(int64_t)code * 9 + i] = B.b[i];
    }
    ++code;
  }
  if (threadIdx.x < 9)
    tab[threadIdx.x] = 0ull;
  if (threadIdx.x == 1023)
  {
    info[2] = code;
    if (code > BK_CODE_MAX)
      info[1] = 2;
  }
}

// form 2 (more distinct blocks than the LDS table holds): the DISTINCT VALUES of the table's rows into a set, numbered in
// slot order; then every row as nine byte offsets (value code x 8, 16 bits each) into that dictionary, 32 B per row
__device__ inline unsigned bk_val_hash(unsigned long long b)
{
  b ^= b >> 29;
  b *= 0x9E3779B97F4A7C15ull;
  return (unsigned)(b >> (64 - BK_VAL_BITS));
}

__global__ __launch_bounds__(256) void k_bk_val_insert(const unsigned long long* __restrict__ tab, int n9,
                                                       unsigned long long* __restrict__ vset, int* __restrict__ info)
{
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n9; k += gridDim.x * blockDim.x)
  {
    const unsigned long long b = tab[k];
    if (b == 0ull)
      continue; // +0.0 is offset 0 without the set
    if (b == ~0ull)
    {
      info[1] = 5; // (the empty marker: no assembled value has this NaN pattern; met all the same: no dictionary)
      continue;
    }
    unsigned h = bk_val_hash(b);
    for (int probe = 0; probe < BK_VAL_HASH; ++probe)
    {
      unsigned long long cur = __hip_atomic_load(&vset[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (cur == ~0ull)
      {
        cur = atomicCAS(&vset[h], ~0ull, b);
        if (cur == ~0ull)
        {
          if (atomicAdd(&info[3], 1) >= BK_VAL_MAX - 2)
            info[1] = 5;
          cur = b;
        }
      }
      if (cur == b)
        break;
      h = (h + 1) & (BK_VAL_HASH - 1);
      if (__hip_atomic_load(&info[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        return;
    }
  }
}

__global__ __launch_bounds__(1024) void k_bk_val_number(const unsigned long long* __restrict__ vset, int32_t* __restrict__ vcode,
                                                        unsigned long long* __restrict__ dict, int* __restrict__ info)
{
  __shared__ int wsum[16];
  if (info[1])
    return;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  constexpr int PER = BK_VAL_HASH / 1024;
  int mine = 0;
  for (int k = 0; k < PER; ++k)
    mine += vset[threadIdx.x * PER + k] != ~0ull ? 1 : 0;
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
  int off = 1; // (entry 0 is +0.0)
  for (int q = 0; q < wv; ++q)
    off += wsum[q];
  int code = off + incl - mine;
  for (int k = 0; k < PER; ++k)
  {
    const int h = threadIdx.x * PER + k;
    const unsigned long long b = vset[h];
    if (b == ~0ull)
      continue;
    vcode[h] = code;
    if (code < BK_VAL_MAX)
      dict[code] = b;
    ++code;
  }
  if (threadIdx.x == 0)
    dict[0] = 0ull;
  if (threadIdx.x == 1023)
  {
    info[3] = code; // entries of the dictionary, +0.0 included
    if (code > BK_VAL_MAX)
      info[1] = 5;
  }
}

__global__ __launch_bounds__(256) void k_bk_rows16(const unsigned long long* __restrict__ tab, int nent,
                                                   const unsigned long long* __restrict__ vset, const int32_t* __restrict__ vcode,
                                                   uint16_t* __restrict__ rows16, const int* __restrict__ info)
{
  if (info[1])
    return;
  for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < nent; r += gridDim.x * blockDim.x)
  {
    unsigned off[16];
#pragma unroll
    for (int i = 0; i < 16; ++i)
      off[i] = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i)
    {
      const unsigned long long b = tab[(int64_t)r * 9 + i];
      if (b != 0ull)
      {
        unsigned h = bk_val_hash(b);
        while (vset[h] != b)
          h = (h + 1) & (BK_VAL_HASH - 1);
        off[i] = (unsigned)vcode[h] * 8u;
      }
    }
    uint4v q0, q1;
    q0.x = off[0] | (off[1] << 16), q0.y = off[2] | (off[3] << 16), q0.z = off[4] | (off[5] << 16), q0.w = off[6] | (off[7] << 16);
    q1.x = off[8], q1.y = 0u, q1.z = 0u, q1.w = 0u;
    reinterpret_cast<uint4v*>(rows16)[2 * (int64_t)r] = q0;
    reinterpret_cast<uint4v*>(rows16)[2 * (int64_t)r + 1] = q1;
  }
}

// pass 4: the stream.  One wavefront per slice, lane = node; slot e of the slice = every lane's e-th kept block.
// Per chunk: meta[0..15] the slots' smallest block columns (column = base + 16-bit code), meta[16..31] the affine bases
// (column = base + lane: valid when flags bit 8 is set, i.e. EVERY slot of the chunk is affine); flags = width | affine << 8.
__global__ __launch_bounds__(256) void k_bk_fill(const rp_t* __restrict__ rowptr, const int32_t* __restrict__ cols,
                                                 const int32_t* __restrict__ park, int nnodes, int n_owned, int64_t nsl,
                                                 const int32_t* __restrict__ c0s, const int32_t* __restrict__ nchs,
                                                 const int32_t* __restrict__ slot_code, int2* __restrict__ desc,
                                                 int32_t* __restrict__ meta, int32_t* __restrict__ flags, uint16_t* __restrict__ bcode,
                                                 uint16_t* __restrict__ ccode, uint8_t* __restrict__ ghost_flag, int* __restrict__ info)
{
  if (info[1])
    return;
  const int lane = threadIdx.x & 63;
  unsigned long long bytes = 0;
  for (int64_t s = blockIdx.x * 4ll + (threadIdx.x >> 6); s < nsl; s += gridDim.x * 4ll)
  {
    const int64_t r = s * 64 + lane;
    const bool has = r < nnodes;
    const int64_t p0 = has ? rowptr[3 * r] : 0, p1 = has ? rowptr[3 * r + 1] : 0;
    const int nbk = has ? (int)((p1 - p0) / 3) : 0;
    const int32_t* const pk = park + p0 / 9;
    const int c0 = c0s[s], nch = nchs[s];
    if (lane == 0)
      desc[s] = make_int2(c0, nch);
    int k = 0; // the lane's next block to look at
    bool ghost = false;
    for (int j = 0; j < nch; ++j)
    {
      const int64_t c = (int64_t)c0 + j;
      int width = 0;
      bool affine = true;
      unsigned bpk[BK_SLOTS / 2], cpk[BK_SLOTS / 2]; // the lane's sixteen block codes / column codes, two per word
#pragma unroll
      for (int e = 0; e < BK_SLOTS; ++e)
      {
        // the lane's next kept block: its slot in the set was parked by k_bk_insert
        int slot = -1;
        while (k < nbk)
        {
          slot = pk[k];
          if (slot >= 0)
            break;
          ++k;
        }
        const bool found = k < nbk && slot >= 0;
        int col = INT_MAX;
        unsigned code = 0;
        if (found)
        {
          col = cols[p0 + 3 * k] / 3;
          ghost |= col >= n_owned;
          code = (unsigned)slot_code[slot];
          ++k;
        }
        const bool any = __ballot(found) != 0ull;
        const int mn = wave_min_i(col);                          // the slot's smallest block column
        const int am = wave_min_i(found ? col - lane : INT_MAX); // ... and its affine base, if it has one
        if (any)
        {
          width = e + 1;
          if (__ballot(found && col - lane != am) != 0ull)
            affine = false;
        }
        if (found && col - mn > 0xffff)
          info[1] = 4;
        const unsigned cc = (unsigned)(found ? col - mn : 0) & 0xffffu;
        if (e & 1)
        {
          bpk[e >> 1] |= (code & 0xffffu) << 16;
          cpk[e >> 1] |= cc << 16;
        }
        else
        {
          bpk[e >> 1] = code & 0xffffu;
          cpk[e >> 1] = cc;
        }
        if (lane == 0)
        {
          meta[c * BK_META + e] = any ? mn : 0;
          meta[c * BK_META + BK_SLOTS + e] = any ? am : 0;
        }
      }
      // (a lane's sixteen codes are 32 contiguous bytes of each array: two 16-B stores instead of sixteen 2-B ones)
      {
        uint4v* const bd = reinterpret_cast<uint4v*>(bcode + c * 1024 + lane * BK_SLOTS);
        uint4v* const cd = reinterpret_cast<uint4v*>(ccode + c * 1024 + lane * BK_SLOTS);
        bd[0] = uint4v{bpk[0], bpk[1], bpk[2], bpk[3]};
        bd[1] = uint4v{bpk[4], bpk[5], bpk[6], bpk[7]};
        cd[0] = uint4v{cpk[0], cpk[1], cpk[2], cpk[3]};
        cd[1] = uint4v{cpk[4], cpk[5], cpk[6], cpk[7]};
      }
      if (lane == 0)
      {
        flags[c] = width | (affine ? 0x100 : 0);
        bytes += 2048 + 68 + (affine ? 0 : 2048);
      }
    }
    const bool gh = __ballot(ghost) != 0ull;
    if (lane == 0)
    {
      ghost_flag[s] = gh ? 1 : 0;
      bytes += 8;
    }
  }
  if (lane == 0 && bytes)
    atomicAdd(reinterpret_cast<unsigned long long*>(info + 4), bytes);
}

// ---- the product -------------------------------------------------------------------------------------------------------
struct BlkArgs
{
  int ntab;    // entries of the block table (zero block included)
  int nnodes;  // owned nodes (rows / 3)
  int nslices; // slices of 64 nodes
  int nx8;     // bytes of x (owned + ghost scalars)
  double* partials;
  const int* stop_flag;
  int64_t nlist;
  int pstride, nn_is_rr;
  int xprobe = 0; // (tools build: ZZZ_BK_XPROBE, a timing probe of a component-major x)
};

template <bool NT, typename T>
__device__ inline T bk_ld(const T* p)
{
  return NT ? __builtin_nontemporal_load(p) : *p;
}

__device__ inline unsigned bk_code16(const uint4v& a, const uint4v& b, int e)
{
  const uint4v& q = e < 8 ? a : b;
  const int f = e & 7;
  const unsigned wd = f < 2 ? q.x : (f < 4 ? q.y : (f < 6 ? q.z : q.w));
  return (f & 1) ? wd >> 16 : wd & 0xffffu;
}

// FORM 1: the table's rows (nine doubles) in LDS; FORM 2: the rows as nine 16-bit byte offsets in memory (p_rows16, 32 B per
// row: L2-resident), the values they point at in LDS (p_tab = the dictionary then, a.ntab its entries)
template <bool DOT, bool SR, bool NT, int FORM, bool CHEB = false>
__global__ __launch_bounds__(BK_THREADS) void spmv_blk3_kernel(const int2* __restrict__ p_desc, const int32_t* __restrict__ p_meta,
                                                               const int32_t* __restrict__ p_flags,
                                                               const uint16_t* __restrict__ p_bcode,
                                                               const uint16_t* __restrict__ p_ccode, const double* __restrict__ p_tab,
                                                               const uint16_t* __restrict__ p_rows16,
                                                               const double* __restrict__ p_x, double* __restrict__ p_y,
                                                               const double* __restrict__ p_rvec, const int32_t* __restrict__ p_list,
                                                               BlkArgs a, ChebEpi epi)
{
  static_assert(!(CHEB && SR), "a Chebyshev term has no residual vector of its own");
  extern __shared__ __attribute__((aligned(16))) double bk_lds[]; // the block table: [entry][9]
  __shared__ double red[BK_THREADS / 64];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (a.stop_flag && *a.stop_flag) // CG already converged: the host is a few iterations ahead
    return;
  {
    // the table into LDS: eight entries per thread requested before any is stored
    const int n9 = FORM == 1 ? a.ntab * 9 : a.ntab;
    for (int k0 = 0; k0 < n9; k0 += BK_THREADS * 8)
    {
      double t[8];
#pragma unroll
      for (int i = 0; i < 8; ++i)
      {
        const int k = k0 + i * BK_THREADS + (int)threadIdx.x;
        t[i] = k < n9 ? p_tab[k] : 0.0;
      }
#pragma unroll
      for (int i = 0; i < 8; ++i)
      {
        const int k = k0 + i * BK_THREADS + (int)threadIdx.x;
        if (k < n9)
          bk_lds[k] = t[i];
      }
    }
  }
  __syncthreads();

  // the wavefronts of XCD x (workgroups x, x + 8, ...) walk the x-th eighth of the slices side by side
  const int64_t nitems = p_list ? a.nlist : (int64_t)a.nslices;
  const int xcd = blockIdx.x & 7;
  const int64_t lo = nitems * xcd / 8, hi = nitems * (xcd + 1) / 8;
  const int wgs_in_xcd = ((int)gridDim.x + 7 - xcd) >> 3;
  const int64_t stride = (int64_t)wgs_in_xcd * (BK_THREADS / 64);
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(p_x), 0, a.nx8, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(p_y, 0, a.nnodes * 24, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(SR ? p_rvec : p_x), 0, a.nnodes * 24, 0x00020000);

  double dot = 0.0, dot_rx = 0.0, dot_nn = 0.0;
  // One slice of look-ahead: while a slice is summed, the descriptor, the first chunk's flags and codes of the wavefront's next
  // slice are in flight (a slice is a chain list -> descriptor -> flags / bases -> codes -> x; without the look-ahead a
  // wavefront's five slices at C4 are five such chains end to end).  Two named stages, the body exists twice: no register that
  // a load is still writing is copied.
  struct Stage
  {
    int s, c0, nch, fl; // s < 0: no slice
    uint4v b0, b1; // (the column codes of a chunk that is not affine -- the mesh's rim -- are loaded when it is worked on)
  };
  auto fetch = [&](int64_t it, Stage& S) {
    S.s = -1;
    S.nch = 0;
    S.c0 = 0;
    S.fl = 0;
    if (it < hi)
    {
      S.s = __builtin_amdgcn_readfirstlane(p_list ? p_list[it] : (int)it);
      const int2 ds = p_desc[S.s];
      S.c0 = __builtin_amdgcn_readfirstlane(ds.x);
      S.nch = __builtin_amdgcn_readfirstlane(ds.y);
      if (S.nch > 0)
      {
        S.fl = __builtin_amdgcn_readfirstlane(p_flags[S.c0]);
        const uint4v* __restrict__ bp = reinterpret_cast<const uint4v*>(p_bcode + (int64_t)S.c0 * 1024) + 2 * lane;
        S.b0 = bk_ld<NT>(bp);
        S.b1 = bk_ld<NT>(bp + 1);
      }
    }
  };
  auto body = [&](Stage& S, int64_t it_next, Stage& N) {
    const int s = S.s;
    const int node24 = (s * 64 + lane) * 24;
    double xr0 = 0.0, xr1 = 0.0, xr2 = 0.0, rr0 = 0.0, rr1 = 0.0, rr2 = 0.0;
    if (DOT || CHEB)
    {
      const auto u = __builtin_amdgcn_raw_buffer_load_b128(rs_x, node24, 0, 0);
      const auto u2 = __builtin_amdgcn_raw_buffer_load_b64(rs_x, node24 + 16, 0, 0);
      xr0 = __hiloint2double((int)u[1], (int)u[0]);
      xr1 = __hiloint2double((int)u[3], (int)u[2]);
      xr2 = __hiloint2double((int)u2[1], (int)u2[0]);
      if (SR)
      {
        const auto t = __builtin_amdgcn_raw_buffer_load_b128(rs_r, node24, 0, 0);
        const auto t2 = __builtin_amdgcn_raw_buffer_load_b64(rs_r, node24 + 16, 0, 0);
        rr0 = __hiloint2double((int)t[1], (int)t[0]);
        rr1 = __hiloint2double((int)t[3], (int)t[2]);
        rr2 = __hiloint2double((int)t2[1], (int)t2[0]);
      }
    }
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    for (int j = 0; j < S.nch; ++j)
    {
      const int c = S.c0 + j;
      int fl = S.fl;
      uint4v b0 = S.b0, b1 = S.b1, q0 = {0u, 0u, 0u, 0u}, q1 = {0u, 0u, 0u, 0u};
      if (j > 0) // (a slice of more than 16 blocks per node: P2 / P3, irregular meshes)
      {
        fl = __builtin_amdgcn_readfirstlane(p_flags[c]);
        const uint4v* __restrict__ bp = reinterpret_cast<const uint4v*>(p_bcode + (int64_t)c * 1024) + 2 * lane;
        b0 = bk_ld<NT>(bp);
        b1 = bk_ld<NT>(bp + 1);
      }
      if (!(fl & 0x100))
      {
        const uint4v* __restrict__ cp = reinterpret_cast<const uint4v*>(p_ccode + (int64_t)c * 1024) + 2 * lane;
        q0 = bk_ld<NT>(cp);
        q1 = bk_ld<NT>(cp + 1);
      }
      const int width = fl & 31;
      const bool affine = (fl & 0x100) != 0;
      const int32_t* __restrict__ mb = p_meta + (int64_t)c * BK_META + (affine ? BK_SLOTS : 0);
      int base[BK_SLOTS];
#pragma unroll
      for (int e = 0; e < BK_SLOTS; ++e)
        base[e] = mb[e];
      // G slots at a time (8; form 2: 4 -- its table rows are in registers too): their x values (and rows) requested, then block
      // by block the nine values, nine mul + add in column order
      auto group = [&](auto stag, auto gtag, bool ahead) {
        constexpr int START = decltype(stag)::value, G = decltype(gtag)::value;
        double x0[G], x1[G], x2[G];
        uint4v rw0[FORM == 2 ? G : 1];
        unsigned rw1[FORM == 2 ? G : 1];
#pragma unroll
        for (int f = 0; f < G; ++f)
        {
          const int e = START + f;
          x0[f] = x1[f] = x2[f] = 0.0;
          if (FORM == 2)
          {
            rw0[f] = uint4v{0u, 0u, 0u, 0u};
            rw1[f] = 0u;
            if (e < width)
            {
              const uint16_t* __restrict__ rp = p_rows16 + (size_t)bk_code16(b0, b1, e) * 16u;
              rw0[f] = *reinterpret_cast<const uint4v*>(rp);
              rw1[f] = *reinterpret_cast<const unsigned*>(rp + 8);
            }
          }
          if (e < width)
          {
            const int col = base[e] + (affine ? lane : (int)bk_code16(q0, q1, e));
            const int off = col * 24;
#ifdef ZZZ_EXPERIMENTS
            if (a.xprobe) // timing probe (wrong results): x as three component planes, three dense 8-B loads per block
            {
              const int plane = a.nnodes * 8;
              const auto w0 = __builtin_amdgcn_raw_buffer_load_b64(rs_x, col * 8, 0, 0);
              const auto w1 = __builtin_amdgcn_raw_buffer_load_b64(rs_x, col * 8 + plane, 0, 0);
              const auto w2 = __builtin_amdgcn_raw_buffer_load_b64(rs_x, col * 8 + 2 * plane, 0, 0);
              x0[f] = __hiloint2double((int)w0[1], (int)w0[0]);
              x1[f] = __hiloint2double((int)w1[1], (int)w1[0]);
              x2[f] = __hiloint2double((int)w2[1], (int)w2[0]);
              continue;
            }
#endif
            const auto u = __builtin_amdgcn_raw_buffer_load_b128(rs_x, off, 0, 0);
            const auto u2 = __builtin_amdgcn_raw_buffer_load_b64(rs_x, off + 16, 0, 0);
            x0[f] = __hiloint2double((int)u[1], (int)u[0]);
            x1[f] = __hiloint2double((int)u[3], (int)u[2]);
            x2[f] = __hiloint2double((int)u2[1], (int)u2[0]);
          }
        }
        if (ahead) // the next slice's stream goes out behind this slice's last gathers
        {
          __builtin_amdgcn_sched_barrier(0);
          fetch(it_next, N);
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int f = 0; f < G; ++f)
        {
          const int e = START + f;
          if (e < width)
          {
            double v0, v1, v2, v3, v4, v5, v6, v7, v8;
            if (FORM == 1)
            {
              const double* __restrict__ t = bk_lds + bk_code16(b0, b1, e) * 9u;
              v0 = t[0], v1 = t[1], v2 = t[2], v3 = t[3], v4 = t[4], v5 = t[5], v6 = t[6], v7 = t[7], v8 = t[8];
            }
            else
            {
              const uint4v r0 = rw0[f];
              const unsigned r1 = rw1[f];
              auto at = [&](unsigned off) { return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(bk_lds) + off); };
              v0 = at(r0.x & 0xffffu), v1 = at(r0.x >> 16), v2 = at(r0.y & 0xffffu), v3 = at(r0.y >> 16), v4 = at(r0.z & 0xffffu);
              v5 = at(r0.z >> 16), v6 = at(r0.w & 0xffffu), v7 = at(r0.w >> 16), v8 = at(r1 & 0xffffu);
            }
            a0 += v0 * x0[f];
            a0 += v1 * x1[f];
            a0 += v2 * x2[f];
            a1 += v3 * x0[f];
            a1 += v4 * x1[f];
            a1 += v5 * x2[f];
            a2 += v6 * x0[f];
            a2 += v7 * x1[f];
            a2 += v8 * x2[f];
          }
        }
      };
      const bool last = j + 1 == S.nch;
      using I0 = std::integral_constant<int, 0>;
      using I4 = std::integral_constant<int, 4>;
      using I8 = std::integral_constant<int, 8>;
      using I12 = std::integral_constant<int, 12>;
      if (FORM == 1)
      {
        group(I0(), I8(), last && width <= 8);
        if (width > 8)
          group(I8(), I8(), last);
      }
      else
      {
        group(I0(), I4(), last && width <= 4);
        if (width > 4)
          group(I4(), I4(), last && width <= 8);
        if (width > 8)
          group(I8(), I4(), last && width <= 12);
        if (width > 12)
          group(I12(), I4(), last);
      }
    }
    if (S.nch == 0)
      fetch(it_next, N);
    if (CHEB)
    {
      // a term of the Chebyshev-Jacobi polynomial as the epilogue (ChebEpi, zzz_internal.h; the generic kernel's arithmetic
      // row by row): x is d, y the next d; DOT marks the last term (sums of <r,z> and the norm, nothing else stored but z)
      if (s * 64 + lane < a.nnodes)
      {
        const int r0 = (s * 64 + lane) * 3;
        const double sum[3] = {a0, a1, a2}, xr[3] = {xr0, xr1, xr2};
        double dn[3];
#pragma unroll
        for (int c = 0; c < 3; ++c)
        {
          const double gi = -1.0 * (epi.dinv[r0 + c] * sum[c]) + epi.g[r0 + c];
          dn[c] = epi.c1 * xr[c] + epi.c2 * gi;
          const double zi = epi.z[r0 + c] + dn[c];
          epi.z[r0 + c] = zi;
          if (DOT)
          {
            const double ri = epi.r[r0 + c];
            dot_rx += ri * zi;
            dot_nn += a.nn_is_rr ? ri * ri : zi * zi;
          }
          else
          {
            epi.g[r0 + c] = gi;
            p_y[r0 + c] = dn[c];
          }
        }
      }
    }
    else
    {
      uint4v o;
      o.x = (unsigned)__double2loint(a0), o.y = (unsigned)__double2hiint(a0);
      o.z = (unsigned)__double2loint(a1), o.w = (unsigned)__double2hiint(a1);
      __builtin_amdgcn_raw_buffer_store_b128(o, rs_y, node24, 0, 0); // (nodes beyond the last: the range check)
      uint2v o2;
      o2.x = (unsigned)__double2loint(a2), o2.y = (unsigned)__double2hiint(a2);
      __builtin_amdgcn_raw_buffer_store_b64(o2, rs_y, node24 + 16, 0, 0);
    }
    if (!CHEB && DOT && s * 64 + lane < a.nnodes)
    {
      dot += a0 * xr0;
      dot += a1 * xr1;
      dot += a2 * xr2;
      if (SR)
      {
        dot_rx += rr0 * xr0;
        dot_rx += rr1 * xr1;
        dot_rx += rr2 * xr2;
        dot_nn += a.nn_is_rr ? rr0 * rr0 : xr0 * xr0;
        dot_nn += a.nn_is_rr ? rr1 * rr1 : xr1 * xr1;
        dot_nn += a.nn_is_rr ? rr2 * rr2 : xr2 * xr2;
      }
    }
  };
  {
    Stage A, B;
    // consecutive slices of the XCD's eighth go to consecutive WORKGROUPS (= CUs), a wavefront's next slice lies 16 x that
    // many on: the remainder of slices / wavefronts then spreads over the CUs one by one (with consecutive slices in one
    // workgroup a few CUs took a whole extra round of sixteen slices: 96 against 81 at C4 -- and the CU's LDS is the bound)
    int64_t it = lo + (blockIdx.x >> 3) + (int64_t)wgs_in_xcd * wv;
    fetch(it, A);
    while (A.s >= 0)
    {
      body(A, it + stride, B);
      it += stride;
      if (B.s < 0)
        break;
      body(B, it + stride, A);
      it += stride;
    }
  }
  if (DOT)
  {
    const double sres = CHEB ? 0.0 : block_reduce_sum(dot, red);
    double s1 = 0.0, s2 = 0.0;
    if (SR || CHEB)
    {
      s1 = block_reduce_sum(dot_rx, red);
      s2 = block_reduce_sum(dot_nn, red);
    }
    if (threadIdx.x == 0)
    {
      if (!CHEB)
        a.partials[blockIdx.x] = sres;
      if (SR || CHEB)
      {
        a.partials[a.pstride + blockIdx.x] = s1;
        a.partials[2 * a.pstride + blockIdx.x] = s2;
      }
    }
  }
}

// ---- host side ---------------------------------------------------------------------------------------------------------
// The block-row form of a freshly assembled matrix of block size 3 (called at the stream's first use, behind the dictionaries).
// Declined (bk_on stays false, nothing else changes): another block size, sorted rows, more distinct blocks than the table
// holds, columns beyond 16-bit codes, ZZZ_SELLP_BLK=0.
// Set-up at C4 (1.33 M nodes): k_bk_insert 1.95 ms (the one walk over the values: lane per node, 1 080 B apart), k_bk_number
// 0.37 ms, k_bk_fill 0.50 ms.
int sellp_blk_build(zzz_ctx* ctx)
{
  ctx->bk_on = false;
  if (!ctx->sellp_blk || ctx->bs != 3 || ctx->sp_sorted || ctx->nrows <= 0 || ctx->nrows % 3 != 0)
    return ZZZ_OK;
  if ((double)(ctx->n_owned + ctx->n_ghost) * 24.0 >= 2147483647.0)
    return ZZZ_OK; // (32-bit byte offsets into x)
  // ZZZ_SELLP_BLK: 0 never, 1 (default) from 100 000 nodes on -- below that the generic product is as fast or faster (a few
  // slices per wavefront, the table copy per workgroup: 30^3 sub-cubes 13.7 against 8.9 us, 50^3 15.6 against 16.6), 2 always
  if (ctx->sellp_blk == 1 && ctx->nrows / 3 < 100000)
    return ZZZ_OK;
  hipStream_t s = ctx->stream;
  const int nnodes = (int)(ctx->nrows / 3);
  const int64_t nsl = ((int64_t)nnodes + 63) / 64;
  DevBuf<int32_t>& info = ctx->bk_info;
  ZZZ_HIP(ctx, info.reserve(8));
  ZZZ_HIP(ctx, hipMemsetAsync(info.p, 0, 8 * sizeof(int32_t), s));
  ZZZ_HIP(ctx, ctx->bk_nch.alloc((size_t)nsl + 1));
  ZZZ_HIP(ctx, ctx->bk_c0.alloc((size_t)nsl + 1));
  ZZZ_HIP(ctx, ctx->bk_hash_tag.alloc((size_t)BK_HASH));
  ZZZ_HIP(ctx, ctx->bk_hash_owner.alloc((size_t)BK_HASH));
  ZZZ_HIP(ctx, ctx->bk_slot_code.alloc((size_t)BK_HASH));
  ZZZ_HIP(ctx, ctx->bk_tab.alloc((size_t)BK_CODE_MAX * 9));
  ZZZ_HIP(ctx, ctx->bk_gflag.alloc((size_t)nsl));
  ZZZ_HIP(ctx, ctx->bk_hash_tag2.alloc((size_t)BK_HASH));
  ZZZ_HIP(ctx, ctx->bk_park.alloc((size_t)(ctx->nnz / 9) + 1));
  ZZZ_HIP(ctx, hipMemsetAsync(ctx->bk_hash_tag.p, 0, sizeof(unsigned long long) * BK_HASH, s));
  ZZZ_HIP(ctx, hipMemsetAsync(ctx->bk_hash_tag2.p, 0, sizeof(unsigned long long) * BK_HASH, s));
  ZZZ_HIP(ctx, hipMemsetAsync(ctx->bk_hash_owner.p, 0xff, sizeof(unsigned long long) * BK_HASH, s));
  ZZZ_HIP(ctx, hipMemsetAsync(ctx->bk_nch.p + nsl, 0, sizeof(int32_t), s)); // closes the scan
  const unsigned long long* vals = reinterpret_cast<const unsigned long long*>(ctx->vals.p);
  const unsigned grid = (unsigned)std::min<int64_t>((nsl + 3) / 4, 256 * 16);
  hipLaunchKernelGGL(k_bk_insert, dim3(grid), dim3(256), 0, s, ctx->rowptr.p, vals, nnodes, nsl, ctx->bk_hash_tag.p,
                     ctx->bk_hash_tag2.p, ctx->bk_hash_owner.p, ctx->bk_park.p, ctx->bk_nch.p, info.p, BK_CODE_MAX);
  hipLaunchKernelGGL(k_bk_number, dim3(1), dim3(1024), 0, s, ctx->bk_hash_tag.p, ctx->bk_hash_owner.p, ctx->rowptr.p, vals,
                     ctx->bk_slot_code.p, reinterpret_cast<unsigned long long*>(ctx->bk_tab.p), info.p);
  {
    size_t tb = 0;
    ZZZ_HIP(ctx, rocprim::exclusive_scan(nullptr, tb, ctx->bk_nch.p, ctx->bk_c0.p, (int32_t)0, (size_t)nsl + 1, rocprim::plus<int32_t>(), s));
    ZZZ_HIP(ctx, ctx->scr_tmp.grow_keep(tb, ctx->retired));
    ZZZ_HIP(ctx, rocprim::exclusive_scan(ctx->scr_tmp.p, tb, ctx->bk_nch.p, ctx->bk_c0.p, (int32_t)0, (size_t)nsl + 1, rocprim::plus<int32_t>(), s));
  }
  ZZZ_HIP(ctx, hipGetLastError());
  int32_t h[8];
  int32_t total = 0;
  ZZZ_HIP(ctx, hipMemcpyAsync(h, info.p, sizeof(h), hipMemcpyDeviceToHost, s));
  ZZZ_HIP(ctx, hipMemcpyAsync(&total, ctx->bk_c0.p + nsl, sizeof(total), hipMemcpyDeviceToHost, s));
  ZZZ_HIP(ctx, hipStreamSynchronize(s));
  if (h[1] || h[2] <= 0 || h[2] > BK_CODE_MAX || total <= 0)
    return ZZZ_OK;
  ctx->bk_form = h[2] <= BK_TAB_MAX ? 1 : 2;
  ctx->bk_ndict = 0;
  if (ctx->bk_form == 2)
  {
    // more blocks than the LDS table holds: their distinct VALUES into a dictionary for LDS, the rows as offsets into it
    ZZZ_HIP(ctx, ctx->bk_vset.alloc((size_t)BK_VAL_HASH));
    ZZZ_HIP(ctx, ctx->bk_vcode.alloc((size_t)BK_VAL_HASH));
    ZZZ_HIP(ctx, ctx->bk_vdict.alloc((size_t)BK_VAL_MAX));
    ZZZ_HIP(ctx, ctx->bk_rows16.alloc((size_t)h[2] * 16));
    ZZZ_HIP(ctx, hipMemsetAsync(ctx->bk_vset.p, 0xff, sizeof(unsigned long long) * BK_VAL_HASH, s));
    const unsigned long long* tabb = reinterpret_cast<const unsigned long long*>(ctx->bk_tab.p);
    hipLaunchKernelGGL(k_bk_val_insert, dim3((unsigned)std::min<int64_t>(((int64_t)h[2] * 9 + 255) / 256, 2048)), dim3(256), 0, s, tabb,
                       h[2] * 9, ctx->bk_vset.p, info.p);
    hipLaunchKernelGGL(k_bk_val_number, dim3(1), dim3(1024), 0, s, ctx->bk_vset.p, ctx->bk_vcode.p,
                       reinterpret_cast<unsigned long long*>(ctx->bk_vdict.p), info.p);
    hipLaunchKernelGGL(k_bk_rows16, dim3((unsigned)std::min<int64_t>((h[2] + 255) / 256, 2048)), dim3(256), 0, s, tabb, h[2],
                       ctx->bk_vset.p, ctx->bk_vcode.p, ctx->bk_rows16.p, info.p);
    ZZZ_HIP(ctx, hipGetLastError());
    int32_t h2[8];
    ZZZ_HIP(ctx, hipMemcpyAsync(h2, info.p, sizeof(h2), hipMemcpyDeviceToHost, s));
    ZZZ_HIP(ctx, hipStreamSynchronize(s));
    if (h2[1] || h2[3] <= 0 || h2[3] > BK_VAL_MAX)
      return ZZZ_OK;
    ctx->bk_ndict = h2[3];
  }
  // (the chunk index times 1024 codes stays below 2^31 elements only as int64: the kernels index with 64 bits)
  ZZZ_HIP(ctx, ctx->bk_desc.alloc(2 * (size_t)nsl + 2));
  ZZZ_HIP(ctx, ctx->bk_meta.alloc((size_t)total * BK_META));
  ZZZ_HIP(ctx, ctx->bk_flags.alloc((size_t)total + 1));
  ZZZ_HIP(ctx, ctx->bk_code.alloc((size_t)total * 1024));
  ZZZ_HIP(ctx, ctx->bk_ccode.alloc((size_t)total * 1024));
  hipLaunchKernelGGL(k_bk_fill, dim3(grid), dim3(256), 0, s, ctx->rowptr.p, ctx->cols.p, ctx->bk_park.p, nnodes, (int)ctx->n_owned, nsl,
                     ctx->bk_c0.p, ctx->bk_nch.p, ctx->bk_slot_code.p, reinterpret_cast<int2*>(ctx->bk_desc.p), ctx->bk_meta.p,
                     ctx->bk_flags.p, ctx->bk_code.p, ctx->bk_ccode.p, ctx->bk_gflag.p, info.p);
  ZZZ_HIP(ctx, hipGetLastError());
  ZZZ_HIP(ctx, hipMemcpyAsync(h, info.p, sizeof(h), hipMemcpyDeviceToHost, s));
  ZZZ_HIP(ctx, hipStreamSynchronize(s));
  if (h[1])
    return ZZZ_OK;
  // interior / boundary slices for the halo-compute overlap of a partitioned matrix (its own lists: a slice here is 64 NODES)
  ctx->bk_n_interior = ctx->bk_n_boundary = 0;
  ctx->bk_have_split = false;
  if (ctx->n_ghost > 0 || ctx->have_group_split)
  {
    std::vector<uint8_t> gf((size_t)nsl);
    ZZZ_HIP(ctx, hipMemcpyAsync(gf.data(), ctx->bk_gflag.p, gf.size(), hipMemcpyDeviceToHost, s));
    ZZZ_HIP(ctx, hipStreamSynchronize(s));
    std::vector<int32_t> in, bd;
    for (int64_t q = 0; q < nsl; ++q)
      (gf[(size_t)q] ? bd : in).push_back((int32_t)q);
    ZZZ_HIP(ctx, ctx->bk_list_interior.alloc(in.size()));
    ZZZ_HIP(ctx, ctx->bk_list_boundary.alloc(bd.size()));
    if (!in.empty())
      ZZZ_HIP(ctx, hipMemcpyAsync(ctx->bk_list_interior.p, in.data(), in.size() * sizeof(int32_t), hipMemcpyHostToDevice, s));
    if (!bd.empty())
      ZZZ_HIP(ctx, hipMemcpyAsync(ctx->bk_list_boundary.p, bd.data(), bd.size() * sizeof(int32_t), hipMemcpyHostToDevice, s));
    ZZZ_HIP(ctx, hipStreamSynchronize(s));
    ctx->bk_n_interior = (int64_t)in.size();
    ctx->bk_n_boundary = (int64_t)bd.size();
    ctx->bk_have_split = true;
  }
  unsigned long long bytes = 0;
  memcpy(&bytes, h + 4, sizeof(bytes));
  ctx->bk_entries = h[2];
  ctx->bk_chunks = total;
  ctx->bk_slices = nsl;
  ctx->bk_bytes = (int64_t)bytes + (ctx->bk_form == 1 ? (int64_t)h[2] * 72 : (int64_t)h[2] * 32 + (int64_t)ctx->bk_ndict * 8);
  if (!ctx->bk_lds_attr)
  {
#define ZZZ_BK_ATTR(DOT, SR, NT)                                                                                                   \
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&spmv_blk3_kernel<DOT, SR, NT, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                            BK_TAB_MAX * 72)
    ZZZ_BK_ATTR(true, true, true);
    ZZZ_BK_ATTR(true, true, false);
    ZZZ_BK_ATTR(true, false, true);
    ZZZ_BK_ATTR(true, false, false);
    ZZZ_BK_ATTR(false, false, true);
    ZZZ_BK_ATTR(false, false, false);
#define ZZZ_BK_ATTRC(DOT, NT)                                                                                                      \
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&spmv_blk3_kernel<DOT, false, NT, 1, true>),                            \
                            hipFuncAttributeMaxDynamicSharedMemorySize, BK_TAB_MAX * 72)
    ZZZ_BK_ATTRC(true, true);
    ZZZ_BK_ATTRC(true, false);
    ZZZ_BK_ATTRC(false, true);
    ZZZ_BK_ATTRC(false, false);
#undef ZZZ_BK_ATTRC
#undef ZZZ_BK_ATTR
    ZZZ_HIP(ctx, hipGetLastError());
    ctx->bk_lds_attr = true;
  }
  ctx->bk_on = true;
  return ZZZ_OK;
}

// Does the block-row kernel serve a launch on this context?  (No Chebyshev epilogue, no folded all-reduce: the caller checks.)
bool sellp_blk_serves(const zzz_ctx* ctx) { return ctx->bk_on && ctx->sellp_blk && ctx->bs == 3 && !ctx->sp_sorted; }

// workgroups of a launch over `items` slices: one per CU, persistent
int sellp_blk_grid(const zzz_ctx* ctx, int64_t items)
{
  (void)ctx;
  int64_t g = (items + BK_THREADS / 64 - 1) / (BK_THREADS / 64);
  g = (g + 7) / 8 * 8;
  return (int)std::max<int64_t>(8, std::min<int64_t>(g, 256));
}

bool launch_sellp_blk(zzz_ctx* ctx, bool dot, bool nt, int grid, const double* x, double* y, double* partials, const int* stop,
                      const int32_t* list, int64_t nlist, const double* rvec, int nn_is_rr, const ChebEpi* epi)
{
  if (!sellp_blk_serves(ctx))
    return false;
  BlkArgs a;
  const bool f1 = ctx->bk_form == 1;
  a.ntab = f1 ? ctx->bk_entries : ctx->bk_ndict;
  a.nnodes = (int)(ctx->nrows / 3);
  a.nslices = (int)ctx->bk_slices;
  a.nx8 = (int)((ctx->n_owned + ctx->n_ghost) * 24);
  a.partials = partials;
  a.stop_flag = stop;
  a.nlist = nlist;
  a.pstride = SPMV_PSTRIDE;
  a.nn_is_rr = nn_is_rr;
#ifdef ZZZ_EXPERIMENTS
  a.xprobe = ctx->timing_only && getenv("ZZZ_BK_XPROBE") ? atoi(getenv("ZZZ_BK_XPROBE")) : 0; // (inside zzz_spmv_time only)
#endif
  const size_t lds = f1 ? (size_t)ctx->bk_entries * 72 : (size_t)ctx->bk_ndict * 8;
  const double* tabp = f1 ? ctx->bk_tab.p : ctx->bk_vdict.p;
#define ZZZ_BK_GO3(DOT, SR, NT, FORM, CHEB, EPI)                                                                                   \
  hipLaunchKernelGGL((spmv_blk3_kernel<DOT, SR, NT, FORM, CHEB>), dim3(grid), dim3(BK_THREADS), lds, ctx->stream,                  \
                     reinterpret_cast<const int2*>(ctx->bk_desc.p), ctx->bk_meta.p, ctx->bk_flags.p, ctx->bk_code.p, ctx->bk_ccode.p,  \
                     tabp, ctx->bk_rows16.p, x, y, rvec, list, a, EPI)
#define ZZZ_BK_GO2(DOT, SR, NT, FORM) ZZZ_BK_GO3(DOT, SR, NT, FORM, false, ChebEpi())
#define ZZZ_BK_GO(DOT, SR, NT)                                                                                                     \
  do                                                                                                                               \
  {                                                                                                                                \
    if (f1)                                                                                                                        \
      ZZZ_BK_GO2(DOT, SR, NT, 1);                                                                                                  \
    else                                                                                                                           \
      ZZZ_BK_GO2(DOT, SR, NT, 2);                                                                                                  \
  } while (0)
#define ZZZ_BK_GOC(DOT, NT)                                                                                                        \
  do                                                                                                                               \
  {                                                                                                                                \
    if (f1)                                                                                                                        \
      ZZZ_BK_GO3(DOT, false, NT, 1, true, *epi);                                                                                   \
    else                                                                                                                           \
      ZZZ_BK_GO3(DOT, false, NT, 2, true, *epi);                                                                                   \
  } while (0)
  if (epi)
  {
    if (dot)
    {
      if (nt)
        ZZZ_BK_GOC(true, true);
      else
        ZZZ_BK_GOC(true, false);
    }
    else
    {
      if (nt)
        ZZZ_BK_GOC(false, true);
      else
        ZZZ_BK_GOC(false, false);
    }
  }
  else if (dot && rvec)
  {
    if (nt)
      ZZZ_BK_GO(true, true, true);
    else
      ZZZ_BK_GO(true, true, false);
  }
  else if (dot)
  {
    if (nt)
      ZZZ_BK_GO(true, false, true);
    else
      ZZZ_BK_GO(true, false, false);
  }
  else
  {
    if (nt)
      ZZZ_BK_GO(false, false, true);
    else
      ZZZ_BK_GO(false, false, false);
  }
#undef ZZZ_BK_GOC
#undef ZZZ_BK_GO3
#undef ZZZ_BK_GO2
#undef ZZZ_BK_GO
  return true;
}
ZZZ_PRELOAD_TU(sellp_blk)
} // namespace zzz

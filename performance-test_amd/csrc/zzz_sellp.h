// Shared declarations of the operator-stream translation units (zzz_sellp.hip: packer, dictionaries, the generic
// product; zzz_sellp_pipe.hip: the software-pipelined product).  Not part of the ABI.
#pragma once
#include <climits>
#include <cstdint>

#include "zzz_device.h"
#include "zzz_internal.h"

namespace zzz
{
typedef double dbl2 __attribute__((ext_vector_type(2)));
typedef unsigned uint4v __attribute__((ext_vector_type(4)));
typedef int int4v __attribute__((ext_vector_type(4)));
typedef unsigned uint2v __attribute__((ext_vector_type(2)));
constexpr int SP_BLOCK = 256;
constexpr int SP_SIGMA = 512; // sorting window (rows) of the sorted form: one workgroup

// x windows (k_sp_windows)
constexpr int SP_WIN_NSEG = 24;    // segments per group at most
constexpr int SP_WIN_GAP = 8;      // gaps of up to this many columns are filled (fewer segments, a few unused slots)
constexpr int SP_WIN_WORDS = 8192; // bitmap words of k_sp_windows (32 KiB of LDS): two mesh planes of up to 131 k entries each
constexpr int SP_WIN_SPAN = (SP_WIN_WORDS - 2) * 32; // columns between a group's smallest and largest at most

// per-slice value dictionaries (k_sp_sd_build)
constexpr int SD_SLOTS = 2048, SD_MAX = 1024;

// Column-code class of a chunk, two bits per chunk in the slice's mode word (sp_smode: chunk j of a slice at bits 2 j, 2 j + 1;
// written by the packers from emit_chunk's mode).  What the pipelined product has to LOAD per lane for the chunk's columns:
constexpr int SP_CLS_NONE = 0;  // affine or periodic chunk: nothing
constexpr int SP_CLS_C8 = 1;    // 8-bit codes, first half of the chunk's code block
constexpr int SP_CLS_C16 = 2;   // 16-bit codes, the chunk's code block
constexpr int SP_CLS_C8T = 3;   // 8-bit codes in the free tail of the chunk's value block
constexpr int SP_SMODE_CHUNKS = 32; // chunks per slice the mode word describes (longer slices: the generic product)

// tile index for (workgroup b, step i): XCD x = b % 8 owns items [x*T/8, (x+1)*T/8)   (as in zzz_spmv.hip)
__device__ inline int64_t sp_xcd_item(int64_t n, int b, int nb, int i)
{
  const int xcd = b & 7;
  const int64_t lo = n * xcd / 8, hi = n * (xcd + 1) / 8;
  const int wg_in_xcd = b >> 3, n_in_xcd = (nb + 7 - xcd) >> 3;
  const int64_t t = lo + wg_in_xcd + (int64_t)i * n_in_xcd;
  return t < hi ? t : -1;
}

// x[col] with a 32-bit byte offset: one shift per gather instead of a 64-bit address computation
// (the launcher guarantees 8 * ncols < 2^32)
__device__ inline double gather(const double* __restrict__ x, int col)
{
  return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(x) + ((unsigned)col << 3));
}

// a grid of at most `cap` workgroups for `items` items at `per` each
inline int grid_cap(int64_t items, int per, int cap)
{
  int64_t g = (items + per - 1) / per;
  if (g > cap)
    g = cap;
  if (g < 1)
    g = 1;
  return (int)g;
}

// the stream's value dictionaries (zzz_sellp_dict.hip), built at the stream's first use after an assembly
int sp_dict_build(zzz_ctx* ctx);
int sp_sd_build(zzz_ctx* ctx);

// workgroups per CU of the product for streams of one-chunk slices (zzz_sellp_pipe.hip: two rows per lane, <= 96 registers per
// lane); sp_grid sizes the persistent grid by it
constexpr int SP_ONE_WGS = 4; // (five fit its 89 registers: 73.4 us against 70-72.5 at C2 -- more wavefronts in flight do not help)
constexpr int SP_ONE_WGS_SR = 4; // ... with the single-reduction form's extra sums
int sellp_pipe_wgs(const zzz_ctx* ctx, bool sr); // workgroups per CU if spmv_one_kernel serves the context's stream; 0: it does not
int sellp_pairs_build(zzz_ctx* ctx); // marks the affine slice pairs of a stream of one-chunk slices (spmv_one_kernel)
bool launch_sellp_pipe(zzz_ctx* ctx, bool dot, bool nt, int grid, const double* x, double* y, double* partials, const int* stop,
                       const int32_t* group_list, int64_t nlist, const double* rvec, int nn_is_rr);
// special forms (block rows, block windows): build what applies; true if one of them serves the products of this matrix
bool sellp_special_build(zzz_ctx* ctx);
int sell_pack_generic(zzz_ctx* ctx); // the generic operator stream of the current values (zzz_sellp_pack.hip)
int sellp_need_generic(zzz_ctx* ctx); // ... packed now if a launch needs it and sell_update left it out
// block-window form for long scalar rows (zzz_sellp_win.hip)
int sellp_win_build(zzz_ctx* ctx);
bool sellp_win_serves(const zzz_ctx* ctx);
int sellp_win_grid(const zzz_ctx* ctx, int64_t items);
bool launch_sellp_win(zzz_ctx* ctx, bool dot, bool nt, int grid, const double* x, double* y, double* partials, const int* stop,
                      const int32_t* list, int64_t nlist, const double* rvec, int nn_is_rr, const ChebEpi* epi = nullptr);
// block-row form for block size 3 (zzz_sellp_blk.hip)
int sellp_blk_build(zzz_ctx* ctx);
bool sellp_blk_serves(const zzz_ctx* ctx);
int sellp_blk_grid(const zzz_ctx* ctx, int64_t items);
bool launch_sellp_blk(zzz_ctx* ctx, bool dot, bool nt, int grid, const double* x, double* y, double* partials, const int* stop,
                      const int32_t* list, int64_t nlist, const double* rvec, int nn_is_rr, const ChebEpi* epi = nullptr);
} // namespace zzz

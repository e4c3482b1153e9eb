// Internal locality numbering of the owned dofs.
//
// The reference hands over whatever numbering DOLFINx's partitioner and graph reordering produced
// (src/mesh.cpp:153-162,182-186: the cell partitioner and create_mesh's reordering; the function space of
// src/poisson_problem.cpp:43-44 is numbered on top of that).  The operator stream of the CG product
// (zzz_sellp.hip) and the row-gather assembly are fastest when consecutive rows are consecutive entities of one
// type along a mesh line, so the library does not depend on the caller for that: when the dofmap arrives, the
// owned block dofs are given an INTERNAL order computed from geometry alone,
//
//    key = (z cell index, in-plane / in-layer, entity dimension, direction mask, y cell index, x cell index, sub-dof)
//
// on the tensor-product lattice spanned by the distinct vertex coordinates of the mesh (any box mesh of the
// reference's create_box family, uniform or graded, whole or a partition of it).  Everything behind the C-ABI
// (connectivity, pattern, CSR of record on the device, operator stream, Krylov vectors) lives in that order; every
// index or vector that crosses the ABI is translated (zzz_api.hip), so the caller sees its own numbering only:
// CSR rows and columns in caller order with ascending columns, vectors in caller order.  Matrix and vector VALUES do
// not depend on the dof numbering (each entry is the sum of its cells' contributions in ascending cell order, and
// the cell order is the caller's); the product sums a row in ascending INTERNAL column order, i.e. it is bit-identical
// to the serial CSR loop on the internally ordered system P A P^T (zzz_internal_order_download gives P).
// A feed that is already in this order (every structured feed of this repository) yields the identity and costs
// one sort at set-up; meshes that are not a lattice are left in the caller's order unless ZZZ_RENUMBER=2 asks for
// the coordinate-bin order.  Set-up work ("ZZZ FunctionSpace" of the reference), not part of any ZZZ Assemble/Solve timer.
#include <cstring>

#include "zzz_internal.h"

#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <cstdlib>
#include <vector>

namespace zzz
{
namespace
{
constexpr int LATTICE_MAX = 1 << 18; // distinct coordinate values per axis the key has room for

__global__ void k_axis(const double* __restrict__ x, int64_t n, int a, double* __restrict__ out)
{
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = x[3 * i + a];
}

// head[i] = 1 where a new distinct value starts (sorted input, tolerance tol)
__global__ void k_heads(const double* __restrict__ v, int64_t n, double tol, int32_t* __restrict__ head)
{
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    head[i] = (i == 0 || v[i] - v[i - 1] > tol) ? 1 : 0;
}

__global__ void k_take_heads(const double* __restrict__ v, const int32_t* __restrict__ head, const int32_t* __restrict__ pos,
                             int64_t n, int cap, double* __restrict__ out)
{
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    if (head[i] && pos[i] < cap)
      out[pos[i]] = v[i];
}

struct Lattice
{
  const double* v[3]; // ascending distinct vertex coordinates per axis
  int m[3];
  double tol[3];
  // uniform bins instead of a lattice (ZZZ_RENUMBER=2 on a mesh that is not a lattice)
  int bins;
  double lo[3], h;
};

// cell index i (largest lattice value <= c) and the position of c inside the cell, 0 <= f < 1
__device__ inline void locate(const Lattice& L, int a, double c, int& i, double& f)
{
  if (L.bins)
  {
    const double t = (c - L.lo[a]) / L.h;
    int b = (int)floor(t + 1e-9);
    b = b < 0 ? 0 : (b >= L.bins ? L.bins - 1 : b);
    i = b;
    f = 0.0;
    return;
  }
  const double* v = L.v[a];
  int lo = 0, hi = L.m[a] - 1; // invariant: v[lo] <= c + tol
  const double ct = c + L.tol[a];
  while (lo < hi)
  {
    const int mid = (lo + hi + 1) >> 1;
    if (v[mid] <= ct)
      lo = mid;
    else
      hi = mid - 1;
  }
  i = lo;
  f = 0.0;
  if (lo + 1 < L.m[a])
  {
    const double w = v[lo + 1] - v[lo];
    f = (c - v[lo]) / w;
    if (f < 1e-6)
      f = 0.0;
  }
}

// One thread per (cell, local dof): the geometric key of that dof; every cell of a dof computes the same key, so the
// racing 8-byte stores all carry one value.  Reference nodes of the gll_warped P1..P3 tetrahedron in Basix's entity
// order (src/poisson_problem.cpp:35-38 [EXT]): vertices; edges (2,3) (1,3) (1,2) (0,3) (0,2) (0,1) with sub-dofs at
// (1 -+ 1/sqrt 5)/2 (P3) or 1/2 (P2) from the edge's first local vertex; faces (1,2,3) (0,2,3) (0,1,3) (0,1,2) at the centroid.
__global__ void k_dof_keys(const double* __restrict__ x, const int32_t* __restrict__ cell_verts,
                           const int32_t* __restrict__ cell_dofs, int64_t ncells, int nd, int order, int32_t n_owned,
                           Lattice L, unsigned long long* __restrict__ keys)
{
  const int EV[6][2] = {{2, 3}, {1, 3}, {1, 2}, {0, 3}, {0, 2}, {0, 1}};
  const int FV[4][3] = {{1, 2, 3}, {0, 2, 3}, {0, 1, 3}, {0, 1, 2}};
  const int npe = order - 1;
  const int64_t total = ncells * nd;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x)
  {
    const int32_t d = cell_dofs[t];
    if (d >= n_owned)
      continue;
    const int64_t c = t / nd;
    const int i = (int)(t - c * nd);
    double lam[4] = {0, 0, 0, 0};
    int dim;
    if (i < 4)
    {
      lam[i] = 1.0;
      dim = 0;
    }
    else if (i < 4 + 6 * npe)
    {
      const int e = (i - 4) / npe, s = (i - 4) % npe;
      const double tt = order == 2 ? 0.5 : (s == 0 ? 0.5 * (1.0 - 0.4472135954999579) : 0.5 * (1.0 + 0.4472135954999579));
      lam[EV[e][0]] = 1.0 - tt;
      lam[EV[e][1]] = tt;
      dim = 1;
    }
    else
    {
      const int f = i - 4 - 6 * npe;
      lam[FV[f][0]] = lam[FV[f][1]] = lam[FV[f][2]] = 1.0 / 3.0;
      dim = 2;
    }
    double X[3] = {0, 0, 0};
    for (int v = 0; v < 4; ++v)
      if (lam[v] != 0.0)
      {
        const int32_t gv = cell_verts[4 * c + v];
        for (int a = 0; a < 3; ++a)
          X[a] += lam[v] * x[3 * (int64_t)gv + a];
      }
    int ci[3];
    double fr[3];
    for (int a = 0; a < 3; ++a)
      locate(L, a, X[a], ci[a], fr[a]);
    int mask = 0;
    for (int a = 0; a < 3; ++a)
      if (fr[a] > 0.0)
        mask |= 1 << a;
    int sub = 0;
    if (dim == 1)
    {
      // position along the edge, counted from its lowest vertex: the fraction of any axis the edge moves along
      const int a0 = (mask & 1) ? 0 : ((mask & 2) ? 1 : 2);
      sub = fr[a0] > 0.5 + 1e-6 ? 1 : 0;
    }
    else if (dim == 2)
    {
      // centroid = anchor + (2 S1 + S2) / 3 for the face through anchor, anchor + S1, anchor + S1 + S2
      const int nhi = (fr[0] > 0.5) + (fr[1] > 0.5) + (fr[2] > 0.5);
      if (mask != 7)
      {
        const int a0 = (mask & 1) ? 0 : 1; // lowest axis of the face's plane: S1 there <=> sub 0
        sub = fr[a0] > 0.5 ? 0 : 1;
      }
      else if (nhi == 1)
        sub = fr[0] > 0.5 ? 0 : (fr[1] > 0.5 ? 1 : 2);
      else
        sub = 3 + (fr[0] < 0.5 ? 0 : (fr[1] < 0.5 ? 1 : 2));
    }
    const unsigned long long layer = (mask & 4) ? 1 : 0;
    const unsigned long long type = (unsigned long long)(dim * 8 + mask); // 0..23
    const unsigned long long key = ((unsigned long long)ci[2] << 45) | (layer << 44) | (type << 39)
                                   | ((unsigned long long)ci[1] << 21) | ((unsigned long long)ci[0] << 3) | (unsigned long long)sub;
    keys[d] = key;
  }
}

// Key of a cell on the vertex lattice: (simplex type, z, y, x of its lattice cube).  The type is read off the geometry: the
// four corners of the cube the vertices sit on, as 3-bit codes (x = 1, y = 2, z = 4) in ascending order -- for the six
// Kuhn simplices of create_box that is the order (0,1,3,7) (0,1,5,7) (0,2,3,7) (0,2,6,7) (0,4,5,7) (0,4,6,7), the
// structured feed's own.  A cell that is not inside one lattice cube gets no key (flag).
__global__ void k_cell_keys(const double* __restrict__ x, const int32_t* __restrict__ cell_verts, int64_t ncells, Lattice L,
                            unsigned long long* __restrict__ keys, int32_t* __restrict__ flag)
{
  for (int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; c < ncells; c += (int64_t)gridDim.x * blockDim.x)
  {
    int idx[4][3];
    bool off_lattice = false;
    for (int v = 0; v < 4; ++v)
    {
      const int32_t gv = cell_verts[4 * c + v];
      for (int a = 0; a < 3; ++a)
      {
        double f;
        locate(L, a, x[3 * (int64_t)gv + a], idx[v][a], f);
        off_lattice |= f != 0.0;
      }
    }
    int lo[3];
    for (int a = 0; a < 3; ++a)
      lo[a] = min(min(idx[0][a], idx[1][a]), min(idx[2][a], idx[3][a]));
    int code[4];
    for (int v = 0; v < 4; ++v)
    {
      code[v] = 0;
      for (int a = 0; a < 3; ++a)
      {
        const int d = idx[v][a] - lo[a];
        off_lattice |= d > 1;
        code[v] |= (d & 1) << a;
      }
    }
    // four values: a sorting network
    auto cswap = [](int& p, int& q) {
      const int mn = min(p, q), mx = max(p, q);
      p = mn;
      q = mx;
    };
    cswap(code[0], code[1]);
    cswap(code[2], code[3]);
    cswap(code[0], code[2]);
    cswap(code[1], code[3]);
    cswap(code[1], code[2]);
    if (off_lattice || lo[0] >= (1 << 17) || lo[1] >= (1 << 17) || lo[2] >= (1 << 17))
    {
      flag[0] = 1;
      keys[c] = ~0ull;
      continue;
    }
    const unsigned long long type = (unsigned long long)((code[0] << 9) | (code[1] << 6) | (code[2] << 3) | code[3]);
    keys[c] = (type << 51) | ((unsigned long long)lo[2] << 34) | ((unsigned long long)lo[1] << 17) | (unsigned long long)lo[0];
  }
}

// out[i][0..w) = in[perm[i]][0..w)
__global__ void k_gather_rows(const int32_t* __restrict__ in, const int32_t* __restrict__ perm, int64_t n, int w,
                              int32_t* __restrict__ out)
{
  const int64_t total = n * w;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x)
  {
    const int64_t i = t / w;
    out[t] = in[(int64_t)perm[i] * w + (t - i * w)];
  }
}

__global__ void k_not_identity(const int32_t* __restrict__ perm, int64_t n, int32_t* __restrict__ flag)
{
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    if (perm[i] != (int32_t)i)
      flag[0] = 1;
}

__global__ void k_iota(int32_t* __restrict__ v, int64_t n)
{
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    v[i] = (int32_t)i;
}

// iperm[perm[i]] = i; flag[0] |= (perm[i] != i); flag[1] |= two neighbours in the sorted order carry one key
__global__ void k_invert(const int32_t* __restrict__ perm, const unsigned long long* __restrict__ skeys, int64_t n,
                         int32_t* __restrict__ iperm, int32_t* __restrict__ flag)
{
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
  {
    const int32_t p = perm[i];
    iperm[p] = (int32_t)i;
    if (p != (int32_t)i)
      flag[0] = 1;
    if (i > 0 && skeys[i] == skeys[i - 1])
      flag[1] = 1;
  }
}

__global__ void k_translate(int32_t* __restrict__ cell_dofs, int64_t n, const int32_t* __restrict__ iperm, int32_t n_owned)
{
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
  {
    const int32_t d = cell_dofs[i];
    if (d < n_owned)
      cell_dofs[i] = iperm[d];
  }
}

int grid_of(int64_t n) { return (int)std::min<int64_t>((n + 255) / 256, 16384); }
} // namespace

static int renumber_cells(zzz_ctx* ctx, const Lattice& L);

// forgets the dof order only (the cells may already have been put into theirs)
static void renumber_clear_dofs(zzz_ctx* ctx)
{
  ctx->renumbered = false;
  ctx->perm.release();
  ctx->iperm.release();
  ctx->h_perm.clear();
  ctx->h_iperm.clear();
  ctx->csr_slot.clear();
  ctx->csr_slot.shrink_to_fit();
}

void renumber_clear(zzz_ctx* ctx)
{
  ctx->renumbered = false;
  ctx->perm.release();
  ctx->iperm.release();
  ctx->h_perm.clear();
  ctx->h_iperm.clear();
  ctx->csr_slot.clear();
  ctx->csr_slot.shrink_to_fit();
  ctx->renumber_kind = 0;
  ctx->cells_renumbered = false;
  ctx->h_cperm.clear();
  ctx->h_cperm.shrink_to_fit();
}

// Called by zzz_dofmap_upload with the caller's connectivity on the device: decides on the internal order, and when
// it is not the caller's, translates the device connectivity (owned dofs only; ghosts keep their places, which the
// forward scatter defines).
int renumber_build(zzz_ctx* ctx)
{
  renumber_clear(ctx);
  int mode = 1;
  if (const char* e = getenv("ZZZ_RENUMBER")) // 0: keep the caller's order; 1: lattice meshes (default); 2: any mesh
    mode = atoi(e);
  if (mode <= 0 || ctx->n_owned < 2)
    return ZZZ_OK;
  hipStream_t s = ctx->stream;
  const int64_t nv = ctx->nverts, nb = ctx->n_owned;

  // ---- the vertex lattice: distinct coordinate values per axis ------------------------------------------------
  DevBuf<double> col, sorted, lat[3];
  DevBuf<int32_t> head, pos;
  DevBuf<unsigned char> tmp;
  ZZZ_HIP(ctx, col.alloc((size_t)nv));
  ZZZ_HIP(ctx, sorted.alloc((size_t)nv));
  ZZZ_HIP(ctx, head.alloc((size_t)nv));
  ZZZ_HIP(ctx, pos.alloc((size_t)nv));
  size_t tb_sort = 0, tb_scan = 0;
  ZZZ_HIP(ctx, rocprim::radix_sort_keys(nullptr, tb_sort, col.p, sorted.p, (size_t)nv, 0, 64, s));
  ZZZ_HIP(ctx, rocprim::exclusive_scan(nullptr, tb_scan, head.p, pos.p, (int32_t)0, (size_t)nv, rocprim::plus<int32_t>(), s));
  ZZZ_HIP(ctx, tmp.alloc(std::max(tb_sort, tb_scan)));
  Lattice L{};
  bool lattice = true;
  double lo[3], hi[3];
  for (int a = 0; a < 3; ++a)
  {
    hipLaunchKernelGGL(k_axis, dim3(grid_of(nv)), dim3(256), 0, s, ctx->x.p, nv, a, col.p);
    ZZZ_HIP(ctx, rocprim::radix_sort_keys(tmp.p, tb_sort, col.p, sorted.p, (size_t)nv, 0, 64, s));
    ZZZ_HIP(ctx, hipMemcpyAsync(&lo[a], sorted.p, sizeof(double), hipMemcpyDeviceToHost, s));
    ZZZ_HIP(ctx, hipMemcpyAsync(&hi[a], sorted.p + nv - 1, sizeof(double), hipMemcpyDeviceToHost, s));
    ZZZ_HIP(ctx, hipStreamSynchronize(s));
    const double tol = 1e-9 * std::max(hi[a] - lo[a], 1e-300);
    hipLaunchKernelGGL(k_heads, dim3(grid_of(nv)), dim3(256), 0, s, sorted.p, nv, tol, head.p);
    ZZZ_HIP(ctx, rocprim::exclusive_scan(tmp.p, tb_scan, head.p, pos.p, (int32_t)0, (size_t)nv, rocprim::plus<int32_t>(), s));
    int32_t last_pos = 0, last_head = 0;
    ZZZ_HIP(ctx, hipMemcpyAsync(&last_pos, pos.p + nv - 1, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    ZZZ_HIP(ctx, hipMemcpyAsync(&last_head, head.p + nv - 1, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    ZZZ_HIP(ctx, hipStreamSynchronize(s));
    const int64_t m = (int64_t)last_pos + last_head;
    L.m[a] = (int)std::min<int64_t>(m, LATTICE_MAX);
    L.tol[a] = tol;
    if (m > LATTICE_MAX)
      lattice = false;
    ZZZ_HIP(ctx, lat[a].alloc((size_t)L.m[a]));
    hipLaunchKernelGGL(k_take_heads, dim3(grid_of(nv)), dim3(256), 0, s, sorted.p, head.p, pos.p, nv, L.m[a], lat[a].p);
    L.v[a] = lat[a].p;
  }
  // a lattice holds (nearly) all of its points: an unstructured cloud has as many distinct values per axis as points
  if (lattice && (double)L.m[0] * (double)L.m[1] * (double)L.m[2] > 8.0 * (double)nv)
    lattice = false;
  if (lattice)
    if (int rc = renumber_cells(ctx, L))
      return rc;
  if (!lattice)
  {
    if (mode < 2)
      return ZZZ_OK; // not a lattice: the caller's order stays
    double vol = 1.0;
    for (int a = 0; a < 3; ++a)
      vol *= std::max(hi[a] - lo[a], 1e-300);
    L.h = std::cbrt(vol / (double)std::max<int64_t>(nv, 1));
    double ext = 0.0;
    for (int a = 0; a < 3; ++a)
    {
      L.lo[a] = lo[a];
      ext = std::max(ext, hi[a] - lo[a]);
    }
    L.bins = (int)std::min<double>(std::ceil(ext / L.h) + 1.0, (double)LATTICE_MAX);
    L.h = std::max(L.h, ext / (double)(L.bins - 1 > 0 ? L.bins - 1 : 1));
  }

  // ---- keys, sort, inverse ------------------------------------------------------------------------------------
  DevBuf<unsigned long long> keys, skeys;
  DevBuf<int32_t> ident, flag;
  ZZZ_HIP(ctx, keys.alloc((size_t)nb));
  ZZZ_HIP(ctx, skeys.alloc((size_t)nb));
  ZZZ_HIP(ctx, ident.alloc((size_t)nb));
  ZZZ_HIP(ctx, flag.alloc(2));
  ZZZ_HIP(ctx, ctx->perm.alloc((size_t)nb));
  ZZZ_HIP(ctx, ctx->iperm.alloc((size_t)nb));
  ZZZ_HIP(ctx, hipMemsetAsync(keys.p, 0xff, (size_t)nb * sizeof(unsigned long long), s)); // a dof no cell touches sorts last
  ZZZ_HIP(ctx, hipMemsetAsync(flag.p, 0, 2 * sizeof(int32_t), s));
  const int64_t N = ctx->ncells * ctx->nd;
  hipLaunchKernelGGL(k_dof_keys, dim3(grid_of(N)), dim3(256), 0, s, ctx->x.p, ctx->cell_verts.p, ctx->cell_dofs.p, ctx->ncells,
                     ctx->nd, ctx->order, (int32_t)nb, L, keys.p);
  hipLaunchKernelGGL(k_iota, dim3(grid_of(nb)), dim3(256), 0, s, ident.p, nb);
  size_t tb_pairs = 0;
  ZZZ_HIP(ctx, rocprim::radix_sort_pairs(nullptr, tb_pairs, keys.p, skeys.p, ident.p, ctx->perm.p, (size_t)nb, 0, 64, s));
  ZZZ_HIP(ctx, tmp.alloc(tb_pairs));
  ZZZ_HIP(ctx, rocprim::radix_sort_pairs(tmp.p, tb_pairs, keys.p, skeys.p, ident.p, ctx->perm.p, (size_t)nb, 0, 64, s)); // stable
  hipLaunchKernelGGL(k_invert, dim3(grid_of(nb)), dim3(256), 0, s, ctx->perm.p, skeys.p, nb, ctx->iperm.p, flag.p);
  int32_t hflag[2] = {0, 0};
  ZZZ_HIP(ctx, hipMemcpyAsync(hflag, flag.p, sizeof(hflag), hipMemcpyDeviceToHost, s));
  ZZZ_HIP(ctx, hipStreamSynchronize(s));
  ZZZ_HIP(ctx, hipGetLastError());
  // two dofs with one key on a lattice: the mesh is not of the family the key describes (e.g. another splitting of
  // the sub-cubes, a curved geometry): leave the caller's order alone rather than guess
  if (lattice && hflag[1] && mode < 2)
  {
    renumber_clear_dofs(ctx);
    return ZZZ_OK;
  }
  if (!hflag[0])
  {
    renumber_clear_dofs(ctx); // the caller's numbering IS the internal order
    ctx->renumber_kind = lattice ? 1 : 2;
    return ZZZ_OK;
  }
  hipLaunchKernelGGL(k_translate, dim3(grid_of(N)), dim3(256), 0, s, ctx->cell_dofs.p, N, ctx->iperm.p, (int32_t)nb);
  ctx->h_perm.resize((size_t)nb);
  ctx->h_iperm.resize((size_t)nb);
  ZZZ_HIP(ctx, hipMemcpyAsync(ctx->h_perm.data(), ctx->perm.p, (size_t)nb * sizeof(int32_t), hipMemcpyDeviceToHost, s));
  ZZZ_HIP(ctx, hipMemcpyAsync(ctx->h_iperm.data(), ctx->iperm.p, (size_t)nb * sizeof(int32_t), hipMemcpyDeviceToHost, s));
  ZZZ_HIP(ctx, hipStreamSynchronize(s));
  ctx->renumbered = true;
  ctx->renumber_kind = lattice ? 1 : 2;
  return ZZZ_OK;
}

// The cells in the library's own order (lattice meshes only; called by renumber_build with the vertex lattice in hand).
// The assembly walks "the a-th cell of my dof" for 64 neighbouring rows at a time and the adjacency build looks for
// monotone runs (zzz_pattern.hip): both want the cells simplex type by simplex type, cube by cube -- whatever order the
// caller's mesh library left them in.  An entry of A or b is then summed in ascending INTERNAL cell order: the same
// sum as the reference's cell loop up to the order of its terms.
static int renumber_cells(zzz_ctx* ctx, const Lattice& L)
{
  hipStream_t s = ctx->stream;
  const int64_t nc = ctx->ncells;
  if (nc < 2)
    return ZZZ_OK;
  DevBuf<unsigned long long> keys, skeys;
  DevBuf<int32_t> ident, cperm, flag, tmpi;
  DevBuf<unsigned char> tmp;
  ZZZ_HIP(ctx, keys.alloc((size_t)nc));
  ZZZ_HIP(ctx, skeys.alloc((size_t)nc));
  ZZZ_HIP(ctx, ident.alloc((size_t)nc));
  ZZZ_HIP(ctx, cperm.alloc((size_t)nc));
  ZZZ_HIP(ctx, flag.alloc(2));
  ZZZ_HIP(ctx, hipMemsetAsync(flag.p, 0, 2 * sizeof(int32_t), s));
  hipLaunchKernelGGL(k_cell_keys, dim3(grid_of(nc)), dim3(256), 0, s, ctx->x.p, ctx->cell_verts.p, nc, L, keys.p, flag.p);
  hipLaunchKernelGGL(k_iota, dim3(grid_of(nc)), dim3(256), 0, s, ident.p, nc);
  size_t tb = 0;
  ZZZ_HIP(ctx, rocprim::radix_sort_pairs(nullptr, tb, keys.p, skeys.p, ident.p, cperm.p, (size_t)nc, 0, 64, s));
  ZZZ_HIP(ctx, tmp.alloc(tb));
  ZZZ_HIP(ctx, rocprim::radix_sort_pairs(tmp.p, tb, keys.p, skeys.p, ident.p, cperm.p, (size_t)nc, 0, 64, s)); // stable
  hipLaunchKernelGGL(k_not_identity, dim3(grid_of(nc)), dim3(256), 0, s, cperm.p, nc, flag.p + 1);
  int32_t hf[2] = {0, 0};
  ZZZ_HIP(ctx, hipMemcpyAsync(hf, flag.p, sizeof(hf), hipMemcpyDeviceToHost, s));
  ZZZ_HIP(ctx, hipStreamSynchronize(s));
  ZZZ_HIP(ctx, hipGetLastError());
  if (hf[0] || !hf[1])
    return ZZZ_OK; // a cell that is no lattice simplex (keep the caller's order), or the order is already this one
  const int nd = ctx->nd;
  ZZZ_HIP(ctx, tmpi.alloc((size_t)(nc * std::max(nd, 4))));
  hipLaunchKernelGGL(k_gather_rows, dim3(grid_of(nc * 4)), dim3(256), 0, s, ctx->cell_verts.p, cperm.p, nc, 4, tmpi.p);
  ZZZ_HIP(ctx, hipMemcpyAsync(ctx->cell_verts.p, tmpi.p, (size_t)(4 * nc) * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
  hipLaunchKernelGGL(k_gather_rows, dim3(grid_of(nc * nd)), dim3(256), 0, s, ctx->cell_dofs.p, cperm.p, nc, nd, tmpi.p);
  ZZZ_HIP(ctx, hipMemcpyAsync(ctx->cell_dofs.p, tmpi.p, (size_t)(nd * nc) * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
  ctx->h_cperm.resize((size_t)nc);
  ZZZ_HIP(ctx, hipMemcpyAsync(ctx->h_cperm.data(), cperm.p, (size_t)nc * sizeof(int32_t), hipMemcpyDeviceToHost, s));
  ZZZ_HIP(ctx, hipStreamSynchronize(s));
  ZZZ_HIP(ctx, hipGetLastError());
  ctx->cells_renumbered = true;
  return ZZZ_OK;
}

// ---- translation of what crosses the ABI (host side: these are set-up / inspection calls) -------------------------

// caller-ordered vector (nloc * bs entries, or n_owned * bs when owned_only) -> internal order
void to_internal(const zzz_ctx* ctx, const double* in, double* out, bool owned_only)
{
  const int bs = ctx->bs;
  const int64_t nb = ctx->n_owned, nall = owned_only ? nb : nb + ctx->n_ghost;
  const int32_t* perm = ctx->h_perm.data();
  for (int64_t i = 0; i < nall; ++i)
  {
    const int64_t src = i < nb ? perm[i] : i;
    for (int k = 0; k < bs; ++k)
      out[i * bs + k] = in[src * bs + k];
  }
}

void to_caller(const zzz_ctx* ctx, const double* in, double* out, bool owned_only)
{
  const int bs = ctx->bs;
  const int64_t nb = ctx->n_owned, nall = owned_only ? nb : nb + ctx->n_ghost;
  const int32_t* perm = ctx->h_perm.data();
  for (int64_t i = 0; i < nall; ++i)
  {
    const int64_t dst = i < nb ? perm[i] : i;
    for (int k = 0; k < bs; ++k)
      out[dst * bs + k] = in[i * bs + k];
  }
}

// The CSR of record (internal order on the device) in the caller's order: rows permuted, columns mapped back and
// sorted ascending within each row.  csr_slot[k] = position inside its (internal) row of the k-th entry of the
// caller's CSR: kept, so that values travel both ways without sorting again.
int csr_to_caller(zzz_ctx* ctx, std::vector<rp_t>& rowptr_c, int32_t* cols_c, double* vals_c, bool need_cols)
{
  const int bs = ctx->bs;
  const int64_t nrows = ctx->nrows, nnz = ctx->nnz, nb = ctx->n_owned;
  std::vector<rp_t> rp((size_t)nrows + 1);
  ZZZ_HIP(ctx, hipMemcpy(rp.data(), ctx->rowptr.p, rp.size() * sizeof(rp_t), hipMemcpyDeviceToHost));
  rowptr_c.assign((size_t)nrows + 1, 0);
  const int32_t *perm = ctx->h_perm.data(), *iperm = ctx->h_iperm.data();
  for (int64_t r = 0; r < nrows; ++r)
  {
    const int64_t ri = (int64_t)iperm[r / bs] * bs + r % bs;
    rowptr_c[(size_t)r + 1] = rowptr_c[(size_t)r] + (rp[(size_t)ri + 1] - rp[(size_t)ri]);
  }
  const bool have_slot = (int64_t)ctx->csr_slot.size() == nnz;
  if (!need_cols && !vals_c)
    return ZZZ_OK;
  std::vector<int32_t> ci;
  if (need_cols || !have_slot)
  {
    ci.resize((size_t)nnz);
    ZZZ_HIP(ctx, hipMemcpy(ci.data(), ctx->cols.p, (size_t)nnz * sizeof(int32_t), hipMemcpyDeviceToHost));
  }
  std::vector<double> vi;
  if (vals_c)
  {
    vi.resize((size_t)nnz);
    if (ctx->have_matrix)
      ZZZ_HIP(ctx, hipMemcpy(vi.data(), ctx->vals.p, (size_t)nnz * sizeof(double), hipMemcpyDeviceToHost));
    else
      std::fill(vi.begin(), vi.end(), 0.0);
  }
  if (!have_slot)
    ctx->csr_slot.resize((size_t)nnz);
  int bad = 0;
  // (serial on purpose: these are inspection / set-up calls, and a process that also holds the CPU oracle would otherwise
  // run two OpenMP runtimes -- clang's here, gcc's there -- side by side)
  {
    std::vector<std::pair<int32_t, int32_t>> tmp;
    for (int64_t r = 0; r < nrows; ++r)
    {
      const int64_t ri = (int64_t)iperm[r / bs] * bs + r % bs;
      const rp_t a = rp[(size_t)ri], len = rp[(size_t)ri + 1] - a, o = rowptr_c[(size_t)r];
      if (len > 65535)
      {
        bad = 1;
        continue;
      }
      if (!have_slot || need_cols)
      {
        tmp.resize((size_t)len);
        for (rp_t k = 0; k < len; ++k)
        {
          const int32_t c = ci[(size_t)(a + k)];
          const int32_t cb = c / bs;
          tmp[(size_t)k] = {(cb < nb ? perm[cb] : cb) * bs + c % bs, (int32_t)k};
        }
        std::sort(tmp.begin(), tmp.end());
        for (rp_t k = 0; k < len; ++k)
        {
          if (need_cols)
            cols_c[o + k] = tmp[(size_t)k].first;
          ctx->csr_slot[(size_t)(o + k)] = (uint16_t)tmp[(size_t)k].second;
        }
      }
      if (vals_c)
        for (rp_t k = 0; k < len; ++k)
          vals_c[o + k] = vi[(size_t)(a + ctx->csr_slot[(size_t)(o + k)])];
    }
  }
  if (bad)
  {
    ctx->csr_slot.clear();
    return fail(ctx, ZZZ_ERR_LIMIT, "a matrix row has more than 65535 entries");
  }
  return ZZZ_OK;
}

// values in the caller's CSR order -> internal CSR order (host buffer of nnz doubles)
int csr_values_to_internal(zzz_ctx* ctx, const double* vals_c, std::vector<double>& vals_i)
{
  std::vector<rp_t> rowptr_c;
  if ((int64_t)ctx->csr_slot.size() != ctx->nnz)
    if (int rc = csr_to_caller(ctx, rowptr_c, nullptr, nullptr, false))
      return rc;
  if ((int64_t)ctx->csr_slot.size() != ctx->nnz)
  {
    // csr_to_caller skips the slot map when nothing else was asked for: ask for the values
    std::vector<double> dummy((size_t)ctx->nnz);
    if (int rc = csr_to_caller(ctx, rowptr_c, nullptr, dummy.data(), false))
      return rc;
  }
  const int bs = ctx->bs;
  const int64_t nrows = ctx->nrows;
  std::vector<rp_t> rp((size_t)nrows + 1);
  ZZZ_HIP(ctx, hipMemcpy(rp.data(), ctx->rowptr.p, rp.size() * sizeof(rp_t), hipMemcpyDeviceToHost));
  if (rowptr_c.empty())
  {
    rowptr_c.assign((size_t)nrows + 1, 0);
    for (int64_t r = 0; r < nrows; ++r)
    {
      const int64_t ri = (int64_t)ctx->h_iperm[(size_t)(r / bs)] * bs + r % bs;
      rowptr_c[(size_t)r + 1] = rowptr_c[(size_t)r] + (rp[(size_t)ri + 1] - rp[(size_t)ri]);
    }
  }
  vals_i.resize((size_t)ctx->nnz);
  for (int64_t r = 0; r < nrows; ++r)
  {
    const int64_t ri = (int64_t)ctx->h_iperm[(size_t)(r / bs)] * bs + r % bs;
    const rp_t a = rp[(size_t)ri], len = rp[(size_t)ri + 1] - a, o = rowptr_c[(size_t)r];
    for (rp_t k = 0; k < len; ++k)
      vals_i[(size_t)(a + ctx->csr_slot[(size_t)(o + k)])] = vals_c[o + k];
  }
  return ZZZ_OK;
}
ZZZ_PRELOAD_TU(renumber)
} // namespace zzz

// C-ABI entry points of libzzz_hip.so (declared, with the reference interfaces they replace, in
// include/zzz_abi.h).  There is no CPU fallback: without a usable GPU every entry point fails.
#include "zzz_internal.h"
#include "zzz_sellp.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <numeric>

namespace zzz
{
static std::mutex g_err_mutex;
static std::string g_err;

void set_global_error(const char* msg)
{
  std::lock_guard<std::mutex> lk(g_err_mutex);
  g_err = msg;
}

int fail(zzz_ctx* ctx, int code, const char* fmt, ...)
{
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (ctx)
    ctx->err = buf;
  else
    set_global_error(buf);
  return code;
}

static int ndofs_cell(int order) { return order == 1 ? 4 : order == 2 ? 10 : order == 3 ? 20 : -1; }

template <typename T>
static int upload(zzz_ctx* ctx, DevBuf<T>& d, const T* h, size_t n, size_t pad = 0)
{
  ZZZ_HIP(ctx, d.alloc(n + pad));
  if (pad)
    ZZZ_HIP(ctx, hipMemsetAsync(d.p + n, 0, pad * sizeof(T), ctx->stream));
  if (n)
    ZZZ_HIP(ctx, hipMemcpyAsync(d.p, h, n * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
  ZZZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return ZZZ_OK;
}
// bc marker (nothing constrained yet), coefficient and Krylov vectors, reduction buffers
int alloc_problem_vectors(zzz_ctx* ctx)
{
  const size_t nv = (size_t)ctx->nloc();
  ZZZ_HIP(ctx, ctx->bc.alloc(nv));
  ZZZ_HIP(ctx, hipMemsetAsync(ctx->bc.p, 0, nv, ctx->stream));
  DevBuf<double>* vecs[] = {&ctx->b, &ctx->u, &ctx->r, &ctx->z, &ctx->p, &ctx->w, &ctx->dinv, &ctx->coeff[0], &ctx->coeff[1]};
  for (DevBuf<double>* v : vecs)
  {
    ZZZ_HIP(ctx, v->alloc(nv));
    ZZZ_HIP(ctx, hipMemsetAsync(v->p, 0, nv * sizeof(double), ctx->stream));
  }
  ZZZ_HIP(ctx, ctx->part_a.alloc(3 * 4096));
  ZZZ_HIP(ctx, ctx->part_b.alloc(4096));
  ZZZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return ZZZ_OK;
}
} // namespace zzz

using namespace zzz;

extern "C" {

int zzz_device_count(void)
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess)
    return 0;
  return n;
}

int zzz_device_memory(int device, size_t* free_bytes, size_t* total_bytes)
{
  if (!free_bytes || !total_bytes)
    return fail(nullptr, ZZZ_ERR_ARG, "zzz_device_memory: NULL argument");
  int prev = 0;
  (void)hipGetDevice(&prev);
  if (hipSetDevice(device) != hipSuccess || hipMemGetInfo(free_bytes, total_bytes) != hipSuccess)
  {
    (void)hipGetLastError();
    (void)hipSetDevice(prev);
    return fail(nullptr, ZZZ_ERR_HIP, "hipMemGetInfo failed on device %d", device);
  }
  (void)hipSetDevice(prev);
  return ZZZ_OK;
}

int zzz_ctx_create(int device, zzz_ctx** out)
{
  if (!out)
    return fail(nullptr, ZZZ_ERR_ARG, "zzz_ctx_create: out is NULL");
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return fail(nullptr, ZZZ_ERR_NO_GPU, "no HIP device available (%s); libzzz_hip has no CPU fallback",
                e == hipSuccess ? "device count 0" : hipGetErrorString(e));
  if (device < 0 || device >= n)
    return fail(nullptr, ZZZ_ERR_ARG, "device %d out of range [0, %d)", device, n);
  e = hipSetDevice(device);
  if (e != hipSuccess)
    return fail(nullptr, ZZZ_ERR_HIP, "hipSetDevice(%d): %s", device, hipGetErrorString(e));
  zzz_ctx* ctx = new zzz_ctx();
  ctx->device = device;
  e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
  if (e != hipSuccess)
  {
    delete ctx;
    return fail(nullptr, ZZZ_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e));
  }
  if (ctx->state.alloc(1) != hipSuccess || ctx->red.alloc(16) != hipSuccess
      || hipHostMalloc((void**)&ctx->h_state, 8 * sizeof(zzz::CgState), hipHostMallocDefault) != hipSuccess)
  {
    zzz_ctx_destroy(ctx);
    return fail(nullptr, ZZZ_ERR_HIP, "context allocation failed");
  }
  memset(ctx->h_state, 0, 8 * sizeof(zzz::CgState));
  // the code objects of every translation unit, now (see ZZZ_PRELOAD_TU); once per process and device is enough, and a
  // failure here is not an error: the launch that needs the unit would report it
  {
    static bool loaded[64] = {};
    if (device < 64 && !loaded[device])
    {
      loaded[device] = true;
      zzz::preload_cubegen();
      zzz::preload_renumber();
      zzz::preload_pattern();
      zzz::preload_assemble();
      zzz::preload_sellp_pack();
      zzz::preload_sellp_dict();
      zzz::preload_sellp_pipe();
      zzz::preload_sellp_blk();
      zzz::preload_sellp_win();
      zzz::preload_sellp();
      zzz::preload_spmv();
      zzz::preload_cg();
      zzz::preload_comm();
      zzz::preload_matfree();
      zzz::preload_nullspace();
      (void)hipGetLastError();
    }
  }
  // tuning knobs for A/B measurements (defaults are the measured best)
#ifdef ZZZ_EXPERIMENTS
  if (const char* e = getenv("ZZZ_SPMV_TILE")) // (tools build only: the 4096-nonzero tiles lost)
    ctx->spmv_tile = atoi(e) == 4096 ? 4096 : 2048;
#endif
  if (const char* e = getenv("ZZZ_SPMV_VARIANT"))
  {
    ctx->spmv_variant = atoi(e) & 27;
    ctx->spmv_auto = false;
  }
  if (const char* e = getenv("ZZZ_SELLP")) // operator stream: 0 off, 1 automatic, 2 natural row order, 3 sorted rows
  {
    const int v = atoi(e);
    ctx->sellp_mode = v >= 0 && v <= 4 ? v : 1; // (4: natural row order through the long-row packer whatever the row lengths: tests)
#ifdef ZZZ_EXPERIMENTS
    if (v == 5) // experiment: component-major slices for block size 3 (zzz_sellp_pack.hip: k_sp_cm)
      ctx->sellp_mode = 5;
#endif
    ctx->sellp_long_rows = v == 4;
    if (v == 4)
      ctx->sellp_mode = 2;
  }
  if (const char* e = getenv("ZZZ_SELLP_DROP"))
    ctx->sellp_drop = atoi(e) != 0;
  if (const char* e = getenv("ZZZ_SELLP_DICT"))
    ctx->sellp_dict = atoi(e);
  if (const char* e = getenv("ZZZ_SELLP_BWIN")) // long scalar rows, x from LDS windows: 0 never, 1 by size, 2 always
    ctx->sellp_bwin = atoi(e);
  if (const char* e = getenv("ZZZ_SELLP_EARLY")) // 0: generic stream at every assembly, special forms at the first product (A/B)
    ctx->sellp_early = atoi(e) != 0;
  if (const char* e = getenv("ZZZ_ASM_NODE3")) // 0: elasticity P1 matrix assembly with a thread per scalar row (A/B)
    ctx->asm_node3 = atoi(e) != 0;
  if (const char* e = getenv("ZZZ_SELLP_BLK")) // 0: block size 3 stays on the generic product (A/B against the block-row form)
    ctx->sellp_blk = atoi(e);
  if (const char* e = getenv("ZZZ_SELLP_PIPE")) // 0: the generic product kernel always (A/B against the pipelined one)
    ctx->sellp_pipe = atoi(e);
  if (const char* e = getenv("ZZZ_CG_DINV_CODES"))
    ctx->cg_dinv_codes = atoi(e);
  if (const char* e = getenv("ZZZ_SELLP_FORMS")) // A/B knob, a mask of the code-free chunk forms (default 7): 1 affine chunks
  {                                              // (column = slot base + lane), 2 one-chunk slices placed by column so that
    const int v = atoi(e);                       // short boundary rows fit the affine form, 4 periodic chunks (block size 3)
    if (!(v & 1))
      ctx->sellp_tail |= 2;
    ctx->sellp_align = (v & 2) != 0;
    ctx->sellp_periodic = (v & 4) != 0;
  }
  if (const char* e = getenv("ZZZ_SPMV_LPR")) // lanes per row of the SpMV row phase: 1, 2, 4, 8, 16
  {
    const int v = atoi(e);
    for (int sft = 0; sft <= 4; ++sft)
      if (v == (1 << sft))
        ctx->spmv_lpr_forced = sft;
  }
  if (const char* e = getenv("ZZZ_COLS16")) // 0: keep the SpMV on the int32 columns; 10..13: force the offset width
  {
    const int v = atoi(e);
    ctx->cols16_enabled = v != 0;
    if (v >= 10 && v <= 13)
      ctx->cols16_offb_forced = v;
  }
  if (const char* e = getenv("ZZZ_OVERLAP"))
    ctx->overlap = atoi(e) != 0;
  *out = ctx;
  return ZZZ_OK;
}

void zzz_ctx_destroy(zzz_ctx* ctx)
{
  if (!ctx)
    return;
  (void)hipSetDevice(ctx->device);
  if (ctx->stream)
    (void)hipStreamSynchronize(ctx->stream);
  comm_destroy(ctx);
  for (hipEvent_t ev : ctx->ev)
    (void)hipEventDestroy(ev);
  for (hipEvent_t ev : ctx->ev_halo)
    (void)hipEventDestroy(ev);
  if (ctx->sp_event)
    (void)hipEventDestroy(ctx->sp_event);
  if (ctx->h_state)
    (void)hipHostFree(ctx->h_state);
  if (ctx->adj_flag_host)
    (void)hipHostFree(ctx->adj_flag_host);
  if (ctx->stream)
    (void)hipStreamDestroy(ctx->stream);
  for (void* q : ctx->retired)
    (void)hipFree(q);
  delete ctx;
}

const char* zzz_last_error(const zzz_ctx* ctx)
{
  if (ctx)
    return ctx->err.c_str();
  std::lock_guard<std::mutex> lk(g_err_mutex);
  static thread_local std::string copy;
  copy = g_err;
  return copy.c_str();
}

int zzz_sync(zzz_ctx* ctx)
{
  if (!ctx)
    return fail(nullptr, ZZZ_ERR_ARG, "NULL context");
  ZZZ_HIP(ctx, hipSetDevice(ctx->device));
  ZZZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return ZZZ_OK;
}

#define ZZZ_ENTER(ctx)                                          \
  do                                                            \
  {                                                             \
    if (!(ctx))                                                 \
      return fail(nullptr, ZZZ_ERR_ARG, "NULL context");        \
    ZZZ_HIP(ctx, hipSetDevice((ctx)->device));                  \
  } while (0)

int zzz_mesh_upload(zzz_ctx* ctx, int64_t nverts, const double* x, int64_t ncells, const int32_t* cell_verts)
{
  ZZZ_ENTER(ctx);
  if (nverts <= 0 || ncells <= 0 || !x || !cell_verts)
    return fail(ctx, ZZZ_ERR_ARG, "zzz_mesh_upload: empty mesh or NULL array");
  if (nverts > INT32_MAX / 4 || ncells > (INT32_MAX - 8) / 20)
    return fail(ctx, ZZZ_ERR_LIMIT, "mesh too large for int32 indexing (%lld vertices, %lld cells)", (long long)nverts,
                (long long)ncells);
  for (int64_t i = 0; i < 4 * ncells; ++i)
    if (cell_verts[i] < 0 || cell_verts[i] >= nverts)
      return fail(ctx, ZZZ_ERR_ARG, "cell_verts[%lld] = %d out of range", (long long)i, cell_verts[i]);
  ctx->nverts = nverts;
  ctx->ncells = ncells;
  ctx->h_cell_verts.assign(cell_verts, cell_verts + 4 * ncells);
  int rc = upload(ctx, ctx->x, x, (size_t)(3 * nverts));
  if (rc)
    return rc;
  rc = upload(ctx, ctx->cell_verts, cell_verts, (size_t)(4 * ncells));
  ctx->have_pattern = ctx->have_matrix = false;
  ctx->xq_valid = false;
  ctx->mf.valid = ctx->mf.failed = false;
  return rc;
}

int zzz_dofmap_upload(zzz_ctx* ctx, int order, int bs, const int32_t* cell_dofs, int64_t n_owned, int64_t n_ghost)
{
  ZZZ_ENTER(ctx);
  const int nd = ndofs_cell(order);
  if (nd < 0) // form_poisson_a.at(order - 1) throws std::out_of_range in the reference
    return fail(ctx, ZZZ_ERR_ARG, "order %d not supported (1..3)", order);
  if (bs != 1 && bs != 3)
    return fail(ctx, ZZZ_ERR_ARG, "block size %d not supported (1 or 3)", bs);
  if (ctx->ncells == 0)
    return fail(ctx, ZZZ_ERR_ARG, "zzz_dofmap_upload before zzz_mesh_upload");
  if (n_owned <= 0 || n_ghost < 0 || !cell_dofs)
    return fail(ctx, ZZZ_ERR_ARG, "zzz_dofmap_upload: bad sizes");
  const int64_t nloc = n_owned + n_ghost;
  if (nloc * bs > INT32_MAX - 8)
    return fail(ctx, ZZZ_ERR_LIMIT, "%lld local scalar dofs exceed int32", (long long)(nloc * bs));
  for (int64_t i = 0; i < ctx->ncells * nd; ++i)
    if (cell_dofs[i] < 0 || cell_dofs[i] >= nloc)
      return fail(ctx, ZZZ_ERR_ARG, "cell_dofs[%lld] = %d out of range", (long long)i, cell_dofs[i]);
  ctx->order = order;
  ctx->bs = bs;
  ctx->nd = nd;
  ctx->n_owned = n_owned;
  ctx->n_ghost = n_ghost;
  ctx->h_cell_dofs.assign(cell_dofs, cell_dofs + ctx->ncells * nd);
  int rc = upload(ctx, ctx->cell_dofs, cell_dofs, (size_t)(ctx->ncells * nd));
  if (rc)
    return rc;
  // facet mask: no exterior facets until uploaded
  ZZZ_HIP(ctx, ctx->facet_mask.alloc((size_t)ctx->ncells));
  ZZZ_HIP(ctx, hipMemsetAsync(ctx->facet_mask.p, 0, (size_t)ctx->ncells, ctx->stream));
  ctx->nfacets = 0;
  rc = alloc_problem_vectors(ctx);
  if (rc)
    return rc;
  ctx->have_bc = false;
  ctx->have_coeff[0] = ctx->have_coeff[1] = false;
  ctx->have_pattern = ctx->have_matrix = false;
  ctx->xq_valid = false;
  ctx->mf.valid = ctx->mf.failed = false;
  ctx->near_null_ld = 0;
  // the library's own locality order of the owned dofs (zzz_renumber.hip): from here on the device connectivity is in
  // internal numbering and every entry point below translates at the boundary
  rc = renumber_build(ctx);
  if (rc)
    return rc;
  ctx->adj_runs_n = -1;
  pattern_reserve(ctx);
  return ensure_p1_coords(ctx); // function-space data (dof coordinates), not assembly work
}

int zzz_bc_upload(zzz_ctx* ctx, int64_t nbc, const int32_t* bc_dofs)
{
  ZZZ_ENTER(ctx);
  if (ctx->order == 0)
    return fail(ctx, ZZZ_ERR_ARG, "zzz_bc_upload before zzz_dofmap_upload");
  if (nbc < 0 || (nbc > 0 && !bc_dofs))
    return fail(ctx, ZZZ_ERR_ARG, "zzz_bc_upload: bad arguments");
  std::vector<uint8_t> m((size_t)ctx->nloc(), 0);
  for (int64_t i = 0; i < nbc; ++i)
  {
    if (bc_dofs[i] < 0 || bc_dofs[i] >= ctx->nloc())
      return fail(ctx, ZZZ_ERR_ARG, "bc_dofs[%lld] = %d out of range", (long long)i, bc_dofs[i]);
    int64_t d = bc_dofs[i];
    if (ctx->renumbered && d / ctx->bs < ctx->n_owned)
      d = (int64_t)ctx->h_iperm[(size_t)(d / ctx->bs)] * ctx->bs + d % ctx->bs;
    m[(size_t)d] = 1;
  }
  ZZZ_HIP(ctx, hipMemcpyAsync(ctx->bc.p, m.data(), m.size(), hipMemcpyHostToDevice, ctx->stream));
  ZZZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ctx->have_bc = true;
  ctx->mf.valid = ctx->mf.failed = false; // the plan of the matrix-free action carries the Dirichlet markers of its dof lists
  return ZZZ_OK;
}

int zzz_facets_upload(zzz_ctx* ctx, int64_t nfacets, const int32_t* pairs)
{
  ZZZ_ENTER(ctx);
  if (ctx->order == 0)
    return fail(ctx, ZZZ_ERR_ARG, "zzz_facets_upload before zzz_dofmap_upload");
  if (nfacets < 0 || (nfacets > 0 && !pairs))
    return fail(ctx, ZZZ_ERR_ARG, "zzz_facets_upload: bad arguments");
  std::vector<uint8_t> m((size_t)ctx->ncells, 0);
  for (int64_t i = 0; i < nfacets; ++i)
  {
    const int32_t c = pairs[2 * i], f = pairs[2 * i + 1];
    if (c < 0 || c >= ctx->ncells || f < 0 || f > 3)
      return fail(ctx, ZZZ_ERR_ARG, "facet %lld = (%d, %d) out of range", (long long)i, c, f);
    m[c] |= (uint8_t)(1u << f);
  }
  if (ctx->cells_renumbered) // the device holds the cells in the library's order: mask of internal cell i = caller's cperm[i]
  {
    std::vector<uint8_t> mi(m.size());
    for (size_t i = 0; i < m.size(); ++i)
      mi[i] = m[(size_t)ctx->h_cperm[i]];
    m.swap(mi);
  }
  ZZZ_HIP(ctx, hipMemcpyAsync(ctx->facet_mask.p, m.data(), m.size(), hipMemcpyHostToDevice, ctx->stream));
  ZZZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ctx->nfacets = nfacets;
  return ZZZ_OK;
}

int zzz_coeff_upload(zzz_ctx* ctx, int which, const double* values)
{
  ZZZ_ENTER(ctx);
  if (ctx->order == 0)
    return fail(ctx, ZZZ_ERR_ARG, "zzz_coeff_upload before zzz_dofmap_upload");
  if ((which != ZZZ_COEFF_F && which != ZZZ_COEFF_G) || !values)
    return fail(ctx, ZZZ_ERR_ARG, "zzz_coeff_upload: bad arguments");
  std::vector<double> tmp;
  if (ctx->renumbered)
  {
    // coefficient G is scalar-valued whatever the block size of the space (src/poisson_problem.cpp:96-106): both are
    // stored with nloc * bs entries and Poisson has bs = 1, so one layout serves
    tmp.resize((size_t)ctx->nloc());
    to_internal(ctx, values, tmp.data(), false);
    values = tmp.data();
  }
  ZZZ_HIP(ctx, hipMemcpyAsync(ctx->coeff[which].p, values, (size_t)ctx->nloc() * sizeof(double), hipMemcpyHostToDevice,
                              ctx->stream));
  ZZZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ctx->have_coeff[which] = true;
  return ZZZ_OK;
}

static int pattern_build_host(zzz_ctx* ctx);

int zzz_csr_pattern_build(zzz_ctx* ctx)
{
  ZZZ_ENTER(ctx);
  if (ctx->order == 0)
    return fail(ctx, ZZZ_ERR_ARG, "zzz_csr_pattern_build before zzz_dofmap_upload");
  ctx->have_pattern = ctx->have_matrix = false;
  ++ctx->pattern_version;
  ctx->bw_on = false;
  ctx->have_adj_li = false;
  ctx->have_asm_pos = false;
  ++ctx->mat_version;
  const char* mode = getenv("ZZZ_PATTERN"); // "host": the C++ host builder (kept for meshes whose
                                            // vertex valence exceeds the device kernel's LDS budget)
  int rc;
  if (mode && strcmp(mode, "host") == 0)
    rc = pattern_build_host(ctx);
  else
  {
    bool fallback = false;
    rc = pattern_build_device(ctx, &fallback);
    if (rc && fallback)
      rc = pattern_build_host(ctx);
  }
  if (rc)
    return rc;
  rc = build_adjT(ctx);
  if (!rc)
    rc = ensure_tables(ctx); // reference tensors of the element: resident before the assembly timers start
  if (rc)
    return rc;
  ctx->have_sell = ctx->sell_current = ctx->sp_pending = false; // the operator stream is packed from the values: after assembly
  ctx->sp_bounds_ok = false;
  if (ctx->sellp_mode != 0)
  {
    rc = sellp_pattern_bounds(ctx);
    if (rc)
      return rc;
  }
  ctx->have_pattern = true;
  return ZZZ_OK;
}

// Host (C++/OpenMP) pattern builder: index bookkeeping only, no floating-point work.
static int pattern_build_host(zzz_ctx* ctx)
{
  if (ctx->h_cell_dofs.empty()) // feed was generated on the device
  {
    ctx->h_cell_dofs.resize((size_t)(ctx->ncells * ctx->nd));
    ZZZ_HIP(ctx, hipMemcpy(ctx->h_cell_dofs.data(), ctx->cell_dofs.p, ctx->h_cell_dofs.size() * sizeof(int32_t),
                           hipMemcpyDeviceToHost));
  }
  const int nd = ctx->nd, bs = ctx->bs;
  const int64_t nb = ctx->n_owned, nc = ctx->ncells;
  std::vector<int32_t> cd_internal;
  const bool internal = ctx->renumbered || ctx->cells_renumbered;
  if (internal) // the host copy keeps the caller's dof numbering and cell order (zzz_ghost_layer_build reads it)
  {
    cd_internal.resize((size_t)(nc * nd));
    ZZZ_HIP(ctx, hipMemcpy(cd_internal.data(), ctx->cell_dofs.p, cd_internal.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
  }
  const int32_t* cd = internal ? cd_internal.data() : ctx->h_cell_dofs.data();
  // owned block dof -> incident cells, ascending (counting sort over cells)
  std::vector<int32_t> off((size_t)nb + 1, 0);
  for (int64_t c = 0; c < nc; ++c)
    for (int i = 0; i < nd; ++i)
    {
      const int32_t d = cd[c * nd + i];
      if (d < nb)
        off[(size_t)d + 1]++;
    }
  for (int64_t i = 0; i < nb; ++i)
  {
    if ((int64_t)off[i] + off[i + 1] > INT32_MAX)
      return fail(ctx, ZZZ_ERR_LIMIT, "dof->cell adjacency exceeds int32");
    off[i + 1] += off[i];
  }
  std::vector<int32_t> adj((size_t)off[nb]);
  {
    std::vector<int32_t> pos(off.begin(), off.end() - 1);
    for (int64_t c = 0; c < nc; ++c)
      for (int i = 0; i < nd; ++i)
      {
        const int32_t d = cd[c * nd + i];
        if (d < nb)
          adj[(size_t)pos[d]++] = (int32_t)c;
      }
  }
  // block pattern: per owned block dof, the sorted union of the dofs of its cells
  std::vector<int32_t> bptr((size_t)nb + 1, 0);
  std::vector<std::vector<int32_t>> chunks;
  const int64_t CH = 1 << 16;
  const int64_t nch = (nb + CH - 1) / CH;
  chunks.resize((size_t)nch);
  bool overflow = false;
#pragma omp parallel for schedule(dynamic, 1)
  for (int64_t ch = 0; ch < nch; ++ch)
  {
    std::vector<int32_t> tmp;
    std::vector<int32_t>& out = chunks[(size_t)ch];
    const int64_t lo = ch * CH, hi = std::min(nb, lo + CH);
    for (int64_t r = lo; r < hi; ++r)
    {
      tmp.clear();
      for (int32_t a = off[r]; a < off[r + 1]; ++a)
        tmp.insert(tmp.end(), cd + (int64_t)adj[a] * nd, cd + (int64_t)adj[a] * nd + nd);
      std::sort(tmp.begin(), tmp.end());
      tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
      bptr[(size_t)r + 1] = (int32_t)tmp.size();
      out.insert(out.end(), tmp.begin(), tmp.end());
    }
  }
  int64_t nblk = 0;
  for (int64_t r = 0; r < nb; ++r)
    nblk += bptr[r + 1];
  const int64_t nnz = nblk * bs * bs;
  if (overflow)
    return fail(ctx, ZZZ_ERR_LIMIT, "pattern too large (%lld nonzeros)", (long long)nnz);
  const int64_t nrows = nb * bs;
  std::vector<rp_t> rowptr((size_t)nrows + 1);
  std::vector<int32_t> cols((size_t)nnz);
  rowptr[0] = 0;
  for (int64_t r = 0; r < nb; ++r)
    for (int c = 0; c < bs; ++c)
      rowptr[(size_t)(r * bs + c) + 1] = rowptr[(size_t)(r * bs + c)] + (rp_t)bptr[r + 1] * bs;
  int maxrow = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(max : maxrow)
  for (int64_t ch = 0; ch < nch; ++ch)
  {
    const std::vector<int32_t>& in = chunks[(size_t)ch];
    const int64_t lo = ch * CH, hi = std::min(nb, lo + CH);
    size_t q = 0;
    for (int64_t r = lo; r < hi; ++r)
    {
      const int nbc = bptr[r + 1];
      maxrow = std::max(maxrow, nbc * bs);
      for (int c = 0; c < bs; ++c)
      {
        int32_t* dst = cols.data() + rowptr[(size_t)(r * bs + c)];
        for (int k = 0; k < nbc; ++k)
          for (int d = 0; d < bs; ++d)
            dst[k * bs + d] = in[q + k] * bs + d;
      }
      q += (size_t)nbc;
    }
  }
  chunks.clear();
  ctx->nrows = nrows;
  ctx->ncols = ctx->nloc();
  ctx->nnz = nnz;
  ctx->max_row_nnz = maxrow;
  int rc = upload(ctx, ctx->rowptr, rowptr.data(), rowptr.size());
  if (!rc)
    rc = upload(ctx, ctx->cols, cols.data(), cols.size(), 8);
  if (!rc)
    rc = upload(ctx, ctx->adj_off, off.data(), off.size());
  if (!rc)
    rc = upload(ctx, ctx->adj_cells, adj.data(), adj.size());
  if (rc)
    return rc;
  ZZZ_HIP(ctx, ctx->vals.alloc((size_t)nnz + 8));
  ZZZ_HIP(ctx, hipMemsetAsync(ctx->vals.p, 0, ((size_t)nnz + 8) * sizeof(double), ctx->stream));
  ZZZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  rc = build_tiles_device(ctx, maxrow / bs);
  if (rc)
    return rc;
  ZZZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return ZZZ_OK;
}

int zzz_csr_sizes(const zzz_ctx* ctx, int64_t* nrows, int64_t* ncols, int64_t* nnz)
{
  if (!ctx)
    return fail(nullptr, ZZZ_ERR_ARG, "NULL context");
  if (!ctx->have_pattern)
    return fail(const_cast<zzz_ctx*>(ctx), ZZZ_ERR_ARG, "no sparsity pattern yet");
  if (nrows)
    *nrows = ctx->nrows;
  if (ncols)
    *ncols = ctx->ncols;
  if (nnz)
    *nnz = ctx->nnz;
  return ZZZ_OK;
}

int zzz_csr_download(zzz_ctx* ctx, int32_t* rowptr, int32_t* cols, double* vals)
{
  ZZZ_ENTER(ctx);
  if (!ctx->have_pattern)
    return fail(ctx, ZZZ_ERR_ARG, "no sparsity pattern yet");
  ZZZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->renumbered)
  {
    if (rowptr && ctx->nnz > INT32_MAX)
      return fail(ctx, ZZZ_ERR_LIMIT, "%lld nonzeros do not fit 32-bit row pointers: use zzz_csr_rowptr64_download",
                  (long long)ctx->nnz);
    std::vector<rp_t> rpc;
    if (int rc = csr_to_caller(ctx, rpc, cols, vals, cols != nullptr))
      return rc;
    if (rowptr)
      for (size_t i = 0; i < rpc.size(); ++i)
        rowptr[i] = (int32_t)rpc[i];
    return ZZZ_OK;
  }
  if (rowptr)
  {
    // the matrix of record keeps 64-bit row pointers; this entry point serves the 32-bit form of DOLFINx / the parity
    // tests and refuses matrices it cannot express (zzz_csr_rowptr64_download takes any)
    if (ctx->nnz > INT32_MAX)
      return fail(ctx, ZZZ_ERR_LIMIT, "%lld nonzeros do not fit 32-bit row pointers: use zzz_csr_rowptr64_download",
                  (long long)ctx->nnz);
    std::vector<rp_t> h((size_t)ctx->nrows + 1);
    ZZZ_HIP(ctx, hipMemcpy(h.data(), ctx->rowptr.p, h.size() * sizeof(rp_t), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < h.size(); ++i)
      rowptr[i] = (int32_t)h[i];
  }
  if (cols)
    ZZZ_HIP(ctx, hipMemcpy(cols, ctx->cols.p, (size_t)ctx->nnz * sizeof(int32_t), hipMemcpyDeviceToHost));
  if (vals)
  {
    if (ctx->have_matrix)
      ZZZ_HIP(ctx, hipMemcpy(vals, ctx->vals.p, (size_t)ctx->nnz * sizeof(double), hipMemcpyDeviceToHost));
    else // a created, not yet assembled matrix is zero (the device array is written whole by the first assembly)
      memset(vals, 0, (size_t)ctx->nnz * sizeof(double));
  }
  return ZZZ_OK;
}

int zzz_csr_rowptr64_download(zzz_ctx* ctx, int64_t* rowptr)
{
  ZZZ_ENTER(ctx);
  if (!ctx->have_pattern || !rowptr)
    return fail(ctx, ZZZ_ERR_ARG, "zzz_csr_rowptr64_download: no sparsity pattern yet / NULL array");
  ZZZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->renumbered)
  {
    std::vector<rp_t> rpc;
    if (int rc = csr_to_caller(ctx, rpc, nullptr, nullptr, false))
      return rc;
    std::copy(rpc.begin(), rpc.end(), rowptr);
    return ZZZ_OK;
  }
  ZZZ_HIP(ctx, hipMemcpy(rowptr, ctx->rowptr.p, ((size_t)ctx->nrows + 1) * sizeof(int64_t), hipMemcpyDeviceToHost));
  return ZZZ_OK;
}

int zzz_csr_upload_values(zzz_ctx* ctx, const double* vals)
{
  ZZZ_ENTER(ctx);
  if (!ctx->have_pattern || !vals)
    return fail(ctx, ZZZ_ERR_ARG, "zzz_csr_upload_values: no pattern or NULL values");
  ++ctx->mat_version;
  std::vector<double> vi;
  if (ctx->renumbered)
  {
    ZZZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (int rc = csr_values_to_internal(ctx, vals, vi))
      return rc;
    vals = vi.data();
  }
  ZZZ_HIP(ctx, hipMemcpyAsync(ctx->vals.p, vals, (size_t)ctx->nnz * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  ZZZ_HIP(ctx, hipStreamSynchronize(ctx->stream)); // the staging copy above may be a local
  ctx->sp_rownnz_fresh = ctx->sp_compact_fresh = false; // what an assembly left belongs to the values just replaced
  int rc = sell_update(ctx, false);
  if (rc)
    return rc;
  ZZZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ctx->have_matrix = true;
  return ZZZ_OK;
}

int zzz_assemble_matrix(zzz_ctx* ctx, int form)
{
  ZZZ_ENTER(ctx);
  if (!ctx->have_pattern)
    return fail(ctx, ZZZ_ERR_ARG, "zzz_assemble_matrix before zzz_csr_pattern_build");
  if (form != ZZZ_FORM_POISSON && form != ZZZ_FORM_ELASTICITY)
    return fail(ctx, ZZZ_ERR_ARG, "unknown form %d", form);
  int rc = launch_assemble_matrix(ctx, form);
  if (!rc)
    rc = sell_update(ctx, false); // MatAssemblyEnd-like finalisation: refresh the SpMV copy
  ctx->sp_rownnz_fresh = ctx->sp_compact_fresh = false; // (whether or not this packing path had a use for them)
  if (!rc)
    ctx->have_matrix = true;
  return rc;
}

int zzz_assemble_vector(zzz_ctx* ctx, int form)
{
  ZZZ_ENTER(ctx);
  if (!ctx->have_pattern)
    return fail(ctx, ZZZ_ERR_ARG, "zzz_assemble_vector before zzz_csr_pattern_build");
  if (form != ZZZ_FORM_POISSON && form != ZZZ_FORM_ELASTICITY)
    return fail(ctx, ZZZ_ERR_ARG, "unknown form %d", form);
  if (!ctx->have_coeff[ZZZ_COEFF_F] || (form == ZZZ_FORM_POISSON && !ctx->have_coeff[ZZZ_COEFF_G]))
    return fail(ctx, ZZZ_ERR_ARG, "coefficient(s) of L not uploaded");
  return launch_assemble_vector(ctx, form);
}

static zzz::DevBuf<double>* pick_vec(zzz_ctx* ctx, int which)
{
  return which == ZZZ_VEC_B ? &ctx->b : which == ZZZ_VEC_U ? &ctx->u : nullptr;
}

int zzz_vec_download(zzz_ctx* ctx, int which, double* out)
{
  ZZZ_ENTER(ctx);
  DevBuf<double>* v = pick_vec(ctx, which);
  if (!v || !v->p || !out)
    return fail(ctx, ZZZ_ERR_ARG, "zzz_vec_download: bad vector / not allocated");
  ZZZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->renumbered)
  {
    std::vector<double> tmp((size_t)(ctx->n_owned * ctx->bs));
    ZZZ_HIP(ctx, hipMemcpy(tmp.data(), v->p, tmp.size() * sizeof(double), hipMemcpyDeviceToHost));
    to_caller(ctx, tmp.data(), out, true);
    return ZZZ_OK;
  }
  ZZZ_HIP(ctx, hipMemcpy(out, v->p, (size_t)(ctx->n_owned * ctx->bs) * sizeof(double), hipMemcpyDeviceToHost));
  return ZZZ_OK;
}

int zzz_vec_upload(zzz_ctx* ctx, int which, const double* in)
{
  ZZZ_ENTER(ctx);
  DevBuf<double>* v = pick_vec(ctx, which);
  if (!v || !v->p || !in)
    return fail(ctx, ZZZ_ERR_ARG, "zzz_vec_upload: bad vector / not allocated");
  std::vector<double> tmp;
  if (ctx->renumbered)
  {
    tmp.resize((size_t)(ctx->n_owned * ctx->bs));
    to_internal(ctx, in, tmp.data(), true);
    in = tmp.data();
  }
  ZZZ_HIP(ctx, hipMemcpyAsync(v->p, in, (size_t)(ctx->n_owned * ctx->bs) * sizeof(double), hipMemcpyHostToDevice,
                              ctx->stream));
  ZZZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return ZZZ_OK;
}

int zzz_vec_norm(zzz_ctx* ctx, int which, double* out)
{
  ZZZ_ENTER(ctx);
  DevBuf<double>* v = pick_vec(ctx, which);
  if (!v || !v->p || !out)
    return fail(ctx, ZZZ_ERR_ARG, "zzz_vec_norm: bad vector / not allocated");
  return vec_norm_local(ctx, v->p, ctx->n_owned * ctx->bs, out);
}

int zzz_spmv(zzz_ctx* ctx, const double* x, double* y)
{
  ZZZ_ENTER(ctx);
  if (!ctx->have_matrix || !x || !y)
    return fail(ctx, ZZZ_ERR_ARG, "zzz_spmv: no matrix or NULL vector");
  const size_t n = (size_t)(ctx->n_owned * ctx->bs);
  std::vector<double> xin;
  if (ctx->renumbered)
  {
    xin.resize(n);
    to_internal(ctx, x, xin.data(), true);
    x = xin.data();
  }
  ZZZ_HIP(ctx, hipMemsetAsync(ctx->p.p, 0, ctx->p.n * sizeof(double), ctx->stream));
  ZZZ_HIP(ctx, hipMemcpyAsync(ctx->p.p, x, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  ZZZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  int rc;
  if (ctx->comm && ctx->overlap && ctx->have_tile_split)
    rc = launch_spmv_overlapped(ctx, ctx->p.p, ctx->w.p, nullptr, nullptr);
  else
  {
    if (ctx->comm)
    {
      rc = comm_halo_forward(ctx, ctx->p.p);
      if (rc)
        return rc;
    }
    rc = launch_spmv(ctx, ctx->p.p, ctx->w.p, nullptr, nullptr);
  }
  if (rc)
    return rc;
  if (ctx->renumbered)
  {
    ZZZ_HIP(ctx, hipMemcpyAsync(xin.data(), ctx->w.p, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    ZZZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    to_caller(ctx, xin.data(), y, true);
    return ctx->comm ? comm_p2p_check(ctx) : ZZZ_OK; // a ghost value that never arrived through the peer-memory window
  }
  ZZZ_HIP(ctx, hipMemcpyAsync(y, ctx->w.p, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  ZZZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return ctx->comm ? comm_p2p_check(ctx) : ZZZ_OK;
}

int zzz_spmv_time(zzz_ctx* ctx, int reps, int variant, double* avg_ms)
{
  ZZZ_ENTER(ctx);
  if (!ctx->have_matrix || reps < 1 || !avg_ms)
    return fail(ctx, ZZZ_ERR_ARG, "zzz_spmv_time: no matrix or bad arguments");
  const int saved = ctx->spmv_variant;
  const bool saved_auto = ctx->spmv_auto;
  if (variant >= 0)
  {
    ctx->spmv_variant = variant & 27; // bit 0 nt, bit 1 pipelined tiles, bit 3 SELL, bit 4 int32 columns
    ctx->spmv_auto = false;
  }
  hipEvent_t e0, e1;
  ZZZ_HIP(ctx, hipEventCreate(&e0));
  ZZZ_HIP(ctx, hipEventCreate(&e1));
  ZZZ_HIP(ctx, hipMemsetAsync(ctx->state.p, 0, sizeof(zzz::CgState), ctx->stream));
  int np = 0;
  struct TimingOnly // the products below are timed, their results discarded (zzz_sellp.hip: the ZZZ_EXP_WIN probe)
  {
    zzz_ctx* c;
    explicit TimingOnly(zzz_ctx* cc) : c(cc) { c->timing_only = true; }
    ~TimingOnly() { c->timing_only = false; }
  } timing_only(ctx);
  int rc = launch_spmv(ctx, ctx->p.p, ctx->w.p, ctx->part_a.p, &np); // warm-up
  ZZZ_HIP(ctx, hipEventRecord(e0, ctx->stream));
  for (int i = 0; i < reps && !rc; ++i)
    rc = launch_spmv(ctx, ctx->p.p, ctx->w.p, ctx->part_a.p, &np);
  ZZZ_HIP(ctx, hipEventRecord(e1, ctx->stream));
  ZZZ_HIP(ctx, hipEventSynchronize(e1));
  float ms = 0;
  ZZZ_HIP(ctx, hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  ctx->spmv_variant = saved;
  ctx->spmv_auto = saved_auto;
  *avg_ms = ms / reps;
  return rc;
}

int zzz_action(zzz_ctx* ctx, const double* x, double* y)
{
  ZZZ_ENTER(ctx);
  if (ctx->order == 0 || !x || !y)
    return fail(ctx, ZZZ_ERR_ARG, "zzz_action: no dofmap or NULL vector");
  const size_t n = (size_t)(ctx->n_owned * ctx->bs);
  std::vector<double> xin;
  if (ctx->renumbered)
  {
    xin.resize(n);
    to_internal(ctx, x, xin.data(), true);
    x = xin.data();
  }
  ZZZ_HIP(ctx, hipMemsetAsync(ctx->p.p, 0, ctx->p.n * sizeof(double), ctx->stream));
  ZZZ_HIP(ctx, hipMemcpyAsync(ctx->p.p, x, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  ZZZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->comm)
  {
    int rc = comm_halo_forward(ctx, ctx->p.p);
    if (rc)
      return rc;
  }
  int rc = launch_matfree_action(ctx, ctx->p.p, ctx->w.p, nullptr, nullptr);
  if (rc)
    return rc;
  if (ctx->renumbered)
  {
    ZZZ_HIP(ctx, hipMemcpyAsync(xin.data(), ctx->w.p, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    ZZZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    to_caller(ctx, xin.data(), y, true);
    return ctx->comm ? comm_p2p_check(ctx) : ZZZ_OK;
  }
  ZZZ_HIP(ctx, hipMemcpyAsync(y, ctx->w.p, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  ZZZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return ctx->comm ? comm_p2p_check(ctx) : ZZZ_OK;
}

int zzz_matfree_setup(zzz_ctx* ctx)
{
  ZZZ_ENTER(ctx);
  return mf_plan_build(ctx);
}

int zzz_matfree_diagonal(zzz_ctx* ctx, double* diag)
{
  ZZZ_ENTER(ctx);
  if (ctx->order == 0 || !diag)
    return fail(ctx, ZZZ_ERR_ARG, "zzz_matfree_diagonal: no dofmap or NULL vector");
  if (int rc = launch_matfree_diagonal(ctx, ctx->w.p))
    return rc;
  const size_t n = (size_t)(ctx->n_owned * ctx->bs);
  std::vector<double> tmp(ctx->renumbered ? n : 0);
  double* dst = ctx->renumbered ? tmp.data() : diag;
  ZZZ_HIP(ctx, hipMemcpyAsync(dst, ctx->w.p, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  ZZZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->renumbered)
    to_caller(ctx, tmp.data(), diag, true);
  return ZZZ_OK;
}

int zzz_matfree_info(zzz_ctx* ctx, int64_t info[8])
{
  if (!ctx || !info)
    return fail(ctx, ZZZ_ERR_ARG, "zzz_matfree_info: bad arguments");
  const zzz::MfPlan& M = ctx->mf;
  info[0] = M.valid ? 1 : 0;
  info[1] = M.nblocks;
  info[2] = M.nc;
  info[3] = M.threads;
  info[4] = M.nloc_max;
  info[5] = M.nshared;
  info[6] = M.nslots;
  info[7] = M.bytes_per_action;
  return ZZZ_OK;
}

int zzz_action_time(zzz_ctx* ctx, int reps, double* avg_ms)
{
  ZZZ_ENTER(ctx);
  if (reps <= 0 || !avg_ms || ctx->order == 0)
    return fail(ctx, ZZZ_ERR_ARG, "zzz_action_time: bad arguments");
  hipEvent_t e0, e1;
  ZZZ_HIP(ctx, hipEventCreate(&e0));
  ZZZ_HIP(ctx, hipEventCreate(&e1));
  ZZZ_HIP(ctx, hipMemsetAsync(ctx->state.p, 0, sizeof(zzz::CgState), ctx->stream));
  int np = 0;
  int rc = launch_matfree_action(ctx, ctx->p.p, ctx->w.p, ctx->part_a.p, &np); // warm-up (and the plan, if missing)
  ZZZ_HIP(ctx, hipEventRecord(e0, ctx->stream));
  for (int i = 0; i < reps && !rc; ++i)
    rc = launch_matfree_action(ctx, ctx->p.p, ctx->w.p, ctx->part_a.p, &np);
  ZZZ_HIP(ctx, hipEventRecord(e1, ctx->stream));
  ZZZ_HIP(ctx, hipEventSynchronize(e1));
  float ms = 0;
  ZZZ_HIP(ctx, hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *avg_ms = ms / reps;
  return rc;
}

int zzz_cg_solve(zzz_ctx* ctx, const zzz_solver_opts* o, int* iters, double* rnorm)
{
  ZZZ_ENTER(ctx);
  if (!o)
    return fail(ctx, ZZZ_ERR_ARG, "zzz_cg_solve: NULL options");
  if (o->op == ZZZ_OP_CSR && !ctx->have_matrix)
    return fail(ctx, ZZZ_ERR_ARG, "zzz_cg_solve: matrix not assembled");
  if (o->op != ZZZ_OP_CSR && o->op != ZZZ_OP_MATFREE)
    return fail(ctx, ZZZ_ERR_ARG, "unknown operator kind %d", o->op);
  if (o->variant != ZZZ_CG_PETSC && o->variant != ZZZ_CG_CGH)
    return fail(ctx, ZZZ_ERR_ARG, "unknown CG variant %d", o->variant);
  if (o->variant == ZZZ_CG_CGH && o->pc != ZZZ_PC_NONE)
    return fail(ctx, ZZZ_ERR_ARG, "src/cg.h has no preconditioner: use pc = ZZZ_PC_NONE");
  if (o->pc != ZZZ_PC_NONE && o->pc != ZZZ_PC_JACOBI && o->pc != ZZZ_PC_CHEBYSHEV_JACOBI)
    return fail(ctx, ZZZ_ERR_ARG, "unsupported preconditioner %d (none, jacobi, chebyshev-jacobi)", o->pc);
  if (o->pc == ZZZ_PC_CHEBYSHEV_JACOBI && o->op != ZZZ_OP_CSR)
    return fail(ctx, ZZZ_ERR_ARG, "the Chebyshev-Jacobi preconditioner needs the assembled operator");
  if (o->pc == ZZZ_PC_CHEBYSHEV_JACOBI && (o->variant != ZZZ_CG_PETSC || o->pc_degree < 0 || o->pc_degree > 64 || o->pc_esteig_its > 64))
    return fail(ctx, ZZZ_ERR_ARG, "the Chebyshev-Jacobi preconditioner applies to KSPCG (either form), degree 1..64, estimate <= 64 steps");
  if (o->norm < 0 || o->norm > 2)
    return fail(ctx, ZZZ_ERR_ARG, "unknown norm type %d", o->norm);
  if (o->single_reduction && (o->variant != ZZZ_CG_PETSC || o->op != ZZZ_OP_CSR))
    return fail(ctx, ZZZ_ERR_ARG, "-ksp_cg_single_reduction applies to KSPCG on the assembled operator only");
  if (o->max_it < 0 || o->max_it > (1 << 24))
    return fail(ctx, ZZZ_ERR_ARG, "max_it %d out of range", o->max_it);
  return cg_solve(ctx, o, iters, rnorm);
}

int zzz_cg_history(zzz_ctx* ctx, int n, double* out)
{
  if (!ctx || !out || n < 0)
    return fail(ctx, ZZZ_ERR_ARG, "zzz_cg_history: bad arguments");
  const size_t m = std::min((size_t)n, ctx->history.size());
  std::copy(ctx->history.begin(), ctx->history.begin() + (long)m, out);
  return ZZZ_OK;
}

int zzz_spmv_info(zzz_ctx* ctx, int64_t info[8])
{
  ZZZ_ENTER(ctx);
  if (!info || !ctx->have_pattern)
    return fail(ctx, ZZZ_ERR_ARG, "zzz_spmv_info: no pattern");
  // the packed column stream of the tile kernel is encoded on first use OF THAT KERNEL: a matrix whose product runs on
  // the operator stream never pays for it (several hundred MB at the c4 / c5_rank sizes); info[0..2] then describe a
  // tile kernel that was never prepared: 0 / the configured offset bits / every tile on int32 columns
  if (!sellp_active(ctx))
    if (int rc = ensure_cols16(ctx))
      return rc;
  info[0] = ctx->have_cols16 ? 1 : 0;
  info[1] = ctx->cols16_offb;
  info[2] = ctx->have_cols16 ? ctx->cols16_fallback_tiles : ctx->ntiles;
  info[3] = ctx->ntiles;
  if (sellp_active(ctx) && ctx->sp_win_max > 0)
  {
    // x windows of the operator stream (no tile kernel in use then): bytes of x the product loads into LDS per launch
    // in place of per-entry gathers, and the LDS doubles a workgroup holds
    info[2] = ctx->sp_win_bytes;
    info[3] = -(int64_t)ctx->sp_win_max;
  }
  info[4] = sellp_active(ctx) ? 1 : (int64_t)1 << ctx->spmv_lpr_shift; // the stream sums a row serially
  info[5] = sellp_active(ctx) ? (ctx->sp_sorted ? 2 : 1) : 0;
  info[6] = sellp_active(ctx) ? sellp_stream_bytes(ctx) : 0; // bytes of the operator stream read per product
  info[7] = sellp_active(ctx) ? ctx->sp_chunks * 512 : 0;     // its entries, padding included
  return ZZZ_OK;
}

int zzz_spmv_values_info2(zzz_ctx* ctx, int n, int64_t* out)
{
  ZZZ_ENTER(ctx);
  if (!out || n < 0 || !ctx->have_pattern)
    return fail(ctx, ZZZ_ERR_ARG, "zzz_spmv_values_info: no pattern");
  int64_t info[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (sellp_active(ctx))
  {
    const bool blk = zzz::sellp_blk_serves(ctx);
    info[4] = !blk && zzz::sellp_pipe_wgs(ctx, false) ? 1 : 0;
    info[5] = blk ? 1 : (info[4] ? zzz::sellp_pipe_wgs(ctx, false) : 8);
    info[0] = ctx->sp_sd_on ? 3 : (ctx->sp_dict_on ? (ctx->sp_dict_n <= zzz::SP_DICT_LDS_ENTRIES ? 2 : 1) : 0);
    info[1] = ctx->sp_dict_on ? ctx->sp_dict_n : 0;
    info[2] = sellp_stream_bytes(ctx);
    info[3] = ctx->sp_bytes + ctx->nslices * 8;
    const bool win = zzz::sellp_win_serves(ctx);
    info[6] = blk ? 1 : (win ? 2 : 0);
    info[7] = blk ? ctx->bk_entries : (win ? ctx->bw_window_entries : 0);
    info[8] = blk ? ctx->bk_chunks : (win ? ctx->bw_chunks : 0);
    info[9] = blk ? ctx->bk_form : (win ? ctx->bw_nblk : 0);
    if (win)
      info[4] = 0, info[5] = 1;
  }
  for (int i = 0; i < std::min(n, 10); ++i)
    out[i] = info[i];
  return ZZZ_OK;
}

int zzz_spmv_values_info(zzz_ctx* ctx, int64_t info[4]) { return zzz_spmv_values_info2(ctx, 4, info); }

int zzz_abi_version(void) { return ZZZ_ABI_VERSION; }

int zzz_internal_order_download(zzz_ctx* ctx, int32_t* perm, int32_t* kind)
{
  if (!ctx || ctx->order == 0)
    return fail(ctx, ZZZ_ERR_ARG, "zzz_internal_order_download: no dofmap yet");
  if (perm)
    for (int64_t i = 0; i < ctx->n_owned; ++i)
      perm[i] = ctx->renumbered ? ctx->h_perm[(size_t)i] : (int32_t)i;
  if (kind)
    *kind = (ctx->renumbered ? ctx->renumber_kind : 0) | (ctx->cells_renumbered ? 16 : 0);
  return ZZZ_OK;
}

int zzz_cg_info(zzz_ctx* ctx, int64_t info[4])
{
  if (!ctx || !info)
    return fail(ctx, ZZZ_ERR_ARG, "zzz_cg_info: bad arguments");
  info[0] = (ctx->last_solve_fused ? 1 : 0) | (ctx->last_solve_dinv_codes > 0 ? 2 : 0) | ((int64_t)ctx->last_solve_dinv_codes << 8);
  info[1] = ctx->last_iters;
  info[2] = ctx->last_reason;
  info[3] = (int64_t)(ctx->last_pc_bound * 1.0e6); // Chebyshev-Jacobi: spectrum bound x 1e6
  return ZZZ_OK;
}

int zzz_profile_get(zzz_ctx* ctx, double* spmv_avg_ms, int64_t* spmv_count)
{
  if (!ctx)
    return fail(nullptr, ZZZ_ERR_ARG, "NULL context");
  if (spmv_avg_ms)
    *spmv_avg_ms = ctx->prof_spmv_ms;
  if (spmv_count)
    *spmv_count = ctx->prof_spmv_n;
  return ZZZ_OK;
}

} // extern "C"

// The reference's native partition at the boundary: cells partitioned with GhostMode::none
// (src/mesh.cpp:182-183), rows of interface dofs completed by MatAssemblyBegin/End and scatter_rev
// (src/poisson_problem.cpp:132-137,154).
//
// This library assembles complete owned rows from local cells (row-gather, no atomics, no per-assembly
// exchange), which needs every cell that touches an owned dof to be local.  zzz_ghost_layer_build gets a
// native feed there ONCE, at set-up: every rank sends the cells that touch a neighbour's dofs to that
// neighbour (connectivity in global indices, vertex coordinates, facet marks, Dirichlet flags and nodal
// coefficient values of their dofs), the receivers append them as ghost cells, ask the owners of the dofs
// they did not know for a place in the forward scatter, and the context is re-uploaded with the extended
// mesh, dofmap and halo plan.  From then on the assembled A and b are those of the reference after its
// MatAssembly / scatter_rev -- identical sums, formed on the owner in ascending cell order -- and nothing is
// exchanged per assembly.  Host code: index bookkeeping, no floating-point work.
#include "zzz_internal.h"

#include <algorithm>
#include <cstring>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

namespace zzz
{
int comm_exchange_bytes(zzz_ctx* ctx, const std::vector<std::vector<char>>& send, std::vector<std::vector<char>>& recv);
int comm_size(const zzz_ctx* ctx);
int comm_rank(const zzz_ctx* ctx);

// Collective agreement on a rank-local status: every rank contributes its code (and, when it failed, its message);
// all ranks return the same verdict -- ZZZ_OK only if every rank was fine, else the first failing rank's code with
// its message -- so that no rank walks into the next exchange while a peer has already returned an error (with RCCL
// the peers would block forever in the grouped send/recv, with the local backend until the barrier times out).
static int agree(zzz_ctx* ctx, int status)
{
  const int nr = comm_size(ctx), me = comm_rank(ctx);
  std::vector<std::vector<char>> out((size_t)nr), in;
  const std::string msg = status ? ctx->err : std::string();
  for (int r = 0; r < nr; ++r)
  {
    const int32_t st = status;
    const char* p = reinterpret_cast<const char*>(&st);
    out[(size_t)r].assign(p, p + sizeof(st));
    out[(size_t)r].insert(out[(size_t)r].end(), msg.begin(), msg.end());
  }
  if (int rc = comm_exchange_bytes(ctx, out, in))
    return rc; // the transport itself failed: nothing left to agree through
  for (int r = 0; r < nr; ++r)
  {
    int32_t st = 0;
    if (in[(size_t)r].size() >= sizeof(st))
      memcpy(&st, in[(size_t)r].data(), sizeof(st));
    if (st)
    {
      if (r == me)
        return fail(ctx, st, "%s", msg.c_str());
      const std::string theirs(in[(size_t)r].begin() + sizeof(st), in[(size_t)r].end());
      return fail(ctx, st, "zzz_ghost_layer_build failed on rank %d: %s", r, theirs.c_str());
    }
  }
  return ZZZ_OK;
}

template <typename T>
static void put(std::vector<char>& b, const T* v, size_t n)
{
  const char* p = reinterpret_cast<const char*>(v);
  b.insert(b.end(), p, p + n * sizeof(T));
}
template <typename T>
static const char* get(const char* p, T* v, size_t n)
{
  memcpy(v, p, n * sizeof(T));
  return p + n * sizeof(T);
}
template <typename T>
static int download(zzz_ctx* ctx, const DevBuf<T>& d, std::vector<T>& h, size_t n)
{
  h.resize(n);
  if (n)
    ZZZ_HIP(ctx, hipMemcpy(h.data(), d.p, n * sizeof(T), hipMemcpyDeviceToHost));
  return ZZZ_OK;
}
} // namespace zzz

using namespace zzz;

extern "C" {

int zzz_global_ids_upload(zzz_ctx* ctx, const int64_t* dof_global, const int64_t* vert_global)
{
  if (!ctx)
    return fail(nullptr, ZZZ_ERR_ARG, "NULL context");
  if (ctx->order == 0 || !dof_global || !vert_global)
    return fail(ctx, ZZZ_ERR_ARG, "zzz_global_ids_upload: upload mesh and dofmap first; NULL array");
  ctx->h_dof_global.assign(dof_global, dof_global + (ctx->n_owned + ctx->n_ghost));
  ctx->h_vert_global.assign(vert_global, vert_global + ctx->nverts);
  return ZZZ_OK;
}

int zzz_global_ids_download(zzz_ctx* ctx, int64_t* dof_global)
{
  if (!ctx || !dof_global)
    return fail(ctx, ZZZ_ERR_ARG, "zzz_global_ids_download: bad arguments");
  if ((int64_t)ctx->h_dof_global.size() != ctx->n_owned + ctx->n_ghost)
    return fail(ctx, ZZZ_ERR_ARG, "no global indices uploaded");
  std::copy(ctx->h_dof_global.begin(), ctx->h_dof_global.end(), dof_global);
  return ZZZ_OK;
}

int zzz_ghost_layer_build(zzz_ctx* ctx)
{
  if (!ctx)
    return fail(nullptr, ZZZ_ERR_ARG, "NULL context");
  ZZZ_HIP(ctx, hipSetDevice(ctx->device));
  if (!ctx->comm)
    return fail(ctx, ZZZ_ERR_ARG, "zzz_ghost_layer_build: attach a communicator first");
  // Every rank-local failure below is made COLLECTIVE before the next exchange (agree): all ranks return an error
  // together, none is left waiting in a send/recv for a peer that has already given up.
  const int64_t n_owned = ctx->n_owned, n_ghost = ctx->n_ghost, nloc = n_owned + n_ghost;
  const int64_t ncells = ctx->ncells, nverts = ctx->nverts;
  const int nd = ctx->nd, bs = ctx->bs, order = ctx->order;
  const int nr = comm_size(ctx), me = comm_rank(ctx);
  std::vector<double> x, cf, cg;
  std::vector<int32_t> cverts, cdofs, send_idx;
  std::vector<uint8_t> bc, fmask;
  const bool haveF = ctx->have_coeff[ZZZ_COEFF_F], haveG = ctx->have_coeff[ZZZ_COEFF_G];
  const int64_t nsend_old = ctx->nneigh ? ctx->send_off[(size_t)ctx->nneigh] : 0;
  int rc = [&]() -> int {
  if (order == 0 || (int64_t)ctx->h_dof_global.size() != nloc || (int64_t)ctx->h_vert_global.size() != nverts)
    return fail(ctx, ZZZ_ERR_ARG, "zzz_ghost_layer_build: mesh, dofmap and global indices (zzz_global_ids_upload) first");
  int64_t nrecv = 0;
  for (int k = 0; k < ctx->nneigh; ++k)
    nrecv += ctx->recv_cnt[(size_t)k];
  if (nrecv != n_ghost)
    return fail(ctx, ZZZ_ERR_ARG, "zzz_ghost_layer_build: upload the forward-scatter plan (zzz_halo_upload) first");
  ZZZ_HIP(ctx, hipStreamSynchronize(ctx->stream));

  // ---- host copies of what the context holds --------------------------------------------------------------
  cverts = ctx->h_cell_verts;
  cdofs = ctx->h_cell_dofs;
  int rc = download(ctx, ctx->x, x, (size_t)(3 * nverts));
  // (the host copies are the caller's; only a device-generated feed has none, and that one is in internal order anyway)
  if (!rc && cverts.empty())
    rc = download(ctx, ctx->cell_verts, cverts, (size_t)(4 * ncells));
  if (!rc && cdofs.empty())
    rc = download(ctx, ctx->cell_dofs, cdofs, (size_t)(nd * ncells));
  if (!rc)
    rc = download(ctx, ctx->bc, bc, (size_t)(nloc * bs));
  if (!rc)
    rc = download(ctx, ctx->facet_mask, fmask, (size_t)ncells);
  if (!rc && haveF)
    rc = download(ctx, ctx->coeff[ZZZ_COEFF_F], cf, (size_t)(nloc * bs));
  if (!rc && haveG)
    rc = download(ctx, ctx->coeff[ZZZ_COEFF_G], cg, (size_t)nloc);
  if (!rc)
    rc = download(ctx, ctx->send_idx, send_idx, (size_t)nsend_old);
  if (!rc && ctx->cells_renumbered)
  {
    std::vector<uint8_t> fm2(fmask.size());
    for (int64_t i = 0; i < ncells; ++i)
      fm2[(size_t)ctx->h_cperm[(size_t)i]] = fmask[(size_t)i];
    fmask.swap(fm2);
  }
  if (!rc && ctx->renumbered)
  {
    // the device holds the library's internal numbering of the owned dofs; this function works in the caller's
    // (h_cell_dofs, h_dof_global) and re-uploads through the translating entry points
    const int32_t* perm = ctx->h_perm.data();
    for (int32_t& d : send_idx)
      d = perm[(size_t)d];
    std::vector<uint8_t> bc2(bc.size());
    for (int64_t i = 0; i < nloc; ++i)
    {
      const int64_t dst = i < n_owned ? perm[(size_t)i] : i;
      for (int k = 0; k < bs; ++k)
        bc2[(size_t)(dst * bs + k)] = bc[(size_t)(i * bs + k)];
    }
    bc.swap(bc2);
    std::vector<double> t;
    if (haveF)
    {
      t.resize(cf.size());
      to_caller(ctx, cf.data(), t.data(), false);
      cf.swap(t);
    }
    if (haveG)
    {
      t.resize(cg.size());
      to_caller(ctx, cg.data(), t.data(), false);
      cg.swap(t);
    }
  }
  return rc;
  }();
  if ((rc = agree(ctx, rc)))
    return rc;
  const bool had_bc = ctx->have_bc;
  std::vector<int64_t> dof_g = ctx->h_dof_global, vert_g = ctx->h_vert_global;
  std::vector<int32_t> old_neigh = ctx->neigh_rank;
  std::vector<int64_t> old_send_off = ctx->send_off, old_recv = ctx->recv_cnt;
  if (old_send_off.empty())
    old_send_off.assign(1, 0);

  // owner of every local block dof
  std::vector<int32_t> owner((size_t)nloc, me);
  {
    int64_t g = n_owned;
    for (int k = 0; k < ctx->nneigh; ++k)
      for (int64_t i = 0; i < old_recv[(size_t)k]; ++i)
        owner[(size_t)g++] = old_neigh[(size_t)k];
  }

  // ---- 1. cells that touch a neighbour's dofs go to that neighbour ------------------------------------------
  std::vector<std::vector<char>> out((size_t)nr), in;
  std::vector<int32_t> nmsg((size_t)nr, 0);
  const int32_t flags = (haveF ? 1 : 0) | (haveG ? 2 : 0);
  for (int r = 0; r < nr; ++r)
  {
    const int32_t hdr[2] = {0, flags};
    put(out[(size_t)r], hdr, 2);
  }
  for (int64_t c = 0; c < ncells; ++c)
  {
    int dest[20], ndest = 0;
    for (int i = 0; i < nd; ++i)
    {
      const int o = owner[(size_t)cdofs[(size_t)(c * nd + i)]];
      if (o == me)
        continue;
      bool seen = false;
      for (int q = 0; q < ndest; ++q)
        seen |= dest[q] == o;
      if (!seen)
        dest[ndest++] = o;
    }
    for (int q = 0; q < ndest; ++q)
    {
      std::vector<char>& b = out[(size_t)dest[q]];
      int64_t gd[20], gv[4];
      int32_t ow[20];
      double X[12];
      uint8_t bf[60];
      double fv[60], gvv[20];
      for (int i = 0; i < nd; ++i)
      {
        const int32_t l = cdofs[(size_t)(c * nd + i)];
        gd[i] = dof_g[(size_t)l];
        ow[i] = owner[(size_t)l];
        for (int k = 0; k < bs; ++k)
        {
          bf[i * bs + k] = bc[(size_t)l * bs + k];
          fv[i * bs + k] = haveF ? cf[(size_t)l * bs + k] : 0.0;
        }
        gvv[i] = haveG ? cg[(size_t)l] : 0.0;
      }
      for (int v = 0; v < 4; ++v)
      {
        const int32_t lv = cverts[(size_t)(4 * c + v)];
        gv[v] = vert_g[(size_t)lv];
        for (int a = 0; a < 3; ++a)
          X[3 * v + a] = x[(size_t)(3 * lv + a)];
      }
      put(b, gd, (size_t)nd);
      put(b, ow, (size_t)nd);
      put(b, gv, 4);
      put(b, X, 12);
      put(b, &fmask[(size_t)c], 1);
      put(b, bf, (size_t)(nd * bs));
      if (haveF)
        put(b, fv, (size_t)(nd * bs));
      if (haveG)
        put(b, gvv, (size_t)nd);
      nmsg[(size_t)dest[q]]++;
    }
  }
  for (int r = 0; r < nr; ++r)
    memcpy(out[(size_t)r].data(), &nmsg[(size_t)r], sizeof(int32_t));
  rc = comm_exchange_bytes(ctx, out, in);
  if (rc)
    return rc;

  // ---- 2. append the received cells ------------------------------------------------------------------------
  std::unordered_map<int64_t, int32_t> dof_l, vert_l;
  dof_l.reserve((size_t)nloc * 2);
  vert_l.reserve((size_t)nverts * 2);
  for (int64_t l = 0; l < nloc; ++l)
    dof_l[dof_g[(size_t)l]] = (int32_t)l;
  for (int64_t v = 0; v < nverts; ++v)
    vert_l[vert_g[(size_t)v]] = (int32_t)v;
  struct NewGhost
  {
    int64_t g;
    int32_t owner;
    uint8_t bcf[3];
    double f[3], gv;
  };
  std::vector<NewGhost> ng; // temporary local index nloc + position
  std::vector<std::vector<char>> req((size_t)nr), reqin;
  std::vector<std::vector<int32_t>> ng_of((size_t)nr); // per owner: positions in ng, in request order
  rc = [&]() -> int {
  for (int r = 0; r < nr; ++r)
  {
    if (in[(size_t)r].size() < 2 * sizeof(int32_t))
      continue;
    const char* p = in[(size_t)r].data();
    int32_t hdr[2];
    p = get(p, hdr, 2);
    if (hdr[0] > 0 && hdr[1] != flags)
      return fail(ctx, ZZZ_ERR_ARG, "zzz_ghost_layer_build: rank %d uploaded other coefficients than rank %d", r, me);
    for (int32_t q = 0; q < hdr[0]; ++q)
    {
      int64_t gd[20], gv[4];
      int32_t ow[20];
      double X[12], fv[60], gvv[20];
      uint8_t fm, bf[60];
      p = get(p, gd, (size_t)nd);
      p = get(p, ow, (size_t)nd);
      p = get(p, gv, 4);
      p = get(p, X, 12);
      p = get(p, &fm, 1);
      p = get(p, bf, (size_t)(nd * bs));
      if (haveF)
        p = get(p, fv, (size_t)(nd * bs));
      if (haveG)
        p = get(p, gvv, (size_t)nd);
      for (int v = 0; v < 4; ++v)
      {
        auto it = vert_l.find(gv[v]);
        int32_t lv;
        if (it == vert_l.end())
        {
          lv = (int32_t)vert_g.size();
          vert_l[gv[v]] = lv;
          vert_g.push_back(gv[v]);
          x.insert(x.end(), X + 3 * v, X + 3 * v + 3);
        }
        else
          lv = it->second;
        cverts.push_back(lv);
      }
      for (int i = 0; i < nd; ++i)
      {
        auto it = dof_l.find(gd[i]);
        int32_t l;
        if (it == dof_l.end())
        {
          if (ow[i] == me)
            return fail(ctx, ZZZ_ERR_ARG, "zzz_ghost_layer_build: rank %d sent an unknown dof %lld as owned by rank %d", r,
                        (long long)gd[i], me);
          l = (int32_t)(nloc + (int64_t)ng.size());
          dof_l[gd[i]] = l;
          NewGhost e;
          e.g = gd[i];
          e.owner = ow[i];
          for (int k = 0; k < 3; ++k)
          {
            e.bcf[k] = k < bs ? bf[i * bs + k] : 0;
            e.f[k] = (k < bs && haveF) ? fv[i * bs + k] : 0.0;
          }
          e.gv = haveG ? gvv[i] : 0.0;
          ng.push_back(e);
        }
        else
          l = it->second;
        cdofs.push_back(l);
      }
      fmask.push_back(fm);
    }
  }
  // ---- 3. ask the owners of the new ghosts for a place in the forward scatter --------------------------------
  for (size_t q = 0; q < ng.size(); ++q)
  {
    if (ng[q].owner < 0 || ng[q].owner >= nr || ng[q].owner == me)
      return fail(ctx, ZZZ_ERR_ARG, "zzz_ghost_layer_build: bad owner %d of dof %lld", ng[q].owner, (long long)ng[q].g);
    ng_of[(size_t)ng[q].owner].push_back((int32_t)q);
    put(req[(size_t)ng[q].owner], &ng[q].g, 1);
  }
  return ZZZ_OK;
  }();
  if ((rc = agree(ctx, rc)))
    return rc;
  const int64_t ncells_new = (int64_t)fmask.size();
  rc = comm_exchange_bytes(ctx, req, reqin);
  if (rc)
    return rc;
  std::vector<std::vector<int32_t>> extra_send((size_t)nr);
  std::vector<int32_t> neigh = old_neigh;
  std::vector<int64_t> send_off, recv_cnt;
  std::vector<int32_t> send_new;
  int64_t n_ghost_new = 0;
  rc = [&]() -> int {
  for (int r = 0; r < nr; ++r)
  {
    const size_t cnt = reqin[(size_t)r].size() / sizeof(int64_t);
    for (size_t q = 0; q < cnt; ++q)
    {
      int64_t g;
      memcpy(&g, reqin[(size_t)r].data() + q * sizeof(int64_t), sizeof(g));
      auto it = dof_l.find(g);
      if (it == dof_l.end() || it->second >= n_owned)
        return fail(ctx, ZZZ_ERR_ARG, "zzz_ghost_layer_build: rank %d asks rank %d for dof %lld, which it does not own", r, me,
                    (long long)g);
      extra_send[(size_t)r].push_back(it->second);
    }
  }

  // ---- 4. new neighbour list, ghost numbering (grouped by neighbour: old ghosts, then new ones) --------------
  for (int r = 0; r < nr; ++r)
    if ((!ng_of[(size_t)r].empty() || !extra_send[(size_t)r].empty())
        && std::find(neigh.begin(), neigh.end(), r) == neigh.end())
      neigh.push_back(r);
  const int nn = (int)neigh.size();
  send_off.assign((size_t)nn + 1, 0);
  recv_cnt.assign((size_t)nn, 0);
  std::vector<int32_t> remap((size_t)nloc + ng.size()); // old / temporary local block index -> new
  for (int64_t l = 0; l < n_owned; ++l)
    remap[(size_t)l] = (int32_t)l;
  int64_t gpos = n_owned, old_g = n_owned;
  for (int k = 0; k < nn; ++k)
  {
    const int r = neigh[(size_t)k];
    const bool was = k < (int)old_neigh.size();
    if (was)
    {
      for (int64_t i = old_send_off[(size_t)k]; i < old_send_off[(size_t)k + 1]; ++i)
        send_new.push_back(send_idx[(size_t)i]);
      for (int64_t i = 0; i < old_recv[(size_t)k]; ++i)
        remap[(size_t)old_g++] = (int32_t)gpos++;
      recv_cnt[(size_t)k] = old_recv[(size_t)k];
    }
    for (int32_t l : extra_send[(size_t)r])
      send_new.push_back(l);
    for (int32_t q : ng_of[(size_t)r])
      remap[(size_t)nloc + (size_t)q] = (int32_t)gpos++;
    recv_cnt[(size_t)k] += (int64_t)ng_of[(size_t)r].size();
    send_off[(size_t)k + 1] = (int64_t)send_new.size();
  }
  const int64_t nloc_new = gpos;
  n_ghost_new = nloc_new - n_owned;
  if (nloc_new * bs > INT32_MAX - 8)
    return fail(ctx, ZZZ_ERR_LIMIT, "%lld local scalar dofs with the ghost layer exceed int32", (long long)(nloc_new * bs));
  for (int32_t& l : cdofs)
    l = remap[(size_t)l];
  std::vector<int64_t> dof_g_new((size_t)nloc_new);
  std::vector<uint8_t> bc_new((size_t)(nloc_new * bs), 0);
  std::vector<double> cf_new(haveF ? (size_t)(nloc_new * bs) : 0), cg_new(haveG ? (size_t)nloc_new : 0);
  for (int64_t l = 0; l < nloc; ++l)
  {
    const int64_t t = remap[(size_t)l];
    dof_g_new[(size_t)t] = dof_g[(size_t)l];
    for (int k = 0; k < bs; ++k)
    {
      bc_new[(size_t)(t * bs + k)] = bc[(size_t)(l * bs + k)];
      if (haveF)
        cf_new[(size_t)(t * bs + k)] = cf[(size_t)(l * bs + k)];
    }
    if (haveG)
      cg_new[(size_t)t] = cg[(size_t)l];
  }
  for (size_t q = 0; q < ng.size(); ++q)
  {
    const int64_t t = remap[(size_t)nloc + q];
    dof_g_new[(size_t)t] = ng[q].g;
    for (int k = 0; k < bs; ++k)
    {
      bc_new[(size_t)(t * bs + k)] = ng[q].bcf[k];
      if (haveF)
        cf_new[(size_t)(t * bs + k)] = ng[q].f[k];
    }
    if (haveG)
      cg_new[(size_t)t] = ng[q].gv;
  }

  // ---- 5. the context with the ghost layer: same entry points a ghost-layer feed goes through -----------------
  const int64_t nverts_new = (int64_t)vert_g.size();
  rc = zzz_mesh_upload(ctx, nverts_new, x.data(), ncells_new, cverts.data());
  if (!rc)
    rc = zzz_dofmap_upload(ctx, order, bs, cdofs.data(), n_owned, n_ghost_new);
  if (rc)
    return rc;
  if (had_bc)
  {
    std::vector<int32_t> bcl;
    for (int64_t i = 0; i < nloc_new * bs; ++i)
      if (bc_new[(size_t)i])
        bcl.push_back((int32_t)i);
    rc = zzz_bc_upload(ctx, (int64_t)bcl.size(), bcl.empty() ? nullptr : bcl.data());
    if (rc)
      return rc;
  }
  {
    std::vector<int32_t> fp;
    for (int64_t c = 0; c < ncells_new; ++c)
      for (int f = 0; f < 4; ++f)
        if ((fmask[(size_t)c] >> f) & 1u)
        {
          fp.push_back((int32_t)c);
          fp.push_back(f);
        }
    rc = zzz_facets_upload(ctx, (int64_t)fp.size() / 2, fp.empty() ? nullptr : fp.data());
    if (rc)
      return rc;
  }
  if (haveF)
    rc = zzz_coeff_upload(ctx, ZZZ_COEFF_F, cf_new.data());
  if (!rc && haveG)
    rc = zzz_coeff_upload(ctx, ZZZ_COEFF_G, cg_new.data());
  if (!rc)
    rc = zzz_halo_upload(ctx, nn, neigh.data(), send_off.data(), send_new.empty() ? nullptr : send_new.data(), recv_cnt.data());
  if (rc)
    return rc;
  ctx->h_dof_global = dof_g_new;
  ctx->h_vert_global = vert_g;
  ctx->owned_cells = ncells;
  return ZZZ_OK;
  }();
  return agree(ctx, rc); // a rank whose re-upload failed must not leave the others believing the layer exists
}

int zzz_local_sizes(const zzz_ctx* ctx, int64_t sizes[6])
{
  if (!ctx || !sizes)
    return fail(nullptr, ZZZ_ERR_ARG, "zzz_local_sizes: bad arguments");
  sizes[0] = ctx->nverts;
  sizes[1] = ctx->ncells;
  sizes[2] = ctx->n_owned;
  sizes[3] = ctx->n_ghost;
  sizes[4] = ctx->owned_cells ? ctx->owned_cells : ctx->ncells;
  sizes[5] = ctx->nneigh;
  return ZZZ_OK;
}

} // extern "C"

// Host feed of `--mesh_type unstructured`: the ring-with-spurs mesh of src/mesh.cpp:209-453 (17 hexahedral blocks joined
// into a ring, a curling, tapering spur of 6 blocks on the outer face of each; 119 blocks x 6 tetrahedra), at the size the
// run asks for.  The reference reaches that size by Plaza refinement of the 714 coarse tetrahedra plus a bisection search
// over marked edges (DOLFINx refinement::refine, :407-452), which is not available here.  This feed subdivides every block
// into m x m x m trilinear sub-blocks instead and cuts each one with the block's own 6-tetrahedra pattern (the table
// `cube` of :233-235, all six share the diagonal 2-4), so that faces between blocks -- ring/ring and ring/spur -- carry
// the same diagonals from both sides, as in the reference's coarse mesh: conforming (checked below: every interior face
// has two cells), same geometry, not a lattice (curved, tapered, valence changes where the spurs meet the ring), but not
// the reference's refined mesh entity for entity.
//
// Partitions (zzzh_part_create_spoke_part): the dofs are cut into `nparts` sectors of equal size by the polar angle of
// their coordinates about the ring's axis; a partition owns one sector's dofs, holds every cell that touches one of them
// (owned-row assembly needs no exchange: the ghost-cell layer of the cube feed) and the other dofs of those cells as
// ghosts, grouped by owner.  Global numbering: owner-major, the generator's order inside an owner -- with one partition
// nothing is renumbered.  The ring closes on itself, spurs curl across sector borders: neighbour lists are whatever the
// mesh says, not "rank +- 1".  Every process builds the whole mesh and keeps its own part (a host feed for tests and
// single-node runs, not a distributed mesh generator: src/mesh.cpp:345-356 does the same on rank 0 before distributing).
//
// Dofs: generic, by sorting (vertices; edges as sorted vertex pairs; faces as sorted triples), Basix's local order, cells
// with their vertices ascending so that a cell's edge directions are the global ones.  Numbering: block by block in
// lattice order (what a mesh generator leaves); the library's own renumbering and the tests' permutations go on top.
#include "../../include/zzz_host.h"
#include "cube_layout.h"
#include "part_struct.h"

#include <algorithm>
#include <array>
#include <climits>
#include <cmath>
#include <cstdio>
#include <map>
#include <numeric>
#include <string>
#include <vector>

namespace
{
constexpr double PI = 3.14159265358979323846;

// coarse mesh of src/mesh.cpp:219-343: points, and per block its 8 points in the reference's local order
void coarse_mesh(std::vector<std::array<double, 3>>& x, std::vector<std::array<int, 8>>& hexes)
{
  constexpr int n = 17;
  constexpr double r0 = 0.25, r1 = 0.5, h0 = 1.2, h1 = 1.0;
  constexpr int lspur = 6;
  constexpr double l0 = 0.5, dth = 0.15, tap = 0.9;
  x.clear();
  hexes.clear();
  for (int i = 0; i < n; ++i) // the ring (:247-277)
  {
    std::array<int, 8> pts;
    for (int j = 0; j < 8; ++j)
      pts[(size_t)j] = (i * 4 + j) % (n * 4);
    hexes.push_back(pts);
    const double th = 2 * PI * i / n;
    x.push_back({r0 * std::cos(th), r0 * std::sin(th), h0});
    x.push_back({r0 * std::cos(th), r0 * std::sin(th), -h0});
    x.push_back({r1 * std::cos(th), r1 * std::sin(th), -h1});
    x.push_back({r1 * std::cos(th), r1 * std::sin(th), h1});
  }
  for (int i = 0; i < n; ++i) // the spurs (:280-324)
  {
    const double th0 = 2 * PI * (i + 0.5) / n;
    std::array<int, 8> pts = {(i * 4 + 2) % (n * 4), (i * 4 + 3) % (n * 4), (i * 4 + 7) % (n * 4), (i * 4 + 6) % (n * 4), 0, 0, 0, 0};
    for (int k = 0; k < lspur; ++k)
    {
      for (int j = 0; j < 4; ++j)
      {
        pts[(size_t)(j + 4)] = (int)x.size();
        std::array<double, 3> p = x[(size_t)pts[(size_t)j]];
        p[0] += l0 * std::cos(th0 + k * dth);
        p[1] += l0 * std::sin(th0 + k * dth);
        p[2] *= std::pow(tap, k);
        x.push_back(p);
      }
      hexes.push_back(pts);
      for (int j = 0; j < 4; ++j)
        pts[(size_t)j] = pts[(size_t)(j + 4)];
    }
  }
  // "Check geometric sizes and rescale" (:327-343) as written: the minima start at 0 and are taken over |x|, so
  // nothing is shifted, and every coordinate is divided by 0.9 max |x_0|
  double x0max = 0;
  for (auto& p : x)
    x0max = std::max(std::abs(p[0]), x0max);
  for (auto& p : x)
    for (int a = 0; a < 3; ++a)
      p[(size_t)a] /= 0.9 * x0max;
}

// local corner -> (u, v, w) of the block's trilinear map: 0..3 one cross-section (cyclic), 4..7 the next
constexpr int CU[8] = {0, 1, 1, 0, 0, 1, 1, 0}, CV[8] = {0, 0, 1, 1, 0, 0, 1, 1}, CW[8] = {0, 0, 0, 0, 1, 1, 1, 1};
constexpr int TET[6][4] = {{0, 1, 2, 4}, {1, 2, 4, 5}, {2, 4, 5, 6}, {0, 2, 3, 4}, {6, 7, 4, 2}, {2, 3, 4, 7}}; // src/mesh.cpp:233-235

// Basix's local entity order on the tetrahedron (include/zzz_abi.h)
constexpr int EDGE_V[6][2] = {{2, 3}, {1, 3}, {1, 2}, {0, 3}, {0, 2}, {0, 1}};
constexpr int FACE_V[4][3] = {{1, 2, 3}, {0, 2, 3}, {0, 1, 3}, {0, 1, 2}};
} // namespace

static zzzh_part* spoke_whole(int problem, int order, int m, int bc_mode)
{
  if (problem != ZZZH_POISSON && problem != ZZZH_ELASTICITY)
  {
    zzzh_set_error("unknown problem");
    return nullptr;
  }
  if (order < 1 || order > 3 || m < 1 || m > 200)
  {
    zzzh_set_error("spoke mesh: order 1..3 and 1 <= m <= 200 sub-blocks per block edge");
    return nullptr;
  }
  const int bs = problem == ZZZH_ELASTICITY ? 3 : 1;
  const int nd = order == 1 ? 4 : (order == 2 ? 10 : 20);
  std::vector<std::array<double, 3>> cx;
  std::vector<std::array<int, 8>> hexes;
  coarse_mesh(cx, hexes);
  const int nh = (int)hexes.size(), m1 = m + 1;

  // ---- vertices: lattice points of every block; those on a block's surface are shared and found by what they ARE:
  // the coarse points they interpolate with their (integer) weights
  typedef std::vector<std::pair<int, int64_t>> Key;
  std::map<Key, int32_t> shared;
  std::vector<double> x;
  std::vector<uint8_t> on_surface; // per vertex: on the surface of some block
  std::vector<int32_t> vid((size_t)nh * m1 * m1 * m1);
  for (int h = 0; h < nh; ++h)
    for (int c = 0; c <= m; ++c)
      for (int b = 0; b <= m; ++b)
        for (int a = 0; a <= m; ++a)
        {
          double p[3] = {0, 0, 0};
          Key key;
          for (int k = 0; k < 8; ++k)
          {
            const int64_t wa = CU[k] ? a : m - a, wb = CV[k] ? b : m - b, wc = CW[k] ? c : m - c, w = wa * wb * wc;
            if (w)
            {
              key.push_back({hexes[(size_t)h][(size_t)k], w});
              for (int d = 0; d < 3; ++d)
                p[d] += (double)w * cx[(size_t)hexes[(size_t)h][(size_t)k]][(size_t)d];
            }
          }
          const bool surface = a == 0 || a == m || b == 0 || b == m || c == 0 || c == m;
          int32_t id = -1;
          if (surface)
          {
            // a coarse point may appear twice among a block's corners only with the same weight pattern elsewhere: sort
            std::sort(key.begin(), key.end());
            auto it = shared.find(key);
            if (it != shared.end())
              id = it->second;
          }
          if (id < 0)
          {
            id = (int32_t)(x.size() / 3);
            const double s = 1.0 / ((double)m * m * m);
            x.push_back(p[0] * s);
            x.push_back(p[1] * s);
            x.push_back(p[2] * s);
            on_surface.push_back(surface ? 1 : 0);
            if (surface)
              shared[key] = id;
          }
          vid[(((size_t)h * m1 + c) * m1 + b) * m1 + a] = id;
        }
  const int64_t nverts = (int64_t)x.size() / 3;

  // ---- cells: per sub-block the block's 6-tetrahedra pattern, vertices ascending
  const int64_t ncells = (int64_t)nh * m * m * m * 6;
  std::vector<int32_t> cells((size_t)(4 * ncells));
  {
    int64_t c4 = 0;
    for (int h = 0; h < nh; ++h)
      for (int c = 0; c < m; ++c)
        for (int b = 0; b < m; ++b)
          for (int a = 0; a < m; ++a)
          {
            int32_t corner[8];
            for (int k = 0; k < 8; ++k)
              corner[k] = vid[(((size_t)h * m1 + (c + CW[k])) * m1 + (b + CV[k])) * m1 + (a + CU[k])];
            for (int t = 0; t < 6; ++t)
            {
              int32_t v[4] = {corner[TET[t][0]], corner[TET[t][1]], corner[TET[t][2]], corner[TET[t][3]]};
              std::sort(v, v + 4);
              if (v[0] == v[1] || v[1] == v[2] || v[2] == v[3])
              {
                zzzh_set_error("spoke mesh: degenerate cell");
                return nullptr;
              }
              for (int k = 0; k < 4; ++k)
                cells[(size_t)(c4++)] = v[k];
            }
          }
  }

  // ---- edges and faces by sorting; exterior facets = faces with one cell
  std::vector<std::array<int64_t, 2>> ekey; // (packed vertex pair, cell * 6 + local edge)
  std::vector<int32_t> cell_edge(order >= 2 ? (size_t)(6 * ncells) : 0), cell_face(order == 3 ? (size_t)(4 * ncells) : 0);
  int64_t nedges = 0, nfaces = 0;
  if (order >= 2)
  {
    ekey.resize((size_t)(6 * ncells));
    for (int64_t c = 0; c < ncells; ++c)
      for (int e = 0; e < 6; ++e)
      {
        const int64_t a = cells[(size_t)(4 * c + EDGE_V[e][0])], b = cells[(size_t)(4 * c + EDGE_V[e][1])];
        ekey[(size_t)(6 * c + e)] = {(a << 32) | b, 6 * c + e}; // a < b: the cell's vertices ascend
      }
    std::sort(ekey.begin(), ekey.end());
    for (size_t i = 0; i < ekey.size(); ++i)
    {
      if (i == 0 || ekey[i][0] != ekey[i - 1][0])
        ++nedges;
      cell_edge[(size_t)ekey[i][1]] = (int32_t)(nedges - 1);
    }
  }
  // (v0 << 32 | v1, v2, cell * 4 + local facet).  Below order 3 only the exterior facets are wanted, and those lie on
  // block surfaces: faces with a vertex inside a block have two cells and stay out of the sort (122 M keys -> 3 M at m = 35)
  std::vector<std::array<int64_t, 3>> fkey;
  fkey.reserve(order == 3 ? (size_t)(4 * ncells) : (size_t)nh * 6 * m * m * 8);
  for (int64_t c = 0; c < ncells; ++c)
    for (int f = 0; f < 4; ++f)
    {
      const int64_t a = cells[(size_t)(4 * c + FACE_V[f][0])], b = cells[(size_t)(4 * c + FACE_V[f][1])],
                    d = cells[(size_t)(4 * c + FACE_V[f][2])];
      if (order == 3 || (on_surface[(size_t)a] && on_surface[(size_t)b] && on_surface[(size_t)d]))
        fkey.push_back({(a << 32) | b, d, 4 * c + f});
    }
  std::sort(fkey.begin(), fkey.end());
  std::vector<int32_t> facets;
  for (size_t i = 0; i < fkey.size();)
  {
    size_t j = i + 1;
    while (j < fkey.size() && fkey[j][0] == fkey[i][0] && fkey[j][1] == fkey[i][1])
      ++j;
    if (j - i > 2)
    {
      zzzh_set_error("spoke mesh: a face with more than two cells (not conforming)");
      return nullptr;
    }
    if (order == 3)
      for (size_t k = i; k < j; ++k)
        cell_face[(size_t)fkey[k][2]] = (int32_t)nfaces;
    if (j - i == 1)
    {
      const int64_t cf = fkey[i][2];
      facets.push_back((int32_t)(cf / 4));
      facets.push_back((int32_t)(cf % 4));
    }
    ++nfaces;
    i = j;
  }
  {
    std::vector<std::pair<int32_t, int32_t>> fp(facets.size() / 2);
    for (size_t k = 0; k < fp.size(); ++k)
      fp[k] = {facets[2 * k], facets[2 * k + 1]};
    std::sort(fp.begin(), fp.end());
    for (size_t k = 0; k < fp.size(); ++k)
    {
      facets[2 * k] = fp[k].first;
      facets[2 * k + 1] = fp[k].second;
    }
  }

  // ---- dofmap and dof coordinates (gll_warped Lagrange nodes, src/poisson_problem.cpp:35-38)
  const int64_t ndofs = nverts + (order >= 2 ? (order - 1) * nedges : 0) + (order == 3 ? nfaces : 0);
  if (ndofs * bs > INT32_MAX - 8)
  {
    zzzh_set_error("spoke mesh: too many dofs for int32 local indexing");
    return nullptr;
  }
  zzzh_part* P = new zzzh_part();
  P->problem = problem;
  P->order = order;
  P->bs = bs;
  P->nd = nd;
  P->nparts = 1;
  P->part = 0;
  P->nx = P->ny = P->nz = m;
  P->x = x;
  P->cells = cells;
  P->cell_dofs.resize((size_t)(nd * ncells));
  P->dof_x.assign((size_t)(3 * ndofs), 0.0);
  P->facets = facets;
  std::vector<uint8_t> dof_on_boundary((size_t)ndofs, 0);
  const double t0 = 0.5 * (1.0 - 1.0 / std::sqrt(5.0)), t1 = 0.5 * (1.0 + 1.0 / std::sqrt(5.0));
  for (int64_t c = 0; c < ncells; ++c)
  {
    int32_t* cd = &P->cell_dofs[(size_t)(nd * c)];
    const int32_t* v = &cells[(size_t)(4 * c)];
    for (int k = 0; k < 4; ++k)
    {
      cd[k] = v[k];
      for (int a = 0; a < 3; ++a)
        P->dof_x[3 * (size_t)v[k] + a] = x[3 * (size_t)v[k] + a];
    }
    if (order >= 2)
      for (int e = 0; e < 6; ++e)
      {
        const int32_t a = v[EDGE_V[e][0]], b = v[EDGE_V[e][1]];
        const int64_t base = nverts + (int64_t)(order - 1) * cell_edge[(size_t)(6 * c + e)];
        for (int k = 0; k < order - 1; ++k)
        {
          const double t = order == 2 ? 0.5 : (k == 0 ? t0 : t1); // from the lower to the higher vertex
          cd[4 + (order - 1) * e + k] = (int32_t)(base + k);
          for (int d = 0; d < 3; ++d)
            P->dof_x[3 * (size_t)(base + k) + d] = (1 - t) * x[3 * (size_t)a + d] + t * x[3 * (size_t)b + d];
        }
      }
    if (order == 3)
      for (int f = 0; f < 4; ++f)
      {
        const int64_t dof = nverts + 2 * nedges + cell_face[(size_t)(4 * c + f)];
        cd[16 + f] = (int32_t)dof;
        for (int d = 0; d < 3; ++d)
          P->dof_x[3 * (size_t)dof + d]
              = (x[3 * (size_t)v[FACE_V[f][0]] + d] + x[3 * (size_t)v[FACE_V[f][1]] + d] + x[3 * (size_t)v[FACE_V[f][2]] + d]) / 3.0;
      }
  }
  // dofs on the closure of the exterior facets (bc_mode 1), or of those exterior facets ALL of whose vertices satisfy the
  // reference's marker (bc_mode 0: mesh::locate_entities(mesh, 2, marker) + locate_dofs_topological,
  // src/poisson_problem.cpp:58-75 -- a line of marked vertices that spans no facet constrains nothing)
  auto mark = [&](int mode) {
    std::fill(dof_on_boundary.begin(), dof_on_boundary.end(), (uint8_t)0);
    for (size_t k = 0; k < facets.size() / 2; ++k)
    {
      const int64_t c = facets[2 * k];
      const int f = facets[2 * k + 1];
      const int32_t* cd = &P->cell_dofs[(size_t)(nd * c)];
      if (mode != 1)
      {
        bool all = true;
        for (int q = 0; q < 3; ++q)
          all = all && zzzcube::is_dirichlet(problem, &x[3 * (size_t)cells[(size_t)(4 * c + FACE_V[f][q])]]);
        if (!all)
          continue;
      }
      for (int q = 0; q < 3; ++q)
        dof_on_boundary[(size_t)cd[FACE_V[f][q]]] = 1;
      if (order >= 2)
        for (int e = 0; e < 6; ++e) // edges of the facet: those that do not touch the opposite vertex f
          if (EDGE_V[e][0] != f && EDGE_V[e][1] != f)
            for (int k2 = 0; k2 < order - 1; ++k2)
              dof_on_boundary[(size_t)cd[4 + (order - 1) * e + k2]] = 1;
      if (order == 3)
        dof_on_boundary[(size_t)cd[16 + f]] = 1;
    }
    return std::count(dof_on_boundary.begin(), dof_on_boundary.end(), (uint8_t)1);
  };
  // Dirichlet dofs.  bc_mode 0: the reference's markers (src/poisson_problem.cpp:60-71, src/elasticity_problem.cpp:127-138:
  // |x| or |x - 1| < 1e-8, |y| < 1e-8) on whole facets -- on this geometry that set is EMPTY (the reference then solves a
  // singular system); bc_mode 1: every dof of the exterior boundary (a well-posed problem for tests and measurements);
  // bc_mode 2: the markers if they select anything, else the whole boundary
  int mode_used = bc_mode == 1 ? 1 : 0;
  if (mark(mode_used) == 0 && bc_mode == 2)
    mark(mode_used = 1);
  for (int64_t l = 0; l < ndofs; ++l)
  {
    if (dof_on_boundary[(size_t)l])
      for (int k = 0; k < bs; ++k)
        P->bc_dofs.push_back((int32_t)(l * bs + k));
  }
  P->coeff[0].resize((size_t)(ndofs * bs));
  if (problem == ZZZH_POISSON)
  {
    P->coeff[1].resize((size_t)ndofs);
    for (int64_t l = 0; l < ndofs; ++l)
    {
      P->coeff[0][(size_t)l] = zzzcube::poisson_f(&P->dof_x[3 * (size_t)l]);
      P->coeff[1][(size_t)l] = zzzcube::poisson_g(&P->dof_x[3 * (size_t)l]);
    }
  }
  else
    for (int64_t l = 0; l < ndofs; ++l)
      zzzcube::elasticity_f(&P->dof_x[3 * (size_t)l], &P->coeff[0][3 * (size_t)l]);
  P->global_dofs.resize((size_t)ndofs);
  std::iota(P->global_dofs.begin(), P->global_dofs.end(), (int64_t)0);
  P->global_verts.resize((size_t)nverts);
  std::iota(P->global_verts.begin(), P->global_verts.end(), (int64_t)0);
  P->send_off.push_back(0);
  int64_t* Sz = P->sizes;
  for (int i = 0; i < ZZZH_NSIZES; ++i)
    Sz[i] = 0;
  Sz[ZZZH_NVERTS] = nverts;
  Sz[ZZZH_NCELLS] = ncells;
  Sz[ZZZH_NOWNED] = ndofs;
  Sz[ZZZH_ND] = nd;
  Sz[ZZZH_BS] = bs;
  Sz[ZZZH_NFACETS] = (int64_t)P->facets.size() / 2;
  Sz[ZZZH_NBC] = (int64_t)P->bc_dofs.size();
  Sz[ZZZH_GLOBAL_DOFS] = ndofs * bs;
  Sz[ZZZH_GLOBAL_CELLS] = ncells;
  Sz[ZZZH_OWNED_CELLS] = ncells;
  Sz[ZZZH_GLOBAL_NBC] = (int64_t)P->bc_dofs.size();
  Sz[ZZZH_BC_MODE] = mode_used;
  return P;
}

// partition `part` of `nparts` cut out of the whole mesh G (see the head of this file)
static zzzh_part* spoke_extract(const zzzh_part& G, int nparts, int part)
{
  const int nd = G.nd, bs = G.bs;
  const int64_t n = G.sizes[ZZZH_NOWNED], ncells = G.sizes[ZZZH_NCELLS], nverts = G.sizes[ZZZH_NVERTS];
  // owners: equal chunks of the dofs sorted by polar angle (ties: generator order)
  std::vector<int32_t> owner((size_t)n);
  {
    std::vector<std::pair<double, int32_t>> key((size_t)n);
    for (int64_t l = 0; l < n; ++l)
      key[(size_t)l] = {std::atan2(G.dof_x[3 * (size_t)l + 1], G.dof_x[3 * (size_t)l]), (int32_t)l};
    std::sort(key.begin(), key.end());
    for (int64_t i = 0; i < n; ++i)
      owner[(size_t)key[(size_t)i].second] = (int32_t)(i * nparts / n);
  }
  // global numbering: owner-major, generator order inside an owner
  std::vector<int64_t> first((size_t)nparts + 1, 0), newg((size_t)n);
  for (int64_t l = 0; l < n; ++l)
    ++first[(size_t)owner[(size_t)l] + 1];
  for (int q = 0; q < nparts; ++q)
    first[(size_t)q + 1] += first[(size_t)q];
  {
    std::vector<int64_t> next(first.begin(), first.end() - 1);
    for (int64_t l = 0; l < n; ++l)
      newg[(size_t)l] = next[(size_t)owner[(size_t)l]]++;
  }
  // local cells (any dof owned here), ghosts, and what the others ghost from here
  std::vector<int64_t> lcells;
  std::vector<std::pair<int32_t, int64_t>> ghosts, wanted; // (owner, new global id) / (ghosting part, new global id of a dof owned here)
  int64_t owned_cells = 0;
  for (int64_t c = 0; c < ncells; ++c)
  {
    const int32_t* cd = &G.cell_dofs[(size_t)(nd * c)];
    bool mine = false, mixed = false;
    for (int i = 0; i < nd; ++i)
    {
      mine = mine || owner[(size_t)cd[i]] == part;
      mixed = mixed || owner[(size_t)cd[i]] != owner[(size_t)cd[0]];
    }
    if (!mine)
      continue;
    lcells.push_back(c);
    if (owner[(size_t)cd[0]] == part)
      ++owned_cells; // (a cell counts where its first dof lives: the parts' counts add up to the mesh)
    if (!mixed)
      continue;
    for (int i = 0; i < nd; ++i)
    {
      const int32_t q = owner[(size_t)cd[i]];
      if (q != part)
      {
        ghosts.push_back({q, newg[(size_t)cd[i]]});
        for (int j = 0; j < nd; ++j) // the cell is local to q as well: q ghosts every dof of it owned here
          if (owner[(size_t)cd[j]] == part)
            wanted.push_back({q, newg[(size_t)cd[j]]});
      }
    }
  }
  auto uniq = [](std::vector<std::pair<int32_t, int64_t>>& v) {
    std::sort(v.begin(), v.end());
    v.erase(std::unique(v.begin(), v.end()), v.end());
  };
  uniq(ghosts);
  uniq(wanted);
  const int64_t n_owned = first[(size_t)part + 1] - first[(size_t)part], n_ghost = (int64_t)ghosts.size(), nloc = n_owned + n_ghost;
  // local index of a global (generator-numbered) dof
  std::vector<int64_t> g_of_newg((size_t)n);
  for (int64_t l = 0; l < n; ++l)
    g_of_newg[(size_t)newg[(size_t)l]] = l;
  std::vector<int32_t> loc((size_t)n, -1);
  std::vector<int64_t> gen((size_t)nloc); // local -> generator number
  for (int64_t i = 0; i < n_owned; ++i)
    gen[(size_t)i] = g_of_newg[(size_t)(first[(size_t)part] + i)];
  for (int64_t i = 0; i < n_ghost; ++i)
    gen[(size_t)(n_owned + i)] = g_of_newg[(size_t)ghosts[(size_t)i].second];
  for (int64_t i = 0; i < nloc; ++i)
    loc[(size_t)gen[(size_t)i]] = (int32_t)i;

  zzzh_part* P = new zzzh_part();
  P->problem = G.problem;
  P->order = G.order;
  P->bs = bs;
  P->nd = nd;
  P->nparts = nparts;
  P->part = part;
  P->nx = G.nx;
  P->ny = G.ny;
  P->nz = G.nz;
  // vertices of the local cells, ascending in the global numbering (cells keep their vertices ascending)
  std::vector<int32_t> vloc((size_t)nverts, -1);
  for (int64_t c : lcells)
    for (int k = 0; k < 4; ++k)
      vloc[(size_t)G.cells[(size_t)(4 * c + k)]] = 0;
  int64_t nv = 0;
  for (int64_t v = 0; v < nverts; ++v)
    if (vloc[(size_t)v] == 0)
    {
      vloc[(size_t)v] = (int32_t)nv++;
      P->global_verts.push_back(v);
      for (int a = 0; a < 3; ++a)
        P->x.push_back(G.x[3 * (size_t)v + a]);
    }
  std::vector<int32_t> cloc((size_t)ncells, -1);
  P->cells.reserve(4 * lcells.size());
  P->cell_dofs.reserve((size_t)nd * lcells.size());
  for (size_t k = 0; k < lcells.size(); ++k)
  {
    const int64_t c = lcells[k];
    cloc[(size_t)c] = (int32_t)k;
    for (int q = 0; q < 4; ++q)
      P->cells.push_back(vloc[(size_t)G.cells[(size_t)(4 * c + q)]]);
    for (int i = 0; i < nd; ++i)
      P->cell_dofs.push_back(loc[(size_t)G.cell_dofs[(size_t)(nd * c + i)]]);
  }
  for (size_t k = 0; k < G.facets.size() / 2; ++k) // (sorted by cell, and the local cells keep their order)
    if (cloc[(size_t)G.facets[2 * k]] >= 0)
    {
      P->facets.push_back(cloc[(size_t)G.facets[2 * k]]);
      P->facets.push_back(G.facets[2 * k + 1]);
    }
  std::vector<uint8_t> marked((size_t)(n * bs), 0);
  for (int32_t d : G.bc_dofs)
    marked[(size_t)d] = 1;
  P->dof_x.resize((size_t)(3 * nloc));
  P->global_dofs.resize((size_t)nloc);
  P->coeff[0].resize((size_t)(nloc * bs));
  if (G.problem == ZZZH_POISSON)
    P->coeff[1].resize((size_t)nloc);
  for (int64_t i = 0; i < nloc; ++i)
  {
    const int64_t l = gen[(size_t)i];
    P->global_dofs[(size_t)i] = newg[(size_t)l];
    for (int a = 0; a < 3; ++a)
      P->dof_x[3 * (size_t)i + a] = G.dof_x[3 * (size_t)l + a];
    for (int k = 0; k < bs; ++k)
    {
      P->coeff[0][(size_t)(i * bs + k)] = G.coeff[0][(size_t)(l * bs + k)];
      if (marked[(size_t)(l * bs + k)])
        P->bc_dofs.push_back((int32_t)(i * bs + k));
    }
    if (G.problem == ZZZH_POISSON)
      P->coeff[1][(size_t)i] = G.coeff[1][(size_t)l];
  }
  // forward-scatter plan: neighbours ascending = the order of the ghost groups; to q go the dofs q ghosts from here, in
  // q's ghost order (ascending global number inside an owner's group)
  P->send_off.push_back(0);
  {
    size_t ig = 0, iw = 0;
    while (ig < ghosts.size() || iw < wanted.size())
    {
      const int32_t q = std::min(ig < ghosts.size() ? ghosts[ig].first : INT32_MAX, iw < wanted.size() ? wanted[iw].first : INT32_MAX);
      int64_t nrecv = 0;
      for (; ig < ghosts.size() && ghosts[ig].first == q; ++ig)
        ++nrecv;
      for (; iw < wanted.size() && wanted[iw].first == q; ++iw)
        P->send_idx.push_back((int32_t)(wanted[iw].second - first[(size_t)part]));
      P->neigh.push_back(q);
      P->recv_cnt.push_back(nrecv);
      P->send_off.push_back((int64_t)P->send_idx.size());
    }
  }
  int64_t* Sz = P->sizes;
  for (int i = 0; i < ZZZH_NSIZES; ++i)
    Sz[i] = 0;
  Sz[ZZZH_NVERTS] = nv;
  Sz[ZZZH_NCELLS] = (int64_t)lcells.size();
  Sz[ZZZH_NOWNED] = n_owned;
  Sz[ZZZH_NGHOST] = n_ghost;
  Sz[ZZZH_ND] = nd;
  Sz[ZZZH_BS] = bs;
  Sz[ZZZH_NFACETS] = (int64_t)P->facets.size() / 2;
  Sz[ZZZH_NBC] = (int64_t)P->bc_dofs.size();
  Sz[ZZZH_NNEIGH] = (int64_t)P->neigh.size();
  Sz[ZZZH_NSEND] = (int64_t)P->send_idx.size();
  Sz[ZZZH_GLOBAL_DOFS] = n * bs;
  Sz[ZZZH_GLOBAL_CELLS] = ncells;
  Sz[ZZZH_OWNED_CELLS] = owned_cells;
  Sz[ZZZH_OWN_OFFSET] = first[(size_t)part];
  Sz[ZZZH_GLOBAL_NBC] = G.sizes[ZZZH_GLOBAL_NBC];
  Sz[ZZZH_BC_MODE] = G.sizes[ZZZH_BC_MODE];
  return P;
}

extern "C" zzzh_part* zzzh_part_create_spoke(int problem, int order, int m, int bc_mode)
{
  return spoke_whole(problem, order, m, bc_mode);
}

extern "C" zzzh_part* zzzh_part_create_spoke_part(int problem, int order, int m, int bc_mode, int nparts, int part)
{
  if (nparts < 1 || part < 0 || part >= nparts)
  {
    zzzh_set_error("spoke mesh: bad partition");
    return nullptr;
  }
  zzzh_part* G = spoke_whole(problem, order, m, bc_mode);
  if (!G || nparts == 1)
    return G;
  if (G->sizes[ZZZH_NOWNED] < nparts)
  {
    zzzh_part_destroy(G);
    zzzh_set_error("spoke mesh: fewer dofs than partitions");
    return nullptr;
  }
  zzzh_part* P = spoke_extract(*G, nparts, part);
  zzzh_part_destroy(G);
  return P;
}

// smallest m whose mesh reaches `target` nodes of the order-k space (vertices + (k-1) edges + faces for k = 3), the
// role of the refinement loop of src/mesh.cpp:357-368; counts from the closed forms of one block minus what blocks share
extern "C" int zzzh_spoke_size(int64_t target_nodes, int order)
{
  for (int m = 1; m <= 200; ++m)
  {
    // 119 blocks; 17 + 17 * 6 = 119 interfaces between blocks (each ring block shares two cross-sections -- 17 in all --
    // and each spur block its base)
    const int64_t m1 = m + 1, nb = 119, nif = 119;
    const int64_t V = nb * m1 * m1 * m1 - nif * m1 * m1;
    const int64_t E = nb * (7 * (int64_t)m * m * m + 9 * (int64_t)m * m + 3 * m) - nif * (3 * (int64_t)m * m + 2 * m);
    const int64_t F = nb * (12 * (int64_t)m * m * m + 6 * (int64_t)m * m) - nif * (2 * (int64_t)m * m);
    const int64_t n = order == 1 ? V : (order == 2 ? V + E : V + 2 * E + F);
    if (n >= target_nodes)
      return m;
  }
  return 200;
}

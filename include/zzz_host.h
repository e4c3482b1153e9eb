/*
 * zzz_host.h -- C API of libzzz_host.so: the host-side "feed" of the hot path (pure C++, no GPU).
 *
 * It produces, per mesh partition, exactly the arrays the device library (zzz_abi.h) consumes:
 * what DOLFINx hands to fem::assemble_* in the reference -- geometry, cell connectivity, dofmap,
 * Dirichlet dofs, exterior facets, interpolated coefficients and the ghost-exchange plan.
 * Reference code it replaces (paths relative to FEniCS/performance-test):
 *   mesh size search .............. src/mesh.cpp:44-151 (restated exactly)
 *   create_box + partitioning ..... src/mesh.cpp:153-186 (own structured z-slab partitioner;
 *                                   ParMETIS/SCOTCH/KaHIP and Plaza refinement are out of scope:
 *                                   a mesh "refined r times" is generated directly at its
 *                                   effective size (Nx<<r) x (Ny<<r) x (Nz<<r), which has the same
 *                                   entity and dof counts, src/mesh.cpp:44-54)
 *   create_functionspace .......... src/poisson_problem.cpp:35-44, src/elasticity_problem.cpp:103-111
 *   locate_entities/locate_dofs ... src/poisson_problem.cpp:58-75, src/elasticity_problem.cpp:125-142
 *   Function::interpolate ......... src/poisson_problem.cpp:83-106, src/elasticity_problem.cpp:153-176
 *   create_entities(2) / ds facets  src/main.cpp:147-148
 *   IndexMap / Scatterer plan ..... src/cgpoisson_problem.cpp:187-190
 */
#ifndef ZZZ_HOST_H
#define ZZZ_HOST_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct zzzh_part zzzh_part;

/* src/mesh.cpp:56-74; -1 for an unsupported order */
int64_t zzzh_num_pdofs(int64_t i, int64_t j, int64_t k, int nrefine, int order);
/* src/mesh.cpp:44-54: out = {vertices, edges, faces, cells} */
void zzzh_num_entities(int64_t i, int64_t j, int64_t k, int nrefine, int64_t out[4]);
/* src/mesh.cpp:78-151: out = {Nx, Ny, Nz, r}. strong != 0: ndofs is the total; else per process. */
void zzzh_mesh_size(int64_t ndofs, int strong, int64_t num_processes, int64_t dofs_per_node, int order,
                    int64_t out[4]);

/* The suffix the "Test problem summary" prints after a count (int64_to_human, src/main.cpp:31-50):
 * "" up to 1000, else " (<3 significant digits> thousand|million|billion|trillion)", the count being divided
 * by 1000 while it EXCEEDS 1000 (so 1 000 000 prints " (1e+03 thousand)", as the reference does).  Writes at most
 * `cap` bytes incl. the terminator; returns the length, or -1 for a number too big (the reference throws). */
int zzzh_count_suffix(int64_t n, char* out, int cap);

enum
{
  ZZZH_POISSON = 0,   /* scalar P_k, Dirichlet on x = 0 and x = 1, coefficients f and g */
  ZZZH_ELASTICITY = 1 /* vector P_k (bs 3), Dirichlet on y = 0, coefficient f */
};

/* indices into the sizes array of zzzh_part_sizes */
enum
{
  ZZZH_NVERTS = 0,
  ZZZH_NCELLS,      /* local cells (owned layers + one ghost layer) */
  ZZZH_NOWNED,      /* owned block dofs */
  ZZZH_NGHOST,      /* ghost block dofs */
  ZZZH_ND,          /* dofs per cell */
  ZZZH_BS,
  ZZZH_NFACETS,     /* exterior facets among the local cells */
  ZZZH_NBC,         /* constrained local scalar dofs */
  ZZZH_NNEIGH,
  ZZZH_NSEND,       /* total block dofs sent per forward scatter */
  ZZZH_GLOBAL_DOFS, /* index_map.size_global() * bs (src/main.cpp:178-180) */
  ZZZH_GLOBAL_CELLS,
  ZZZH_OWNED_CELLS, /* cells of this partition's own layers (sum over parts = global cells) */
  ZZZH_OWN_OFFSET,  /* global block index of local owned dof 0 (owned range is contiguous) */
  ZZZH_GLOBAL_NBC,  /* constrained scalar dofs of the whole problem (all partitions) */
  ZZZH_BC_MODE,     /* spoke mesh: the bc_mode in effect (2 resolves to 0 or 1); cube: 0 */
  ZZZH_NSIZES
};

/* Partition `part` of `nparts` z-slabs of the nx*ny*nz unit cube (6 tetrahedra per sub-cube).
 * Returns NULL on bad arguments (message via zzzh_last_error). */
zzzh_part* zzzh_part_create(int problem, int order, int64_t nx, int64_t ny, int64_t nz, int nparts, int part);
/* The same partition as the reference's cell partitioner leaves it (GhostMode::none, src/mesh.cpp:182-183): the
 * partition's own cells only, ghost dofs = those dofs of the own cells that a neighbour owns.  The rows of owned
 * dofs on the partition interface are then incomplete locally (the reference completes them in MatAssemblyBegin/
 * End and scatter_rev, src/poisson_problem.cpp:132-137,154); feed for zzz_ghost_layer_build (include/zzz_abi.h). */
zzzh_part* zzzh_part_create_native(int problem, int order, int64_t nx, int64_t ny, int64_t nz, int nparts, int part);
/* `--mesh_type unstructured`: the ring-with-spurs mesh of create_spoke_mesh (src/mesh.cpp:209-453), every one of its 119
 * hexahedral blocks cut into m x m x m sub-blocks x 6 tetrahedra (in place of the reference's Plaza refinement and edge
 * bisection, :357-452: same geometry and coarse topology, conforming, not a lattice -- but not the reference's refined
 * mesh entity for entity).  One partition.  Generic dofmaps by sorting (P1-P3, scalar and vector-valued).
 * bc_mode 0: the reference's Dirichlet markers (src/poisson_problem.cpp:60-71, src/elasticity_problem.cpp:127-138) --
 * on this geometry possibly an empty set, as in the reference; 1: every dof of the exterior boundary; 2: the reference's
 * markers if they select anything, else the whole exterior boundary (sizes[ZZZH_BC_MODE] says which). */
zzzh_part* zzzh_part_create_spoke(int problem, int order, int m, int bc_mode);
/* Partition `part` of `nparts` of that mesh (mpirun -np N of the reference's CI, .github/workflows/ccpp.yml:102-117): the
 * dofs cut into nparts sectors of equal size by polar angle about the ring's axis; a partition owns a sector's dofs, holds
 * every cell touching one of them and the remaining dofs of those cells as ghosts (grouped by owner, ascending).  Global
 * numbering owner-major (one partition: the generator's, nothing renumbered).  Neighbours are whatever the mesh says --
 * the ring closes on itself.  Every caller builds the whole mesh and keeps its part. */
zzzh_part* zzzh_part_create_spoke_part(int problem, int order, int m, int bc_mode, int nparts, int part);
/* smallest m whose mesh has (about) `target_nodes` nodes of the order-k space: the refinement loop of src/mesh.cpp:357-368 */
int zzzh_spoke_size(int64_t target_nodes, int order);
void zzzh_part_destroy(zzzh_part* p);
const char* zzzh_last_error(void);

void zzzh_part_sizes(const zzzh_part* p, int64_t sizes[ZZZH_NSIZES]);
const double* zzzh_part_x(const zzzh_part* p);             /* nverts*3 */
const int32_t* zzzh_part_cells(const zzzh_part* p);        /* ncells*4, vertices ascending */
const int32_t* zzzh_part_cell_dofs(const zzzh_part* p);    /* ncells*nd */
const int32_t* zzzh_part_facets(const zzzh_part* p);       /* nfacets*2 */
const int32_t* zzzh_part_bc_dofs(const zzzh_part* p);      /* nbc local scalar dofs */
const double* zzzh_part_dof_x(const zzzh_part* p);         /* (nowned+nghost)*3 */
const int64_t* zzzh_part_global_dofs(const zzzh_part* p);  /* (nowned+nghost) global block index */
const int64_t* zzzh_part_global_verts(const zzzh_part* p); /* nverts global vertex index */
const double* zzzh_part_coeff(const zzzh_part* p, int which); /* 0: f, 1: g (Poisson only) */
const int32_t* zzzh_part_neigh(const zzzh_part* p);        /* nneigh ranks */
const int64_t* zzzh_part_send_off(const zzzh_part* p);     /* nneigh+1 */
const int32_t* zzzh_part_send_idx(const zzzh_part* p);     /* nsend owned block dofs */
const int64_t* zzzh_part_recv_cnt(const zzzh_part* p);     /* nneigh */

#ifdef __cplusplus
}
#endif
#endif

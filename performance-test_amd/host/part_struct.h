// The partition object behind include/zzz_host.h (shared by mesh_part.cpp and spoke_mesh.cpp).
#pragma once
#include "../../include/zzz_host.h"

#include <cstdint>
#include <vector>

struct zzzh_part
{
  int problem, order, bs, nd, nparts, part;
  int64_t nx, ny, nz;
  int64_t sizes[ZZZH_NSIZES];
  std::vector<double> x, dof_x, coeff[2];
  std::vector<int32_t> cells, cell_dofs, facets, bc_dofs, neigh, send_idx;
  std::vector<int64_t> global_dofs, global_verts, send_off, recv_cnt;
};
void zzzh_set_error(const char* msg); // mesh_part.cpp: what zzzh_last_error returns

// dolfinx-scaling-test -- MI355X-native drop-in for the hot path of FEniCS/performance-test.
//
// Keeps the reference driver's surface (src/main.cpp): the double-hyphen options (:57-74, unknown
// options ignored as with allow_unregistered(), :79), PETSc-style single-hyphen solver options
// (-ksp_type cg -pc_type jacobi -ksp_rtol ..., consumed by solver.set_from_options() at
// src/poisson_problem.cpp:169), the ZZZ timers (README.md:148-161) and the stdout summary
// (:186-205, :232-233).  Everything between the timers is this repository's own code: the host
// feed (mesh_part.cpp, C++) and the device library (libzzz_hip.so) behind the C-ABI of
// include/zzz_abi.h.  One host thread drives one GPU ("process" = GPU: `--ngpus N` plays the role
// of `mpirun -np N`); RCCL carries the halo and the CG all-reduces.  There is no CPU compute path.
#include "../../include/zzz_abi.h"
#include "../../include/zzz_host.h"

#include <algorithm>
#include <barrier>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iomanip>
#include <iostream>
#include <map>
#include <mutex>
#include <sstream>
#include <stdexcept>
#include <string>
#include <atomic>
#include <ctime>
#include <fstream>
#include <unistd.h>
#include <thread>
#include <vector>

namespace
{
// ---- timers: dolfinx::common::Timer + list_timings (src/main.cpp:130,226) ------------------------
struct TimerRegistry
{
  std::mutex m;
  std::vector<std::string> order;
  std::map<std::string, std::pair<int, double>> t; // reps, total seconds
  void add(const std::string& name, double s)
  {
    std::lock_guard<std::mutex> lk(m);
    auto it = t.find(name);
    if (it == t.end())
    {
      order.push_back(name);
      t[name] = {1, s};
    }
    else
    {
      it->second.first++;
      it->second.second += s;
    }
  }
  void list() const
  {
    size_t w = 30;
    for (auto& n : order)
      w = std::max(w, n.size());
    std::cout << "\n[MPI_MAX] Summary of timings" << std::string(w - 17, ' ') << " |  reps  wall avg  wall tot\n";
    std::cout << std::string(w + 30, '-') << "\n";
    for (auto& n : order)
    {
      auto& e = t.at(n);
      std::cout << std::left << std::setw((int)w + 11) << n << " | " << std::right << std::setw(5) << e.first << "  "
                << std::fixed << std::setprecision(6) << e.second / e.first << "  " << e.second << "\n";
    }
    std::cout.unsetf(std::ios::fixed);
    std::cout << std::setprecision(6) << std::endl;
  }
};
TimerRegistry g_timers;

// Per-thread stopwatch; rank 0 registers the MAX over ranks (the driver passes it after a barrier).
struct Timer
{
  std::string name;
  std::chrono::steady_clock::time_point t0;
  bool running = true;
  explicit Timer(std::string n) : name(std::move(n)), t0(std::chrono::steady_clock::now()) {}
  double stop()
  {
    running = false;
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }
};

// the " (10 million)" suffix of the summary block (zzzh_count_suffix, include/zzz_host.h)
std::string count_suffix(std::int64_t n)
{
  char buf[64];
  if (zzzh_count_suffix(n, buf, (int)sizeof(buf)) < 0)
    throw std::runtime_error("number too big");
  return buf;
}

struct Options
{
  // src/main.cpp:57-74
  std::string problem_type = "poisson", mesh_type = "cube", scaling_type = "weak", output, scatterer = "neighbor";
  bool mem_profile = false, use_subcomm = false, help = false;
  std::size_t ndofs = 50000, order = 1;
  // this driver only: number of GPUs ("processes") and the communicator between them
  int ngpus = 1;
  std::string comm = "rccl"; // "local": host-mediated exchange, all ranks on GPU 0 (validation on one GPU)
  std::string allreduce = "peer"; // CG scalar all-reduces: "peer" memory mailboxes (falls back) | "comm" (RCCL / local)
  // poisson only: "assembled" = the reference's path (AIJ matrix, MatMult); "matfree" = KSPCG + PCJACOBI on the matrix-free
  // operator of cgpoisson: no matrix, the diagonal from the element matrices (an extension; faster from P2 up)
  std::string op = "assembled";
  // PETSc options database (README.md:66-82)
  std::string ksp_type = "cg", pc_type = "jacobi", ksp_norm_type = "preconditioned";
  double ksp_rtol = 1e-5, ksp_atol = 1e-50, ksp_divtol = 1e4; // PETSc defaults (KSPCreate)
  int ksp_max_it = 10000;
  int pc_degree = 0;      // -pc_chebyshev_jacobi_degree (0: the library's default, 3)
  double pc_ratio = 0.0;  // -pc_chebyshev_jacobi_ratio  (0: the library's default, 60)
  int pc_esteig = 0;      // -pc_chebyshev_jacobi_esteig (0: the library's default, 10 Lanczos steps; < 0: Gershgorin alone)
  bool ksp_view = false, log_view = false, options_left = false, ksp_monitor = false, ksp_cg_single_reduction = false;
  bool ksp_error_if_not_converged = false, ksp_converged_reason = false;
  std::vector<std::string> unused;
};

void usage()
{
  // boost::program_options layout of the reference's help (src/main.cpp:84-89)
  std::cout << "Allowed options:\n"
               "  -h [ --help ]                   print usage message\n"
               "  --problem_type arg (=poisson)   problem (poisson, cgpoisson, or elasticity)\n"
               "  --mesh_type arg (=cube)         mesh (cube or unstructured)\n"
               "  --memory_profiling              turn on memory logging\n"
               "  --subcomm_partition             Use sub-communicator for partitioning\n"
               "  --scaling_type arg (=weak)      scaling (weak or strong)\n"
               "  --output arg                    output directory (no output unless this is set)\n"
               "  --ndofs arg (=50000)            number of degrees of freedom\n"
               "  --order arg (=1)                polynomial order\n"
               "  --scatterer arg (=neighbor)     scatterer for CG (neighbor or p2p)\n"
               "  --ngpus arg (=1)                number of GPUs (takes the place of mpirun -np)\n"
               "  --comm arg (=rccl)              rccl | local (host-mediated, all ranks on GPU 0: validation)\n"
               "  --allreduce arg (=peer)         peer (xGMI peer-memory mailboxes, else falls back) | comm\n"
               "  --operator arg (=assembled)     poisson: assembled (the AIJ matrix) | matfree (KSPCG on the matrix-free\n"
               "                                  operator, Jacobi from the element matrices' diagonals; no matrix)\n"
               "PETSc-style solver options honoured: -ksp_type cg -pc_type {jacobi,none,chebyshev_jacobi} -ksp_rtol -ksp_atol\n"
               "  -ksp_divtol -pc_chebyshev_jacobi_degree (=3) -pc_chebyshev_jacobi_ratio (=60) -pc_chebyshev_jacobi_esteig (=10)\n"
               "  -ksp_max_it -ksp_norm_type {preconditioned,unpreconditioned,natural} -ksp_view -ksp_monitor\n"
               "  -ksp_cg_single_reduction -ksp_converged_reason -ksp_error_if_not_converged\n"
               "  -log_view -options_left\n"
            << std::endl;
}

Options parse(int argc, char** argv)
{
  Options o;
  auto value = [&](int& i, const std::string& arg, const std::string& key) -> std::string {
    const size_t eq = arg.find('=');
    if (eq != std::string::npos)
      return arg.substr(eq + 1);
    if (i + 1 >= argc)
      throw std::runtime_error("the required argument for option '--" + key + "' is missing");
    return argv[++i];
  };
  for (int i = 1; i < argc; ++i)
  {
    const std::string arg = argv[i];
    if (arg == "-h" || arg == "--help")
      o.help = true;
    else if (arg.rfind("--", 0) == 0)
    {
      const std::string key = arg.substr(2, arg.find('=') == std::string::npos ? std::string::npos : arg.find('=') - 2);
      if (key == "problem_type")
        o.problem_type = value(i, arg, key);
      else if (key == "mesh_type")
        o.mesh_type = value(i, arg, key);
      else if (key == "scaling_type")
        o.scaling_type = value(i, arg, key);
      else if (key == "output")
        o.output = value(i, arg, key);
      else if (key == "scatterer")
        o.scatterer = value(i, arg, key);
      else if (key == "ndofs")
        o.ndofs = std::stoull(value(i, arg, key));
      else if (key == "order")
        o.order = std::stoull(value(i, arg, key));
      else if (key == "ngpus")
        o.ngpus = std::stoi(value(i, arg, key));
      else if (key == "comm")
        o.comm = value(i, arg, key);
      else if (key == "allreduce")
        o.allreduce = value(i, arg, key);
      else if (key == "operator")
        o.op = value(i, arg, key);
      else if (key == "memory_profiling")
        o.mem_profile = true;
      else if (key == "subcomm_partition")
        o.use_subcomm = true;
      else
        o.unused.push_back(arg); // allow_unregistered(), src/main.cpp:79
    }
    else if (arg.size() > 1 && arg[0] == '-')
    {
      // PETSc options database: "-name [value]"
      const std::string key = arg.substr(1);
      auto next = [&]() -> std::string {
        if (i + 1 < argc && !(argv[i + 1][0] == '-' && !std::isdigit((unsigned char)argv[i + 1][1]) && argv[i + 1][1] != '.'))
          return argv[++i];
        return "";
      };
      if (key == "ksp_type")
        o.ksp_type = next();
      else if (key == "pc_type")
        o.pc_type = next();
      else if (key == "pc_chebyshev_jacobi_degree")
        o.pc_degree = std::stoi(next());
      else if (key == "pc_chebyshev_jacobi_ratio")
        o.pc_ratio = std::stod(next());
      else if (key == "pc_chebyshev_jacobi_esteig")
        o.pc_esteig = std::stoi(next());
      else if (key == "ksp_rtol")
        o.ksp_rtol = std::stod(next());
      else if (key == "ksp_atol")
        o.ksp_atol = std::stod(next());
      else if (key == "ksp_divtol")
        o.ksp_divtol = std::stod(next());
      else if (key == "ksp_max_it")
        o.ksp_max_it = std::stoi(next());
      else if (key == "ksp_norm_type")
        o.ksp_norm_type = next();
      else if (key == "ksp_view")
        o.ksp_view = true;
      else if (key == "ksp_monitor")
        o.ksp_monitor = true;
      else if (key == "ksp_error_if_not_converged")
        o.ksp_error_if_not_converged = true;
      else if (key == "ksp_converged_reason")
        o.ksp_converged_reason = true;
      else if (key == "ksp_cg_single_reduction")
      {
        // PETSc bool option: bare flag, or followed by true/false/1/0
        o.ksp_cg_single_reduction = true;
        if (i + 1 < argc)
        {
          std::string v = argv[i + 1];
          if (v == "true" || v == "1" || v == "yes")
            ++i;
          else if (v == "false" || v == "0" || v == "no")
          {
            o.ksp_cg_single_reduction = false;
            ++i;
          }
        }
      }
      else if (key == "log_view")
        o.log_view = true;
      else if (key == "options_left")
        o.options_left = true;
      else
      {
        std::string v = next();
        o.unused.push_back(arg + (v.empty() ? "" : " " + v));
      }
    }
    else
      o.unused.push_back(arg);
  }
  return o;
}

#define ZCK(ctx, call)                                                                          \
  do                                                                                            \
  {                                                                                             \
    int rc_ = (call);                                                                           \
    if (rc_ != 0)                                                                               \
      throw std::runtime_error(std::string(#call) + " failed: " + zzz_last_error(ctx));         \
  } while (0)

struct Shared
{
  Options opt;
  int nranks = 1;
  std::int64_t dims[4] = {0, 0, 0, 0};
  unsigned char uid[ZZZ_UNIQUE_ID_BYTES] = {0};
  void* local_group = nullptr; // --comm local
  std::vector<unsigned char> p2p_handles; // nranks x ZZZ_P2P_HANDLE_BYTES
  std::vector<int> p2p_enabled;
  int spoke_m = 0;            // --mesh_type unstructured: sub-blocks per block edge
  std::vector<std::int64_t> out_rows; // --output: owned dofs per rank
  int out_bs = 1;
  std::vector<double> tmax;   // scratch for max-over-ranks timing
  std::vector<double> tcg;    // cgpoisson: time of the linalg::cg call alone (the Gdof/s line)
  std::vector<double> tplan;  // cgpoisson: set-up of the matrix-free plan (inside ZZZ Solve, outside the Gdof/s timer)
  std::vector<int> iters;
  std::vector<double> norm, rnorm0, rnorm;
  std::vector<std::string> error;
  std::int64_t num_dofs = 0, num_cells = 0;
};

// problem() + the ZZZ Solve block of solve(): one rank (GPU)
void run_rank(Shared& S, std::barrier<>& bar, int rank)
{
  const Options& o = S.opt;
  const bool root = rank == 0;
  zzz_ctx* ctx = nullptr;
  // A rank that fails keeps arriving at the DRIVER's barriers (catch, record, drain).  Its peers may meanwhile be
  // waiting for it inside a collective of the library: with the local communicator the group is broken (every
  // waiting rank gets an error and drains too); with RCCL nothing can wake them, so the process ends at once
  // with a non-zero status instead of hanging.
  bool failed = false;
  // what a rank does once it has failed anywhere (set-up phase, ZZZ Solve, the closing norm): record, and wake or end
  // the peers that may be waiting for it inside a collective
  auto rank_failed = [&](const std::exception& e) {
    failed = true;
    S.error[rank] = e.what();
    if (S.nranks > 1)
    {
      if (S.local_group)
        zzz_local_group_abort(S.local_group);
      else
      {
        std::cerr << "rank " << rank << ": " << e.what() << "\n(other ranks may be waiting in an RCCL collective: aborting)"
                  << std::endl;
        std::_Exit(2);
      }
    }
  };
  auto phase = [&](const char* tname, auto&& body) {
    Timer t(tname ? tname : "");
    if (!failed)
    {
      try
      {
        body();
        if (ctx)
          ZCK(ctx, zzz_sync(ctx)); // GPU work is asynchronous: a timer brackets a device sync
      }
      catch (const std::exception& e)
      {
        rank_failed(e);
      }
    }
    const double s = t.stop();
    S.tmax[rank] = s;
    bar.arrive_and_wait();
    if (root && tname)
      g_timers.add(tname, *std::max_element(S.tmax.begin(), S.tmax.end()));
    bar.arrive_and_wait();
  };

  const int problem = o.problem_type == "elasticity" ? ZZZH_ELASTICITY : ZZZH_POISSON;
  const int form = problem == ZZZH_ELASTICITY ? ZZZ_FORM_ELASTICITY : ZZZ_FORM_POISSON;
  const bool cgpoisson = o.problem_type == "cgpoisson";
  const bool matfree_op = o.op == "matfree";

  // The structured feed (mesh, function space, Dirichlet set, coefficients, halo plan) is generated
  // on the GPU in closed form (zzz_cube_generate); the reference's setup timers are kept as rows.
  std::int64_t info[6] = {0, 0, 0, 0, 0, 0};
  phase("ZZZ Create Mesh", [&] {
    ZCK(nullptr, zzz_ctx_create(S.local_group ? 0 : rank, &ctx));
    if (S.local_group)
      ZCK(ctx, zzz_comm_init_local(ctx, S.local_group, rank));
    else if (S.nranks > 1)
      ZCK(ctx, zzz_comm_init(ctx, S.nranks, rank, S.uid));
    const int r = (int)S.dims[3];
    if (o.mesh_type == "cube")
      ZCK(ctx, zzz_cube_generate(ctx, form, (int)o.order, S.dims[0] << r, S.dims[1] << r, S.dims[2] << r, S.nranks, rank, info));
    else
    {
      // create_spoke_mesh (src/mesh.cpp:209-453) through the host feed and the upload entry points; with several ranks each
      // takes its sector of the cut by polar angle (host/spoke_mesh.cpp).  The reference's Dirichlet markers (|x| or
      // |x - 1| < 1e-8; |y| < 1e-8) may select nothing on this geometry -- the reference then solves a singular system;
      // here the whole exterior boundary is constrained instead (bc_mode 2), and it says so
      zzzh_part* P = zzzh_part_create_spoke_part(problem, (int)o.order, S.spoke_m, 2, S.nranks, rank);
      if (!P)
        throw std::runtime_error(zzzh_last_error());
      std::int64_t sz[ZZZH_NSIZES];
      zzzh_part_sizes(P, sz);
      if (root && sz[ZZZH_BC_MODE] == 1)
        std::cout << "Unstructured mesh: the reference's Dirichlet markers select no facet of this geometry; the whole "
                     "exterior boundary is constrained (" << sz[ZZZH_GLOBAL_NBC] << " dofs)" << std::endl;
      struct Guard
      {
        zzzh_part* p;
        ~Guard() { zzzh_part_destroy(p); }
      } guard{P};
      ZCK(ctx, zzz_mesh_upload(ctx, sz[ZZZH_NVERTS], zzzh_part_x(P), sz[ZZZH_NCELLS], zzzh_part_cells(P)));
      ZCK(ctx, zzz_dofmap_upload(ctx, (int)o.order, (int)sz[ZZZH_BS], zzzh_part_cell_dofs(P), sz[ZZZH_NOWNED], sz[ZZZH_NGHOST]));
      if (S.nranks > 1)
        ZCK(ctx, zzz_halo_upload(ctx, (int)sz[ZZZH_NNEIGH], zzzh_part_neigh(P), zzzh_part_send_off(P), zzzh_part_send_idx(P),
                                 zzzh_part_recv_cnt(P)));
      ZCK(ctx, zzz_bc_upload(ctx, sz[ZZZH_NBC], zzzh_part_bc_dofs(P)));
      ZCK(ctx, zzz_facets_upload(ctx, sz[ZZZH_NFACETS], zzzh_part_facets(P)));
      ZCK(ctx, zzz_coeff_upload(ctx, ZZZ_COEFF_F, zzzh_part_coeff(P, 0)));
      if (problem == ZZZH_POISSON)
        ZCK(ctx, zzz_coeff_upload(ctx, ZZZ_COEFF_G, zzzh_part_coeff(P, 1)));
      info[0] = sz[ZZZH_GLOBAL_DOFS];
      info[1] = sz[ZZZH_GLOBAL_CELLS];
    }
  });
  if (root && !failed)
  {
    S.num_dofs = info[0];
    S.num_cells = info[1];
  }
  // CG scalar all-reduces through peer memory (xGMI stores into the peers' mailboxes) when every rank
  // can map every peer; otherwise the communicator's all-reduce stays in use
  if (S.nranks > 1 && o.allreduce == "peer")
  {
    phase(nullptr, [&] { ZCK(ctx, zzz_comm_p2p_export(ctx, S.p2p_handles.data() + (size_t)rank * ZZZ_P2P_HANDLE_BYTES)); });
    phase(nullptr, [&] { ZCK(ctx, zzz_comm_p2p_attach(ctx, S.p2p_handles.data(), &S.p2p_enabled[rank])); });
  }
  phase("ZZZ FunctionSpace", [&] {});
  phase("ZZZ Create facets and facet->cell connectivity", [&] {});

  Timer umbrella("ZZZ Assemble"); // poisson/cgpoisson only in the reference (src/poisson_problem.cpp:49)
  phase("ZZZ Create boundary conditions", [&] {});
  phase("ZZZ Create RHS function", [&] {});
  if (problem == ZZZH_ELASTICITY)
    phase("ZZZ Create forms", [&] {});
  // fem::petsc::create_matrix: untimed in the reference, inside the ZZZ Assemble umbrella
  phase(nullptr, [&] { ZCK(ctx, zzz_csr_pattern_build(ctx)); });
  if (matfree_op)
    // no matrix: what takes its place is the plan of the matrix-free kernel (the diagonal is PCSetUp's, in ZZZ Solve)
    phase("ZZZ Assemble matrix", [&] { ZCK(ctx, zzz_matfree_setup(ctx)); });
  else if (!cgpoisson)
    phase("ZZZ Assemble matrix", [&] { ZCK(ctx, zzz_assemble_matrix(ctx, form)); });
  phase("ZZZ Assemble vector", [&] { ZCK(ctx, zzz_assemble_vector(ctx, form)); });
  if (problem == ZZZH_ELASTICITY)
    // build_near_nullspace (src/elasticity_problem.cpp:233-244): six orthonormalised rigid-body modes; GAMG would consume
    // them (MatSetNearNullSpace), Jacobi-CG does not
    phase("ZZZ Create near-nullspace", [&] {
      double dev = 0;
      ZCK(ctx, zzz_near_nullspace_build(ctx, &dev));
    });
  {
    S.tmax[rank] = umbrella.stop();
    bar.arrive_and_wait();
    if (root && problem == ZZZH_POISSON)
      g_timers.add("ZZZ Assemble", *std::max_element(S.tmax.begin(), S.tmax.end()));
    bar.arrive_and_wait();
  }

  if (root && !failed)
  {
    // src/main.cpp:173-206
    std::cout << "----------------------------------------------------------------" << std::endl;
    std::cout << "Test problem summary" << std::endl;
    std::cout << "  dolfinx version: n/a (libzzz_hip 0.1.0, MI355X/gfx950-native hot path)" << std::endl;
    std::cout << "  dolfinx hash:    n/a" << std::endl;
    std::cout << "  ufl hash:        n/a (hand-written HIP element kernels)" << std::endl;
    std::cout << "  petsc version:   n/a (own CSR + CG on HIP/RCCL)" << std::endl;
    std::cout << "  Problem type:    " << o.problem_type << std::endl;
    std::cout << "  Scaling type:    " << o.scaling_type << std::endl;
    std::cout << "  Num processes:   " << S.nranks << std::endl;
    std::cout << "  Num cells:       " << S.num_cells << count_suffix(S.num_cells) << std::endl;
    std::cout << "  Total degrees of freedom:               " << S.num_dofs << count_suffix(S.num_dofs) << std::endl;
    std::cout << "  Average degrees of freedom per process: " << S.num_dofs / S.nranks << std::endl;
    std::cout << "----------------------------------------------------------------" << std::endl;
  }

  zzz_solver_opts so;
  std::memset(&so, 0, sizeof(so));
  if (cgpoisson)
  {
    // linalg::cg(*u.x(), b, action, 100, 1e-6), src/cgpoisson_problem.cpp:233
    so.variant = ZZZ_CG_CGH;
    so.pc = ZZZ_PC_NONE;
    so.op = ZZZ_OP_MATFREE;
    so.max_it = 100;
    so.rtol = 1e-6;
  }
  else
  {
    so.variant = ZZZ_CG_PETSC;
    so.pc = o.pc_type == "none" ? ZZZ_PC_NONE : o.pc_type == "chebyshev_jacobi" ? ZZZ_PC_CHEBYSHEV_JACOBI : ZZZ_PC_JACOBI;
    so.pc_degree = o.pc_degree;
    so.pc_ratio = o.pc_ratio;
    so.pc_esteig_its = o.pc_esteig;
    so.norm = o.ksp_norm_type == "unpreconditioned" ? ZZZ_NORM_UNPRECONDITIONED
              : o.ksp_norm_type == "natural"        ? ZZZ_NORM_NATURAL
                                                    : ZZZ_NORM_PRECONDITIONED;
    so.op = matfree_op ? ZZZ_OP_MATFREE : ZZZ_OP_CSR;
    so.max_it = o.ksp_max_it;
    so.rtol = o.ksp_rtol;
    so.atol = o.ksp_atol;
    so.dtol = o.ksp_divtol;
    so.single_reduction = o.ksp_cg_single_reduction ? 1 : 0;
    so.error_if_not_converged = o.ksp_error_if_not_converged ? 1 : 0;
  }
  double solve_s = 0, cg_s = 0;
  {
    Timer ts("ZZZ Solve");
    if (!failed)
    {
      try
      {
        double rn[2] = {0, 0};
        // cgpoisson: what solver_function sets up before it calls linalg::cg -- coefficient storage of `un`, the
        // Scatterer and its buffers (src/cgpoisson_problem.cpp:181-190) -- is inside ZZZ Solve and outside the
        // timer of the Gdof/s line (:232-235); here that is the plan of the matrix-free kernel
        if (cgpoisson)
        {
          Timer tplan("plan");
          ZCK(ctx, zzz_matfree_setup(ctx));
          ZCK(ctx, zzz_sync(ctx));
          S.tplan[rank] = tplan.stop();
        }
        Timer tcg("cg");
        ZCK(ctx, zzz_cg_solve(ctx, &so, &S.iters[rank], rn));
        ZCK(ctx, zzz_sync(ctx));
        S.tcg[rank] = tcg.stop();
        S.rnorm[rank] = rn[0];
        S.rnorm0[rank] = rn[1];
      }
      catch (const std::exception& e)
      {
        rank_failed(e);
      }
    }
    S.tmax[rank] = ts.stop();
    bar.arrive_and_wait();
    solve_s = *std::max_element(S.tmax.begin(), S.tmax.end());
    cg_s = *std::max_element(S.tcg.begin(), S.tcg.end());
    if (root)
      g_timers.add("ZZZ Solve", solve_s);
    bar.arrive_and_wait();
  }
  if (root && cgpoisson && !failed)
  {
    // src/cgpoisson_problem.cpp:236-241
    const double gdofs = (S.iters[0] * (double)S.num_dofs) / cg_s / 1e9;
    std::cout << "CG matrix-free action processed: " << gdofs << " Gdof/s\n";
    // no line of the reference: what the Gdof/s figure leaves out here (the plan is built once per dofmap; ~40 actions' worth)
    std::cout << "CG matrix-free plan set-up (once per dofmap, not in the Gdof/s figure): "
              << *std::max_element(S.tplan.begin(), S.tplan.end()) * 1e3 << " ms\n";
  }
  // --output <dir> (src/main.cpp:213-223: io::XDMFFile(...).write_mesh / write_function under `ZZZ Output`).  A minimal,
  // HDF5-free XDMF: one Polyvertex grid per process -- the coordinates of its owned dofs and the solution there as raw
  // little-endian float64 files (u_p<rank>.bin, x_p<rank>.bin), tied together by <dir>/solution.xdmf (ParaView reads
  // it; the cell connectivity is not written: the mesh lives on the device and only the solution is the run's result)
  if (!o.output.empty())
  {
    phase("ZZZ Output", [&] {
      std::int64_t ls[6] = {0, 0, 0, 0, 0, 0};
      ZCK(ctx, zzz_local_sizes(ctx, ls));
      const std::int64_t n_owned = ls[2], n_ghost = ls[3], bsz = problem == ZZZH_ELASTICITY ? 3 : 1;
      std::vector<double> u((size_t)((n_owned + n_ghost) * bsz));
      ZCK(ctx, zzz_vec_download(ctx, ZZZ_VEC_U, u.data()));
      zzzh_part* P = o.mesh_type == "cube"
                         ? zzzh_part_create(problem, (int)o.order, S.dims[0] << S.dims[3], S.dims[1] << S.dims[3],
                                            S.dims[2] << S.dims[3], S.nranks, rank)
                         : zzzh_part_create_spoke_part(problem, (int)o.order, S.spoke_m, 2, S.nranks, rank);
      if (!P)
        throw std::runtime_error(zzzh_last_error());
      const std::string base = o.output + "/";
      {
        std::ofstream fu(base + "u_p" + std::to_string(rank) + ".bin", std::ios::binary);
        fu.write(reinterpret_cast<const char*>(u.data()), (std::streamsize)(n_owned * bsz * 8));
        std::ofstream fx(base + "x_p" + std::to_string(rank) + ".bin", std::ios::binary);
        fx.write(reinterpret_cast<const char*>(zzzh_part_dof_x(P)), (std::streamsize)(n_owned * 3 * 8));
        if (!fu || !fx)
        {
          zzzh_part_destroy(P);
          throw std::runtime_error("--output: cannot write under '" + o.output + "'");
        }
      }
      zzzh_part_destroy(P);
      S.out_rows[rank] = n_owned;
      S.out_bs = (int)bsz;
    });
    if (root && !failed)
    {
      std::ofstream fx(o.output + "/solution.xdmf");
      fx << "<?xml version=\"1.0\" ?>\n<Xdmf Version=\"3.0\">\n <Domain>\n  <Grid Name=\"u\" GridType=\"Collection\" "
            "CollectionType=\"Spatial\">\n";
      for (int r = 0; r < S.nranks; ++r)
      {
        const std::int64_t n = S.out_rows[r];
        fx << "   <Grid Name=\"p" << r << "\" GridType=\"Uniform\">\n    <Topology TopologyType=\"Polyvertex\" NumberOfElements=\""
           << n << "\"/>\n    <Geometry GeometryType=\"XYZ\"><DataItem Format=\"Binary\" DataType=\"Float\" Precision=\"8\" "
              "Endian=\"Little\" Dimensions=\"" << n << " 3\">x_p" << r << ".bin</DataItem></Geometry>\n    <Attribute Name=\"u\" "
              "AttributeType=\"" << (S.out_bs == 3 ? "Vector" : "Scalar") << "\" Center=\"Node\"><DataItem Format=\"Binary\" "
              "DataType=\"Float\" Precision=\"8\" Endian=\"Little\" Dimensions=\"" << n << (S.out_bs == 3 ? " 3" : "")
           << "\">u_p" << r << ".bin</DataItem></Attribute>\n   </Grid>\n";
      }
      fx << "  </Grid>\n </Domain>\n</Xdmf>\n";
    }
  }
  if (!failed)
  {
    try
    {
      ZCK(ctx, zzz_vec_norm(ctx, ZZZ_VEC_U, &S.norm[rank])); // la::norm(*u->x()), src/main.cpp:229 (collective)
    }
    catch (const std::exception& e)
    {
      rank_failed(e);
    }
  }
  bar.arrive_and_wait();
  if (root && o.ksp_converged_reason && !failed)
  {
    // KSPConvergedReasonView's line; a diverged solve is reported, not fatal (src/poisson_problem.cpp:172-178)
    std::int64_t ci[4] = {0, 0, 0, 0};
    zzz_cg_info(ctx, ci);
    const char* name = ci[2] == 2 ? "CONVERGED_RTOL" : ci[2] == 3 ? "CONVERGED_ATOL" : ci[2] == -3 ? "DIVERGED_ITS"
                       : ci[2] == -4 ? "DIVERGED_DTOL" : ci[2] == -9 ? "DIVERGED_NANORINF" : "unknown";
    std::cout << "Linear solve " << (ci[2] > 0 ? "converged" : "did not converge") << " due to " << name << " iterations "
              << S.iters[0] << std::endl;
  }
  if (root && o.ksp_monitor && !failed)
  {
    std::vector<double> h((size_t)S.iters[0] + 1);
    zzz_cg_history(ctx, (int)h.size(), h.data());
    for (size_t k = 0; k < h.size(); ++k)
      std::cout << std::setw(3) << k << " KSP Residual norm " << std::scientific << std::setprecision(12) << h[k] << "\n";
    std::cout.unsetf(std::ios::scientific);
  }
  if (ctx)
    zzz_ctx_destroy(ctx);
}

void solve(int argc, char** argv)
{
  Options o = parse(argc, argv);
  if (o.help)
  {
    usage();
    return;
  }
  bool strong;
  if (o.scaling_type == "strong")
    strong = true;
  else if (o.scaling_type == "weak")
    strong = false;
  else
    throw std::runtime_error("Scaling type '" + o.scaling_type + "` unknown"); // src/main.cpp:115
  if (o.problem_type != "poisson" && o.problem_type != "cgpoisson" && o.problem_type != "elasticity")
    throw std::runtime_error("Unknown problem type: " + o.problem_type); // src/main.cpp:170
  // src/main.cpp:131-141: "cube", anything else is the unstructured (spoke) mesh
  if (o.ksp_type != "cg")
    throw std::runtime_error("-ksp_type " + o.ksp_type + ": only cg is built");
  if (o.pc_type != "jacobi" && o.pc_type != "none" && o.pc_type != "chebyshev_jacobi")
    throw std::runtime_error("-pc_type " + o.pc_type +
                             ": only jacobi, none and chebyshev_jacobi are built (hypre/gamg are out of scope)");
  if (o.order < 1 || o.order > 3)
    throw std::out_of_range("vector::_M_range_check: order must be 1..3"); // form_*.at(order - 1)
  const int ndev = zzz_device_count();
  if (ndev < 1)
    throw std::runtime_error("no GPU visible: this build has no CPU path");
  if (o.comm != "rccl" && o.comm != "local")
    throw std::runtime_error("--comm " + o.comm + ": rccl or local");
  if (o.allreduce != "peer" && o.allreduce != "comm")
    throw std::runtime_error("--allreduce " + o.allreduce + ": peer or comm");
  if (o.op != "assembled" && o.op != "matfree")
    throw std::runtime_error("--operator " + o.op + ": assembled or matfree");
  if (o.op == "matfree" && o.problem_type != "poisson")
    throw std::runtime_error("--operator matfree applies to --problem_type poisson (cgpoisson is matrix-free already; the "
                             "matrix-free kernel holds the Poisson form only)");
  if (o.op == "matfree" && (o.pc_type == "chebyshev_jacobi" || o.ksp_cg_single_reduction))
    throw std::runtime_error("--operator matfree: -pc_type jacobi or none, classical CG");
  if (o.ngpus < 1 || (o.comm == "rccl" && o.ngpus > ndev))
    throw std::runtime_error("--ngpus " + std::to_string(o.ngpus) + " but " + std::to_string(ndev) + " GPU(s) visible");

  Shared S;
  S.opt = o;
  S.nranks = o.ngpus;
  const int ndofs_per_node = (o.problem_type == "elasticity") ? 3 : 1; // src/main.cpp:128
  if (o.mesh_type == "cube")
  {
    zzzh_mesh_size((std::int64_t)o.ndofs, strong ? 1 : 0, S.nranks, ndofs_per_node, (int)o.order, S.dims);
    if (S.dims[0] < 1 || S.dims[1] < 1 || S.dims[2] < 1)
      throw std::runtime_error("mesh size search returned a non-positive dimension (ndofs too small)");
    // src/mesh.cpp:190-194
    std::cout << "UnitCube (" << S.dims[0] << "x" << S.dims[1] << "x" << S.dims[2] << ") to be refined " << S.dims[3]
              << " times" << std::endl;
  }
  else
  {
    // create_spoke_mesh's target (src/mesh.cpp:213-216) and what stands in for its refinement loop (:357-452)
    std::int64_t target = (std::int64_t)o.ndofs / ndofs_per_node;
    if (!strong)
      target *= S.nranks;
    S.spoke_m = zzzh_spoke_size(target, (int)o.order);
    std::cout << "Create unstructured mesh: 119 blocks of the ring-with-spurs geometry, each cut " << S.spoke_m << "x" << S.spoke_m
              << "x" << S.spoke_m << " (x 6 tetrahedra)" << std::endl;
  }
  if (o.comm == "local")
  {
    if (zzz_local_group_create(S.nranks, &S.local_group) != 0)
      throw std::runtime_error(zzz_last_error(nullptr));
  }
  else if (S.nranks > 1)
    if (zzz_comm_unique_id(S.uid) != 0)
      throw std::runtime_error(zzz_last_error(nullptr));
  S.tmax.assign(S.nranks, 0.0);
  S.tcg.assign(S.nranks, 0.0);
  S.tplan.assign(S.nranks, 0.0);
  S.out_rows.assign(S.nranks, 0);
  S.iters.assign(S.nranks, 0);
  S.norm.assign(S.nranks, 0.0);
  S.rnorm.assign(S.nranks, 0.0);
  S.rnorm0.assign(S.nranks, 0.0);
  S.error.assign(S.nranks, "");
  S.p2p_handles.assign((size_t)S.nranks * ZZZ_P2P_HANDLE_BYTES, 0);
  S.p2p_enabled.assign(S.nranks, 0);

  // --memory_profiling (src/main.cpp:102-107,236-240; src/mem.cpp:18-38): a thread logs VSIZE and RSS of the
  // process from /proc/self/stat every 100 ms; the used HBM of GPU 0 is appended (the device side is where
  // this build keeps its data)
  std::atomic<bool> mem_quit{false};
  std::thread mem_thread;
  if (o.mem_profile)
    mem_thread = std::thread([&mem_quit] {
      const long page_kb = sysconf(_SC_PAGE_SIZE) / 1024;
      while (!mem_quit.load())
      {
        std::ifstream f("/proc/self/stat");
        std::string tok;
        for (int i = 0; i < 22 && (f >> tok); ++i)
        {
        }
        unsigned long long vsize = 0, rss = 0;
        f >> vsize >> rss;
        size_t hbm_free = 0, hbm_total = 0;
        zzz_device_memory(0, &hbm_free, &hbm_total);
        const auto now = std::chrono::system_clock::now();
        const std::time_t tt = std::chrono::system_clock::to_time_t(now);
        const int ms = (int)(std::chrono::duration_cast<std::chrono::milliseconds>(now.time_since_epoch()).count() % 1000);
        char stamp[32];
        std::strftime(stamp, sizeof(stamp), "%Y-%m-%d %H:%M:%S", std::localtime(&tt));
        std::fprintf(stderr, "[%s.%03d] [MEM] [warning] VSIZE=%llu, RSS=%llu, HBM=%zu\n", stamp, ms, vsize / 1024, rss * page_kb,
                     (hbm_total - hbm_free) / 1024);
        std::this_thread::sleep_for(std::chrono::milliseconds(100));
      }
    });

  std::barrier<> bar(S.nranks);
  std::vector<std::thread> th;
  for (int r = 1; r < S.nranks; ++r)
    th.emplace_back(run_rank, std::ref(S), std::ref(bar), r);
  run_rank(S, bar, 0);
  for (auto& t : th)
    t.join();
  if (mem_thread.joinable())
  {
    mem_quit.store(true);
    mem_thread.join();
  }
  if (S.local_group)
    zzz_local_group_destroy(S.local_group);
  for (int r = 0; r < S.nranks; ++r)
    if (!S.error[r].empty())
      throw std::runtime_error("rank " + std::to_string(r) + ": " + S.error[r]);

  if (o.ksp_view)
    std::cout << "KSP Object: type: cg\n  maximum iterations=" << o.ksp_max_it << ", initial guess is zero\n  tolerances:  relative="
              << o.ksp_rtol << ", absolute=" << o.ksp_atol << "\n  using " << o.ksp_norm_type
              << " norm type for convergence test\n"
              << (o.ksp_cg_single_reduction ? "  using single-reduction variant\n" : "") << "PC Object: type: " << o.pc_type
              << (o.op == "matfree" ? "\n  linear system matrix: type=shell (matrix-free action, diagonal from the element matrices) on "
                                    : "\n  linear system matrix: type=csr (fp64 values, int32 indices) on ")
              << S.nranks << " MI355X\n"
              << (S.nranks > 1 ? (S.p2p_enabled[0] ? "  scalar all-reduces: peer-memory mailboxes; halo: peer-memory window where the plan fits\n"
                                                    : "  scalar all-reduces and halo: communicator\n")
                               : "");
  g_timers.list(); // dolfinx::list_timings, src/main.cpp:226
  // src/main.cpp:229-234
  std::cout << "*** Number of Krylov iterations: " << S.iters[0] << std::endl;
  std::cout << "*** Solution norm:  " << S.norm[0] << std::endl;
  if (o.options_left && !o.unused.empty())
  {
    std::cout << "#PETSc Option Table entries:\n";
    for (auto& u : o.unused)
      std::cout << "option left (not used by this build): " << u << "\n";
  }
}
} // namespace

int main(int argc, char* argv[])
{
  // Init MPI / Init logging / Init PETSc of the reference (src/main.cpp:245-258) have no counterpart
  // here; the rows are kept so that tools reading the timing table find them.
  for (const char* n : {"Init MPI", "Init logging", "Init PETSc"})
    g_timers.add(n, 0.0);
  try
  {
    solve(argc, argv);
  }
  catch (const std::exception& e)
  {
    // the reference lets exceptions terminate the process (non-zero exit)
    std::cerr << "terminate called after throwing an instance of 'std::runtime_error'\n  what():  " << e.what() << std::endl;
    return 134;
  }
  return 0;
}

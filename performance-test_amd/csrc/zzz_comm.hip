// Inter-GPU communication over RCCL (xGMI): the 8-byte all-reduces of the CG dot products
// (MPI_Allreduce inside la::inner_product / la::squared_norm, src/cg.h:53,65,74) and the forward
// halo scatter (common::Scatterer::scatter_fwd, src/cgpoisson_problem.cpp:225-229).
//
// librccl is loaded with dlopen the first time a communicator is asked for, so single-GPU runs do
// not depend on it.  All calls are enqueued on the context's stream: the CG loop stays free of
// host synchronisation.  The halo is a grouped ncclSend/ncclRecv per neighbour (<= 7 peers on one
// node = one xGMI link each); owned values are packed by a gather kernel, received values land
// directly in the ghost segment, which the partitioner orders by (neighbour, sender order).
#include "zzz_device.h"
#include "zzz_internal.h"

#include <dlfcn.h>
#include <pthread.h>
#include <unistd.h>

#include <cstdlib>
#include <cstring>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <string>
#include <vector>

namespace zzz
{
typedef struct ncclComm* ncclComm_t;
typedef struct
{
  char internal[128];
} ncclUniqueId;
typedef int ncclResult_t;
enum
{
  ncclFloat64 = 8,
  ncclSum = 0
};
static_assert(sizeof(ncclUniqueId) == ZZZ_UNIQUE_ID_BYTES, "unique id size");

struct Rccl
{
  void* h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommSplit)(ncclComm_t, int, int, ncclComm_t*, void*) = nullptr; // optional
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

static Rccl g_rccl;

static const char* load_rccl_once();

static const char* load_rccl()
{
  // several driver threads may attach communicators at the same time
  static std::mutex m;
  std::lock_guard<std::mutex> lk(m);
  return load_rccl_once();
}

static std::string g_rccl_err;  // why the last load failed (dlerror text), under load_rccl's mutex
static std::string g_rccl_path; // file the bound librccl was loaded from (dladdr)

static const char* load_rccl_once()
{
  if (g_rccl.h)
    return nullptr;
  const char* names[] = {"librccl.so.1", "librccl.so"};
  void* h = nullptr;
  g_rccl_err.clear();
  for (const char* n : names)
  {
    if ((h = dlopen(n, RTLD_NOW | RTLD_LOCAL)))
      break;
    const char* de = dlerror(); // ONE call: dlerror clears the message it returns
    g_rccl_err += std::string(g_rccl_err.empty() ? "" : "; ") + (de ? de : "dlopen failed");
  }
  if (!h)
    return "cannot dlopen librccl.so.1";
#define ZZZ_SYM(field, name)                                   \
  *(void**)(&g_rccl.field) = dlsym(h, name);                   \
  if (!g_rccl.field)                                           \
  {                                                            \
    g_rccl = Rccl();                                           \
    dlclose(h);                                                \
    g_rccl_err = "symbol missing";                             \
    return "librccl lacks " name;                              \
  }
  ZZZ_SYM(GetUniqueId, "ncclGetUniqueId")
  ZZZ_SYM(CommInitRank, "ncclCommInitRank")
  ZZZ_SYM(CommDestroy, "ncclCommDestroy")
  ZZZ_SYM(AllReduce, "ncclAllReduce")
  ZZZ_SYM(Send, "ncclSend")
  ZZZ_SYM(Recv, "ncclRecv")
  ZZZ_SYM(GroupStart, "ncclGroupStart")
  ZZZ_SYM(GroupEnd, "ncclGroupEnd")
  ZZZ_SYM(GetErrorString, "ncclGetErrorString")
#undef ZZZ_SYM
  *(void**)(&g_rccl.CommSplit) = dlsym(h, "ncclCommSplit"); // absent in old RCCL: then no halo overlap
  g_rccl.h = h;
  Dl_info di;
  if (dladdr(reinterpret_cast<void*>(g_rccl.AllReduce), &di) && di.dli_fname)
    g_rccl_path = di.dli_fname;
  return nullptr;
}

// RCCL (ROCm 7.2) prints a version banner with printf to STDOUT while a communicator is created.
// Programs whose stdout is parsed (bench.py prints one JSON line) must not see it: send fd 1 to
// stderr for the duration of the call.
// Several rank threads of one process (the driver's --ngpus N) enter the rendezvous calls together: the redirect
// is counted under a mutex -- the first one in saves the real stdout and redirects, the last one out restores it.
struct StdoutToStderr
{
  static std::mutex& mtx()
  {
    static std::mutex m;
    return m;
  }
  static int& users()
  {
    static int n = 0;
    return n;
  }
  static int& saved_fd()
  {
    static int fd = -1;
    return fd;
  }
  StdoutToStderr()
  {
    std::lock_guard<std::mutex> lk(mtx());
    if (users()++ == 0)
    {
      fflush(stdout);
      saved_fd() = dup(1);
      if (saved_fd() >= 0)
        dup2(2, 1);
    }
  }
  ~StdoutToStderr()
  {
    std::lock_guard<std::mutex> lk(mtx());
    if (--users() == 0)
    {
      fflush(stdout);
      if (saved_fd() >= 0)
      {
        dup2(saved_fd(), 1);
        close(saved_fd());
        saved_fd() = -1;
      }
    }
  }
};

// A second, host-mediated backend: the ranks are contexts driven by threads of ONE process (they may
// even share a GPU).  Every collective synchronises the rank's stream, meets the others at a barrier
// and exchanges through host mailboxes.  Slow by design -- it exists so that the partitioned
// assemble + CG logic (ghost layout, halo plan, multi-rank scalar logic, lock-step convergence
// polling) can be run with the real kernels on a single-GPU box; production runs use RCCL.
// A barrier that can be broken: a rank that fails inside a collective (or the driver, for a rank that failed
// elsewhere) aborts the group and every waiting or later-arriving rank gets an error instead of hanging; waits
// are also bounded (5 minutes) so that a rank that silently never arrives ends in an error, not a hang.
struct AbortableBarrier
{
  std::mutex m;
  std::condition_variable cv;
  int n = 0, count = 0;
  unsigned long generation = 0;
  bool aborted = false;
  bool wait() // true = all ranks arrived
  {
    std::unique_lock<std::mutex> lk(m);
    if (aborted)
      return false;
    const unsigned long gen = generation;
    if (++count == n)
    {
      count = 0;
      ++generation;
      cv.notify_all();
      return true;
    }
    const bool ok = cv.wait_for(lk, std::chrono::seconds(300), [&] { return generation != gen || aborted; });
    if (!ok)
      aborted = true;
    if (aborted)
    {
      cv.notify_all();
      return false;
    }
    return true;
  }
  void abort()
  {
    std::lock_guard<std::mutex> lk(m);
    aborted = true;
    cv.notify_all();
  }
};

struct LocalGroup
{
  int n = 0;
  AbortableBarrier bar;
  std::vector<std::vector<double>> red;  // per rank: values to reduce
  std::vector<std::vector<double>> mail; // per rank: packed send buffer (all neighbours)
  std::vector<std::vector<int32_t>> neigh;
  std::vector<std::vector<int64_t>> send_off;
  std::vector<int> bs;
  std::vector<std::vector<std::vector<char>>> xch; // set-up exchanges: [source][destination] byte buffers
};

// Peer-memory all-reduce of the CG scalars (MPI_Allreduce of la::inner_product / la::squared_norm,
// src/cg.h:53,65,74): every rank owns a small UNCACHED device mailbox, box[parity][source rank] =
// {v0, v1, v2, tag}; a one-workgroup kernel reduces the producer's partials, stores the three sums and then
// (system-scope release) the tag into its slot of every peer's mailbox over xGMI, polls its own mailbox
// until all `nranks` tags of this round have arrived (system-scope acquire) and adds the contributions
// in rank order, so every rank forms the bit-identical sum.  One kernel replaces reduce + ncclAllReduce.
// Two parities suffice: a rank can only be one round ahead of the slowest peer.  The poll is bounded
// (P2P_TIMEOUT_TICKS of the 100 MHz wall clock): on time-out the kernel raises a flag and the solve
// returns an error instead of hanging.
struct P2P
{
  int nranks = 0, rank = 0;
  double* box = nullptr;                // own mailbox (uncached device memory)
  std::vector<double*> peer;            // peer[r] = rank r's mailbox as mapped here
  std::vector<bool> ipc_opened;         // peer[r] came from hipIpcOpenMemHandle
  zzz::DevBuf<double*> peer_dev;        // the same pointers for the kernel
  zzz::DevBuf<int32_t> fail;            // device flag: a poll timed out
  double* tail_mem = nullptr;           // partial arrays + tickets of the folded all-reduce (zzz_tail.h)
  bool tail_on = false;                 // ZZZ_TAIL=1 when the mailbox was created (A/B variant, off by default)
  int64_t seq = 0;                      // round counter = tag; identical call sequence on every rank
  // halo window (behind the mailbox in the same allocation, so the same IPC handle maps it): the forward scatter of
  // the product's input vector as plain device stores into the NEIGHBOUR's memory over xGMI instead of ncclSend /
  // ncclRecv.  Layout: 4 KiB of arrival tags [parity][sender rank] (int64 = exchange number), then
  // [parity][sender rank] data regions of halo_stride doubles each.
  size_t halo_off = 0;                  // doubles from the start of the box; 0 = no window
  size_t halo_stride = 0;               // doubles per (parity, sender) region
  bool halo_on = false;                 // in use (zzz_comm_p2p_halo; needs `enabled`)
  int halo_agreed = -1;                 // the ranks' COMMON verdict on "this halo plan travels through the windows": -1 not
                                        // taken yet (the communicator carries the halo), 0 no, 1 yes.  Taken by the collective
                                        // entry points only (attach, zzz_halo_upload, zzz_comm_p2p_halo, enable / disable), so
                                        // every rank is in the same state at every exchange
  bool inproc = false;                  // some peer is a context of THIS process on THIS GPU (validation on one GPU)
  int64_t halo_seq = 0;                 // exchange counter: the same call sequence on every rank
  zzz::DevBuf<int32_t> halo_ticket;     // arrival counter of the push kernel's workgroups
  bool enabled = false;
  bool verified = false;                // attach passed on every rank: may be switched on and off
};
constexpr long long P2P_TIMEOUT_TICKS = 300000000LL; // 3 s
constexpr int P2P_SLOT = 4;                          // doubles per mailbox slot
constexpr size_t P2P_HALO_HDR = 512;                 // doubles: the halo window's tag header (4 KiB)
constexpr int P2P_HALO_MAX_NEIGH = 32;

struct P2PHandle // ZZZ_P2P_HANDLE_BYTES
{
  hipIpcMemHandle_t ipc; // 64 bytes
  int64_t pid;
  double* raw;           // valid in the exporting process only
  int32_t device, ok;
  uint64_t halo_off, halo_stride; // geometry of the halo window behind the mailbox (0: none): must be the same on every rank
  char pad[128 - 64 - 8 - 8 - 8 - 16];
};
static_assert(sizeof(P2PHandle) == ZZZ_P2P_HANDLE_BYTES, "p2p handle size");

struct Comm
{
  P2P* p2p = nullptr;             // optional: all-reduces through peer memory instead of RCCL
  ncclComm_t comm = nullptr;      // all-reduces, on the context's main stream
  ncclComm_t comm_halo = nullptr; // send/recv of the halo, on the comm stream: one communicator per
                                  // stream, so no communicator is ever alternated between two streams
  LocalGroup* local = nullptr;
  int nranks = 1, rank = 0;
};

#define ZZZ_NCCL(ctx, call)                                                                                    \
  do                                                                                                           \
  {                                                                                                            \
    ncclResult_t r_ = (call);                                                                                  \
    if (r_ != 0)                                                                                               \
      return fail(ctx, ZZZ_ERR_RCCL, "%s failed: %s", #call, g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "?"); \
  } while (0)

__global__ void k_pack(const double* __restrict__ v, const int32_t* __restrict__ idx, double* __restrict__ out,
                       int64_t n, int bs)
{
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n * bs; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = v[(int64_t)idx[i / bs] * bs + i % bs]; // pack_fn, src/cgpoisson_problem.cpp:32-37
}

#define ZZZ_LOCAL_WAIT(ctx, G)                                                                                      \
  do                                                                                                                \
  {                                                                                                                 \
    if (!(G)->bar.wait())                                                                                           \
      return fail(ctx, ZZZ_ERR_RCCL, "local communicator: a rank failed or did not arrive (group aborted)");       \
  } while (0)
// a HIP error between two barriers must not leave the peers waiting: break the barrier, then report
#define ZZZ_LOCAL_HIP(ctx, G, call)                                                                                 \
  do                                                                                                                \
  {                                                                                                                 \
    hipError_t e_ = (call);                                                                                         \
    if (e_ != hipSuccess)                                                                                           \
    {                                                                                                               \
      (G)->bar.abort();                                                                                             \
      return zzz::fail(ctx, ZZZ_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    }                                                                                                               \
  } while (0)

static int local_allreduce(zzz_ctx* ctx, double* dev, int n)
{
  LocalGroup* G = ctx->comm->local;
  const int me = ctx->comm->rank;
  G->red[me].resize((size_t)n);
  ZZZ_LOCAL_HIP(ctx, G, hipMemcpyAsync(G->red[me].data(), dev, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream));
  ZZZ_LOCAL_HIP(ctx, G, hipStreamSynchronize(ctx->stream));
  ZZZ_LOCAL_WAIT(ctx, G);
  std::vector<double> sum((size_t)n, 0.0);
  for (int r = 0; r < G->n; ++r) // rank order: every rank forms the same sum
    for (int i = 0; i < n; ++i)
      sum[i] += G->red[r][i];
  ZZZ_LOCAL_WAIT(ctx, G);
  ZZZ_LOCAL_HIP(ctx, G, hipMemcpyAsync(dev, sum.data(), sizeof(double) * n, hipMemcpyHostToDevice, ctx->stream));
  ZZZ_LOCAL_HIP(ctx, G, hipStreamSynchronize(ctx->stream));
  return ZZZ_OK;
}

// val[j] = sum of partial array j (j < nv <= 3) in ONE pass and one barrier pair: every thread strides over the
// arrays, the wavefronts combine with shuffles, wavefront 0 adds the per-wavefront sums in order (fixed tree).
// On return val[] is valid in every thread.
__device__ inline void reduce3(const double* __restrict__ pa, const double* __restrict__ pb, const double* __restrict__ pc,
                               int np, int nv, double* val /* shared, 3 */)
{
  __shared__ double part[3][16];
  double s0 = 0, s1 = 0, s2 = 0;
  for (int i = threadIdx.x; i < np; i += blockDim.x)
  {
    s0 += pa[i];
    if (nv > 1)
      s1 += pb[i];
    if (nv > 2)
      s2 += pc[i];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1)
  {
    s0 += __shfl_down(s0, o, 64);
    s1 += __shfl_down(s1, o, 64);
    s2 += __shfl_down(s2, o, 64);
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
  if (lane == 0)
  {
    part[0][wv] = s0;
    part[1][wv] = s1;
    part[2][wv] = s2;
  }
  __syncthreads();
  if ((int)threadIdx.x < nv)
  {
    double t = 0;
    for (int w = 0; w < nw; ++w)
      t += part[threadIdx.x][w];
    val[threadIdx.x] = t;
  }
  __syncthreads();
}

// one workgroup: out[j] = sum over ranks of (sum of partial array j), j < nv <= 3
__global__ __launch_bounds__(1024) void k_allreduce_p2p(const int* __restrict__ stop, const double* __restrict__ pa,
                                                        const double* __restrict__ pb, const double* __restrict__ pc,
                                                        int np, int nv, double* __restrict__ out, double* const* peers,
                                                        double* box, int nranks, int rank, long long seq,
                                                        int* __restrict__ fail, long long timeout)
{
  // the stop word is requested together with the partials (one memory round trip instead of two on the critical path
  // of every iteration); a stopped solve then sums stale partials for nothing and leaves
  const int stopv = stop ? *stop : 0; // CG already converged (the same on every rank: identical scalars everywhere)
  __shared__ double val[3];
  __shared__ int timed_out;
  if (threadIdx.x == 0)
    timed_out = 0;
  reduce3(pa, pb, pc, np, nv, val);
  if (stopv)
    return;
  const int par = (int)(seq & 1);
  if ((int)threadIdx.x < nranks)
  {
    // my contribution into slot [par][rank] of peer threadIdx.x (my own mailbox included)
    double* slot = peers[threadIdx.x] + ((size_t)par * nranks + rank) * P2P_SLOT;
    for (int j = 0; j < nv; ++j)
      __hip_atomic_store(slot + j, val[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(reinterpret_cast<long long*>(slot + 3), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    // contribution of rank threadIdx.x in my mailbox
    const double* mine = box + ((size_t)par * nranks + threadIdx.x) * P2P_SLOT;
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(reinterpret_cast<const long long*>(mine + 3), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != seq)
    {
      if (wall_clock64() - t0 > timeout)
      {
        timed_out = 1;
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0)
  {
    if (timed_out)
    {
      *fail = 1;
      for (int j = 0; j < nv; ++j)
        out[j] = __builtin_nan("");
    }
    else
      for (int j = 0; j < nv; ++j)
      {
        double s = 0;
        for (int r = 0; r < nranks; ++r) // rank order: the same sum on every rank
          s += __hip_atomic_load(box + ((size_t)par * nranks + r) * P2P_SLOT + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        out[j] = s;
      }
  }
}

__global__ __launch_bounds__(1024) void k_reduce_partials(const int* __restrict__ stop, const double* __restrict__ pa,
                                                          const double* __restrict__ pb, const double* __restrict__ pc,
                                                          int np, int nv, double* __restrict__ out)
{
  if (stop && *stop)
    return;
  __shared__ double val[3];
  reduce3(pa, pb, pc, np, nv, val);
  if ((int)threadIdx.x < nv)
    out[threadIdx.x] = val[threadIdx.x];
}

bool comm_p2p_enabled(const zzz_ctx* ctx) { return ctx->comm && ctx->comm->p2p && ctx->comm->p2p->enabled; }

// out[0..nv) = all-reduced sums of up to three partial arrays of length np (pb/pc may be null for nv < 2/3).
// stop: device flag that turns the call into a no-op (CgState::converged) or null.
int comm_reduce_allreduce(zzz_ctx* ctx, const int* stop, const double* pa, const double* pb, const double* pc, int np,
                          int nv, double* out)
{
  hipStream_t s = ctx->stream;
  if (comm_p2p_enabled(ctx))
  {
    P2P* P = ctx->comm->p2p;
    const long long seq = ++P->seq;
    hipLaunchKernelGGL(k_allreduce_p2p, dim3(1), dim3(512), 0, s, stop, pa, pb, pc, np, nv, out, P->peer_dev.p, P->box,
                       P->nranks, P->rank, seq, P->fail.p, P2P_TIMEOUT_TICKS);
    ZZZ_HIP(ctx, hipGetLastError());
    return ZZZ_OK;
  }
  hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(512), 0, s, stop, pa, pb, pc, np, nv, out);
  ZZZ_HIP(ctx, hipGetLastError());
  return comm_allreduce_sum(ctx, out, nv);
}

bool comm_tail_args(zzz_ctx* ctx, TailArgs& T, int nv, double* out)
{
#ifndef ZZZ_EXPERIMENTS
  (void)ctx, (void)T, (void)nv, (void)out;
  return false; // the folded all-reduce (zzz_tail.h) exists in the tools build only: measured 1 us slower per reduction point
#else
  // A/B knob ZZZ_TAIL=1 (read when the mailbox is created): fold the all-reduce into the producer's tail.  Measured
  // slower than the kernel of its own (zzz_tail.h has the numbers), so off unless asked for.
  const bool off = !(ctx->comm && ctx->comm->p2p && ctx->comm->p2p->tail_on);
  if (off || !comm_p2p_enabled(ctx) || !ctx->comm->p2p->tail_mem)
    return false;
  P2P* P = ctx->comm->p2p;
  T = TailArgs();
  T.parts = P->tail_mem;
  T.ticket = reinterpret_cast<int*>(P->tail_mem + (size_t)TAIL_PART_DOUBLES);
  T.nv = nv;
  T.out = out;
  T.peers = P->peer_dev.p;
  T.box = P->box;
  T.nranks = P->nranks;
  T.rank = P->rank;
  T.seq = ++P->seq;
  T.fail = P->fail.p;
  T.timeout = P2P_TIMEOUT_TICKS;
  return true;
#endif
}

// did a peer all-reduce time out since the last call?  (checked at the end of a solve)
int comm_p2p_check(zzz_ctx* ctx)
{
  if (!comm_p2p_enabled(ctx))
    return ZZZ_OK;
  int32_t f = 0;
  ZZZ_HIP(ctx, hipMemcpyAsync(&f, ctx->comm->p2p->fail.p, sizeof(f), hipMemcpyDeviceToHost, ctx->stream));
  ZZZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (f)
  {
    ZZZ_HIP(ctx, hipMemsetAsync(ctx->comm->p2p->fail.p, 0, sizeof(int32_t), ctx->stream));
    return fail(ctx, ZZZ_ERR_RCCL, "peer-memory all-reduce timed out (rank %d of %d; all-reduce mailbox or halo window)",
                ctx->comm->p2p->rank, ctx->comm->p2p->nranks);
  }
  return ZZZ_OK;
}

static void p2p_destroy(P2P* P)
{
  if (!P)
    return;
  for (size_t r = 0; r < P->peer.size(); ++r)
    if (P->ipc_opened[r] && P->peer[r])
      (void)hipIpcCloseMemHandle(P->peer[r]);
  if (P->box)
    (void)hipFree(P->box);
  if (P->tail_mem)
    (void)hipFree(P->tail_mem);
  delete P;
}

int comm_allreduce_sum(zzz_ctx* ctx, double* dev, int n)
{
  if (!ctx->comm)
    return ZZZ_OK;
  if (ctx->comm->local)
    return local_allreduce(ctx, dev, n);
  if (!ctx->comm->comm)
    return fail(ctx, ZZZ_ERR_RCCL, "all-reduce: this communicator has no transport (peer-only) and the peer-memory path is off");
  ZZZ_NCCL(ctx, g_rccl.AllReduce(dev, dev, (size_t)n, ncclFloat64, ncclSum, ctx->comm->comm, ctx->stream));
  return ZZZ_OK;
}

// ---- forward halo through peer memory ----------------------------------------------------------------------------
struct HaloPlanDev
{
  int nneigh;
  int rank[P2P_HALO_MAX_NEIGH];
  long long soff[P2P_HALO_MAX_NEIGH + 1]; // scalar entries sent to neighbour k: [soff[k], soff[k+1])
  long long roff[P2P_HALO_MAX_NEIGH + 1]; // ... received from neighbour k, relative to the ghost range
};

// Every owned entry a neighbour needs goes straight into that neighbour's window; the workgroup that finishes last
// publishes the exchange number in each neighbour's tag (release at system scope behind everybody's stores).
__global__ __launch_bounds__(256) void k_halo_push(const int* __restrict__ stop, const double* __restrict__ vec,
                                                   const int32_t* __restrict__ send_idx, int bs, HaloPlanDev H,
                                                   double* const* peers, size_t halo_off, size_t stride, int nranks, int me,
                                                   long long seq, int* __restrict__ ticket)
{
  if (stop && *stop) // converged (the same verdict on every rank): the consumer skips its wait as well
    return;
  const size_t region = ((size_t)(seq & 1) * nranks + me) * stride;
  const long long total = H.soff[H.nneigh];
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x)
  {
    int k = 0;
    while (k + 1 < H.nneigh && i >= H.soff[k + 1])
      ++k;
    double* dst = peers[H.rank[k]] + halo_off + P2P_HALO_HDR + region + (i - H.soff[k]);
    *dst = vec[(long long)send_idx[i / bs] * bs + i % bs];
  }
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0)
  {
    const int t = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (t == (int)gridDim.x - 1)
    {
      __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __threadfence_system();
      for (int k = 0; k < H.nneigh; ++k)
        __hip_atomic_store(reinterpret_cast<long long*>(peers[H.rank[k]] + halo_off) + (size_t)(seq & 1) * nranks + me, seq,
                           __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// Waits (bounded) for the tags of all neighbours, then copies their regions of the own window behind the owned entries.
__global__ __launch_bounds__(256) void k_halo_pull(const int* __restrict__ stop, double* __restrict__ ghost, HaloPlanDev H,
                                                   const double* window, size_t stride, int nranks, long long seq,
                                                   int* __restrict__ fail, long long timeout)
{
  if (stop && *stop)
    return;
  __shared__ int timed_out;
  if (threadIdx.x == 0)
    timed_out = 0;
  __syncthreads();
  if ((int)threadIdx.x < H.nneigh)
  {
    const long long* tag = reinterpret_cast<const long long*>(window) + (size_t)(seq & 1) * nranks + H.rank[threadIdx.x];
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(tag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != seq)
    {
      if (wall_clock64() - t0 > timeout)
      {
        timed_out = 1;
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
  }
  __syncthreads();
  if (timed_out)
  {
    if (threadIdx.x == 0)
      *fail = 1;
    return;
  }
  const long long total = H.roff[H.nneigh];
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x)
  {
    int k = 0;
    while (k + 1 < H.nneigh && i >= H.roff[k + 1])
      ++k;
    const double* src = window + P2P_HALO_HDR + ((size_t)(seq & 1) * nranks + H.rank[k]) * stride + (i - H.roff[k]);
    ghost[i] = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// can THIS rank's halo plan travel through the windows?  (A rank without neighbours has nothing to send: yes.)
static bool halo_peer_local_ok(const zzz_ctx* ctx)
{
  if (!ctx->comm || !ctx->comm->p2p)
    return false;
  const P2P* P = ctx->comm->p2p;
  if (!P->enabled || !P->halo_on || !P->halo_off || ctx->nneigh > P2P_HALO_MAX_NEIGH)
    return false;
  for (int k = 0; k < ctx->nneigh; ++k)
    if ((size_t)((ctx->send_off[(size_t)k + 1] - ctx->send_off[(size_t)k]) * ctx->bs) > P->halo_stride
        || ctx->neigh_rank[(size_t)k] < 0 || ctx->neigh_rank[(size_t)k] >= P->nranks || ctx->neigh_rank[(size_t)k] == P->rank)
      return false;
  return true;
}
// The transport of an exchange must be the SAME on both ends (a rank that pushes into windows while its neighbour sits in
// ncclRecv: a time-out on one side, an unbounded block on the other), and what decides it is rank-local (message sizes
// against the window's regions, neighbour count, the ZZZ_P2P_HALO* knobs).  So the ranks take one common verdict -- one
// round of the mailbox all-reduce over "my plan does not fit" -- whenever an input of it changes, and every exchange
// goes by that verdict.  Collective; a no-op verdict of "no" while the mailboxes are off.
static int halo_peer_agree(zzz_ctx* ctx);
static bool halo_peer_usable(const zzz_ctx* ctx)
{
  return ctx->comm && ctx->comm->p2p && ctx->comm->p2p->enabled && ctx->comm->p2p->halo_agreed == 1 && ctx->nneigh >= 1;
}

static int halo_peer(zzz_ctx* ctx, double* vec, hipStream_t st, const int* stop)
{
  P2P* P = ctx->comm->p2p;
  HaloPlanDev H;
  H.nneigh = ctx->nneigh;
  H.soff[0] = H.roff[0] = 0;
  for (int k = 0; k < ctx->nneigh; ++k)
  {
    H.rank[k] = ctx->neigh_rank[(size_t)k];
    H.soff[k + 1] = ctx->send_off[(size_t)k + 1] * ctx->bs;
    H.roff[k + 1] = H.roff[k] + ctx->recv_cnt[(size_t)k] * ctx->bs;
  }
  const long long seq = ++P->halo_seq;
  const long long ns = H.soff[H.nneigh], nr = H.roff[H.nneigh];
  int gp = (int)std::min<long long>((ns + 255) / 256, 64);
  if (gp < 1)
    gp = 1;
  hipLaunchKernelGGL(k_halo_push, dim3((unsigned)gp), dim3(256), 0, st, stop, vec, ctx->send_idx.p, ctx->bs, H, P->peer_dev.p,
                     P->halo_off, P->halo_stride, P->nranks, P->rank, seq, P->halo_ticket.p);
  // Ranks that share a GPU inside one process (validation) also share its legacy default stream: a rank still in
  // set-up (synchronous copies) would wait for this rank's waiting kernel, and that kernel for the other's push.  They
  // therefore meet on the host first: every push is enqueued before any wait is.
  if (P->inproc && ctx->comm->local)
    ZZZ_LOCAL_WAIT(ctx, ctx->comm->local);
  int gl = (int)std::min<long long>((nr + 255) / 256, 32);
  if (gl < 1)
    gl = 1;
  hipLaunchKernelGGL(k_halo_pull, dim3((unsigned)gl), dim3(256), 0, st, stop, vec + ctx->n_owned * ctx->bs, H,
                     P->box + P->halo_off, P->halo_stride, P->nranks, seq, P->fail.p, P2P_TIMEOUT_TICKS);
  ZZZ_HIP(ctx, hipGetLastError());
  return ZZZ_OK;
}

static int halo_on_stream(zzz_ctx* ctx, double* vec, hipStream_t st);

int comm_halo_forward(zzz_ctx* ctx, double* vec) { return halo_on_stream(ctx, vec, ctx->stream); }

// overlap form: the exchange runs on the comm stream behind everything enqueued on the main stream
// so far (vec is final), the main stream goes on with work that needs no ghost value
int comm_halo_begin(zzz_ctx* ctx, double* vec)
{
  if (!ctx->comm || (ctx->nneigh == 0 && !ctx->comm->local))
    return ZZZ_OK;
  ctx->halo_pending = false;
  const bool peer = halo_peer_usable(ctx);
  // Ranks that are contexts of one process on one GPU share its few hardware queues: a waiting kernel on a second
  // stream per rank could sit in front of the very kernel it waits for.  There (validation only) the exchange stays on
  // the main stream, like the mailbox all-reduce.
  if (peer && ctx->comm->p2p->inproc)
    return halo_on_stream(ctx, vec, ctx->stream);
  if (!peer && (ctx->comm->local || !ctx->comm->comm_halo)) // host-synchronous backend / no second communicator:
    return halo_on_stream(ctx, vec, ctx->stream);           // nothing to overlap, same results
  if (!ctx->comm_stream)
  {
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi); // hi = numerically smallest = most urgent
    ZZZ_HIP(ctx, hipStreamCreateWithPriority(&ctx->comm_stream, hipStreamNonBlocking, prio_hi));
    ZZZ_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_x_ready, hipEventDisableTiming));
    ZZZ_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_halo_done, hipEventDisableTiming));
  }
  ZZZ_HIP(ctx, hipEventRecord(ctx->ev_x_ready, ctx->stream));
  ZZZ_HIP(ctx, hipStreamWaitEvent(ctx->comm_stream, ctx->ev_x_ready, 0));
  int rc = halo_on_stream(ctx, vec, ctx->comm_stream);
  if (rc)
    return rc;
  ZZZ_HIP(ctx, hipEventRecord(ctx->ev_halo_done, ctx->comm_stream));
  ctx->halo_pending = true;
  return ZZZ_OK;
}

int comm_halo_end(zzz_ctx* ctx)
{
  if (!ctx->comm || !ctx->halo_pending || !ctx->comm_stream)
    return ZZZ_OK;
  ctx->halo_pending = false;
  // timed iteration: how long the main stream sits between the interior rows and the halo's arrival
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (ctx->prof_now && ctx->prof_halo_n < 256)
  {
    while ((int)ctx->ev_halo.size() < 2 * (ctx->prof_halo_n + 1))
    {
      hipEvent_t e;
      ZZZ_HIP(ctx, hipEventCreate(&e));
      ctx->ev_halo.push_back(e);
    }
    e0 = ctx->ev_halo[(size_t)(2 * ctx->prof_halo_n)];
    e1 = ctx->ev_halo[(size_t)(2 * ctx->prof_halo_n + 1)];
    ++ctx->prof_halo_n;
    ZZZ_HIP(ctx, hipEventRecord(e0, ctx->stream));
  }
  ZZZ_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_halo_done, 0));
  if (e1)
    ZZZ_HIP(ctx, hipEventRecord(e1, ctx->stream));
  return ZZZ_OK;
}

static int halo_on_stream(zzz_ctx* ctx, double* vec, hipStream_t st)
{
  if (!ctx->comm || (ctx->nneigh == 0 && !ctx->comm->local))
    return ZZZ_OK;
  if (halo_peer_usable(ctx))
    return halo_peer(ctx, vec, st, nullptr);
  const int bs = ctx->bs;
  const int64_t nsend = ctx->send_off[ctx->nneigh];
  // pack only what is not already contiguous in the vector (z-slab partitions send contiguous ranges)
  bool need_pack = ctx->comm->local != nullptr;
  for (int k = 0; k < ctx->nneigh; ++k)
    if (ctx->send_contig[k] < 0)
      need_pack = true;
  if (nsend > 0 && need_pack)
  {
    int64_t g = (nsend * bs + 255) / 256;
    if (g > 1024)
      g = 1024;
    hipLaunchKernelGGL(k_pack, dim3((unsigned)g), dim3(256), 0, st, vec, ctx->send_idx.p, ctx->send_buf.p, nsend, bs);
  }
  if (ctx->comm->local)
  {
    LocalGroup* G = ctx->comm->local;
    const int me = ctx->comm->rank;
    G->mail[me].resize((size_t)(nsend * bs));
    if (nsend > 0)
      ZZZ_LOCAL_HIP(ctx, G, hipMemcpyAsync(G->mail[me].data(), ctx->send_buf.p, sizeof(double) * nsend * bs,
                                           hipMemcpyDeviceToHost, ctx->stream));
    ZZZ_LOCAL_HIP(ctx, G, hipStreamSynchronize(ctx->stream));
    ZZZ_LOCAL_WAIT(ctx, G);
    int64_t gh = ctx->n_owned;
    for (int k = 0; k < ctx->nneigh; ++k)
    {
      const int src = ctx->neigh_rank[k];
      const int64_t nr = ctx->recv_cnt[k];
      // where, in the sender's packed buffer, is the block meant for me?
      int kk = -1;
      for (size_t q = 0; q < G->neigh[src].size(); ++q)
        if (G->neigh[src][q] == me)
          kk = (int)q;
      if (kk < 0 || G->send_off[src][kk + 1] - G->send_off[src][kk] != nr)
      {
        G->bar.abort();
        return fail(ctx, ZZZ_ERR_ARG, "local halo: rank %d does not send %lld block dofs to rank %d", src, (long long)nr, me);
      }
      if (nr > 0)
        ZZZ_LOCAL_HIP(ctx, G, hipMemcpyAsync(vec + gh * bs, G->mail[src].data() + G->send_off[src][kk] * bs,
                                             sizeof(double) * nr * bs, hipMemcpyHostToDevice, ctx->stream));
      gh += nr;
    }
    ZZZ_LOCAL_HIP(ctx, G, hipStreamSynchronize(ctx->stream));
    ZZZ_LOCAL_WAIT(ctx, G);
    return ZZZ_OK;
  }
  ncclComm_t hc = (st != ctx->stream && ctx->comm->comm_halo) ? ctx->comm->comm_halo : ctx->comm->comm;
  if (!hc)
    return fail(ctx, ZZZ_ERR_RCCL, "halo exchange: this communicator has no transport (peer-only)");
  ZZZ_NCCL(ctx, g_rccl.GroupStart());
  int64_t ghost = ctx->n_owned;
  for (int k = 0; k < ctx->nneigh; ++k)
  {
    const int64_t ns = ctx->send_off[k + 1] - ctx->send_off[k], nr = ctx->recv_cnt[k];
    const double* src = ctx->send_contig[k] >= 0 ? vec + ctx->send_contig[k] * bs : ctx->send_buf.p + ctx->send_off[k] * bs;
    if (ns > 0)
      ZZZ_NCCL(ctx, g_rccl.Send(src, (size_t)(ns * bs), ncclFloat64, ctx->neigh_rank[k], hc, st));
    if (nr > 0)
      ZZZ_NCCL(ctx, g_rccl.Recv(vec + ghost * bs, (size_t)(nr * bs), ncclFloat64, ctx->neigh_rank[k], hc, st));
    ghost += nr;
  }
  ZZZ_NCCL(ctx, g_rccl.GroupEnd());
  return ZZZ_OK;
}

int comm_size(const zzz_ctx* ctx) { return ctx->comm ? ctx->comm->nranks : 1; }
int comm_rank(const zzz_ctx* ctx) { return ctx->comm ? ctx->comm->rank : 0; }

// Set-up exchange of byte buffers of arbitrary sizes between all ranks (send[r] goes to rank r, recv[r] came from
// rank r; collective).  Host mailboxes with the local backend; with RCCL the sizes travel first, then the
// payloads, as grouped ncclSend / ncclRecv through device staging buffers.  Not for the solve loop.
int comm_exchange_bytes(zzz_ctx* ctx, const std::vector<std::vector<char>>& send, std::vector<std::vector<char>>& recv)
{
  if (!ctx->comm)
    return fail(ctx, ZZZ_ERR_ARG, "exchange: no communicator");
  const int n = ctx->comm->nranks, me = ctx->comm->rank;
  if ((int)send.size() != n)
    return fail(ctx, ZZZ_ERR_ARG, "exchange: one send buffer per rank expected");
  recv.assign((size_t)n, std::vector<char>());
  if (ctx->comm->local)
  {
    LocalGroup* G = ctx->comm->local;
    G->xch[(size_t)me] = send;
    ZZZ_LOCAL_WAIT(ctx, G);
    for (int r = 0; r < n; ++r)
      recv[(size_t)r] = G->xch[(size_t)r][(size_t)me];
    ZZZ_LOCAL_WAIT(ctx, G);
    return ZZZ_OK;
  }
  if (!ctx->comm->comm)
    return fail(ctx, ZZZ_ERR_RCCL, "exchange: this communicator has no transport (peer-only)");
  hipStream_t st = ctx->stream;
  std::vector<int64_t> ssz((size_t)n), rsz((size_t)n, 0);
  for (int r = 0; r < n; ++r)
    ssz[(size_t)r] = (int64_t)send[(size_t)r].size();
  DevBuf<int64_t> dsz;
  ZZZ_HIP(ctx, dsz.alloc(2 * (size_t)n));
  ZZZ_HIP(ctx, hipMemcpyAsync(dsz.p, ssz.data(), sizeof(int64_t) * n, hipMemcpyHostToDevice, st));
  ZZZ_NCCL(ctx, g_rccl.GroupStart());
  for (int r = 0; r < n; ++r)
    if (r != me)
    {
      ZZZ_NCCL(ctx, g_rccl.Send(dsz.p + r, 8, 0 /* ncclInt8 */, r, ctx->comm->comm, st));
      ZZZ_NCCL(ctx, g_rccl.Recv(dsz.p + n + r, 8, 0, r, ctx->comm->comm, st));
    }
  ZZZ_NCCL(ctx, g_rccl.GroupEnd());
  ZZZ_HIP(ctx, hipMemcpyAsync(rsz.data(), dsz.p + n, sizeof(int64_t) * n, hipMemcpyDeviceToHost, st));
  ZZZ_HIP(ctx, hipStreamSynchronize(st));
  rsz[(size_t)me] = ssz[(size_t)me];
  std::vector<int64_t> soff((size_t)n + 1, 0), roff((size_t)n + 1, 0);
  for (int r = 0; r < n; ++r)
  {
    soff[(size_t)r + 1] = soff[(size_t)r] + (r == me ? 0 : ssz[(size_t)r]);
    roff[(size_t)r + 1] = roff[(size_t)r] + (r == me ? 0 : rsz[(size_t)r]);
  }
  DevBuf<char> dsend, drecv;
  ZZZ_HIP(ctx, dsend.alloc((size_t)soff[(size_t)n] + 8));
  ZZZ_HIP(ctx, drecv.alloc((size_t)roff[(size_t)n] + 8));
  for (int r = 0; r < n; ++r)
    if (r != me && ssz[(size_t)r])
      ZZZ_HIP(ctx, hipMemcpyAsync(dsend.p + soff[(size_t)r], send[(size_t)r].data(), (size_t)ssz[(size_t)r], hipMemcpyHostToDevice, st));
  ZZZ_NCCL(ctx, g_rccl.GroupStart());
  for (int r = 0; r < n; ++r)
    if (r != me)
    {
      if (ssz[(size_t)r])
        ZZZ_NCCL(ctx, g_rccl.Send(dsend.p + soff[(size_t)r], (size_t)ssz[(size_t)r], 0, r, ctx->comm->comm, st));
      if (rsz[(size_t)r])
        ZZZ_NCCL(ctx, g_rccl.Recv(drecv.p + roff[(size_t)r], (size_t)rsz[(size_t)r], 0, r, ctx->comm->comm, st));
    }
  ZZZ_NCCL(ctx, g_rccl.GroupEnd());
  for (int r = 0; r < n; ++r)
  {
    recv[(size_t)r].resize((size_t)rsz[(size_t)r]);
    if (r != me && rsz[(size_t)r])
      ZZZ_HIP(ctx, hipMemcpyAsync(recv[(size_t)r].data(), drecv.p + roff[(size_t)r], (size_t)rsz[(size_t)r], hipMemcpyDeviceToHost, st));
  }
  ZZZ_HIP(ctx, hipStreamSynchronize(st));
  recv[(size_t)me] = send[(size_t)me];
  return ZZZ_OK;
}

// One double of every rank to every rank, through the all-reduce the CG scalars take (peer-memory mailboxes or the
// communicator's all-reduce): rank r's value travels in slot r of a sum of otherwise-zero slots, three slots per round.
// all: nranks values (all[0] = v without a communicator).
int comm_allgather_double(zzz_ctx* ctx, double v, double* all)
{
  if (!ctx->comm || ctx->comm->nranks == 1)
  {
    all[0] = v;
    return ZZZ_OK;
  }
  const int n = ctx->comm->nranks, me = ctx->comm->rank;
  hipStream_t s = ctx->stream;
  for (int q = 0; q < n; q += 3)
  {
    double slot[3] = {0.0, 0.0, 0.0}, got[3] = {0.0, 0.0, 0.0};
    if (me >= q && me < q + 3)
      slot[me - q] = v;
    double* in = ctx->part_a.p; // three one-entry "partial arrays"
    ZZZ_HIP(ctx, hipMemcpyAsync(in, slot, sizeof(slot), hipMemcpyHostToDevice, s));
    ZZZ_HIP(ctx, hipStreamSynchronize(s)); // slot[] leaves scope
    if (int rc = comm_reduce_allreduce(ctx, nullptr, in, in + 1, in + 2, 1, 3, ctx->red.p + 4))
      return rc;
    ZZZ_HIP(ctx, hipMemcpyAsync(got, ctx->red.p + 4, sizeof(got), hipMemcpyDeviceToHost, s));
    ZZZ_HIP(ctx, hipStreamSynchronize(s));
    for (int k = 0; k < 3 && q + k < n; ++k)
      all[q + k] = got[k];
  }
  return comm_p2p_check(ctx);
}

int comm_allgather_max(zzz_ctx* ctx, double* v)
{
  if (!ctx->comm || ctx->comm->nranks == 1)
    return ZZZ_OK;
  std::vector<double> all((size_t)ctx->comm->nranks);
  if (int rc = comm_allgather_double(ctx, *v, all.data()))
    return rc;
  for (double w : all)
    if (w > *v)
      *v = w;
  return ZZZ_OK;
}

int comm_rank_of(const zzz_ctx* ctx, int* nranks)
{
  if (nranks)
    *nranks = ctx->comm ? ctx->comm->nranks : 1;
  return ctx->comm ? ctx->comm->rank : 0;
}

void comm_destroy(zzz_ctx* ctx)
{
  if (ctx->comm_stream)
  {
    (void)hipStreamSynchronize(ctx->comm_stream);
    (void)hipEventDestroy(ctx->ev_x_ready);
    (void)hipEventDestroy(ctx->ev_halo_done);
    (void)hipStreamDestroy(ctx->comm_stream);
    ctx->comm_stream = nullptr;
  }
  if (ctx->comm)
  {
    p2p_destroy(ctx->comm->p2p);
    ctx->comm->p2p = nullptr;
    if (ctx->comm->comm_halo && !ctx->comm->local && g_rccl.CommDestroy)
      (void)g_rccl.CommDestroy(ctx->comm->comm_halo);
    if (ctx->comm->comm && !ctx->comm->local && g_rccl.CommDestroy)
      (void)g_rccl.CommDestroy(ctx->comm->comm);
    delete ctx->comm;
    ctx->comm = nullptr;
  }
}
ZZZ_PRELOAD_TU(comm)
} // namespace zzz

using namespace zzz;

extern "C" {

int zzz_comm_load(void)
{
  if (const char* e = load_rccl())
    return fail(nullptr, ZZZ_ERR_RCCL, "%s: %s", e, g_rccl_err.c_str());
  return ZZZ_OK;
}

int zzz_comm_info(zzz_ctx* ctx, int64_t info[12])
{
  if (!ctx || !info)
    return fail(ctx, ZZZ_ERR_ARG, "zzz_comm_info: bad arguments");
  for (int i = 0; i < 12; ++i)
    info[i] = 0;
  info[0] = ctx->comm ? ctx->comm->nranks : 1;
  info[1] = ctx->comm ? ctx->comm->rank : 0;
  info[2] = ctx->nneigh;
  int64_t nrecv = 0;
  for (int k = 0; k < ctx->nneigh; ++k)
    nrecv += ctx->recv_cnt[(size_t)k];
  info[3] = (ctx->nneigh ? ctx->send_off[(size_t)ctx->nneigh] : 0) * ctx->bs * 8; // bytes sent per forward scatter
  info[4] = nrecv * ctx->bs * 8;
  info[5] = (ctx->comm && ctx->comm->comm_halo) ? 1 : 0; // second communicator (ncclCommSplit): halo on its own stream
  if (halo_peer_usable(ctx))
    info[5] = 2; // the halo travels through peer memory (device stores into the neighbour's window), no communicator
  info[6] = comm_p2p_enabled(ctx) ? 1 : 0;                // CG scalars through the peer-memory mailboxes
  const bool stream_split = sellp_active(ctx) && ctx->have_group_split;
  info[7] = (ctx->comm && ctx->overlap && (stream_split || ctx->have_tile_split)) ? 1 : 0; // halo overlapped with interior rows
  info[8] = stream_split ? ctx->n_groups_interior : ctx->n_tiles_interior;
  info[9] = stream_split ? ctx->n_groups_boundary : ctx->n_tiles_boundary;
  info[10] = ctx->comm && ctx->comm->local ? 1 : 0;
  info[11] = (int64_t)(ctx->prof_halo_wait_ms * 1.0e6); // ns: mean exposed halo wait per product of the last profiled solve
  return ZZZ_OK;
}

/* path of the librccl this process bound (dladdr of ncclAllReduce); "" before zzz_comm_load */
const char* zzz_comm_library_path(void) { return g_rccl_path.c_str(); }

int zzz_comm_unique_id(void* id)
{
  if (!id)
    return fail(nullptr, ZZZ_ERR_ARG, "zzz_comm_unique_id: NULL buffer");
  if (const char* e = load_rccl())
    return fail(nullptr, ZZZ_ERR_RCCL, "%s: %s", e, g_rccl_err.c_str());
  ncclUniqueId u;
  StdoutToStderr quiet;
  ncclResult_t r = g_rccl.GetUniqueId(&u);
  if (r != 0)
    return fail(nullptr, ZZZ_ERR_RCCL, "ncclGetUniqueId failed: %s", g_rccl.GetErrorString(r));
  memcpy(id, &u, sizeof(u));
  return ZZZ_OK;
}

int zzz_comm_init(zzz_ctx* ctx, int nranks, int rank, const void* id)
{
  if (!ctx)
    return fail(nullptr, ZZZ_ERR_ARG, "NULL context");
  ZZZ_HIP(ctx, hipSetDevice(ctx->device));
  if (nranks < 1 || rank < 0 || rank >= nranks || !id)
    return fail(ctx, ZZZ_ERR_ARG, "zzz_comm_init: bad rank %d of %d", rank, nranks);
  if (const char* e = load_rccl())
    return fail(ctx, ZZZ_ERR_RCCL, "%s: %s", e, g_rccl_err.c_str());
  comm_destroy(ctx);
  ncclUniqueId u;
  memcpy(&u, id, sizeof(u));
  Comm* c = new Comm();
  c->nranks = nranks;
  c->rank = rank;
  ncclResult_t r;
  {
    StdoutToStderr quiet;
    r = g_rccl.CommInitRank(&c->comm, nranks, u, rank);
    // second communicator over the same ranks for the halo stream (collective: every rank calls it);
    // without it the halo stays on the main stream and is simply not overlapped
    if (r == 0 && g_rccl.CommSplit)
      if (g_rccl.CommSplit(c->comm, 0, rank, &c->comm_halo, nullptr) != 0)
        c->comm_halo = nullptr;
  }
  if (r != 0)
  {
    delete c;
    return fail(ctx, ZZZ_ERR_RCCL, "ncclCommInitRank failed: %s", g_rccl.GetErrorString(r));
  }
  ctx->comm = c;
  return ZZZ_OK;
}

int zzz::halo_peer_agree(zzz_ctx* ctx)
{
  if (!ctx->comm || !ctx->comm->p2p)
    return ZZZ_OK;
  P2P* P = ctx->comm->p2p;
  P->halo_agreed = -1;
  if (!P->enabled) // (the same on every rank: attach, enable and disable are collective)
    return ZZZ_OK;
  zzz::DevBuf<double> tv;
  ZZZ_HIP(ctx, tv.alloc(8));
  const double in[3] = {halo_peer_local_ok(ctx) ? 0.0 : 1.0, (double)ctx->nneigh, 0.0}; // (no neighbours anywhere: nothing to carry)
  ZZZ_HIP(ctx, hipMemcpyAsync(tv.p, in, sizeof(in), hipMemcpyHostToDevice, ctx->stream));
  if (ctx->comm->local)
  {
    // ranks that are threads of one process: through the host mailboxes.  (A kernel that waits for a peer must not run
    // while that peer is still in set-up: its hipFree / synchronous copies wait for the whole device, this kernel included.)
    if (int rc = comm_allreduce_sum(ctx, tv.p, 2))
      return rc;
    double nb[2] = {1.0, 0.0};
    ZZZ_HIP(ctx, hipMemcpyAsync(nb, tv.p, sizeof(nb), hipMemcpyDeviceToHost, ctx->stream));
    ZZZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    P->halo_agreed = (nb[0] == 0.0 && nb[1] > 0.0) ? 1 : 0;
    return ZZZ_OK;
  }
  if (ctx->comm->comm)
  {
    // A communicator with a transport of its own: the verdict travels through IT -- a blocking collective without a clock.
    // (The mailbox kernel below gives up after its time-out; ranks reach set-up calls minutes apart when one of them builds
    // a mesh on the host, and a rank that gave up would be out of step with the mailboxes' round numbers for good.)
    if (int rc = comm_allreduce_sum(ctx, tv.p, 2))
      return rc;
    double nb[2] = {1.0, 0.0};
    ZZZ_HIP(ctx, hipMemcpyAsync(nb, tv.p, sizeof(nb), hipMemcpyDeviceToHost, ctx->stream));
    ZZZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    P->halo_agreed = (nb[0] == 0.0 && nb[1] > 0.0) ? 1 : 0;
    return ZZZ_OK;
  }
  // peer-only communicator: the mailboxes are all there is; a set-up call may wait long for the slowest rank (ten minutes)
  const long long seq = ++P->seq;
  hipLaunchKernelGGL(k_allreduce_p2p, dim3(1), dim3(1024), 0, ctx->stream, (const int*)nullptr, tv.p, tv.p + 1, tv.p + 2, 1, 3,
                     tv.p + 4, P->peer_dev.p, P->box, P->nranks, P->rank, seq, P->fail.p, 200 * P2P_TIMEOUT_TICKS);
  double nbad[2] = {1.0, 0.0};
  ZZZ_HIP(ctx, hipMemcpyAsync(nbad, tv.p + 4, sizeof(nbad), hipMemcpyDeviceToHost, ctx->stream));
  ZZZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (int rc = comm_p2p_check(ctx))
    return rc;
  P->halo_agreed = (nbad[0] == 0.0 && nbad[1] > 0.0) ? 1 : 0;
  return ZZZ_OK;
}

int zzz_comm_init_peer_only(zzz_ctx* ctx, int nranks, int rank)
{
  if (!ctx)
    return fail(nullptr, ZZZ_ERR_ARG, "NULL context");
  if (nranks < 1 || rank < 0 || rank >= nranks)
    return fail(ctx, ZZZ_ERR_ARG, "zzz_comm_init_peer_only: bad rank %d of %d", rank, nranks);
  comm_destroy(ctx);
  Comm* c = new Comm();
  c->nranks = nranks;
  c->rank = rank;
  ctx->comm = c;
  return ZZZ_OK;
}

int zzz_comm_p2p_export(zzz_ctx* ctx, void* handle)
{
  if (!ctx)
    return fail(nullptr, ZZZ_ERR_ARG, "NULL context");
  ZZZ_HIP(ctx, hipSetDevice(ctx->device));
  if (!ctx->comm || !handle)
    return fail(ctx, ZZZ_ERR_ARG, "zzz_comm_p2p_export: attach a communicator first");
  P2PHandle h;
  memset(&h, 0, sizeof(h));
  p2p_destroy(ctx->comm->p2p);
  ctx->comm->p2p = nullptr;
  P2P* P = new P2P();
  P->nranks = ctx->comm->nranks;
  P->rank = ctx->comm->rank;
#ifdef ZZZ_EXPERIMENTS
  if (const char* e = getenv("ZZZ_TAIL"))
    P->tail_on = atoi(e) == 1;
#endif
  size_t bytes = sizeof(double) * 2 * (size_t)P->nranks * P2P_SLOT;
  bytes = (bytes + 4095) / 4096 * 4096;
  // the halo window behind the mailbox (ZZZ_P2P_HALO_MB, default 64; 0 = none: halo through the communicator)
  size_t halo_mb = 64;
  if (const char* hm = getenv("ZZZ_P2P_HALO_MB"))
    halo_mb = (size_t)std::max(0, atoi(hm));
  size_t halo_bytes = halo_mb << 20;
  if ((size_t)P->nranks * 16 > P2P_HALO_HDR * sizeof(double))
    halo_bytes = 0; // more ranks than the tag header holds
  if (halo_bytes)
  {
    P->halo_off = bytes / sizeof(double);
    P->halo_stride = (halo_bytes / sizeof(double) - P2P_HALO_HDR) / 2 / (size_t)P->nranks;
  }
  // uncached: remote stores must be visible to the local poll without a kernel boundary
  hipError_t e = hipExtMallocWithFlags(reinterpret_cast<void**>(&P->box), bytes + halo_bytes, hipDeviceMallocUncached);
  if (e != hipSuccess)
  {
    (void)hipGetLastError();
    e = hipExtMallocWithFlags(reinterpret_cast<void**>(&P->box), bytes + halo_bytes, hipDeviceMallocFinegrained);
  }
  if (e != hipSuccess && halo_bytes)
  {
    (void)hipGetLastError();
    halo_bytes = 0; // the mailbox alone
    P->halo_off = P->halo_stride = 0;
    e = hipExtMallocWithFlags(reinterpret_cast<void**>(&P->box), bytes, hipDeviceMallocUncached);
  }
  if (e == hipSuccess)
    e = hipMemset(P->box, 0xff, bytes + (halo_bytes ? P2P_HALO_HDR * sizeof(double) : 0)); // tags = -1: no round has that number
  if (e == hipSuccess && halo_bytes && (P->halo_ticket.alloc(4) != hipSuccess || hipMemset(P->halo_ticket.p, 0, 16) != hipSuccess))
  {
    (void)hipGetLastError();
    P->halo_off = P->halo_stride = 0;
  }
  P->halo_on = P->halo_off != 0;
  if (const char* hk = getenv("ZZZ_P2P_HALO")) // A/B knob: 0 keeps the halo on the communicator
    P->halo_on = P->halo_on && atoi(hk) != 0;
#ifdef ZZZ_EXPERIMENTS
  if (e == hipSuccess)
  {
    // partial arrays + tickets of the folded all-reduce (producers on all eight XCDs, one reader)
    const size_t tb = sizeof(double) * (size_t)TAIL_PART_DOUBLES + sizeof(int) * (size_t)TAIL_TICKET_INTS;
    // ordinary device memory: the producers store write-through (sc1), the reader acquires at agent scope (zzz_tail.h).
    // (Uncached memory was tried first: its stores alone stretched a 20-us product of 2048 workgroups to 40 us.)
    hipError_t e2 = hipMalloc(reinterpret_cast<void**>(&P->tail_mem), tb);
    if (e2 == hipSuccess)
      e2 = hipMemset(P->tail_mem, 0, tb);
    if (e2 != hipSuccess)
    {
      (void)hipGetLastError();
      P->tail_mem = nullptr; // the separate all-reduce kernel stays in use
    }
  }
#endif
  if (e == hipSuccess)
  {
    h.ok = 1;
    h.pid = (int64_t)getpid();
    h.raw = P->box;
    h.device = ctx->device;
    h.halo_off = P->halo_off;
    h.halo_stride = P->halo_stride;
    if (hipIpcGetMemHandle(&h.ipc, P->box) != hipSuccess)
    {
      (void)hipGetLastError();
      h.ok = 2; // usable by ranks of this process only
    }
  }
  else
  {
    (void)hipGetLastError();
    P->box = nullptr;
  }
  ctx->comm->p2p = P;
  memcpy(handle, &h, sizeof(h));
  return ZZZ_OK; // h.ok == 0 tells every rank (zzz_comm_p2p_attach) that this one has no mailbox
}

int zzz_comm_p2p_attach(zzz_ctx* ctx, const void* handles, int* enabled)
{
  if (!ctx)
    return fail(nullptr, ZZZ_ERR_ARG, "NULL context");
  ZZZ_HIP(ctx, hipSetDevice(ctx->device));
  if (enabled)
    *enabled = 0;
  if (!ctx->comm || !ctx->comm->p2p || !handles)
    return fail(ctx, ZZZ_ERR_ARG, "zzz_comm_p2p_attach: call zzz_comm_p2p_export first");
  P2P* P = ctx->comm->p2p;
  const P2PHandle* H = static_cast<const P2PHandle*>(handles);
  const int n = P->nranks;
  P->enabled = false;
  P->halo_agreed = -1;
  // a rank stores at peers[r] + halo_off + ...: the windows must have one geometry (every rank reads the same handles
  // here, so every rank drops the windows together).  A rank that allocated the mailbox alone exports 0 / 0.
  for (int r = 0; r < n; ++r)
    if (H[r].ok && (H[r].halo_off != (uint64_t)P->halo_off || H[r].halo_stride != (uint64_t)P->halo_stride))
    {
      P->halo_off = P->halo_stride = 0;
      P->halo_on = false;
    }
  for (int r = 0; r < n; ++r) // (a second pass: a mismatch between two OTHER ranks must switch this one off as well)
    for (int q = 0; q < n; ++q)
      if (H[r].ok && H[q].ok && (H[r].halo_off != H[q].halo_off || H[r].halo_stride != H[q].halo_stride))
      {
        P->halo_off = P->halo_stride = 0;
        P->halo_on = false;
      }
  P->peer.assign((size_t)n, nullptr);
  P->ipc_opened.assign((size_t)n, false);
  bool ok = P->box != nullptr;
  const int64_t me = (int64_t)getpid();
  for (int r = 0; r < n && ok; ++r)
  {
    if (!H[r].ok)
      ok = false;
    else if (r == P->rank)
      P->peer[r] = P->box;
    else if (H[r].pid == me)
    {
      // another context of this process: the pointer itself, after enabling peer access between GPUs
      if (H[r].device != ctx->device)
      {
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, ctx->device, H[r].device) != hipSuccess || !can)
          ok = false;
        else
        {
          hipError_t e = hipDeviceEnablePeerAccess(H[r].device, 0);
          if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled)
            ok = false;
          (void)hipGetLastError();
        }
      }
      P->peer[r] = H[r].raw;
      if (H[r].device == ctx->device)
        P->inproc = true;
    }
    else if (H[r].ok == 1)
    {
      void* q = nullptr;
      if (hipIpcOpenMemHandle(&q, H[r].ipc, hipIpcMemLazyEnablePeerAccess) != hipSuccess)
      {
        (void)hipGetLastError();
        ok = false;
      }
      else
      {
        P->peer[r] = static_cast<double*>(q);
        P->ipc_opened[r] = true;
      }
    }
    else
      ok = false;
  }
  ZZZ_HIP(ctx, P->peer_dev.alloc((size_t)n));
  ZZZ_HIP(ctx, P->fail.alloc(4));
  ZZZ_HIP(ctx, hipMemset(P->fail.p, 0, 4 * sizeof(int32_t)));
  if (ok)
    ZZZ_HIP(ctx, hipMemcpy(P->peer_dev.p, P->peer.data(), sizeof(double*) * (size_t)n, hipMemcpyHostToDevice));
  // Every rank must take the same decision, and it must rest on evidence: all ranks run the same
  // rounds of the kernel on known values (a rank that could not map a peer sits them out and the
  // others time out), then agree through the communicator that is already attached.
  double verdict = ok ? 1.0 : 0.0;
  if (ok)
  {
    zzz::DevBuf<double> tv;
    ZZZ_HIP(ctx, tv.alloc(8));
    const int rounds = 8;
    for (int k = 0; k < rounds && verdict == 1.0; ++k)
    {
      const double in[3] = {(double)(P->rank + 1) * (k + 1), 0.5 * (P->rank + 1), (double)k};
      ZZZ_HIP(ctx, hipMemcpyAsync(tv.p, in, sizeof(in), hipMemcpyHostToDevice, ctx->stream));
      const long long seq = ++P->seq;
      hipLaunchKernelGGL(k_allreduce_p2p, dim3(1), dim3(1024), 0, ctx->stream, (const int*)nullptr, tv.p, tv.p + 1, tv.p + 2,
                         1, 3, tv.p + 4, P->peer_dev.p, P->box, n, P->rank, seq, P->fail.p,
                         k == 0 ? 10 * P2P_TIMEOUT_TICKS : P2P_TIMEOUT_TICKS); // ranks reach round 0 far apart
      double outv[3];
      ZZZ_HIP(ctx, hipMemcpyAsync(outv, tv.p + 4, sizeof(outv), hipMemcpyDeviceToHost, ctx->stream));
      ZZZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
      const double tri = 0.5 * n * (n + 1);
      if (!(outv[0] == tri * (k + 1) && outv[1] == 0.5 * tri && outv[2] == (double)k * n))
        verdict = 0.0;
    }
    ZZZ_HIP(ctx, hipMemset(P->fail.p, 0, 4 * sizeof(int32_t)));
  }
  if (ctx->comm->comm || ctx->comm->local)
  {
    // product of the verdicts = 1 only if every rank passed (sum all-reduce of the failures)
    zzz::DevBuf<double> v;
    ZZZ_HIP(ctx, v.alloc(1));
    const double bad = 1.0 - verdict;
    ZZZ_HIP(ctx, hipMemcpyAsync(v.p, &bad, sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    int rc = comm_allreduce_sum(ctx, v.p, 1);
    if (rc)
      return rc;
    double nbad = 1.0;
    ZZZ_HIP(ctx, hipMemcpyAsync(&nbad, v.p, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    ZZZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (nbad != 0.0)
      verdict = 0.0;
  }
  P->enabled = P->verified = verdict == 1.0;
  if (enabled)
    *enabled = P->enabled ? 1 : 0;
  return halo_peer_agree(ctx); // (a halo plan uploaded before the mailboxes existed)
}

int zzz_comm_p2p_disable(zzz_ctx* ctx)
{
  if (!ctx)
    return fail(nullptr, ZZZ_ERR_ARG, "NULL context");
  if (ctx->comm && ctx->comm->p2p)
  {
    ctx->comm->p2p->enabled = false;
    ctx->comm->p2p->halo_agreed = -1;
  }
  return ZZZ_OK;
}

int zzz_comm_p2p_enable(zzz_ctx* ctx, int* enabled)
{
  if (!ctx)
    return fail(nullptr, ZZZ_ERR_ARG, "NULL context");
  if (ctx->comm && ctx->comm->p2p && ctx->comm->p2p->verified)
    ctx->comm->p2p->enabled = true;
  if (enabled)
    *enabled = comm_p2p_enabled(ctx) ? 1 : 0;
  return halo_peer_agree(ctx);
}

int zzz_comm_p2p_halo(zzz_ctx* ctx, int on, int* in_use)
{
  if (!ctx)
    return fail(nullptr, ZZZ_ERR_ARG, "NULL context");
  if (ctx->comm && ctx->comm->p2p)
    ctx->comm->p2p->halo_on = on != 0 && ctx->comm->p2p->halo_off != 0;
  const int rc = halo_peer_agree(ctx); // collective: the ranks' common verdict
  if (in_use)
    *in_use = (ctx->comm && ctx->comm->p2p && ctx->comm->p2p->halo_agreed == 1) ? 1 : 0;
  return rc;
}

int zzz_local_group_create(int nranks, void** group)
{
  if (nranks < 1 || !group)
    return fail(nullptr, ZZZ_ERR_ARG, "zzz_local_group_create: bad arguments");
  LocalGroup* G = new LocalGroup();
  G->n = nranks;
  G->bar.n = nranks;
  G->red.resize((size_t)nranks);
  G->mail.resize((size_t)nranks);
  G->neigh.resize((size_t)nranks);
  G->send_off.resize((size_t)nranks);
  G->bs.assign((size_t)nranks, 1);
  G->xch.resize((size_t)nranks);
  *group = G;
  return ZZZ_OK;
}

void zzz_local_group_destroy(void* group)
{
  LocalGroup* G = static_cast<LocalGroup*>(group);
  if (!G)
    return;
  delete G;
}

void zzz_local_group_abort(void* group)
{
  if (LocalGroup* G = static_cast<LocalGroup*>(group))
    G->bar.abort();
}

int zzz_comm_init_local(zzz_ctx* ctx, void* group, int rank)
{
  if (!ctx)
    return fail(nullptr, ZZZ_ERR_ARG, "NULL context");
  LocalGroup* G = static_cast<LocalGroup*>(group);
  if (!G || rank < 0 || rank >= G->n)
    return fail(ctx, ZZZ_ERR_ARG, "zzz_comm_init_local: bad group or rank %d", rank);
  comm_destroy(ctx);
  Comm* c = new Comm();
  c->local = G;
  c->nranks = G->n;
  c->rank = rank;
  ctx->comm = c;
  return ZZZ_OK;
}

int zzz_halo_upload(zzz_ctx* ctx, int nneigh, const int32_t* neigh_rank, const int64_t* send_off,
                    const int32_t* send_idx, const int64_t* recv_cnt)
{
  if (!ctx)
    return fail(nullptr, ZZZ_ERR_ARG, "NULL context");
  ZZZ_HIP(ctx, hipSetDevice(ctx->device));
  if (ctx->order == 0)
    return fail(ctx, ZZZ_ERR_ARG, "zzz_halo_upload before zzz_dofmap_upload");
  if (nneigh < 0 || (nneigh > 0 && (!neigh_rank || !send_off || !recv_cnt)))
    return fail(ctx, ZZZ_ERR_ARG, "zzz_halo_upload: bad arguments");
  int64_t nrecv = 0;
  for (int k = 0; k < nneigh; ++k)
  {
    if (send_off[k + 1] < send_off[k] || recv_cnt[k] < 0)
      return fail(ctx, ZZZ_ERR_ARG, "zzz_halo_upload: negative count for neighbour %d", k);
    nrecv += recv_cnt[k];
  }
  if (nrecv != ctx->n_ghost)
    return fail(ctx, ZZZ_ERR_ARG, "halo receives %lld block dofs but the dofmap has %lld ghosts", (long long)nrecv,
                (long long)ctx->n_ghost);
  const int64_t nsend = nneigh ? send_off[nneigh] : 0;
  for (int64_t i = 0; i < nsend; ++i)
    if (send_idx[i] < 0 || send_idx[i] >= ctx->n_owned)
      return fail(ctx, ZZZ_ERR_ARG, "send_idx[%lld] = %d is not an owned block dof", (long long)i, send_idx[i]);
  std::vector<int32_t> send_internal;
  if (ctx->renumbered && nsend)
  {
    send_internal.assign(send_idx, send_idx + nsend);
    for (int32_t& d : send_internal)
      d = ctx->h_iperm[(size_t)d];
    send_idx = send_internal.data();
  }
  ctx->nneigh = nneigh;
  ctx->neigh_rank.assign(neigh_rank, neigh_rank + nneigh);
  ctx->send_off.assign(send_off, send_off + nneigh + 1);
  ctx->recv_cnt.assign(recv_cnt, recv_cnt + nneigh);
  if (nneigh == 0)
    ctx->send_off.assign(1, 0);
  ctx->send_contig.assign((size_t)nneigh, -1);
  for (int k = 0; k < nneigh; ++k)
  {
    bool contig = send_off[k + 1] > send_off[k];
    for (int64_t i = send_off[k] + 1; i < send_off[k + 1] && contig; ++i)
      contig = send_idx[i] == send_idx[i - 1] + 1;
    if (contig)
      ctx->send_contig[(size_t)k] = send_idx[send_off[k]];
  }
  ZZZ_HIP(ctx, ctx->send_idx.alloc((size_t)nsend));
  ZZZ_HIP(ctx, ctx->send_buf.alloc((size_t)(nsend * ctx->bs)));
  if (nsend)
    ZZZ_HIP(ctx, hipMemcpy(ctx->send_idx.p, send_idx, (size_t)nsend * sizeof(int32_t), hipMemcpyHostToDevice));
  if (ctx->comm && ctx->comm->local)
  {
    LocalGroup* G = ctx->comm->local;
    const int me = ctx->comm->rank;
    G->neigh[me] = ctx->neigh_rank;
    G->send_off[me] = ctx->send_off;
    G->bs[me] = ctx->bs;
  }
  return halo_peer_agree(ctx); // a new plan: the ranks decide again whether it travels through the windows
}

} // extern "C"

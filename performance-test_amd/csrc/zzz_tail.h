// The CG scalar all-reduce folded into the TAIL of the kernel that produces the partial sums (the product, or
// k_update_xr): MPI_Allreduce of la::inner_product / la::squared_norm, src/cg.h:53,65,74.
//
// Without it a reduction point of a multi-GPU iteration is [producer kernel] -> boundary -> [k_allreduce_p2p, one
// workgroup: sum the per-workgroup partials, exchange through the peer mailboxes] -> boundary -> [consumer kernel]:
// 4.8-6.4 us of kernel plus a launch boundary on the critical path of a ~50-us iteration at the 8-GPU per-rank size.
// Here every workgroup of the producer leaves its partials in UNCACHED memory (write-through, coherent across the
// eight XCDs without cache maintenance), takes a ticket, and the workgroup whose ticket is the last one does what
// k_allreduce_p2p does -- the same summation tree, the same mailbox protocol, the same bits -- while the rest of
// the grid has already left the chip.  One poller per rank (the variant with every workgroup polling the mailbox
// cost 130 us).
//
// MEASURED (round 3, 1.25 M rows, 1-rank communicator + mailboxes, rocprofv3 + in-kernel wall-clock stamps): the tail is a
// chain of dependent memory-side round trips of ~0.5-0.7 us each -- write-through store acknowledged (s_waitcnt), shard
// ticket, top ticket, acquire, the read of 2048 x 3 partials (48 KB, 2-4 us under the producers' 64-VGPR cap), mailbox
// store, mailbox poll -- 7-10 us in all, against 6.3 us for k_allreduce_p2p plus ~1 us of launch boundary: the folded
// form is ~1 us SLOWER per reduction point (single-reduction iteration 51.6 vs 50.5 us; 54.4 with the slots spread over pages).  A kernel boundary is cheaper
// than two extra round trips to the memory side.  The code stays as an A/B variant, OFF by default (ZZZ_TAIL=1), pinned
// bit for bit to the separate kernel by tests/test_gpu_parity.py.  Hand-off form: MI355X_MICROARCH.md "Valid forms": sc1 stores -> s_waitcnt vmcnt(0) ->
// agent-scope atomic add; the workgroup whose add came last reads with sc1 loads behind a workgroup barrier.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace zzz
{
constexpr int TAIL_STRIDE = 4096; // most partials per array (>= SPMV_PSTRIDE, >= VGRID_MAX)
constexpr int TAIL_P2P_SLOT = 4;  // doubles per mailbox slot (P2P_SLOT of zzz_comm.hip)
// Arrivals are counted in two levels: an atomic on uncached memory executes at the memory side, ~12 ns each and one
// after the other per address (MI355X_MICROARCH.md, fanin / dequeue rows) -- 2048 workgroups on ONE counter added 25 us
// to a 20-us product in the guide's measurements of that shape.  Workgroup i takes a ticket of shard i mod TAIL_SHARDS (its own 128-B line, so the shards
// proceed in parallel), the last of a shard one of the top counter: <= 64 + 32 serialised adds instead of 2048.
constexpr int TAIL_SHARDS = 32;
constexpr int TAIL_PART_DOUBLES = 3 * TAIL_STRIDE;              // three contiguous arrays: the reader's loads coalesce
constexpr int TAIL_TICKET_STRIDE = 64;                          // ints: one 256-B block per counter
constexpr int TAIL_TICKET_INTS = (TAIL_SHARDS + 1) * TAIL_TICKET_STRIDE; // the top counter comes last
__host__ __device__ inline int tail_slot(int j, int idx) { return j * TAIL_STRIDE + idx; }

struct TailArgs
{
  double* parts = nullptr; // TAIL_PART_DOUBLES, laid out by tail_slot; null = no folded all-reduce (the kernel behaves as before)
  int* ticket = nullptr;   // TAIL_TICKET_INTS, zero between uses
  int expected = 0;        // arrivals (workgroups over all launches of this producer) that complete the reduction
  int base = 0;            // index of this launch's workgroup 0 in the partial arrays
  int nv = 0;              // values to reduce (1..3)
  double* out = nullptr;   // all-reduced values (ordinary device memory: the consumer is a later kernel)
  double* const* peers = nullptr;
  double* box = nullptr;
  int nranks = 1, rank = 0;
  long long seq = 0;
  int* fail = nullptr;
  long long timeout = 0;
};

// Called by EVERY thread of the workgroup at the end of the producer; thread 0 holds the workgroup's partial sums
// v0..v2 (in OUTPUT order).  blockDim.x must be 256.
#ifdef ZZZ_TAIL_DEBUG
#define ZZZ_TT(i) if (threadIdx.x == 0) tt[i] = wall_clock64()
#else
#define ZZZ_TT(i)
#endif
#ifndef ZZZ_EXPERIMENTS
// the product library: no folded all-reduce (comm_tail_args never arms it); the call sites compile to nothing
__device__ inline void tail_arrive(const TailArgs&, double, double, double) {}
#else
__device__ inline void tail_arrive(const TailArgs& T, double v0, double v1, double v2)
{
#ifdef ZZZ_TAIL_DEBUG
  __shared__ long long tt[8];
#endif
  ZZZ_TT(0);
  __shared__ int last_flag;
  __shared__ double part[3][8];
  __shared__ double val[3];
  __shared__ int timed_out;
  if (threadIdx.x == 0)
  {
    const int idx = T.base + (int)blockIdx.x;
    __hip_atomic_store(T.parts + tail_slot(0, idx), v0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (T.nv > 1)
      __hip_atomic_store(T.parts + tail_slot(1, idx), v1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (T.nv > 2)
      __hip_atomic_store(T.parts + tail_slot(2, idx), v2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the stores have left before the ticket is taken
    const int shard = idx % TAIL_SHARDS;
    const int in_shard = T.expected / TAIL_SHARDS + (shard < T.expected % TAIL_SHARDS ? 1 : 0);
    const int nshards = T.expected < TAIL_SHARDS ? T.expected : TAIL_SHARDS;
    int last = 0;
    if (__hip_atomic_fetch_add(T.ticket + TAIL_TICKET_STRIDE * shard, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == in_shard - 1)
    {
      __hip_atomic_store(T.ticket + TAIL_TICKET_STRIDE * shard, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // ready for the next producer
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (__hip_atomic_fetch_add(T.ticket + TAIL_TICKET_STRIDE * TAIL_SHARDS, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nshards - 1)
      {
        __hip_atomic_store(T.ticket + TAIL_TICKET_STRIDE * TAIL_SHARDS, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = 1;
      }
    }
    last_flag = last;
    timed_out = 0;
    ZZZ_TT(1);
    if (last)
    {
      // consumer side of the hand-off: one agent-scope acquire on this CU, waited for, before the workgroup reads
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  __syncthreads();
  if (!last_flag)
    return;
  ZZZ_TT(2);
  // ---- the last workgroup: k_allreduce_p2p's body with 512 virtual threads on 256 real ones (same tree) -------
  // Plain loads behind the acquire, issued together.  (Atomic or volatile loads are issued ONE AT A TIME by the compiler,
  // each waited for: 48 of them made this tail 24 us long -- the whole cost of the first versions of this function,
  // which was first blamed on the ticket counter, then on the memory type, then on the stores.)
  const int np = T.expected;
  double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
  {
    // one array at a time, its (at most 16) values per thread requested together: the producer kernels are capped at
    // 64 VGPRs, which is what 3 x 16 values would need on their own
    constexpr int MAXI = TAIL_STRIDE / 512; // values per virtual thread
    const double* __restrict__ parts = T.parts;
    for (int j = 0; j < T.nv; ++j)
    {
      double bl[MAXI], bh[MAXI];
      // unconditional loads from clamped slots (a predicated load sits in a branch of its own and is waited for at the
      // join: serial again); what lies beyond np is dropped by the sums below
#pragma unroll
      for (int k = 0; k < MAXI; ++k)
      {
        const int il = (int)threadIdx.x + 512 * k, ih = il + 256;
        bl[k] = parts[tail_slot(j, il < np ? il : 0)];
        bh[k] = parts[tail_slot(j, ih < np ? ih : 0)];
      }
      double sl = 0, sh = 0;
#pragma unroll
      for (int k = 0; k < MAXI; ++k)
      {
        const int il = (int)threadIdx.x + 512 * k, ih = il + 256;
        if (il < np)
          sl += bl[k];
        if (ih < np)
          sh += bh[k];
      }
      lo[j] = sl;
      hi[j] = sh;
    }
  }
  ZZZ_TT(3);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1)
#pragma unroll
    for (int j = 0; j < 3; ++j)
    {
      lo[j] += __shfl_down(lo[j], o, 64);
      hi[j] += __shfl_down(hi[j], o, 64);
    }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0)
    for (int j = 0; j < 3; ++j)
    {
      part[j][wv] = lo[j];
      part[j][wv + 4] = hi[j];
    }
  __syncthreads();
  if ((int)threadIdx.x < T.nv)
  {
    double t = 0;
    for (int w = 0; w < 8; ++w)
      t += part[threadIdx.x][w];
    val[threadIdx.x] = t;
  }
  __syncthreads();
  ZZZ_TT(4);
  const int par = (int)(T.seq & 1);
  if ((int)threadIdx.x < T.nranks)
  {
    double* slot = T.peers[threadIdx.x] + ((size_t)par * T.nranks + T.rank) * TAIL_P2P_SLOT;
    for (int j = 0; j < T.nv; ++j)
      __hip_atomic_store(slot + j, val[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(reinterpret_cast<long long*>(slot + 3), T.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    const double* mine = T.box + ((size_t)par * T.nranks + threadIdx.x) * TAIL_P2P_SLOT;
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(reinterpret_cast<const long long*>(mine + 3), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != T.seq)
    {
      if (wall_clock64() - t0 > T.timeout)
      {
        timed_out = 1;
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
  }
  __syncthreads();
  ZZZ_TT(5);
#ifdef ZZZ_TAIL_DEBUG
  if (threadIdx.x == 0 && (T.seq & 127) == 5)
    printf("tail seq %lld np %d nv %d: ticket %lld fence+bar %lld loads %lld reduce %lld mailbox %lld (x10 ns)\n", T.seq, T.expected,
           T.nv, tt[1] - tt[0], tt[2] - tt[1], tt[3] - tt[2], tt[4] - tt[3], tt[5] - tt[4]);
#endif
  if (threadIdx.x == 0)
  {
    if (timed_out)
    {
      *T.fail = 1;
      for (int j = 0; j < T.nv; ++j)
        T.out[j] = __builtin_nan("");
    }
    else
      for (int j = 0; j < T.nv; ++j)
      {
        double s = 0;
        for (int r = 0; r < T.nranks; ++r) // rank order: the same sum on every rank
          s += __hip_atomic_load(T.box + ((size_t)par * T.nranks + r) * TAIL_P2P_SLOT + j, __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_SYSTEM);
        T.out[j] = s;
      }
  }
}
#endif // ZZZ_EXPERIMENTS
} // namespace zzz

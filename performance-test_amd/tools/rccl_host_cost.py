#!/usr/bin/env python3
"""Host-side cost of the per-iteration RCCL calls of the multi-GPU CG loop, measured on ONE GPU: a 1-rank
communicator sending to / receiving from itself (ncclGroupStart; ncclSend; ncclRecv; ncclGroupEnd) with the
halo size of the 8-GPU run (one vertex plane of the 10 M-dof cube = 44 919 doubles), and a 3-double ncclAllReduce.
What is timed is how long the HOST needs to enqueue them (the CG loop enqueues ~5 kernels + these per iteration
and must stay ahead of ~85 us of GPU work) and, for reference, the device-side duration of the self-exchange."""
import ctypes as C
import time

hip = C.CDLL("libamdhip64.so")
rccl = C.CDLL("librccl.so.1")


class Uid(C.Structure):
    _fields_ = [("b", C.c_char * 128)]


def ck(rc, what):
    assert rc == 0, f"{what} failed: {rc}"


ck(hip.hipSetDevice(0), "hipSetDevice")
uid = Uid()
ck(rccl.ncclGetUniqueId(C.byref(uid)), "ncclGetUniqueId")
comm = C.c_void_p()
rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, Uid, C.c_int]
ck(rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0), "ncclCommInitRank")
n = 44919
a, b, r = C.c_void_p(), C.c_void_p(), C.c_void_p()
for p, sz in ((a, 8 * n), (b, 8 * n), (r, 64)):
    ck(hip.hipMalloc(C.byref(p), C.c_size_t(sz)), "hipMalloc")
stream = C.c_void_p()
ck(hip.hipStreamCreate(C.byref(stream)), "hipStreamCreate")
rccl.ncclSend.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
rccl.ncclRecv.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
rccl.ncclAllReduce.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
F64, SUM = 8, 0


def halo():
    rccl.ncclGroupStart()
    rccl.ncclSend(a, n, F64, 0, comm, stream)
    rccl.ncclRecv(b, n, F64, 0, comm, stream)
    rccl.ncclGroupEnd()


def allreduce():
    rccl.ncclAllReduce(r, r, 3, F64, SUM, comm, stream)


for name, fn in (("halo group (send + recv of 44 919 doubles)", halo), ("all-reduce of 3 doubles", allreduce)):
    for _ in range(20):
        fn()
    hip.hipStreamSynchronize(stream)
    reps = 2000
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    t1 = time.perf_counter()
    hip.hipStreamSynchronize(stream)
    t2 = time.perf_counter()
    print(f"{name}: host enqueue {1e6 * (t1 - t0) / reps:.1f} us per call; {1e6 * (t2 - t0) / reps:.1f} us per call until the GPU has finished")

"""Plumbing probe for the N > 1 bench launch: torchrun env, gloo init over 127.0.0.1, broadcast, barrier,
all-reduce MAX -- exactly the torch.distributed calls bench.py makes (no GPU work)."""
import os

import torch
import torch.distributed as dist

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group(backend="gloo", init_method="env://", world_size=world, rank=rank)
uid = torch.zeros(128, dtype=torch.uint8)
if rank == 0:
    uid = torch.arange(128, dtype=torch.uint8)
dist.broadcast(uid, src=0)
assert int(uid[127]) == 127
dist.barrier()
t = torch.tensor([float(rank)], dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert t[0] == world - 1
# the peer-memory handle exchange and the warm-up verdict of bench.py
mine = torch.frombuffer(bytearray(bytes([rank]) * 128), dtype=torch.uint8).clone()
allh = [torch.zeros(128, dtype=torch.uint8) for _ in range(world)]
dist.all_gather(allh, mine)
blob = b"".join(bytes(h.numpy().tobytes()) for h in allh)
assert len(blob) == 128 * world and all(blob[128 * r] == r for r in range(world))
flag = torch.tensor([0 if rank == world - 1 else 1], dtype=torch.int32)
dist.all_reduce(flag, op=dist.ReduceOp.MIN)
assert int(flag.item()) == 0
dist.barrier()
dist.destroy_process_group()
if rank == 0:
    print("gloo probe ok", world)

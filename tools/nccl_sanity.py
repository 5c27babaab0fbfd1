#!/usr/bin/env python3
"""The torch.distributed calls bench.py --gpus N makes on the RCCL side, with one rank (the box has one GPU):
init with device_id, a gloo group next to it, barrier, all_reduce on the device, a self send/recv pair.
usage: python -m torch.distributed.run --nproc-per-node 1 --master-addr 127.0.0.1 tools/nccl_sanity.py"""
import os
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
lr = int(os.environ.get("LOCAL_RANK", "0"))
dist.init_process_group(backend="nccl", device_id=torch.device("cuda", lr))
torch.cuda.set_device(lr)
g = dist.new_group(backend="gloo")
dist.barrier()
t = torch.tensor([1.5], dtype=torch.float64, device=f"cuda:{lr}")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
z = torch.zeros(1, dtype=torch.int64)
dist.gather(z, [torch.zeros_like(z)], dst=0, group=g)
a = torch.arange(8, dtype=torch.uint8, device=f"cuda:{lr}")
b = torch.zeros(8, dtype=torch.uint8, device=f"cuda:{lr}")
ok_p2p = None
try:
    for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, a, 0), dist.P2POp(dist.irecv, b, 0)]):
        w.wait()
    torch.cuda.synchronize()
    ok_p2p = bool(torch.equal(a, b))
except Exception as ex:  # a rank sending to itself is not what the bench does; reported, not required
    ok_p2p = repr(ex)
print("nccl sanity:", float(t.item()), "self p2p:", ok_p2p, flush=True)
dist.barrier()
dist.destroy_process_group()

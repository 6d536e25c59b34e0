#!/usr/bin/env python3
"""Host-side cost of one torch.distributed batch_isend_irecv round on RCCL, measured
with a single rank sending to itself (RCCL implements self send/recv as a copy), and
of the D2H read of a small counts tensor.  Sizing input for the slab driver."""
import os, time
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29531")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
for nbytes in (64, 170_000, 1_000_000):
    x = torch.ones(nbytes // 4, device="cuda"); y = torch.zeros_like(x)
    x2 = torch.ones(nbytes // 4, device="cuda"); y2 = torch.zeros_like(x)
    def one():
        ops = [dist.P2POp(dist.isend, x, 0), dist.P2POp(dist.irecv, y, 0),
               dist.P2POp(dist.isend, x2, 0), dist.P2POp(dist.irecv, y2, 0)]
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    for _ in range(20): one()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(500): one()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{nbytes:>8} B x2 each way: host {1e6*(t1-t0)/500:.1f} us/round, incl. GPU drain {1e6*(t2-t0)/500:.1f} us", flush=True)
c = torch.arange(4, device="cuda", dtype=torch.int64)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(500): c.tolist()
print(f"tolist of 4 int64: {1e6*(time.perf_counter()-t0)/500:.1f} us")
t0 = time.perf_counter()
for _ in range(500): torch.tensor([1, 2, 3, 4], dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
print(f"torch.tensor(list, device=cuda): {1e6*(time.perf_counter()-t0)/500:.1f} us")
dist.destroy_process_group()

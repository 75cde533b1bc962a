"""rccl_probe.py - does RCCL come up on this box at world_size 1?  (tools only; the product path is rcognita_amd/parallel.py)"""
import os, sys, time
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch
import torch.distributed as dist
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
t0 = time.perf_counter()
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
print("init", time.perf_counter() - t0, dist.get_backend(), flush=True)
x = torch.arange(65536, device=dev, dtype=torch.float32)
out = [torch.empty_like(x)]
t0 = time.perf_counter(); dist.all_gather(out, x); torch.cuda.synchronize(); print("first all_gather", time.perf_counter() - t0, flush=True)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20):
    dist.all_gather(out, x)
b.record(); torch.cuda.synchronize()
print("all_gather ms each", a.elapsed_time(b) / 20, bool((out[0] == x).all()), flush=True)
flat = torch.empty(65536, device=dev); dist.all_gather_into_tensor(flat, x); torch.cuda.synchronize(); print("into_tensor ok", bool((flat == x).all()))
s = torch.tensor([1., 2, 3, 4, 5, 6], dtype=torch.float64, device=dev); parts = [torch.empty_like(s)]; dist.all_gather(parts, s)
m = torch.tensor([3.0], device=dev, dtype=torch.float64); dist.all_reduce(m, op=dist.ReduceOp.MAX); dist.barrier()
torch.cuda.synchronize(); print("summary", parts[0].tolist(), m.item(), flush=True)
dist.destroy_process_group(); print("destroyed")

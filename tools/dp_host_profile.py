"""Host-time breakdown of one data-parallel step (1-rank RCCL group)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
x = torch.zeros(1560000, device="cuda"); y = torch.zeros(8000, device="cuda"); m = torch.zeros(20108, dtype=torch.int32, device="cuda")
for _ in range(20):
  dist.all_reduce(x); dist.all_reduce(y); dist.all_reduce(m, op=dist.ReduceOp.MAX)
torch.cuda.synchronize()
def t(fn, n=200):
  t0 = time.perf_counter()
  for _ in range(n): fn()
  dt = (time.perf_counter() - t0) / n * 1e6
  torch.cuda.synchronize(); return dt
print("all_reduce(6MB) host us:", t(lambda: dist.all_reduce(x)))
print("all_reduce(32KB) host us:", t(lambda: dist.all_reduce(y)))
print("all_reduce async + wait host us:", t(lambda: dist.all_reduce(y, async_op=True).wait()))
print("all_reduce MAX int32 host us:", t(lambda: dist.all_reduce(m, op=dist.ReduceOp.MAX)))
s2 = torch.cuda.Stream()
def f():
  with torch.cuda.stream(s2): dist.all_reduce(y)
print("all_reduce under other stream host us:", t(f))
e = torch.cuda.Event()
print("event record+wait host us:", t(lambda: (e.record(), s2.wait_event(e))))
print("tensor slice+copy_ host us:", t(lambda: y[:1].copy_(x[:1])))
torch.cuda.synchronize()
t0=time.perf_counter(); 
for _ in range(200): dist.all_reduce(x)
torch.cuda.synchronize(); print("all_reduce(6MB) 1-rank gpu+host us:", (time.perf_counter()-t0)/200*1e6)
dist.destroy_process_group()

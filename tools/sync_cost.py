"""GPU-side cost of HIP event records and cross-stream waits between small kernels."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recoder_amd import _lib
from recoder_amd._lib import check, ptr
import ctypes
lib = _lib.load()
dev = torch.device("cuda")
X = torch.randn(64, 256, device=dev); out = torch.empty(256, device=dev)
s1 = torch.cuda.current_stream(); s2 = torch.cuda.Stream()
def k(stream):
  check(lib.rk_colsum(ptr(X), 64, 256, 256, None, ptr(out), ctypes.c_void_p(stream.cuda_stream)))
def bench(fn, n=300):
  for _ in range(20): fn()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(n): fn()
  torch.cuda.synchronize()
  return (time.perf_counter() - t0) / n * 1e6
e1 = torch.cuda.Event(); e2 = torch.cuda.Event()
print("k;k same stream           : %.1f us" % bench(lambda: (k(s1), k(s1))))
print("k;record;k same stream    : %.1f us" % bench(lambda: (k(s1), e1.record(s1), k(s1))))
def pingpong():
  k(s1); e1.record(s1); s2.wait_event(e1); k(s2); e2.record(s2); s1.wait_event(e2)
print("k(s1)->ev->k(s2)->ev->s1  : %.1f us" % bench(pingpong))
def fork_join():
  e1.record(s1); s2.wait_event(e1); k(s1); k(s2); e2.record(s2); s1.wait_event(e2)
print("fork: k(s1)||k(s2) join   : %.1f us" % bench(fork_join))
rawe = [lib.rk_event_create(0) for _ in range(2)]
hip = ctypes.CDLL("libamdhip64.so")
def raw_pingpong():
  k(s1); hip.hipEventRecord(ctypes.c_void_p(rawe[0]), ctypes.c_void_p(s1.cuda_stream)); hip.hipStreamWaitEvent(ctypes.c_void_p(s2.cuda_stream), ctypes.c_void_p(rawe[0]), 0)
  k(s2); hip.hipEventRecord(ctypes.c_void_p(rawe[1]), ctypes.c_void_p(s2.cuda_stream)); hip.hipStreamWaitEvent(ctypes.c_void_p(s1.cuda_stream), ctypes.c_void_p(rawe[1]), 0)
print("raw hip event pingpong    : %.1f us" % bench(raw_pingpong))

def mk(flags):
  e = ctypes.c_void_p()
  assert hip.hipEventCreateWithFlags(ctypes.byref(e), ctypes.c_uint(flags)) == 0
  return e
big = torch.zeros(25_000_000, device=dev)
for name, fl in (("default", 0), ("notiming", 2), ("notiming+relToDevice", 2 | 0x40000000),
                 ("notiming+noSysFence", 2 | 0x20000000)):
  ev = [mk(fl), mk(fl)]
  def pp(kern=k):
    kern(s1); hip.hipEventRecord(ev[0], ctypes.c_void_p(s1.cuda_stream)); hip.hipStreamWaitEvent(ctypes.c_void_p(s2.cuda_stream), ev[0], 0)
    k(s2); hip.hipEventRecord(ev[1], ctypes.c_void_p(s2.cuda_stream)); hip.hipStreamWaitEvent(ctypes.c_void_p(s1.cuda_stream), ev[1], 0)
  def same():
    k(s1); hip.hipEventRecord(ev[0], ctypes.c_void_p(s1.cuda_stream)); k(s1)
  def bigk(stream):
    with torch.cuda.stream(stream):
      big.add_(1.0)
  print("%-22s pingpong %.1f us | same-stream record %.1f us | big-kernel pingpong %.1f us" %
        (name, bench(pp), bench(same), bench(lambda: pp(bigk), n=100)))
print("big kernel alone + k same stream: %.1f us" % bench(lambda: (bigk(s1), k(s1)), n=100))

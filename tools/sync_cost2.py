"""GPU-side (not host-bound) cost of event patterns on the recording stream: ~25 us kernels."""
import os, sys, time, ctypes
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recoder_amd import _lib
from recoder_amd._lib import check, ptr
lib = _lib.load()
hip = ctypes.CDLL("libamdhip64.so")
dev = torch.device("cuda")
X = torch.randn(64, 256, device=dev); out = torch.empty(256, device=dev)
big = torch.zeros(20_000_000, device=dev)
s1 = torch.cuda.current_stream(); s2 = torch.cuda.Stream()
S1 = ctypes.c_void_p(s1.cuda_stream); S2 = ctypes.c_void_p(s2.cuda_stream)
def small(stream):
  check(lib.rk_colsum(ptr(X), 64, 256, 256, None, ptr(out), ctypes.c_void_p(stream.cuda_stream)))
def K():
  big.add_(1.0)
def mk(flags):
  e = ctypes.c_void_p(); assert hip.hipEventCreateWithFlags(ctypes.byref(e), ctypes.c_uint(flags)) == 0; return e
def bench(fn, n=200):
  for _ in range(20): fn()
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for _ in range(n): fn()
  t1 = time.perf_counter(); torch.cuda.synchronize()
  return (time.perf_counter() - t0) / n * 1e6, (t1 - t0) / n * 1e6
for name, fl in (("notiming", 2), ("noSysFence", 2 | 0x20000000)):
  e = [mk(fl) for _ in range(4)]
  rec = lambda ev, S: hip.hipEventRecord(ev, S)
  wait = lambda S, ev: hip.hipStreamWaitEvent(S, ev, 0)
  def a(): K(); K()
  def b(): K(); rec(e[0], S1); K(); rec(e[1], S1)
  def c(): K(); rec(e[0], S1); wait(S2, e[0]); small(s2); K()
  def d(): K(); rec(e[0], S1); wait(S2, e[0]); small(s2); rec(e[1], S2); K(); wait(S1, e[1])
  def d2(): K(); rec(e[0], S1); wait(S2, e[0]); small(s2); rec(e[1], S2); K(); K(); wait(S1, e[1])
  for nm, fn in (("a K;K", a), ("b K;rec;K;rec", b), ("c K;rec;[s2 waits,small];K", c),
                 ("d c + s1 joins s2 at end", d), ("d2 K;fork;K;K;join", d2)):
    tot, host = bench(fn)
    print("%-11s %-30s gpu %.1f us/iter (host enqueue %.1f)" % (name, nm, tot, host))

"""Hidden nn.Linear layers (reference nn.py:242-249) through rk_linear_fwd / rk_linear_bwd in
isolation: B x N x K per call, HIP-event medians, checked against torch.
    python tools/probes/linear_probe.py [B N K]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from recoder_amd import _lib  # noqa: E402
from recoder_amd._lib import check, ptr  # noqa: E402
from recoder_amd.device import current_stream  # noqa: E402

ACT_TANH = 1


def timeit(fn, n=50, warm=5):
  for _ in range(warm):
    fn()
  torch.cuda.synchronize()
  evs = []
  for _ in range(n):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); fn(); e.record()
    evs.append((s, e))
  torch.cuda.synchronize()
  t = sorted(s.elapsed_time(e) * 1e3 for s, e in evs)
  return t[len(t) // 2]


def main():
  B, N, K = [int(x) for x in sys.argv[1:4]] if len(sys.argv) >= 4 else (500, 200, 200)
  lib = _lib.load()
  f = dict(dtype=torch.float32, device="cuda")
  torch.manual_seed(0)
  X, W, b = torch.randn(B, K, **f), torch.randn(N, K, **f) * 0.1, torch.randn(N, **f) * 0.1
  Y, dX, dW, db = torch.empty(B, N, **f), torch.empty(B, K, **f), torch.empty(N, K, **f), torch.empty(N, **f)
  dY0 = torch.randn(B, N, **f)
  dY = dY0.clone()
  st = current_stream()
  fwd = lambda: check(lib.rk_linear_fwd(ptr(X), ptr(W), ptr(b), B, N, K, 0, ACT_TANH, ptr(Y), st), "fwd")
  def bwd():
    check(lib.rk_linear_bwd(ptr(dY), ptr(Y), ptr(X), ptr(W), B, N, K, 0, ACT_TANH, ptr(dX), ptr(dW), 0,
                            ptr(db), st), "bwd")
  fwd()
  ref = torch.tanh(X.double() @ W.double().t() + b.double())
  print("fwd  max err %.2e" % (Y.double() - ref).abs().max().item())
  dY.copy_(dY0); bwd()
  g = dY0.double() * (1 - ref * ref)
  print("bwd  max err dX %.2e dW %.2e db %.2e" % ((dX.double() - g @ W.double()).abs().max().item(),
        (dW.double() - g.t() @ X.double()).abs().max().item(), (db.double() - g.sum(0)).abs().max().item()))
  import numpy as np
  sys.path.insert(0, os.path.join(ROOT, "tools"))
  from gemm_probe import report
  probe = torch.zeros(8 * 20000, dtype=torch.int64, device="cuda")
  for name, fn in (("linear fwd", fwd), ("linear bwd (dX then dW launch: stamps of both)", bwd)):
    torch.cuda.synchronize(); probe.zero_()
    lib.rk_gemm_probe(ptr(probe)); fn(); torch.cuda.synchronize(); lib.rk_gemm_probe(None)
    report(name, probe)
  print("B=%d N=%d K=%d: rk_linear_fwd %.1f us, rk_linear_bwd %.1f us" % (B, N, K, timeit(fwd), timeit(bwd)))


if __name__ == "__main__":
  main()

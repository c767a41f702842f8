"""Times rk_linear_bwd at C3's layer shape (500 x 200 x 200): launch by launch with HIP events.
argv[1] = 0: dX and dW as two launches (round 2 / early round 3; rk_tune RK_TUNE_LINEAR_PAIR)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from recoder_amd import _lib
from recoder_amd._lib import check, ptr
from recoder_amd.device import current_stream

lib = _lib.load()
PAIR = int(sys.argv[1]) if len(sys.argv) > 1 else 1
lib.rk_linear_pair(PAIR)
dev = torch.device("cuda")
B, N, K = 500, 200, 200
f = lambda *s: torch.randn(*s, device=dev)
X, W, dY, Y = f(B, K), f(N, K) * 0.1, f(B, N), torch.tanh(f(B, N))
dX, dW, db, Xa = torch.empty(B, K, device=dev), torch.empty(N, K, device=dev), torch.empty(N, device=dev), torch.tanh(f(B, K))
st = current_stream()
for name, fn in [("rk_linear_bwd", lambda: lib.rk_linear_bwd(ptr(dY), ptr(Y), ptr(X), ptr(W), B, N, K, 0, 1, ptr(dX), ptr(dW), 0, ptr(db), st)),
                 ("rk_linear_bwd_dact", lambda: lib.rk_linear_bwd_dact(ptr(dY), ptr(Y), ptr(X), ptr(W), B, N, K, 0, 1, ptr(dX), ptr(dW), 0, ptr(db), ptr(Xa), st)),
                 ("dX only", lambda: lib.rk_linear_bwd(ptr(dY), ptr(Y), ptr(X), ptr(W), B, N, K, 0, 1, ptr(dX), None, 0, ptr(db), st)),
                 ("dW only", lambda: lib.rk_linear_bwd(ptr(dY), ptr(Y), ptr(X), ptr(W), B, N, K, 0, 1, None, ptr(dW), 0, ptr(db), st))]:
  for _ in range(20):
    check(fn())
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(200):
    fn()
  e1.record()
  torch.cuda.synchronize()
  print("%-20s %.2f us per call (pair = %d)" % (name, e0.elapsed_time(e1) * 5.0, PAIR))

"""Is a multi-millisecond stall of the GPU a property of this stack or of our step?  A plain torch
loop (a few small kernels, one synchronize per iteration, ~0.2 ms) timed for a few seconds: the
iterations that take more than 3 ms, and when.   python tools/probes/hiccup_probe.py [seconds]"""
import sys, time
import torch
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
x = torch.randn(1024, 1024, device="cuda")
y = torch.empty_like(x)
for _ in range(50):
  torch.mul(x, 1.0001, out=y)
torch.cuda.synchronize()
t0 = time.perf_counter()
ts = []
while time.perf_counter() - t0 < secs:
  a = time.perf_counter()
  for _ in range(20):
    torch.mul(x, 1.0001, out=y)
  torch.cuda.synchronize()
  ts.append((a - t0, time.perf_counter() - a))
import numpy as np
d = np.array([t[1] for t in ts]) * 1e3
print("%d iterations, median %.3f ms, p99 %.3f ms, max %.2f ms" % (len(d), np.median(d), np.percentile(d, 99), d.max()))
print("iterations > 3 ms (at s, ms):", [(round(t[0], 3), round(t[1] * 1e3, 1)) for t in ts if t[1] > 3e-3])

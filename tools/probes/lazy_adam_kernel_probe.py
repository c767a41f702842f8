"""rk_adam_multi's lazy sweep against its dense sweep IN ISOLATION at the C2 shape (two 20108 x 200 tables, item sets
drawn like the synthetic ML-20M matrix) or C3's (41140 items), hot (back to back) and cold (1 GB streamed in between):
    python tools/probes/lazy_adam_kernel_probe.py [period] [hot|cold|both] [c2|c3]"""
import ctypes
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from recoder_amd import _lib, synthetic
from recoder_amd._lib import RkAdamJob, RkReplay, check, ptr

period = int(sys.argv[1]) if len(sys.argv) > 1 else 16
lib = _lib.load()
dev = torch.device("cuda")
cfg = sys.argv[3] if len(sys.argv) > 3 else "c2"
m = synthetic.ml20m_like() if cfg == "c2" else synthetic.msd_like(n_users=200000)
N, h, B = m.shape[1], 200, 500
rng = np.random.RandomState(0)
order = rng.permutation(m.shape[0])
n_steps = 48
sets = [np.unique(m[order[s * B:(s + 1) * B]].indices) for s in range(n_steps + 1)]
poss, grads = [], []
for it in sets:
  pos = np.full(N, -1, np.int32)
  pos[it] = np.arange(len(it), dtype=np.int32)
  poss.append(torch.from_numpy(pos).to(dev))
  grads.append(torch.randn(len(it), h, device=dev) * 0.01)
th = torch.zeros(n_steps * 8, dtype=torch.float32)
for i in range(n_steps):
  assert lib.rk_adam_consts(1e-3, 0.9, 0.999, 1e-8, 2e-5, 10 + i, 1, 8, th.data_ptr() + i * 32) == 0
table = th.to(dev)
cursor = torch.zeros(2, dtype=torch.int64, device=dev)
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
junk = torch.empty(1 << 28, dtype=torch.float32, device=dev)


import os
use_list = os.environ.get("PROBE_LIST") == "1"
lists = []
for t in range(n_steps):
  c = t % period
  lo, hi = c * N // period, (c + 1) * N // period
  need = (poss[t] >= 0) | (poss[t + 1] >= 0)
  need[lo:hi] = False
  idx = torch.nonzero(need).flatten().to(torch.int32)
  idx = idx[torch.randperm(idx.numel(), device=dev)] if os.environ.get("PROBE_LIST_SHUFFLE") == "1" else idx
  lists.append((idx.contiguous(), torch.tensor([idx.numel()], dtype=torch.int32, device=dev)))


def run(lazy, cold):
  tabs = [[torch.randn(N, h, device=dev) * 0.1, torch.zeros(N, h, device=dev), torch.full((N, h), 1e-4, device=dev),
           torch.zeros(N, dtype=torch.int32, device=dev)] for _ in range(2)]
  times, rows = [], []
  for t in range(n_steps):
    cursor.copy_(torch.tensor([t, 0], dtype=torch.int64))
    ctx = RkReplay()
    ctx.cursor, ctx.off, ctx.B = ptr(cursor), 0, 1
    ctx.users_base, ctx.adam_table, ctx.tab_stride = ptr(cursor), ptr(table), 1
    jobs = (RkAdamJob * 2)()
    for k, (p, mm, v, st) in enumerate(tabs):
      j = jobs[k]
      a = j.par
      a.p, a.m, a.v = ptr(p), ptr(mm), ptr(v)
      a.lr, a.beta1, a.beta2, a.eps, a.weight_decay, a.step, a.sparse = 1e-3, 0.9, 0.999, 1e-8, 2e-5, 1, 0
      j.n_rows, j.h, j.g, j.g_parts, j.pos = N, h, ptr(grads[t]), 1, ptr(poss[t])
      if lazy:
        j.lazy_stamp, j.lazy_pos_next, j.lazy_period = ptr(st), ptr(poss[t + 1]), period
        if use_list:            # (PROBE_LIST=1: the need-set outside the chunk as a list, built here with torch)
          j.lazy_need_list, j.lazy_need_count = ptr(lists[t][0]), ptr(lists[t][1])
    if cold:
      junk.fill_(float(t))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    lib.rk_replay_set(ctypes.byref(ctx))
    e0.record()
    check(lib.rk_adam_multi(jobs, 2, None, 0, 1.0, None, stream), "rk_adam_multi")
    e1.record()
    lib.rk_replay_set(None)
    torch.cuda.synchronize()
    times.append(e0.elapsed_time(e1) * 1000)
    if lazy:
      rows.append(int((tabs[0][3] == t + 1).sum()))
  return np.median(times[20:]), (np.mean(rows[20:]) / N if rows else 1.0)


modes = {'hot': (False,), 'cold': (True,)}.get(sys.argv[2] if len(sys.argv) > 2 else 'both', (False, True))
for cold in modes:
  d, _ = run(False, cold)
  z, frac = run(True, cold)
  print("%s: dense %.1f us, lazy(period %d) %.1f us (%.1f %% of the rows swept per step)" %
        ("cold" if cold else "hot ", d, period, z, 100 * frac))

"""rk_adam_multi in isolation with the jobs of the C2 step (two dense [N, h] tables with compact
gradient rows through pos, the decoder side as 2 K slabs, the two biases, the loss reduction),
timed with HIP events; `streamed` MB of other data are read between two sweeps (what a step touches
evicts the tables from the Infinity Cache).  Compare tools/probes/hbm_rw_rate.hip."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from recoder_amd import _lib, synthetic
from recoder_amd._lib import RkAdamJob, check, ptr
from recoder_amd.device import Block, DeviceCSR, current_stream

lib = _lib.load()
dev = torch.device("cuda")
f = dict(dtype=torch.float32, device=dev)
csr = synthetic.ml20m_like(seed=0, n_users=20000)
dcsr = DeviceCSR(csr)
N, h, B = csr.shape[1], 200, 500
blk = Block(B, int(np.sort(dcsr.degrees)[-B:].sum()), N, dev)
blk.collate(dcsr, torch.arange(B, dtype=torch.int64, device=dev))
n_b = blk.counts_host()[0]
tabs = [[torch.randn(N, h, **f) * 0.05, torch.zeros(N, h, **f), torch.zeros(N, h, **f)] for _ in range(2)]
bias = [[torch.zeros(n, **f), torch.zeros(n, **f), torch.zeros(n, **f)] for n in (N, h)]
G_en = torch.randn(blk.n_cap * h, **f) * 1e-3
slabs = torch.randn(4 * blk.n_cap * h, **f) * 1e-3
gb_part = torch.randn(8 * blk.ld_cap, **f) * 1e-3
gb_en = torch.randn(h, **f)
n_part = lib.rk_loss_partials(B, blk.n_cap)
loss_part = torch.zeros(n_part, **f)
loss_out = torch.zeros(4, **f)
nslab = torch.tensor([2], dtype=torch.int32, device=dev)
other = torch.zeros(int(os.environ.get("STREAMED_MB", "170")) * (1 << 18), **f)
st = current_stream()


def job(p, m, v, n_rows, hh, g, pos=None, g_parts=1, g_stride=0, gparts_dev=None, gstride_dev=None):
  j = RkAdamJob()
  a = j.par
  a.p, a.m, a.v = ptr(p), ptr(m), ptr(v)
  a.lr, a.beta1, a.beta2, a.eps, a.weight_decay, a.step, a.sparse = 1e-3, 0.9, 0.999, 1e-8, 2e-5, 7, 0
  j.n_rows, j.h, j.g, j.g_parts, j.g_stride = n_rows, hh, ptr(g), g_parts, g_stride
  j.pos = ptr(pos) if pos is not None else None
  j.gparts_dev = ptr(gparts_dev) if gparts_dev is not None else None
  j.gstride_dev = ptr(gstride_dev) if gstride_dev is not None else None
  return j


def variants():
  full = [job(*tabs[0], N, h, G_en, pos=blk.pos),
          job(*tabs[1], N, h, slabs, pos=blk.pos, g_parts=4, g_stride=blk.n_cap * h, gparts_dev=nslab),
          job(*bias[0], N, 1, gb_part, pos=blk.pos, g_parts=8, gstride_dev=blk.counts[2:3]),
          job(*bias[1], 1, h, gb_en)]
  yield "the step's 4 jobs + loss, biases first", [full[2], full[3], full[0], full[1]], True
  yield "the step's 4 jobs + loss", full, True
  yield "the step's 4 jobs", full, False
  yield "two tables, decoder side as ONE gradient array", [full[0], job(*tabs[1], N, h, G_en, pos=blk.pos)], False
  yield "one table", [full[0]], False


def timeit(jobs, with_loss, reps=30):
  arr = (RkAdamJob * len(jobs))(*jobs)
  ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
  for a, b in ev:
    other.sum()                               # stream `other` through the caches
    a.record()
    check(lib.rk_adam_multi(arr, len(jobs), ptr(loss_part) if with_loss else None, n_part, 500.0,
                            ptr(loss_out) if with_loss else None, st), "rk_adam_multi")
    b.record()
  torch.cuda.synchronize()
  return float(np.median([a.elapsed_time(b) for a, b in ev[5:]])) * 1e3


for name, jobs, wl in variants():
  us = timeit(jobs, wl)
  tables = sum(1 for j in jobs if j.h == h and j.n_rows == N)
  mb = tables * N * h * 24 / 1e6
  print("%-52s %6.1f us   (%.0f MB of p / m / v traffic: %.2f TB/s on those alone)" % (name, us, mb, mb / us * 1e-6 * 1e6 / 1e6))


# ---- the regrouped tail: [decoder-table sweep || encoder backward] | [encoder-table sweep]
def pair_probe():
  dZ = torch.randn(B * h, **f) * 1e-3
  G2 = torch.zeros(blk.n_cap * h, **f)
  gb2 = torch.zeros(h * 8, **f)
  s2 = torch.cuda.Stream()
  full = [job(*tabs[0], N, h, G_en, pos=blk.pos),
          job(*tabs[1], N, h, slabs, pos=blk.pos, g_parts=4, g_stride=blk.n_cap * h, gparts_dev=nslab),
          job(*bias[0], N, 1, gb_part, pos=blk.pos, g_parts=8, gstride_dev=blk.counts[2:3]),
          job(*bias[1], 1, h, gb_en)]
  de = (RkAdamJob * 2)(full[2], full[1])
  en = (RkAdamJob * 2)(full[3], full[0])
  h2 = ctypes.c_void_p(s2.cuda_stream)

  def run(what):
    ts = []
    for _ in range(30):
      other.sum()
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      e0.record()
      if what in ("pair", "enc"):
        s2.wait_stream(torch.cuda.current_stream())
        check(lib.rk_ae_encode_bwd(blk.ref, 0, B, ptr(dZ), h, ptr(G2), 0, ptr(gb2), h2))
      if what in ("pair", "de"):
        check(lib.rk_adam_multi(de, 2, None, 0, 500.0, None, st))
      if what == "en":
        check(lib.rk_adam_multi(en, 2, ptr(loss_part), n_part, 500.0, ptr(loss_out), st))
      if what in ("pair", "enc"):
        torch.cuda.current_stream().wait_stream(s2)
      e1.record()
      e1.synchronize()
      ts.append(e0.elapsed_time(e1) * 1e3)
    return float(np.median(ts[5:]))
  for what, name in (("enc", "encoder backward alone (second stream, fork + join)"), ("de", "decoder-table + bias sweep alone"),
                     ("pair", "decoder-table sweep || encoder backward (two streams)"), ("en", "encoder-table + bias sweep + loss")):
    print("%-60s %6.1f us" % (name, run(what)))


pair_probe()

#!/usr/bin/env python
"""Per-queue kernel timeline of steady-state training steps from a rocprofv3
rocpd database:  python tools/rocpd_timeline.py results.db [step_index] [n_steps]
Times in us relative to the start of the step's ae_encode_fwd kernel."""
import re
import sqlite3
import sys


def short(name):
  name = name.replace("(anonymous namespace)::", "").replace("void ", "")
  m = re.match(r"gemm_kernel<([^>]*)>", name)
  if m:
    a = m.group(1).replace(" ", "").split(",")
    kind = {"1": "decode", "2": "decode", "3": "dz", "0": "store"}.get(a[6], a[6])
    if a[4] == "1":
      kind = "dw"
    return "gemm[%s %s]" % (kind, "x".join(a[:4]))
  return re.sub(r"\(.*", "", name)[:40]


def main(path, step=150, n=1):
  c = sqlite3.connect(path)
  rows = c.execute("select name, queue_id, start, end from kernels order by start").fetchall()
  starts = [i for i, r in enumerate(rows) if "ae_encode_fwd" in r[0]]
  step = min(step, len(starts) - n - 2)
  i0, i1 = starts[step], starts[step + n]
  t0 = rows[i0][2]
  t1 = rows[i1][2]
  print("step wall %.1f us" % ((t1 - t0) / 1e3 / n))
  sel = [r for r in rows if r[2] >= t0 - 60000 and r[2] < t1]
  queues = sorted({r[1] for r in sel})
  for q in queues:
    print("queue %s" % q)
    prev = None
    for name, qq, s, e in sel:
      if qq != q:
        continue
      gap = "" if prev is None else "  (gap %.1f)" % ((s - prev) / 1e3)
      print("  %8.1f -> %8.1f  %6.1f  %s%s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, short(name), gap))
      prev = e


if __name__ == "__main__":
  main(sys.argv[1], *(int(x) for x in sys.argv[2:]))

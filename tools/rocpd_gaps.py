#!/usr/bin/env python
"""Where the time between kernels goes: per-queue busy time, and the gaps of the training queue,
over a window of steady-state steps of a rocprofv3 rocpd database.
    python tools/rocpd_gaps.py results.db [first_step] [n_steps]"""
import re
import sqlite3
import sys
from collections import defaultdict


def short(name):
  name = name.replace("(anonymous namespace)::", "").replace("void ", "")
  return re.sub(r"\(.*", "", name)[:44]


def main(path, first=60, n=40):
  c = sqlite3.connect(path)
  rows = c.execute("select name, queue_id, start, end from kernels order by start").fetchall()
  starts = [i for i, r in enumerate(rows) if "ae_encode_fwd" in r[0]]
  first = min(first, len(starts) - n - 2)
  t0, t1 = rows[starts[first]][2], rows[starts[first + n]][2]
  q_main = rows[starts[first]][1]
  print("%d steps: %.1f us per step" % (n, (t1 - t0) / 1e3 / n))
  sel = [r for r in rows if t0 <= r[2] < t1]
  busy = defaultdict(float)
  per = defaultdict(lambda: [0, 0.0])
  for name, q, s, e in sel:
    busy[q] += (e - s) / 1e3
    per[(q, short(name))][0] += 1
    per[(q, short(name))][1] += (e - s) / 1e3
  for q in sorted(busy):
    print("queue %s%s: busy %.1f us per step" % (q, " (training)" if q == q_main else "", busy[q] / n))
    for (qq, nm), (cnt, tot) in sorted(per.items(), key=lambda kv: -kv[1][1]):
      if qq == q:
        print("     %-46s %6.2f per step  x %5.2f us" % (nm, cnt / n, tot / cnt))
  # gaps on the training queue, attributed to the kernel that FOLLOWS the gap
  gaps = defaultdict(lambda: [0, 0.0])
  prev = None
  for name, q, s, e in sel:
    if q != q_main:
      continue
    if prev is not None:
      g = (s - prev) / 1e3
      gaps[short(name)][0] += 1
      gaps[short(name)][1] += max(g, 0.0)
    prev = e
  print("gaps on the training queue (before the named kernel), us per step:")
  for nm, (cnt, tot) in sorted(gaps.items(), key=lambda kv: -kv[1][1]):
    print("     %-46s %6.2f  (avg %.2f over %d)" % (nm, tot / n, tot / max(cnt, 1), cnt))


if __name__ == "__main__":
  main(sys.argv[1], *(int(x) for x in sys.argv[2:]))

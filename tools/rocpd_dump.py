#!/usr/bin/env python
"""Plain kernel timeline of a rocprofv3 rocpd database (queue, start, duration, gap to the previous
kernel on the same queue), from the n-th ae_encode_fwd launch on:
    python tools/rocpd_dump.py results.db [first_step] [n_steps]"""
import re
import sqlite3
import sys


def short(name):
  name = name.replace("(anonymous namespace)::", "").replace("void ", "")
  return re.sub(r"\(.*", "", name)[:40]


def main(path, first=0, n=30):
  c = sqlite3.connect(path)
  rows = c.execute("select name, queue_id, start, end from kernels order by start").fetchall()
  starts = [i for i, r in enumerate(rows) if "ae_encode_fwd" in r[0]]
  i0 = starts[first] - 30 if first < len(starts) else 0
  i1 = starts[min(first + n, len(starts) - 1)]
  t0 = rows[max(i0, 0)][2]
  last = {}
  for name, q, s, e in rows[max(i0, 0):i1 + 8]:
    gap = (s - last[q]) / 1e3 if q in last else 0.0
    last[q] = e
    print("q%-3s %10.1f  dur %7.1f  gap %8.1f  %s" % (q, (s - t0) / 1e3, (e - s) / 1e3, gap, short(name)))


if __name__ == "__main__":
  main(sys.argv[1], *(int(x) for x in sys.argv[2:]))

#!/usr/bin/env python
"""Dump the per-kernel summary (the `--stats` view) of a rocprofv3 rocpd
database as a markdown table:  python tools/rocpd_stats.py results.db > out.md"""
import sqlite3
import sys


def main(path):
  c = sqlite3.connect(path)
  rows = c.execute("select name, total_calls, total_duration, average, percentage "
                   "from top_kernels order by total_duration desc").fetchall()
  print("| kernel | calls | total (us) | avg (us) | % |")
  print("|---|---:|---:|---:|---:|")
  for name, calls, tot, avg, pct in rows:
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    if len(name) > 110:
      name = name[:107] + "..."
    print("| `%s` | %d | %.1f | %.2f | %.2f |" % (name, calls, tot, avg, pct))


if __name__ == "__main__":
  main(sys.argv[1])

#!/usr/bin/env python
"""Per-kernel averages of the PMC counters in a rocprofv3 rocpd database.
    python tools/rocpd_pmc.py results.db [name-filter]"""
import sqlite3
import sys
from collections import defaultdict


def main(path, flt=None):
  c = sqlite3.connect(path)
  rows = c.execute("select kernel_name, counter_name, value, duration, vgpr_count, accum_vgpr_count, "
                   "lds_block_size, grid_size, workgroup_size, dispatch_id from counters_collection").fetchall()
  agg = defaultdict(lambda: defaultdict(list))
  meta = {}
  for name, cn, val, dur, vg, ag, lds, grid, wg, did in rows:
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    if flt and flt not in name:
      continue
    agg[name][cn].append(val)
    meta[name] = (vg, ag, lds, grid, wg)
    agg[name]["_dur_us"].append(dur / 1e3)
  for name in sorted(agg, key=lambda n: -sum(agg[n]["_dur_us"])):
    vg, ag, lds, grid, wg = meta[name]
    print("== %s" % name[:100])
    print("   vgpr %s agpr %s lds %s grid %s wg %s" % (vg, ag, lds, grid, wg))
    for cn in sorted(agg[name]):
      v = agg[name][cn]
      print("   %-28s avg %14.1f  (n=%d)" % (cn, sum(v) / len(v), len(v)))


if __name__ == "__main__":
  main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)

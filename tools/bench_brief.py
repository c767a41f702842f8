#!/usr/bin/env python
"""One line per bench.py JSON line on stdin: ms/step and the bracketed launch groups (us).
    python bench.py ... | python tools/bench_brief.py [label]"""
import json
import sys

label = sys.argv[1] if len(sys.argv) > 1 else ""
for line in sys.stdin:
  if not line.startswith("{"):
    continue
  j = json.loads(line)
  r = j.get("roofline", {})
  ks = r.get("kernels") or r.get("all") or []
  print(label, "%.5f ms/step" % j["ms_per_step"], "value %.0f" % j["value"],
        [(k["name"].replace("rk_", ""), round(k["avg_us"], 1)) for k in ks])

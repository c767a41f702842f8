#!/usr/bin/env python
"""How many torch threads serve the CPU baseline best on this host: bench.py's cpu_baseline (the
oracle = pinned PyTorch-CPU restatement of the reference op sequence, collation included) for a few
seconds at each thread count.    python tools/probes/cpu_threads_sweep.py [seconds per point]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
  secs = sys.argv[1] if len(sys.argv) > 1 else "6"
  import argparse
  import bench
  bench.ARGS = argparse.Namespace(cpu_seconds=float(secs), cpu_threads=16)
  cfg = bench.CONFIGS["c2"]
  csr = bench.make_csr(cfg)
  print("host cores: %d; C2 workload, %s s per point" % (os.cpu_count(), secs), flush=True)
  for t in (1, 2, 4, 8, 16, 32, 64, 128, 256):
    if t > os.cpu_count():
      break
    bench.ARGS.cpu_threads = t
    r = bench.cpu_baseline(cfg, csr, 100000, warmup=2)
    print("threads %3d: %8.0f users/s  (%s)" % (r["cores"], r["value"], r["sample"].split(" of ")[0]), flush=True)


if __name__ == "__main__":
  main()

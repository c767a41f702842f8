#!/usr/bin/env python
"""HBM traffic per launch of the hot-path entries from two rocprofv3 PMC passes
(FETCH_SIZE and WRITE_SIZE collected in SEPARATE runs, MI355X_MICROARCH.md §HBM):

    bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024

FETCH_SIZE is doubled: on gfx950 this rocprofv3 reports exactly half the bytes of a
wide (16 B/lane) coalesced read stream; calibrated here on the Adam sweep, whose
byte count is known exactly (reads 3 x N*h*4 + gradient rows, writes 3 x N*h*4).

    python tools/pmc_traffic.py fetch.db write.db > profiles/rNN_pmc_traffic.json
"""
import json
import sqlite3
import sys
from collections import defaultdict

ENTRY_KERNELS = {
  "rk_adam_multi": ["adam_multi_kernel"],
  "rk_decode_loss": ["decode_planes_kernel", "gemm_kernel<2, 2, 1, 2, 0, 0, 1"],
  "rk_decode_bwd_dz": ["dz_planes_kernel", "gemm_kernel<4, 1, 1, 4, 0, 1, 2", "splitk_reduce_kernel"],
  "rk_decode_bwd_dw": ["dw3_kernel", "dw_encbwd_kernel"],   # dW (csrc/dw3.hip), alone or || encoder backward
  "rk_ae_encode_bwd": ["ae_encode_bwd_cols_kernel", "ae_encode_bwd_kernel"],
  "rk_ae_encode_fwd": ["ae_encode_fwd_kernel"],
}


def avg(path, counter):
  c = sqlite3.connect(path)
  acc = defaultdict(list)
  for name, val in c.execute("select kernel_name, value from counters_collection where counter_name=?",
                             (counter,)):
    acc[name.replace("(anonymous namespace)::", "").replace("void ", "")].append(val)
  return {k: sum(v) / len(v) for k, v in acc.items()}


def main(fetch_db, write_db):
  f, w = avg(fetch_db, "FETCH_SIZE"), avg(write_db, "WRITE_SIZE")
  out = {}
  for entry, pats in ENTRY_KERNELS.items():
    fk = sum(v for k, v in f.items() if any(k.startswith(p) for p in pats))
    wk = sum(v for k, v in w.items() if any(k.startswith(p) for p in pats))
    out[entry] = dict(fetch_size_kb=fk, write_size_kb=wk, hbm_bytes_per_launch=(2 * fk + wk) * 1024)
  print(json.dumps({"workload": "bench.py c2 (B=500, h=200, n_b~7.8k)", "formula":
                    "(2*FETCH_SIZE + WRITE_SIZE)*1024, separate --pmc passes", "entries": out}, indent=1))


if __name__ == "__main__":
  main(sys.argv[1], sys.argv[2])

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

# bracketed entry of the step -> kernel-name prefixes (as rocprofv3 prints them, namespaces stripped) of every
# generation of kernels an entry can dispatch: round-4 default C2 step = fdec_kernel | splitk_reduce_kernel |
# pgemm's dw_encbwd_kernel; >= 1024 rows / h > 256 = pg::gemm_kernel<..> (by epilogue); rounds 1-3 behind them
ENTRY_KERNELS = {
  "rk_adam_multi": ["adam_multi_kernel"],
  "rk_decode_loss": ["fdec_kernel", "decode_planes_kernel", "pg::gemm_kernel", "gemm_kernel<2, 2, 1, 2, 0, 0, 1"],
  "rk_decode_bwd_dz": ["splitk_reduce_kernel", "dz_planes_kernel", "gemm_kernel<4, 1, 1, 4, 0, 1, 2"],
  "rk_decode_bwd_dw": ["dw_encbwd_kernel", "dw3_kernel", "dw_reduce_kernel"],   # dW alone or || encoder backward
  "rk_ae_encode_bwd": ["ae_encode_bwd_cols_kernel", "ae_encode_bwd_kernel"],
  "rk_ae_encode_fwd": ["ae_encode_fwd_kernel"],
}
# (pg::gemm_kernel instantiations are told apart by their epilogue: EpiLoss = the decode)
PG_ENTRY = {"EpiLoss": "rk_decode_loss", "EpiStats": "rk_decode_loss", "EpiSlab": None}


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
  seen = set()
  for entry, pats in ENTRY_KERNELS.items():
    def mine(k):
      if k.startswith("pg::gemm_kernel"):
        return entry == "rk_decode_loss" and ("EpiLoss" in k or "EpiStats" in k)
      return any(k.startswith(p) for p in pats)
    ks = sorted(k for k in set(f) | set(w) if mine(k))
    seen.update(ks)
    fk = sum(f.get(k, 0.0) for k in ks)
    wk = sum(w.get(k, 0.0) for k in ks)
    out[entry] = dict(kernels=[k[:80] for k in ks], fetch_size_kb=fk, write_size_kb=wk,
                      hbm_bytes_per_launch=(2 * fk + wk) * 1024)
  # every other kernel of the run, by name (nothing the step dispatches may go unreported)
  other = {k[:80]: dict(fetch_size_kb=f.get(k, 0.0), write_size_kb=w.get(k, 0.0),
                        hbm_bytes_per_launch=(2 * f.get(k, 0.0) + w.get(k, 0.0)) * 1024)
           for k in sorted(set(f) | set(w)) if k not in seen and not k.startswith("at::") and "rocclr" not in k}
  print(json.dumps({"workload": "bench.py c2 (B=500, h=200, n_b~7.8k)", "formula":
                    "(2*FETCH_SIZE + WRITE_SIZE)*1024, separate --pmc passes (MI355X_MICROARCH.md, HBM: FETCH_SIZE "
                    "reports half the bytes of a wide coalesced read on gfx950)", "entries": out,
                    "other_kernels": other}, indent=1))


if __name__ == "__main__":
  main(sys.argv[1], sys.argv[2])

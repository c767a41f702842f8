#!/usr/bin/env python
"""Benchmark of the hot path: training users/sec of the mini-batch
negative-sampling loop (BASELINE.json metric), one JSON line on rank 0.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2]

A "step" is one pass of the hot path over one batch of B users: on-device
collation (rk_collate) + encoder SpMM + decoder GEMM with fused loss +
backward + fused Adam -- everything Recoder._train does per iteration
(reference model.py:383-404).  The CSR is resident in HBM when the timed region
starts.  N > 1 is launched by torch.distributed.run, one rank per GPU; users are
sharded over the ranks (weak scaling: B users per rank per step).

Also reported:
  roofline     -- the dominant kernel entry (by time per step, chosen from the
                  warm-up profile), timed with HIP events inside the timed
                  region, against its algorithmic flops/bytes.
  cpu_baseline -- oracle/recoder_oracle.py (the pinned CPU restatement of the
                  reference op sequence, PyTorch-CPU eager) timed on this host's
                  cores on a bounded sample of the same workload (rank 0, N=1).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
PEAK_MFMA_F32_TF = 157.3       # v_mfma_f32_32x32x2_f32 dense peak
PEAK_MFMA_F16_TF = 2500.0      # v_mfma_f32_32x32x16_f16 dense peak (MI355X_MICROARCH.md, no sparsity)
# decode / dZ run on the f16 pipe with every fp32 operand split into an fp16 pair: 3 MFMA flops per
# algorithmic flop, so the ceiling for ALGORITHMIC flops of those two entries is a third of the pipe
H3_ENTRIES = ("rk_decode_loss", "rk_decode_bwd_dz")
GEMM_F32 = os.environ.get("RK_GEMM_PREC", "")[:1].lower() == "f"

CONFIGS = {
  # C2 of BASELINE.json: ML-20M autoencoder, hidden [200], MSE, 1 x MI355X
  "c2": dict(workload="C2 ML-20M-like synthetic CSR 116677x20108 (lognormal degree mean 73, Zipf(1) "
                      "items, values 1.0, seed 0); DynamicAutoencoder hidden=[200] tanh noise 0.5, "
                      "MSE, dense Adam lr 1e-3 wd 2e-5, negative sampling",
             data="ml20m", kind="ae", hidden_layers=[200], activation_type="tanh", noise_prob=0.5,
             sparse=False, loss="mse", batch_size=500, lr=1e-3, weight_decay=2e-5),
  "c2s": dict(workload="C2 with sparse=True (SparseAdam on the two tables)",
              data="ml20m", kind="ae", hidden_layers=[200], activation_type="tanh", noise_prob=0.5,
              sparse=True, loss="mse", batch_size=500, lr=1e-3, weight_decay=2e-5),
  "small": dict(workload="smoke-size synthetic 5000x3000", data="small", kind="ae",
                hidden_layers=[200], activation_type="tanh", noise_prob=0.5, sparse=False,
                loss="mse", batch_size=500, lr=1e-3, weight_decay=2e-5),
}


def make_csr(cfg):
  from recoder_amd import synthetic
  if cfg["data"] == "ml20m":
    return synthetic.ml20m_like(seed=0)
  if cfg["data"] == "small":
    return synthetic.lognormal_zipf(5000, 3000, 40, seed=0)
  raise ValueError(cfg["data"])


def algorithmic_work(entry, B, h0, n_b, nnz, n_items, cfg_sparse=False):
  """(bound, work per launch, unit) of one C-ABI entry."""
  gemm = 2.0 * B * h0 * n_b
  if entry in ("rk_decode_loss", "rk_decode_bwd_dz", "rk_decode_bwd_dw"):
    return "mfma", gemm / 1e12, "TFLOP/s"          # algorithmic flops of the contraction
  if entry == "rk_ae_encode_fwd":
    return "hbm", (nnz * (h0 * 4 + 12) + B * h0 * 4) / 1e9, "GB/s"
  if entry == "rk_ae_encode_bwd":
    return "hbm", (nnz * (h0 * 4 + 8) + n_b * h0 * 4) / 1e9, "GB/s"
  if entry == "rk_adam_table":
    # both tables + the bias table call this; dominated by the [n_items,h0] sweeps:
    # p, m, v read + written (24 B/elem) + gradient rows + pos
    return "hbm", (n_items * h0 * 24 + n_b * h0 * 4 + n_items * 4) / 1e9, "GB/s"
  if entry == "rk_adam_multi":
    # ONE launch for every update of the step: two [n_items,h0] dense-Adam sweeps (p, m, v
    # read + written = 24 B/elem, + compact gradient rows + pos), the decoder bias table
    # (24 B/elem + pos + 8 row-tile partials per sampled item), the encoder bias, the loss
    if cfg_sparse:
      return "hbm", (2 * n_b * h0 * 28 + n_items * 28 + n_b * 32) / 1e9, "GB/s"
    return "hbm", (2 * (n_items * h0 * 24 + n_b * h0 * 4 + n_items * 4)
                   + n_items * 28 + n_b * 32 + h0 * 28) / 1e9, "GB/s"
  if entry == "rk_adam_rows":
    return "hbm", (n_b * h0 * 28) / 1e9, "GB/s"
  return "hbm", 0.0, "GB/s"


def cpu_baseline(cfg, csr, steps, warmup=4):
  """The oracle (CPU restatement of the reference op sequence) on this host."""
  from oracle import recoder_oracle as orc
  # eager PyTorch-CPU on B x n_b matrices stops scaling (and oversubscribes) far
  # below a big host's core count: 256 threads ran 50x slower than 8
  torch.set_num_threads(min(os.cpu_count(), int(os.environ.get("RK_CPU_THREADS", "16"))))
  max_seconds = float(os.environ.get("RK_CPU_SECONDS", "20"))
  B = cfg["batch_size"]
  torch.manual_seed(0)
  st = orc.init_ae_state(csr.shape[1], cfg["hidden_layers"])
  o = orc.OracleRecoder("ae", st, hidden_layers=cfg["hidden_layers"],
                        activation_type=cfg["activation_type"], noise_prob=cfg["noise_prob"],
                        sparse=cfg["sparse"], loss=cfg["loss"], lr=cfg["lr"],
                        weight_decay=cfg["weight_decay"])
  rng = np.random.RandomState(1)
  order = rng.permutation(csr.shape[0])
  t0 = None
  done = 0
  for i in range(warmup + steps):
    if i == warmup:
      t0 = time.perf_counter()
    users = order[(i * B) % (len(order) - B):][:B]
    b = orc.collate(orc.extract_rows(csr, users), users, B, True)[0]   # collation included
    keep = (rng.random_sample(b.indices.shape[1]) >= cfg["noise_prob"]).astype(np.uint8)
    o.train_step(b, None, keep, None)
    if i >= warmup:
      done += B
      if time.perf_counter() - t0 > max_seconds:
        break
  dt = time.perf_counter() - t0
  return dict(value=done / dt, unit="users/s", cores=torch.get_num_threads(), kind="port",
              sample="%d steps of B=%d of the same workload after %d warm-up steps (%.1f s, bounded "
                     "to ~%.0f s); oracle/recoder_oracle.py = pinned PyTorch-CPU restatement of the "
                     "reference op sequence incl. collation"
                     % (done // B, B, warmup, dt, max_seconds))


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--gpus", type=int, default=1)
  ap.add_argument("--steps", type=int, default=200)
  ap.add_argument("--warmup", type=int, default=20)
  ap.add_argument("--config", default="c2")
  ap.add_argument("--cpu-steps", type=int, default=1000)   # bounded by RK_CPU_SECONDS (20 s)
  ap.add_argument("--no-cpu-baseline", action="store_true")
  # diagnostics only (the JSON line is marked invalid): train every step on the first
  # collated block, i.e. without the collation the real loop overlaps on its side stream
  ap.add_argument("--diag-reuse-block", action="store_true")
  # with --diag-reuse-block: N trivial kernels per step on a side stream (what does a second busy
  # queue cost the training chain, independent of what its kernels do?)
  ap.add_argument("--diag-side-noise", type=int, default=0)
  # diagnostics only (marked invalid): run rank 0's share of an N-way item-parallel step on
  # this one GPU with the collectives replaced by no-ops -> the per-rank compute time at N GPUs
  ap.add_argument("--diag-virtual-world", type=int, default=0)
  args = ap.parse_args()
  cfg = CONFIGS[args.config]

  world = int(os.environ.get("WORLD_SIZE", "1"))
  rank = int(os.environ.get("RANK", "0"))
  local_rank = int(os.environ.get("LOCAL_RANK", "0"))
  assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
  torch.cuda.set_device(local_rank)
  device = torch.device("cuda", local_rank)
  dp = ip = None
  # Multi-GPU (one process per GPU, RCCL): the ITEM dimension is sharded by default
  # (parallel.ItemParallel: every rank runs the whole global batch of world*B users against
  # its items, two [world*B, h] all-reduces per step); RK_PARALLEL=users shards the users
  # instead (gradient-row all-reduce).  RK_FORCE_DP=1 exercises either with a 1-rank group.
  force_dp = os.environ.get("RK_FORCE_DP") == "1"
  if world > 1 or force_dp:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    from recoder_amd.parallel import DataParallel, ItemParallel, shard_range
    if os.environ.get("RK_PARALLEL", "items") == "users":
      dp = DataParallel()
    else:
      ip = ItemParallel()

  from recoder_amd._lib import ENTRY
  from recoder_amd.device import Block, DeviceCSR
  from recoder_amd.engine import FusedEngine
  from recoder_amd.model import Recoder, _top_sum
  from recoder_amd.nn import DynamicAutoencoder

  if args.diag_virtual_world:
    from recoder_amd.parallel import ItemParallel
    ip = ItemParallel(rank=0, world=args.diag_virtual_world, allreduce_fn=lambda t: t,
                      allgather_fn=lambda t: [t] * args.diag_virtual_world)
  vworld = args.diag_virtual_world or world
  csr_full = make_csr(cfg)
  if dp is not None and world > 1:
    lo, hi = shard_range(csr_full.shape[0], rank, world)
    csr = csr_full[lo:hi]
  elif ip is not None:
    csr = ip.shard_csr(csr_full)            # every user, this rank's item columns
  else:
    csr = csr_full
  n_users, n_items = csr.shape
  B = cfg["batch_size"] * (vworld if ip is not None else 1)   # rows this rank runs per step
  h0 = cfg["hidden_layers"][0]

  torch.manual_seed(0)       # same initial weights on every rank
  model = DynamicAutoencoder(hidden_layers=cfg["hidden_layers"], activation_type=cfg["activation_type"],
                             noise_prob=cfg["noise_prob"], sparse=cfg["sparse"])
  rec = Recoder(model=model, use_cuda=True, optimizer_type="adam", loss=cfg["loss"],
                num_items=n_items, num_users=csr_full.shape[0])
  from recoder_amd.data import RecommendationDataset
  ds = RecommendationDataset(csr)
  rec._Recoder__init_training(ds, cfg["lr"], cfg["weight_decay"])
  model.train()
  eng = rec._engine()
  if dp is not None:
    dp.attach(eng)
  if ip is not None:
    ip.user_norm_dev = torch.from_numpy(ItemParallel.user_norms(csr_full)).to(device)
    ip.prepare(device)
    eng.item_parallel = ip
  dcsr = ds.device_csr()                      # CSR resident in HBM before timing
  # the union item set over all ranks can exceed one rank's nnz bound
  from recoder_amd.device import CollatePrefetcher
  nnz_bound = _top_sum(dcsr.degrees, B)
  pf = CollatePrefetcher(
      # capacity of the item set: the union over all ranks under data parallelism; at most the
      # owned items under item parallelism (grids are capacity-sized: keep it tight)
      lambda: Block(B, nnz_bound, n_items, device, negative_sampling=True,
                    n_cap=(nnz_bound * world if dp is not None else
                           min(nnz_bound, -(-n_items // vworld)) if ip is not None else nnz_bound)),
      dcsr, device, collate_fn=(dp.collate if dp is not None else None), group=rec.prefetch_group)
  G = pf.group

  total = args.warmup + args.steps
  rng = np.random.RandomState(100 + (rank if dp is not None else 0))   # item shards share the order
  order = np.concatenate([rng.permutation(n_users) for _ in range((total * B) // n_users + 1)])
  order = order[: total * B].astype(np.int64)
  order_dev = torch.from_numpy(order).to(device)
  loss_buf = torch.zeros(total, dtype=torch.float32, device=device)
  global_rows = B * world if dp is not None else B      # users all ranks consume per step

  def users_of(i):
    return order_dev[i * B:(i + 1) * B]

  def chunk_users(c):
    return [users_of(i) for i in range(c * G, min(total, (c + 1) * G))]

  cur = {}

  def step(i):
    # the collation of the next G steps runs on the prefetcher's side stream while these train
    if args.diag_reuse_block and i >= G:
      if args.diag_side_noise:
        if "noise" not in cur:
          cur["noise"] = (torch.cuda.Stream(device=device), torch.zeros(64, device=device))
        with torch.cuda.stream(cur["noise"][0]):
          for _ in range(args.diag_side_noise):
            cur["noise"][1].add_(1.0)
      eng.train_step(cur["blks"][0], 0, B, out=loss_buf[i:i + 1])
      return
    c = i // G
    if i % G == 0:
      if (c + 1) * G < total:
        pf.submit((c + 1) % 2, chunk_users(c + 1))
      cur["blks"] = pf.acquire(c % 2)
    eng.train_step(cur["blks"][i % G], 0, B, out=loss_buf[i:i + 1],
                   global_rows=global_rows if dp is not None else None)
    if i % G == G - 1 or i == total - 1:
      pf.release(c % 2)

  pf.submit(0, chunk_users(0))

  # ---- warm-up (untimed).  The first half runs the per-entry Python sequencing
  # with every C-ABI entry bracketed by HIP events -> picks the dominant entry;
  # the rest runs the production path (rk_ae_train_step) ----
  step(0)
  torch.cuda.synchronize()
  half = max(2, args.warmup // 2)
  if ip is None:
    eng.use_c_step = False
    eng.lib.enabled = True
    for i in range(1, half):
      step(i)
    prof = eng.lib.summary()
    eng.lib.reset()
    eng.lib.enabled = False
    timed = {k: v for k, v in prof.items() if k in ENTRY}
    # the production path issues all Adam updates as one rk_adam_multi launch
    upd = ("rk_adam_table", "rk_adam_dense", "rk_adam_rows")
    n_prof = max(1, half - 1)
    if any(k in prof for k in upd):     # (the per-entry sequencing batches them into rk_adam_multi too)
      extra = sum(prof[k][0] * prof[k][1] for k in upd if k in prof)
      have = timed.get("rk_adam_multi", (0, 0.0))
      timed["rk_adam_multi"] = (n_prof, (have[0] * have[1] + extra) / n_prof)
    dominant = max(timed, key=lambda k: timed[k][0] * timed[k][1]) if timed else "rk_decode_loss"
  else:
    # item shards only run the one-call step; its Adam sweep covers 1/world of the tables, the
    # decoder contraction is the largest kernel
    half, prof, dominant = 1, {}, "rk_decode_loss"
  only = dominant
  c_path = True                 # rk_ae_train_step handles single-GPU and data-parallel steps
  eng.use_c_step = True
  eng.time_entry = only         # the C driver brackets this entry on its launch stream
  for i in range(half, args.warmup):
    step(i)
  torch.cuda.synchronize()
  eng._c_time_idx = 0

  multi = (dp is not None or ip is not None) and not args.diag_virtual_world
  if multi:
    import torch.distributed as dist
    dist.barrier()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for i in range(args.warmup, total):
    step(i)
  t_enqueue = time.perf_counter() - t0      # host time to enqueue the timed steps
  torch.cuda.synchronize()
  if multi:
    dist.barrier()
  dt = time.perf_counter() - t0
  if multi:
    t = torch.tensor([dt], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
  if ip is not None:
    ip.allreduce_sum(loss_buf)             # every rank holds its items' share of the loss

  losses = loss_buf.cpu().numpy()
  assert np.all(np.isfinite(losses)), "non-finite loss"
  value = args.steps * global_rows / dt     # users consumed by all ranks per second

  if rank == 0:
    # per-step n_b / nnz of the timed steps (host recomputation, outside the timing)
    nbs, nnzs = [], []
    for i in range(args.warmup, min(total, args.warmup + 50)):
      rows = csr[order[i * B:(i + 1) * B]]
      nnzs.append(rows.nnz)
      nbs.append(len(np.unique(rows.indices)))
    n_b, nnz = float(np.mean(nbs)), float(np.mean(nnzs))
    tms = eng.timed_entry_ms()          # sampled launches of the timed region
    calls_per_step = 1.0
    ms_raw = float(np.mean(tms)) if tms else float("nan")
    # an event pair with NOTHING between it reads a few us (the second record's own
    # marker): measured here on the same stream and subtracted, so that the figure is the
    # kernel's duration -- it then agrees with the rocprofv3 kernel-trace average
    ev_over = eng.event_pair_overhead_ms()
    ms = ms_raw - ev_over
    bound, work, unit = algorithmic_work(only, B, h0, n_b, nnz, n_items, bool(cfg["sparse"]))
    achieved = work / (ms * 1e-3) if ms == ms and ms > 0 else float("nan")
    if bound != "mfma":
      peak = PEAK_HBM_GBS
    elif only in H3_ENTRIES and not GEMM_F32:
      peak = PEAK_MFMA_F16_TF / 3.0
    else:
      peak = PEAK_MFMA_F32_TF
    # HBM bytes per launch from the committed rocprofv3 PMC passes of this same workload
    # (profiles/r*_pmc_traffic.json, tools/pmc_traffic.py); null for other configs
    traffic = None
    if args.config == "c2":
      import glob
      files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
      if files:
        ent = json.load(open(files[-1]))["entries"].get(only)
        if ent:
          traffic = ent["hbm_bytes_per_launch"]
    roofline = dict(bound=bound, achieved=achieved, peak=peak, unit=unit, frac=achieved / peak,
                    traffic=traffic, kernel=only, avg_launch_ms=ms, event_pair_overhead_ms=ev_over,
                    avg_launch_ms_uncorrected=ms_raw, calls_per_step=calls_per_step,
                    warmup_profile_ms={k: round(v[0] * v[1] / max(1, half - 1), 4)
                                       for k, v in sorted(prof.items())})
    out = {
      "metric": "train_users_per_sec", "value": value, "unit": "users/s",
      "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
      "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
      # fp32 storage, accumulation and element-wise math everywhere; decode / dZ multiply fp16
      # hi+lo pairs of the fp32 operands on the f16 MFMA (3 products, ~2^-22 relative), dW fp32 MFMA
      "vs_baseline": None, "dtype": "f32" if GEMM_F32 else "f32 (decode/dZ: split-fp16 pairs, f32 acc)",
      "data": ("INVALID (diagnostic: no collation)" if args.diag_reuse_block else
               "INVALID (diagnostic: one rank's share, no collectives)" if args.diag_virtual_world
               else "synthetic"),
      "config": {"workload": cfg["workload"], "batch_size_per_gpu": cfg["batch_size"],
                 "global_batch": global_rows,
                 "parallelism": ("items%d (item-sharded tables, all users on every rank)" % world
                                 if ip is not None else "dp%d" % world),
                 "avg_sampled_items": n_b, "avg_nnz_per_batch": nnz,
                 "first_loss": float(losses[0]), "last_loss": float(losses[-1]),
                 "host_enqueue_ms_per_step": t_enqueue / args.steps * 1e3},
      "roofline": roofline,
    }
    if world == 1 and not force_dp and not args.no_cpu_baseline:
      out["cpu_baseline"] = cpu_baseline(cfg, csr_full, args.cpu_steps)
    print(json.dumps(out))
  if multi:
    dist.destroy_process_group()


if __name__ == "__main__":
  main()

"""Worker of tests/test_dp_two_process.py: one of two torch.distributed processes (gloo, both on GPU
0) training through Recoder.train under the real _setup_data_parallel.
    python -m torch.distributed.run --nproc-per-node 2 tests/dp_two_process_worker.py <out prefix> <case>"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def make_case(case):
  """(csr, model factory, loss, weight decay, B, epochs): shared with the test."""
  from recoder_amd.nn import DynamicAutoencoder, MatrixFactorization
  from tests.test_hip_parity import synth_csr
  csr = synth_csr(1200, 1500, 20, seed=29)
  if case == "ae_dense":
    mk = lambda: DynamicAutoencoder([64], activation_type="tanh", noise_prob=0.0, sparse=False)
    return csr, mk, "mse", 2e-5, 150, 2
  if case == "ae_items":          # RK_PARALLEL=items (set by the test)
    mk = lambda: DynamicAutoencoder([64], activation_type="tanh", noise_prob=0.0, sparse=False)
    return csr, mk, "mse", 2e-5, 150, 2
  if case == "ae_sparse":         # SparseAdam tables: owned-row Adam under users-DP
    mk = lambda: DynamicAutoencoder([48], activation_type="tanh", noise_prob=0.0, sparse=True)
    return csr, mk, "mse", 0.0, 150, 2
  if case == "mf_sparse":
    mk = lambda: MatrixFactorization(32, activation_type="none", sparse=True)
    return csr, mk, "logistic", 0.0, 150, 2
  raise ValueError(case)


def shard_orders(n, world, per, seed=6):
  from recoder_amd.parallel import shard_range
  rng = np.random.RandomState(seed)
  return [rng.permutation(shard_range(n, r, world)[1] - shard_range(n, r, world)[0])[:per].astype(np.int64)
          for r in range(world)]


def main():
  prefix, case = sys.argv[1], sys.argv[2]
  rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
  os.environ["RK_COMM"] = "torch"            # two ranks on one GPU: no RCCL communicator
  os.environ.setdefault("RK_DP_OWNED", "1")  # (auto, the default, would price the two and pick the replicated update)
  os.environ.setdefault("RK_DP_ZERO", "1")   # (ae_dense: the sharded dense Adam over gloo; auto = from 8 ranks)
  torch.cuda.set_device(0)
  dist.init_process_group("gloo", rank=rank, world_size=world)
  from recoder_amd.data import RecommendationDataset
  from recoder_amd.model import Recoder
  csr, mk, loss, wd, B, epochs = make_case(case)
  n = csr.shape[0]
  torch.manual_seed(19 + rank)               # (different seeds: rank 0's weights are broadcast)
  model = mk()
  rec = Recoder(model=model, use_cuda=True, optimizer_type="adam", loss=loss)
  if case == "ae_items":
    # item parallel: every rank sees ALL users of the global batch, in the same order
    per = n // world
    so = shard_orders(n, world, per)
    from recoder_amd.parallel import shard_range
    lo1 = shard_range(n, 1, world)[0]
    glob = []
    for off in range(0, per, B):
      glob += list(so[0][off:off + B]) + list(lo1 + so[1][off:off + B])
    order = np.asarray(glob, dtype=np.int64)
    rec.user_order_hook = lambda epoch, n_: order
    bs = B                                   # per rank; the global batch is world * B
  else:
    so = shard_orders(n, world, n // world)
    rec.user_order_hook = lambda epoch, n_: so[rank]
    bs = B
  rec.train(RecommendationDataset(csr), batch_size=bs, lr=1e-3, weight_decay=wd, num_epochs=epochs,
            negative_sampling=True, model_checkpoint_prefix=prefix + "_ckpt", checkpoint_freq=epochs)
  out = {"losses": np.concatenate(rec.loss_history), "owned": np.asarray(int(bool(getattr(rec._engine(), "owned_rows", False))))}
  for k, v in model.named_parameters():
    out["p/" + k] = v.detach().cpu().numpy()
  np.savez(prefix + "_rank%d.npz" % rank, **out)
  dist.barrier()
  dist.destroy_process_group()


if __name__ == "__main__":
  main()

"""Two REAL torch.distributed processes (gloo collectives, both on GPU 0 -- the GPU boxes have one
GPU) through Recoder.train and the product's own _setup_data_parallel: initial-weight broadcast,
dataset sharding, two-phase collation with the MAX-reduced stamps, gradient all-reduce, owner
publication of rank-private rows, rank-0 checkpoint.  Both ranks must end with the parameters of the
single-process run with batch_size = 2 * B over the interleaved user order."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("case", ["ae_dense", "ae_items", "mf_sparse", "ae_sparse"])
def test_two_processes_equal_single_process(case, tmp_path):
  from recoder_amd.data import RecommendationDataset
  from recoder_amd.model import Recoder
  from recoder_amd.parallel import shard_range
  from tests.dp_two_process_worker import make_case, shard_orders
  from tests.test_hip_parity import close_stats
  prefix = str(tmp_path / "dp")
  env = dict(os.environ, MASTER_ADDR="127.0.0.1", PYTHONPATH=ROOT)
  env["RK_PARALLEL"] = "items" if case == "ae_items" else "users"
  cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
         "--master-addr", "127.0.0.1", "--master-port", "29591",
         os.path.join(ROOT, "tests", "dp_two_process_worker.py"), prefix, case]
  r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
  assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]

  csr, mk, loss, wd, B, epochs = make_case(case)
  n, world = csr.shape[0], 2
  per = n // world
  so = shard_orders(n, world, per)
  lo1 = shard_range(n, 1, world)[0]
  glob = []
  for off in range(0, per, B):
    glob += list(so[0][off:off + B]) + list(lo1 + so[1][off:off + B])
  glob = np.asarray(glob, dtype=np.int64)
  torch.manual_seed(19)                      # rank 0's seed: its weights are what every rank starts from
  model = mk()
  rec = Recoder(model=model, use_cuda=True, optimizer_type="adam", loss=loss)
  rec.user_order_hook = lambda epoch, n_: glob
  rec.train(RecommendationDataset(csr), batch_size=world * B, lr=1e-3, weight_decay=wd, num_epochs=epochs,
            negative_sampling=True)
  base_l = np.concatenate(rec.loss_history)
  got = [np.load(prefix + "_rank%d.npz" % k) for k in range(world)]
  for g in got:
    assert int(g["owned"]) == (1 if case in ("mf_sparse", "ae_sparse") else 0)      # (owned-row Adam ran)
    if case == "ae_items":
      pass      # (each rank logs its own shard's share of the loss; the sum is checked below)
    else:
      assert np.allclose(g["losses"], base_l, rtol=2e-5, atol=0), (g["losses"][:3], base_l[:3])
    for k, v in model.named_parameters():
      frac, mx, scale = close_stats(g["p/" + k], v.detach().cpu().numpy(), 1e-4, 2e-6)
      assert frac < 2e-3, (case, k, frac, mx, scale)
  if case == "ae_items":
    tot = got[0]["losses"] + got[1]["losses"] if not np.allclose(got[0]["losses"], base_l, rtol=2e-5) \
        else got[0]["losses"]
    assert np.allclose(tot, base_l, rtol=2e-5, atol=0), (tot[:3], base_l[:3])
  # the checkpoint: written once (rank 0), loadable, equal to the trained parameters
  ck = prefix + "_ckpt_epoch_%d.model" % epochs
  assert os.path.exists(ck)
  st = torch.load(ck, map_location="cpu", weights_only=False)
  for k, v in model.named_parameters():
    if k in st["model"]:
      frac, mx, scale = close_stats(st["model"][k].numpy(), v.detach().cpu().numpy(), 1e-4, 2e-6)
      assert frac < 2e-3, (case, "ckpt", k, frac)

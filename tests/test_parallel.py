"""CPU, 2 processes, gloo: the data-parallel formulation of recoder_amd/parallel.py
(shard users, union item set through all-reduce(MAX) of the stamp array, SUM
all-reduce of the compact gradient rows + dense gradients + loss, identical Adam on
every replica) equals the single-process run with batch_size = N * B_local."""
import os
import socket
import sys

import numpy as np
import pytest
import scipy.sparse as sp
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
  s = socket.socket()
  s.bind(("127.0.0.1", 0))
  p = s.getsockname()[1]
  s.close()
  return p


def _csr(n_users, n_items, seed):
  rng = np.random.RandomState(seed)
  m = sp.random(n_users, n_items, density=0.08, random_state=rng, format="csr", dtype=np.float32)
  m.data[:] = 1.0
  m.sort_indices()
  return m


def _worker(rank, world, port, out):
  sys.path.insert(0, ROOT)
  os.environ["MASTER_ADDR"] = "127.0.0.1"
  os.environ["MASTER_PORT"] = str(port)
  dist.init_process_group("gloo", rank=rank, world_size=world)
  torch.set_num_threads(1)
  from oracle import recoder_oracle as orc
  from recoder_amd.parallel import DataParallel, shard_range
  # the product class (users sharded; on CPU tensors its collectives go through torch.distributed)
  dp = DataParallel().prepare(torch.device("cpu"))
  assert dp.world == world and dp.rank == rank and not dp.direct

  n_users, n_items, B, h = 64, 90, 8, 12
  csr = _csr(n_users, n_items, 5)
  torch.manual_seed(3)
  st0 = orc.init_ae_state(n_items, [h])
  o = orc.OracleRecoder("ae", st0, hidden_layers=[h], activation_type="tanh", loss="mse",
                        lr=1e-2, weight_decay=1e-4)
  lo, hi = shard_range(n_users, rank, world)
  shard = csr[lo:hi]
  mark = torch.zeros(n_items, dtype=torch.int32)
  losses = []
  for step in range(3):
    users = np.arange(step * B, (step + 1) * B)          # local row ids of this rank's shard
    rows = shard[users]
    stamp = step + 1
    mark[torch.from_numpy(np.unique(rows.indices).astype(np.int64))] = stamp   # phase 1
    dp.union_marks(mark)                                                   # all-reduce MAX
    items = np.nonzero(mark.numpy() == stamp)[0].astype(np.int64)         # phase 2 (same on all ranks)
    pos = np.full(n_items, -1, dtype=np.int64)
    pos[items] = np.arange(len(items))
    coo = rows.tocoo()
    batch = orc.Batch(users=users + lo, items=items,
                      indices=np.stack([coo.row.astype(np.int64), pos[coo.col]]),
                      values=coo.data.astype(np.float32), size=(B, len(items)))
    o.optimizer.zero_grad()
    out_, t = o.forward(batch)
    loss = o._loss(out_, t) / torch.FloatTensor([B * world])             # global normalisation
    loss.backward()
    # compact rows of the two tables + everything else, summed over the ranks
    idx = torch.from_numpy(items)
    views, tables = [], []
    for name, p in o.params.items():
      if name in (orc.AE_EN_W, orc.AE_DE_W):
        r = p.grad[idx].contiguous()
        tables.append((p, r))
        views.append(r)
      elif name == orc.AE_DE_B:
        r = p.grad[idx].contiguous()
        tables.append((p, r))
        views.append(r)
      else:
        views.append(p.grad)
    lt = loss.detach().clone()
    views.append(lt)
    dp.reduce(views)                     # ONE exchange per step: every gradient + the loss
    for p, r in tables:
      p.grad.zero_()
      p.grad[idx] = r
    o.optimizer.step()
    losses.append(float(lt))
  if rank == 0:
    torch.save({"losses": losses, "state": {k: v.detach() for k, v in o.params.items()}}, out)
  dist.barrier()
  dist.destroy_process_group()


def test_two_rank_data_parallel_equals_single_process(tmp_path):
  from oracle import recoder_oracle as orc
  from recoder_amd.parallel import shard_range
  world = 2
  out = str(tmp_path / "dp.pt")
  mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
  got = torch.load(out, weights_only=False)

  n_users, n_items, B, h = 64, 90, 8, 12
  csr = _csr(n_users, n_items, 5)
  torch.manual_seed(3)
  st0 = orc.init_ae_state(n_items, [h])
  o = orc.OracleRecoder("ae", st0, hidden_layers=[h], activation_type="tanh", loss="mse",
                        lr=1e-2, weight_decay=1e-4)
  ref_losses = []
  for step in range(3):
    users = np.concatenate([np.arange(step * B, (step + 1) * B) + shard_range(n_users, r, world)[0]
                            for r in range(world)])
    b = orc.collate(csr[users], users, B * world, True)[0]
    ref_losses.append(o.train_step(b))
  assert np.allclose(got["losses"], ref_losses, rtol=1e-5, atol=0)
  for k, v in o.state().items():
    assert torch.allclose(got["state"][k], v, rtol=1e-4, atol=1e-6), k


def _mf_worker(rank, world, port, out):
  sys.path.insert(0, ROOT)
  os.environ["MASTER_ADDR"] = "127.0.0.1"
  os.environ["MASTER_PORT"] = str(port)
  dist.init_process_group("gloo", rank=rank, world_size=world)
  torch.set_num_threads(1)
  from oracle import recoder_oracle as orc
  from recoder_amd.parallel import allreduce_sum, shard_range, sync_owned_rows, union_marks

  n_users, n_items, B, d = 64, 90, 8, 10
  csr = _csr(n_users, n_items, 6)
  torch.manual_seed(4)
  st0 = orc.init_mf_state(n_items, n_users, d)
  o = orc.OracleRecoder("mf", st0, activation_type="tanh", loss="logistic", lr=1e-2,
                        weight_decay=1e-4)
  lo, hi = shard_range(n_users, rank, world)
  shard = csr[lo:hi]
  mark = torch.zeros(n_items, dtype=torch.int32)
  losses = []
  for step in range(3):
    users = np.arange(step * B, (step + 1) * B)
    rows = shard[users]
    stamp = step + 1
    mark[torch.from_numpy(np.unique(rows.indices).astype(np.int64))] = stamp
    union_marks(mark)
    items = np.nonzero(mark.numpy() == stamp)[0].astype(np.int64)
    pos = np.full(n_items, -1, dtype=np.int64)
    pos[items] = np.arange(len(items))
    coo = rows.tocoo()
    batch = orc.Batch(users=users + lo, items=items,
                      indices=np.stack([coo.row.astype(np.int64), pos[coo.col]]),
                      values=coo.data.astype(np.float32), size=(B, len(items)))
    o.optimizer.zero_grad()
    out_, t = o.forward(batch)
    loss = o._loss(out_, t) / torch.FloatTensor([B * world])
    loss.backward()
    # item rows + gathered bias + loss are summed over the ranks; user rows stay private
    idx = torch.from_numpy(items)
    tables = []
    for name in ("item_embedding_layer.weight", "bias"):
      p = o.params[name]
      tables.append((p, p.grad[idx].contiguous()))
    lt = loss.detach().clone()
    allreduce_sum([r for _, r in tables] + [lt], small_threshold=64)
    for p, r in tables:
      p.grad.zero_()
      p.grad[idx] = r
    o.optimizer.step()
    losses.append(float(lt))
  # owners publish their user rows (parameters and Adam moments)
  w = o.params["user_embedding_layer.weight"]
  st = o.optimizer.state[w]
  sync_owned_rows([w.data, st["exp_avg"], st["exp_avg_sq"]], n_users)
  if rank == 1:
    torch.save({"losses": losses, "state": {k: v.detach() for k, v in o.params.items()},
                "m": st["exp_avg"].clone(), "v": st["exp_avg_sq"].clone()}, out)
  dist.barrier()
  dist.destroy_process_group()


def test_two_rank_matrix_factorization_private_user_rows(tmp_path):
  """MF under data parallelism: item table + bias all-reduced, user rows rank-private and
  published by their owners at the end == single process with batch_size = N * B."""
  from oracle import recoder_oracle as orc
  from recoder_amd.parallel import shard_range
  world = 2
  out = str(tmp_path / "dp_mf.pt")
  mp.spawn(_mf_worker, args=(world, _free_port(), out), nprocs=world, join=True)
  got = torch.load(out, weights_only=False)
  n_users, n_items, B, d = 64, 90, 8, 10
  csr = _csr(n_users, n_items, 6)
  torch.manual_seed(4)
  st0 = orc.init_mf_state(n_items, n_users, d)
  o = orc.OracleRecoder("mf", st0, activation_type="tanh", loss="logistic", lr=1e-2,
                        weight_decay=1e-4)
  ref_losses = []
  for step in range(3):
    users = np.concatenate([np.arange(step * B, (step + 1) * B) + shard_range(n_users, r, world)[0]
                            for r in range(world)])
    b = orc.collate(csr[users], users, B * world, True)[0]
    ref_losses.append(o.train_step(b))
  assert np.allclose(got["losses"], ref_losses, rtol=1e-5, atol=0)
  for k, v in o.state().items():
    assert torch.allclose(got["state"][k], v, rtol=1e-4, atol=1e-6), k
  w = o.params["user_embedding_layer.weight"]
  assert torch.allclose(got["m"], o.optimizer.state[w]["exp_avg"], rtol=1e-4, atol=1e-7)
  assert torch.allclose(got["v"], o.optimizer.state[w]["exp_avg_sq"], rtol=1e-4, atol=1e-9)


def _ip_worker(rank, world, port, out):
  """Item-parallel formulation (parallel.ItemParallel) with real gloo collectives: the
  arithmetic of the three step segments restated with torch autograd on CPU tensors."""
  sys.path.insert(0, ROOT)
  os.environ["MASTER_ADDR"] = "127.0.0.1"
  os.environ["MASTER_PORT"] = str(port)
  dist.init_process_group("gloo", rank=rank, world_size=world)
  torch.set_num_threads(1)
  from oracle import recoder_oracle as orc
  from recoder_amd.parallel import ItemParallel

  n_users, n_items, S, h = 64, 91, 16, 12          # n_items not a multiple of the world size
  csr = _csr(n_users, n_items, 7)
  torch.manual_seed(5)
  st0 = orc.init_ae_state(n_items, [h])
  o = orc.OracleRecoder("ae", st0, hidden_layers=[h], activation_type="tanh", loss="mse",
                        lr=1e-2, weight_decay=1e-4)
  P = o.params
  ip = ItemParallel()
  shard = ip.shard_csr(csr)
  assert np.all(shard.indices % world == rank)
  norms = torch.from_numpy(ItemParallel.user_norms(csr))
  losses = []
  for step in range(3):
    users = np.arange(step * S, (step + 1) * S)          # the SAME global batch on every rank
    rows = shard[users]
    items = np.unique(rows.indices).astype(np.int64)     # owned items only
    idx = torch.from_numpy(items)
    X = torch.from_numpy(np.asarray(rows[:, items].todense(), dtype=np.float32))
    Xn = X / norms[users].clamp_min(1e-12)[:, None]
    o.optimizer.zero_grad()
    # segment 1: partial encoder sums over the local items, all-reduced
    Zp = Xn @ P[orc.AE_EN_W][idx]
    Z0 = ip.allreduce_sum(Zp.detach().clone()).requires_grad_()
    # segment 2: finish the layer, decode + loss on the local items, partial dLoss/dZ0
    z = torch.tanh(Z0 + P[orc.AE_EN_B])
    out_ = z @ P[orc.AE_DE_W][idx].t() + P[orc.AE_DE_B][idx]
    loss = ((out_ - X) ** 2).sum() / S
    loss.backward()
    dZ = ip.allreduce_sum(Z0.grad.clone())
    # segment 3: encoder-bias gradient from the FULL dZ0, encoder rows of the local items
    P[orc.AE_EN_B].grad = dZ.sum(0)
    Zp.backward(dZ)
    o.optimizer.step()        # non-owned rows only see the decay: overwritten by their owners below
    losses.append(float(ip.allreduce_sum(loss.detach().clone())))
  tensors = []
  for name in (orc.AE_EN_W, orc.AE_DE_W, orc.AE_DE_B):
    st = o.optimizer.state[P[name]]
    tensors += [P[name].data, st["exp_avg"], st["exp_avg_sq"]]
  ip.sync_owned(tensors, n_items)
  if rank == 1:
    torch.save({"losses": losses, "state": {k: v.detach() for k, v in P.items()},
                "m_en": o.optimizer.state[P[orc.AE_EN_W]]["exp_avg"].clone()}, out)
  dist.barrier()
  dist.destroy_process_group()


def test_two_rank_item_parallel_equals_single_process(tmp_path):
  from oracle import recoder_oracle as orc
  world = 2
  out = str(tmp_path / "ip.pt")
  mp.spawn(_ip_worker, args=(world, _free_port(), out), nprocs=world, join=True)
  got = torch.load(out, weights_only=False)
  n_users, n_items, S, h = 64, 91, 16, 12
  csr = _csr(n_users, n_items, 7)
  torch.manual_seed(5)
  st0 = orc.init_ae_state(n_items, [h])
  o = orc.OracleRecoder("ae", st0, hidden_layers=[h], activation_type="tanh", loss="mse",
                        lr=1e-2, weight_decay=1e-4)
  ref_losses = []
  for step in range(3):
    users = np.arange(step * S, (step + 1) * S)
    b = orc.collate(csr[users], users, S, True)[0]
    ref_losses.append(o.train_step(b))
  assert np.allclose(got["losses"], ref_losses, rtol=1e-5, atol=0)
  for k, v in o.state().items():
    assert torch.allclose(got["state"][k], v, rtol=1e-4, atol=1e-6), k
  assert torch.allclose(got["m_en"], o.optimizer.state[o.params[orc.AE_EN_W]]["exp_avg"],
                        rtol=1e-4, atol=1e-7)


def test_shard_range_partitions():
  from recoder_amd.parallel import shard_range
  for n, w in [(10, 3), (116677, 8), (7, 8), (64, 2)]:
    edges = [shard_range(n, r, w) for r in range(w)]
    assert edges[0][0] == 0 and edges[-1][1] == n
    assert all(edges[i][1] == edges[i + 1][0] for i in range(w - 1))
    sizes = [b - a for a, b in edges]
    assert max(sizes) - min(sizes) <= 1


def _owned_worker(rank, world, port, out):
  """The owned-row exchange primitives of parallel.DataParallel over gloo (CPU tensors): the partial
  gradient rows of every rank's item range arrive at their owner, the owners' updated rows at everyone."""
  sys.path.insert(0, ROOT)
  os.environ["MASTER_ADDR"] = "127.0.0.1"
  os.environ["MASTER_PORT"] = str(port)
  dist.init_process_group("gloo", rank=rank, world_size=world)
  torch.set_num_threads(1)
  from recoder_amd.parallel import DataParallel
  dp = DataParallel().prepare(torch.device("cpu"))
  n_items, h = 500, 6
  rng = np.random.RandomState(11)
  freq = rng.zipf(1.5, n_items).clip(max=300)
  dp.set_owner_bounds(DataParallel.balanced_bounds(freq, 1000, 64 * world, world))
  items = torch.from_numpy(np.sort(rng.choice(n_items, 173, replace=False)).astype(np.int32))
  n_b = len(items)
  offs = dp.owned_offsets(items, n_b)
  # segment q holds exactly the items of rank q's id range
  b = dp.owner_bounds
  for q in range(world):
    seg = items[offs[q]:offs[q + 1]].numpy()
    assert ((seg >= b[q]) & (seg < b[q + 1])).all()
  G = torch.from_numpy(np.random.RandomState(100 + rank).randn(n_b, h).astype(np.float32)).reshape(-1)
  R, cnt = dp.exchange_rows(G, offs, h)
  assert cnt == offs[rank + 1] - offs[rank]
  want = [torch.from_numpy(np.random.RandomState(100 + q).randn(n_b, h).astype(np.float32))[offs[rank]:offs[rank + 1]]
          for q in range(world)]
  for q in range(world):
    assert torch.equal(R[q * cnt * h:(q + 1) * cnt * h].view(cnt, h), want[q])
  S = sum(want) * (rank + 2.0)                     # "updated rows" of this rank's segment
  T = dp.publish_rows(S, offs, h).view(n_b, h)
  for q in range(world):
    wq = sum(torch.from_numpy(np.random.RandomState(100 + r).randn(n_b, h).astype(np.float32))[offs[q]:offs[q + 1]]
             for r in range(world)) * (q + 2.0)
    assert torch.allclose(T[offs[q]:offs[q + 1]], wq)
  # moments of the owners' id ranges reach every replica
  m = torch.full((n_items, 2), float(rank + 1))
  dp.sync_owned_moments([m])
  for q in range(world):
    assert (m[int(b[q]):int(b[q + 1])] == q + 1).all()
  out[rank] = True
  dist.destroy_process_group()


def test_owned_row_exchange_primitives_two_ranks():
  world = 2
  port = _free_port()
  with mp.Manager() as mgr:
    out = mgr.dict()
    mp.spawn(_owned_worker, args=(world, port, out), nprocs=world, join=True)
    assert out.get(0) and out.get(1)


def _zero_worker(rank, world, port, out):
  """The sharded-dense-Adam (ZeRO-1) primitives of parallel.DataParallel over gloo (CPU tensors): the dense
  gradient layout reduce-scattered over equal row ranges (n_items not a multiple of the world size), an
  "update" of the owned rows, the all-gather of the updated rows in place, the moments' final sync."""
  sys.path.insert(0, ROOT)
  os.environ["MASTER_ADDR"] = "127.0.0.1"
  os.environ["MASTER_PORT"] = str(port)
  dist.init_process_group("gloo", rank=rank, world_size=world)
  torch.set_num_threads(1)
  from recoder_amd.parallel import DataParallel
  dp = DataParallel().prepare(torch.device("cpu"))
  n_items, h = 501, 4                      # (501 = 2 x 251 - 1: the last range is one row short)
  z = dp.setup_zero(n_items)
  b = dp.zero_bounds()
  assert b[0] == 0 and b[-1] == n_items and z["rows_pad"] == z["sh"] * world >= n_items
  assert (z["lo"], z["hi"]) == (b[rank], b[rank + 1])
  dense = lambda r: torch.from_numpy(np.random.RandomState(300 + r).randn(z["rows_pad"], h).astype(np.float32))
  D = dense(rank).reshape(-1).clone()
  shard = torch.zeros(z["sh"] * h)
  dp.zero_reduce_scatter(D, shard)
  want = sum(dense(q) for q in range(world))[rank * z["sh"]:(rank + 1) * z["sh"]]
  assert torch.allclose(shard.view(-1, h), want)
  # the owner "updates" its rows of a replicated table from its shard; everyone gets every owner's rows
  W = torch.arange(n_items * h, dtype=torch.float32).view(n_items, h).clone()
  W[z["lo"]:z["hi"]] += shard.view(-1, h)[:z["hi"] - z["lo"]]
  dp.zero_all_gather([W], h)
  full = torch.arange(n_items * h, dtype=torch.float32).view(n_items, h) + sum(dense(q) for q in range(world))[:n_items]
  assert torch.allclose(W, full)
  m = torch.full((n_items, 2), float(rank + 1))
  dp.sync_owned_moments([m], bounds=b)
  for q in range(world):
    assert (m[b[q]:b[q + 1]] == q + 1).all()
  out[rank] = True
  dist.destroy_process_group()


def test_zero_adam_primitives_two_ranks():
  world = 2
  port = _free_port()
  with mp.Manager() as mgr:
    out = mgr.dict()
    mp.spawn(_zero_worker, args=(world, port, out), nprocs=world, join=True)
    assert out.get(0) and out.get(1)


def test_balanced_owner_bounds():
  """Item-id ranges with equal EXPECTED union rows per step: monotone, covering, balanced on a Zipf
  catalogue where equal-width ranges are not."""
  from recoder_amd.parallel import DataParallel
  rng = np.random.RandomState(0)
  n_items, n_users, rows = 20000, 100000, 4000
  freq = np.sort(rng.zipf(1.3, n_items).clip(max=50000))[::-1]          # ids sorted by popularity
  b = DataParallel.balanced_bounds(freq, n_users, rows, 8)
  assert b[0] == 0 and b[-1] == n_items and (np.diff(b) >= 0).all()
  p = 1.0 - np.power(1.0 - np.minimum(freq / n_users, 1.0), rows)
  mass = np.array([p[b[r]:b[r + 1]].sum() for r in range(8)])
  assert mass.max() / mass.mean() < 1.1
  eq = np.array([p[r * n_items // 8:(r + 1) * n_items // 8].sum() for r in range(8)])
  assert eq.max() / eq.mean() > 2.0

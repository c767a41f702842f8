"""Lazy dense Adam (include/recoder_hip.h rk_adam_job_t.lazy_stamp, csrc/optim.hip table_sweep_lazy).

optim.Adam with a dense embedding gradient (reference model.py:135,398-399) updates EVERY row of a table every
step; the lazy sweep skips the rows that neither carry a gradient nor are read by the next step and catches them up
later by replaying their missed steps.  The bar: p, m and v come out BIT FOR BIT what the dense sweeps leave --
at the kernel (random item sets, constants changing per step, weight decay on / off) and through Recoder.train
(lr milestone, validation pass, checkpoint + resume, tails and step marks that cut groups).
"""
import ctypes

import numpy as np
import pytest
import torch

from tests.test_hip_parity import synth_csr

pytestmark = pytest.mark.gpu


def _consts_table(lib, steps, lrs, wd, first_step, stride=1):
  """[steps][stride] entries of 8 floats; entry i = Adam constants of step first_step + i with lr lrs[i]."""
  th = torch.zeros(steps * stride * 8, dtype=torch.float32)
  for i in range(steps):
    assert lib.rk_adam_consts(float(lrs[i]), 0.9, 0.999, 1e-8, wd, first_step + i, 1, stride * 8,
                              th.data_ptr() + i * stride * 8 * 4) == 0
  return th.cuda()


def _need_lists(lib, poss, N):
  """rk_lazy_need_lists over consecutive pos maps (as the graph stepper calls it behind a collation): per step the
  (list, count) device tensors, checked against numpy."""
  from recoder_amd._lib import RkBlock, check, ptr
  out = []
  for t0 in range(0, len(poss) - 1, 7):                     # (at most RK_COLLATE_MULTI = 8 blocks per call)
    chunk = poss[t0:t0 + 8]
    n = len(chunk)
    blks = [RkBlock() for _ in range(n)]
    for b, pos in zip(blks, chunk):
      b.n_items, b.n_chunks, b.pos = N, -(-N // 2048), ptr(pos)
    arr = (ctypes.POINTER(RkBlock) * n)(*[ctypes.pointer(b) for b in blks])
    lists = [torch.full((N,), -7, dtype=torch.int32, device="cuda") for _ in range(n - 1)]
    counts = [torch.zeros(1, dtype=torch.int32, device="cuda") for _ in range(n - 1)]
    lp = (ctypes.c_void_p * n)(*[x.data_ptr() for x in lists], None)
    cp = (ctypes.c_void_p * n)(*[x.data_ptr() for x in counts], None)
    check(lib.rk_lazy_need_lists(arr, n, lp, cp, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)),
          "rk_lazy_need_lists")
    for i in range(n - 1):
      want = np.nonzero((chunk[i].cpu().numpy() >= 0) | (chunk[i + 1].cpu().numpy() >= 0))[0]
      c = int(counts[i].item())
      assert c == len(want) and np.array_equal(lists[i][:c].cpu().numpy(), want)
      assert bool((lists[i][c:] == -7).all())               # (nothing written past the count)
      out.append((lists[i], counts[i]))
  return out


@pytest.mark.parametrize("use_list", [False, True])
@pytest.mark.parametrize("wd,period,h", [(2e-5, 16, 200), (0.0, 4, 64), (1e-3, 7, 512), (2e-5, 1, 32)])
def test_lazy_sweeps_equal_dense_sweeps_bit_for_bit(wd, period, h, use_list):
  from recoder_amd import _lib
  from recoder_amd._lib import RkAdamJob, RkReplay, check, ptr
  lib = _lib.load()
  dev = torch.device("cuda")
  rng = np.random.RandomState(int(period * 1000 + h))
  N, n_steps, base, first_adam_step = 3001, 41, 1000, 5
  lrs = np.where(np.arange(n_steps) < 20, 1e-3, 1e-4)           # (a milestone in the middle)
  table = _consts_table(lib, n_steps, lrs, wd, first_adam_step)
  # per step: a random item set (Zipf-ish: a hot head + a sparse tail), compact gradient rows, pos maps
  sets, poss, grads = [], [], []
  pop = 1.0 / np.arange(1, N + 1)
  for t in range(n_steps):
    n_b = rng.randint(50, 900)
    items = np.unique(rng.choice(N, size=n_b, p=pop / pop.sum()))
    pos = np.full(N, -1, np.int32)
    pos[items] = np.arange(len(items), dtype=np.int32)
    sets.append(items)
    poss.append(torch.from_numpy(pos).to(dev))
    grads.append(torch.randn(len(items), h, device=dev) * 0.1)
  need = _need_lists(lib, poss, N) if use_list else None    # (need[t]: the rows of steps t and t + 1)
  p0 = torch.randn(N, h, device=dev) * 0.3
  m0 = torch.randn(N, h, device=dev) * 0.01
  v0 = torch.rand(N, h, device=dev) * 1e-4
  cursor = torch.zeros(2, dtype=torch.int64, device=dev)
  stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

  def run(lazy, flush_at=()):
    p, m, v = p0.clone(), m0.clone(), v0.clone()
    stamp = torch.full((N,), base, dtype=torch.int32, device=dev)
    amax = torch.zeros(64, dtype=torch.int32, device=dev)
    skipped = 0
    snaps = {}
    for t in range(n_steps):
      cursor.copy_(torch.tensor([base + t, base], dtype=torch.int64))
      ctx = RkReplay()
      ctx.cursor, ctx.off, ctx.B = ptr(cursor), 0, 1
      ctx.users_base, ctx.adam_table, ctx.tab_stride = ptr(cursor), ptr(table), 1
      ctx.cursor_next, ctx.advance = None, 0
      j = RkAdamJob()
      a = j.par
      a.p, a.m, a.v = ptr(p), ptr(m), ptr(v)
      a.lr, a.beta1, a.beta2, a.eps, a.weight_decay = 1e-3, 0.9, 0.999, 1e-8, wd
      a.step, a.sparse = 1, 0                      # (replay: the slot + 1)
      j.n_rows, j.h, j.g, j.g_parts = N, h, ptr(grads[t]), 1
      j.pos = ptr(poss[t])
      j.amax_out = ptr(amax)
      if lazy:
        nxt = poss[t + 1] if t + 1 < n_steps and (t + 1) not in flush_at else None
        j.lazy_stamp, j.lazy_pos_next, j.lazy_period = ptr(stamp), ptr(nxt), period
        if need is not None and nxt is not None:           # (the sweep takes its rows from the list + the chunk)
          j.lazy_need_list, j.lazy_need_count = ptr(need[t][0]), ptr(need[t][1])
      lib.rk_replay_set(ctypes.byref(ctx))
      try:
        check(lib.rk_adam_multi(ctypes.byref(j), 1, None, 0, 1.0, None, stream), "rk_adam_multi")
      finally:
        lib.rk_replay_set(None)
      if lazy:
        st = stamp.cpu().numpy()
        assert st.max() == base + t + 1 and st.min() >= base + t + 1 - period
        skipped += int((st < base + t + 1).sum())
        # the rows the next step reads are up to date
        if t + 1 < n_steps:
          assert np.all(st[sets[t + 1]] == base + t + 1)
      if (t + 1) in flush_at:       # (lazy: that step had no next block, lazy_pos_next == NULL)
        snaps[t + 1] = (p.clone(), m.clone(), v.clone())
    if lazy:
      sl = (ctypes.c_int32 * 1)(0)
      j2 = RkAdamJob()
      j2.par.p, j2.par.m, j2.par.v = ptr(p), ptr(m), ptr(v)
      j2.n_rows, j2.h, j2.lazy_stamp, j2.lazy_period = N, h, ptr(stamp), 1
      j2.amax_out = ptr(amax)
      check(lib.rk_adam_lazy_flush(ctypes.byref(j2), 1, ptr(table), 1, sl, base + n_steps, base, stream),
            "rk_adam_lazy_flush")
      assert bool((stamp == base + n_steps).all())
      # a second flush finds nothing to do
      check(lib.rk_adam_lazy_flush(ctypes.byref(j2), 1, ptr(table), 1, sl, base + n_steps, base, stream),
            "rk_adam_lazy_flush")
    torch.cuda.synchronize()
    return p, m, v, amax, skipped, snaps

  flush_at = (9, 26)
  pd, md, vd, amax_d, _, snaps_d = run(False, flush_at)
  pl, ml, vl, amax_l, skipped, snaps_l = run(True, flush_at)
  if period > 1:
    assert skipped > N, "the lazy sweeps were expected to skip rows"
  else:
    assert skipped == 0                      # (period 1: the whole table is the round-robin chunk)
  for name, x, y in (("p", pd, pl), ("m", md, ml), ("v", vd, vl)):
    assert torch.equal(x, y), (name, float((x - y).abs().max()))
  for k in flush_at:                         # a step without a next block leaves every row up to date
    for x, y in zip(snaps_d[k], snaps_l[k]):
      assert torch.equal(x, y), k
  # the running bound of |p| (the decoder contractions' split scale): equal once everything is flushed
  assert int(amax_d.max()) == int(amax_l.max())


def _train_case(case):
  from recoder_amd.nn import DynamicAutoencoder
  if case == "ae_dense":
    return (lambda: DynamicAutoencoder([64], activation_type="tanh", noise_prob=0.3, sparse=False)), "mse", 2e-5
  if case == "ae_tied_bce":
    return (lambda: DynamicAutoencoder([32], activation_type="sigmoid", noise_prob=0.0, sparse=False,
                                       is_constrained=True)), "logistic", 1e-5
  if case == "ae_logloss":
    return (lambda: DynamicAutoencoder([32], activation_type="tanh", noise_prob=0.2, sparse=False)), "logloss", 0.0
  if case == "stack_logloss":
    return (lambda: DynamicAutoencoder([48, 24], activation_type="tanh", noise_prob=0.2, sparse=False)), "logloss", 2e-5
  if case == "stack_dropout_mse":
    return (lambda: DynamicAutoencoder([40, 24], activation_type="tanh", noise_prob=0.1, dropout_prob=0.2,
                                       sparse=False)), "mse", 1e-5
  raise KeyError(case)


@pytest.mark.parametrize("case", ["ae_dense", "ae_tied_bce", "ae_logloss", "stack_logloss", "stack_dropout_mse"])
def test_training_with_lazy_adam_is_bitwise_the_dense_sweep(case, monkeypatch, tmp_path):
  """Recoder.train on the graph path with the lazy sweeps (default) and with RK_ADAM_LAZY=0 (every row every step):
  identical losses, parameters and Adam moments -- over an lr milestone, a validation pass between epochs, step
  marks that cut groups, a ragged last batch, a checkpoint and a resume from it in a fresh trainer."""
  from recoder_amd.data import RecommendationDataset
  from recoder_amd.model import Recoder
  mk, loss, wd = _train_case(case)
  csr = synth_csr(1430, 2500, 14, seed=71)           # 1430 = 22 x 64 + 22: ragged tail; items >> a block's set
  val = synth_csr(300, 2500, 14, seed=72)
  order = lambda epoch, n: (np.random.RandomState(80 + epoch).permutation(n).astype(np.int64) if n == csr.shape[0]
                            else np.arange(n, dtype=np.int64))

  def run(period):
    monkeypatch.setenv("RK_ADAM_LAZY", str(period))
    kw = dict(batch_size=64, lr=1e-3, weight_decay=wd, negative_sampling=True, lr_milestones=[2])
    torch.manual_seed(17)
    model = mk()
    rec = Recoder(model=model, use_cuda=True, optimizer_type="adam", loss=loss)
    rec.user_order_hook = order
    seen = []
    # (22 whole batches + a ragged one per epoch, groups of 8: the mark at 16 ends a run() on a whole replayed group
    # WITH look-ahead -- rows stay behind, rk_adam_lazy_flush brings them up; the one at 30 cuts a group)
    rec.step_marks = {16: lambda: seen.append(16) or False, 30: lambda: seen.append(30) or False}
    prefix = str(tmp_path / ("lazy%s" % str(period).replace(",", "_")))
    rec.train(RecommendationDataset(csr), val_dataset=RecommendationDataset(val, val), num_epochs=3, eval_freq=1,
              model_checkpoint_prefix=prefix, checkpoint_freq=3, **kw)
    assert seen == [16, 30]
    gs = rec._graph_stepper
    lists = str(period).endswith(",list")
    period = int(str(period).split(",")[0])
    assert bool(gs.lazy) == (period > 0)
    assert not lists or (gs.need_lists and any(getattr(b, "_need_for", None) for b in gs.blocks[0] + gs.blocks[1]))
    if period > 0:
      assert gs.lazy_flushes >= 1                    # (a run() that ended on a whole group with look-ahead)
      for n in gs.lazy:                              # nothing is left behind
        assert bool((rec._engine().lazy_stamp(n) == gs.global_step).all())
    out = [np.concatenate(rec.loss_history), rec.last_epoch_summary["val_loss"],
           {k: v.detach().cpu().clone() for k, v in model.named_parameters()},
           {k: (int(s.step), s.m.detach().cpu().clone(), s.v.detach().cpu().clone())
            for k, s in rec._engine().states.items()}]
    torch.manual_seed(5)
    model2 = mk()
    rec2 = Recoder(model=model2, use_cuda=True, optimizer_type="adam", loss=loss)
    rec2.init_from_model_file(prefix + "_epoch_3.model")
    rec2.user_order_hook = order
    rec2.train(RecommendationDataset(csr), num_epochs=4, **kw)
    out.append(np.concatenate(rec2.loss_history))
    out.append({k: v.detach().cpu().clone() for k, v in model2.named_parameters()})
    return out

  d = run(0)
  z = run(16)
  s = run(3)                                         # (a short period: the round-robin chunk wraps many times)
  l = run("16,list")                                 # (the sweeps take their rows from rk_lazy_need_lists' lists)
  for o in (z, s, l):
    assert np.array_equal(d[0], o[0]), np.abs(d[0] - o[0]).max()
    assert d[1] == o[1]
    for k in d[2]:
      assert torch.equal(d[2][k], o[2][k]), k
    for k in d[3]:
      assert d[3][k][0] == o[3][k][0], k
      assert torch.equal(d[3][k][1], o[3][k][1]) and torch.equal(d[3][k][2], o[3][k][2]), k
    assert np.array_equal(d[4], o[4])
    for k in d[5]:
      assert torch.equal(d[5][k], o[5][k]), k


def test_lazy_adam_is_off_where_it_does_not_apply(monkeypatch):
  """SparseAdam tables and MatrixFactorization take the plain sweeps (nothing to skip / not built)."""
  from recoder_amd.data import RecommendationDataset
  from recoder_amd.model import Recoder
  from recoder_amd.nn import DynamicAutoencoder, MatrixFactorization
  csr = synth_csr(600, 800, 10, seed=73)
  for mk in (lambda: DynamicAutoencoder([32], activation_type="tanh", sparse=True),
             lambda: MatrixFactorization(16, activation_type="none", sparse=False)):
    torch.manual_seed(3)
    rec = Recoder(model=mk(), use_cuda=True, optimizer_type="adam", loss="mse")
    rec.train(RecommendationDataset(csr), batch_size=64, lr=1e-3, weight_decay=0.0, num_epochs=1, negative_sampling=True)
    assert rec._graph_stepper.lazy == []


def test_need_lists_follow_the_matrix(monkeypatch):
  """The sweeps' need lists (rk_lazy_need_lists) are built where a block's expected item set covers less than 30 %
  of the catalogue (graph.GraphStepper: most rows would be skipped), not where most rows are swept anyway;
  RK_ADAM_LAZY=<period>,list / ,scan force either."""
  from recoder_amd.data import RecommendationDataset
  from recoder_amd.model import Recoder
  from recoder_amd.nn import DynamicAutoencoder

  def stepper(csr, batch, env=None):
    if env is None:
      monkeypatch.delenv("RK_ADAM_LAZY", raising=False)
    else:
      monkeypatch.setenv("RK_ADAM_LAZY", env)
    torch.manual_seed(3)
    rec = Recoder(model=DynamicAutoencoder([32], activation_type="tanh", sparse=False), use_cuda=True,
                  optimizer_type="adam", loss="mse")
    rec.train(RecommendationDataset(csr), batch_size=batch, lr=1e-3, weight_decay=0.0, num_epochs=1, negative_sampling=True)
    return rec._graph_stepper

  wide = synth_csr(1200, 6000, 8, seed=75)       # 16 users x 8 items of 6 000: a block holds ~2 % of the catalogue
  gs = stepper(wide, 16)
  assert gs.lazy and gs.need_lists and gs.need_cover < 0.1
  assert any(getattr(b, "_need_for", None) for b in gs.blocks[0] + gs.blocks[1])
  narrow = synth_csr(1200, 300, 12, seed=76)     # 64 users x 12 items of 300: most of the catalogue in every block
  gs = stepper(narrow, 64)
  assert gs.lazy and not gs.need_lists and gs.need_cover > 0.5
  assert not stepper(wide, 16, "16,scan").need_lists
  assert stepper(narrow, 64, "16,list").need_lists

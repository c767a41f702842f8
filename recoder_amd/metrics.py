"""Ranking metrics + evaluator with the reference's API (recoder/metrics.py).

The metric arithmetic (metrics.py:9-45) is host-side numpy on the top-k lists
the GPU produces (``Recoder.recommend`` -> ``rk_topk_masked``); it is pinned by
the reference's own known-answer tests (tests/test_metrics.py:12-54, restated
in tests/test_host_logic.py here).
"""
import numpy as np

from .data import RecommendationDataLoader


def _hits(x, y, k):
  x = np.asarray(x)[:k]
  return x, np.isin(x, y, assume_unique=True).astype(int)


def average_precision(x, y, k, normalize=True):
  x, hit = _hits(x, y, k)
  precision = hit.cumsum() / (1 + np.arange(len(x)))
  normalization = min(k, len(y)) if normalize else len(y)
  return np.multiply(precision, hit).sum() / normalization


def recall(x, y, k, normalize=True):
  x, hit = _hits(x, y, k)
  normalization = min(k, len(y)) if normalize else len(y)
  return hit.sum() / normalization


def dcg(x, y, k):
  x, hit = _hits(x, y, k)
  return (hit / np.log2(2 + np.arange(len(x)))).sum()


def ndcg(x, y, k):
  return dcg(x, y, k) / dcg(y, y, k)


class Metric(object):
  """Base class for metrics (metrics.py:48-75)."""

  def __init__(self, metric_name):
    self.metric_name = metric_name

  def __str__(self):
    return self.metric_name

  def __hash__(self):
    return self.metric_name.__hash__()

  def evaluate(self, x, y):
    raise NotImplementedError


class AveragePrecision(Metric):
  def __init__(self, k, normalize=True):
    super().__init__(metric_name="AveragePrecision@{}".format(k))
    self.k = k
    self.normalize = normalize

  def evaluate(self, x, y):
    return average_precision(x, y, k=self.k, normalize=self.normalize)


class Recall(Metric):
  def __init__(self, k, normalize=True):
    super().__init__(metric_name="Recall@{}".format(k))
    self.k = k
    self.normalize = normalize

  def evaluate(self, x, y):
    return recall(x, y, k=self.k, normalize=self.normalize)


class NDCG(Metric):
  def __init__(self, k):
    super().__init__(metric_name="NDCG@{}".format(k))
    self.k = k

  def evaluate(self, x, y):
    return ndcg(x, y, k=self.k)


def batch_metrics(recommendations, target_csr, metrics):
  """The three built-in metrics for a whole batch of users at once: {metric: [value per user]},
  or None when it does not apply (a user-defined Metric, ragged recommendation lists) and the
  caller has to loop over the users.  Same arithmetic as average_precision / recall / ndcg above
  (metrics.py:9-45) on a [users, k] hit matrix -- the per-user Python loop made evaluation 100x
  slower than the scoring + top-k it evaluates."""
  if any(type(m) not in (AveragePrecision, Recall, NDCG) for m in metrics):
    return None
  try:
    recs = np.asarray(recommendations)
  except ValueError:
    return None
  if recs.dtype == object or recs.ndim != 2:
    return None
  recs = recs.astype(np.int64, copy=False)
  B, K = recs.shape
  tm = target_csr.tocsr()
  if tm.shape[0] < B:
    return None
  n_items = int(tm.shape[1])
  rows = np.repeat(np.arange(tm.shape[0], dtype=np.int64), np.diff(tm.indptr))
  nz = tm.data != 0
  ykeys = np.sort(rows[nz] * n_items + tm.indices[nz].astype(np.int64))
  ny = np.bincount(rows[nz], minlength=tm.shape[0])[:B].astype(np.int64)
  tkeys = (np.arange(B, dtype=np.int64)[:, None] * n_items + recs).ravel()
  if len(ykeys):
    pos = np.minimum(np.searchsorted(ykeys, tkeys), len(ykeys) - 1)
    hit = (ykeys[pos] == tkeys).reshape(B, K).astype(int)
  else:
    hit = np.zeros((B, K), dtype=int)
  out = {}
  with np.errstate(divide="ignore", invalid="ignore"):      # (users without targets: nan, as per user)
    for m in metrics:
      k = min(int(m.k), K)
      h = hit[:, :k]
      if isinstance(m, Recall):
        norm = np.minimum(m.k, ny) if m.normalize else ny
        out[m] = (h.sum(axis=1) / norm).tolist()
      elif isinstance(m, AveragePrecision):
        precision = h.cumsum(axis=1) / (1 + np.arange(k))
        norm = np.minimum(m.k, ny) if m.normalize else ny
        out[m] = (np.multiply(precision, h).sum(axis=1) / norm).tolist()
      else:
        w = np.log2(2 + np.arange(max(int(m.k), 1)))
        got = (h / w[:k]).sum(axis=1)
        ideal = np.concatenate([[0.0], np.cumsum(1.0 / w)])[np.minimum(m.k, ny)]
        out[m] = (got / ideal).tolist()
  return out


class RecommenderEvaluator(object):
  """Evaluates a recommender on a dataset with input/target interactions
  (metrics.py:135-232).  ``num_workers`` is accepted for compatibility; scoring
  and top-k run on the GPU, the per-user metric arithmetic on the host."""

  def __init__(self, recommender, metrics):
    self.recommender = recommender
    self.metrics = metrics

  def evaluate(self, eval_dataset, batch_size=1, num_users=None, num_workers=0):
    dataloader = RecommendationDataLoader(eval_dataset, batch_size=batch_size,
                                          collate_fn=lambda _: _)
    results = {metric: [] for metric in self.metrics}
    processed = 0
    for inp, target in dataloader:
      as_array = getattr(self.recommender, "recommend_array", None)
      recommendations = as_array(inp) if as_array is not None else self.recommender.recommend(inp)
      tm = target.interactions_matrix.tocsr()
      fast = batch_metrics(recommendations, tm, self.metrics)
      if fast is not None:
        for metric in self.metrics:
          results[metric].extend(fast[metric])
        recommendations = ()
      for i, x in enumerate(recommendations):
        lo, hi = tm.indptr[i], tm.indptr[i + 1]
        y = tm.indices[lo:hi][tm.data[lo:hi] != 0]
        for metric in self.metrics:
          results[metric].append(metric.evaluate(x, y))
      processed += len(target.users)
      if num_users is not None and processed >= num_users:
        break
    return results

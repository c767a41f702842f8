"""Ranking metrics + evaluator with the reference's API (recoder/metrics.py).

The metric arithmetic (metrics.py:9-45) is host-side numpy on the top-k lists
the GPU produces (``Recoder.recommend`` -> ``rk_topk_masked``); it is pinned by
the reference's own known-answer tests (tests/test_metrics.py:12-54, restated
in tests/test_metrics.py here).
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
      recommendations = self.recommender.recommend(inp)
      tm = target.interactions_matrix.tocsr()
      for i, x in enumerate(recommendations):
        lo, hi = tm.indptr[i], tm.indptr[i + 1]
        y = tm.indices[lo:hi][tm.data[lo:hi] != 0]
        for metric in self.metrics:
          results[metric].append(metric.evaluate(x, y))
      processed += len(target.users)
      if num_users is not None and processed >= num_users:
        break
    return results

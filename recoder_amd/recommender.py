"""Recommenders (recoder/recommender.py:7-25,104-118).  The Annoy-based
``SimilarityRecommender`` of the reference is a serving-time heuristic outside
the training hot path and is not provided."""


class Recommender(object):
  def recommend(self, users_hist):
    raise NotImplementedError


class InferenceRecommender(Recommender):
  """Recommends from the predictions of a ``Recoder`` (recommender.py:104-118)."""

  def __init__(self, model, num_recommendations):
    self.model = model
    self.num_recommendations = num_recommendations

  def recommend(self, users_hist):
    return self.model.recommend(users_hist, self.num_recommendations)

  def recommend_array(self, users_hist):
    """The same lists as one [users, k] integer array (metrics.RecommenderEvaluator's fast path)."""
    return self.model.recommend_array(users_hist, self.num_recommendations)

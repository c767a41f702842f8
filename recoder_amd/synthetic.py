"""Seeded synthetic user x item CSR generators for the benchmark configs of
SURVEY.md section 8(d) (the real ML-20M / MSD files are not available offline).

  ml20m_like : 116,677 x 20,108, lognormal degree (mu = ln 73 - 0.5, sigma = 1,
               clipped to [5, n_items/4]), Zipf(1) item popularity, values 1.0
  msd_like   : 471,355 x 41,140, lognormal mean 59, Zipf(1)
  uniform    : fixed-degree rows, uniform popularity (C5-style, scaled)
"""
import numpy as np
import scipy.sparse as sp


def _zipf_csr(n_users, n_items, degrees, rng, zipf_a=1.0):
  if zipf_a is None:
    cum = None
  else:
    pop = 1.0 / np.power(np.arange(1, n_items + 1, dtype=np.float64), zipf_a)
    cum = np.cumsum(pop / pop.sum())
  total = int(degrees.sum())
  rows = np.repeat(np.arange(n_users, dtype=np.int64), degrees)
  if cum is None:
    cols = rng.randint(0, n_items, size=total).astype(np.int64)
  else:
    cols = np.searchsorted(cum, rng.random_sample(total)).astype(np.int64)
    np.minimum(cols, n_items - 1, out=cols)
  m = sp.coo_matrix((np.ones(total, dtype=np.float32), (rows, cols)),
                    shape=(n_users, n_items)).tocsr()
  m.sum_duplicates()          # dedup: a user/item pair is one interaction
  m.data[:] = 1.0
  m.sort_indices()
  return m


def lognormal_zipf(n_users, n_items, mean_deg, seed, sigma=1.0, min_deg=5, zipf_a=1.0):
  rng = np.random.RandomState(seed)
  deg = rng.lognormal(np.log(mean_deg) - 0.5 * sigma * sigma, sigma, n_users)
  deg = np.clip(deg.astype(np.int64), min_deg, max(min_deg, n_items // 4))
  return _zipf_csr(n_users, n_items, deg, rng, zipf_a)


def ml20m_like(seed=0, n_users=116677, n_items=20108):
  return lognormal_zipf(n_users, n_items, 73, seed)


def msd_like(seed=1, n_users=471355, n_items=41140):
  return lognormal_zipf(n_users, n_items, 59, seed)


def uniform(n_users, n_items, degree, seed=3):
  rng = np.random.RandomState(seed)
  deg = np.full(n_users, degree, dtype=np.int64)
  return _zipf_csr(n_users, n_items, deg, rng, zipf_a=None)

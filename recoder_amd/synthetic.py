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


def device_csr(n_users, n_items, degree, seed=3, zipf_a=None, device=None, chunk_users=131072):
  """Seeded user x item matrix generated ON THE DEVICE, never materialised on the host (C5 of
  BASELINE.json: 10 M x 1 M at 0.01 % density is ~1e9 interactions; a data-parallel rank holds
  its 1.25 M-user shard).  Every user draws ``degree`` items -- uniform, or Zipf(zipf_a) through
  the inverse CDF -- duplicates inside a row are dropped, values are all 1.0 (implicit feedback).
  Returns a recoder_amd.device.DeviceCSR."""
  import torch
  from .device import DeviceCSR, require_gpu
  device = device or require_gpu()
  g = torch.Generator(device=device)
  g.manual_seed(int(seed))
  cum = None
  if zipf_a is not None:
    pop = 1.0 / torch.arange(1, n_items + 1, dtype=torch.float64, device=device).pow(zipf_a)
    cum = torch.cumsum(pop / pop.sum(), 0)
  counts, cols = [], []
  for lo in range(0, n_users, chunk_users):
    n = min(chunk_users, n_users - lo)
    if cum is None:
      it = torch.randint(0, n_items, (n, degree), generator=g, device=device, dtype=torch.int64)
    else:
      u = torch.rand((n, degree), generator=g, device=device, dtype=torch.float64)
      it = torch.searchsorted(cum, u).clamp_(max=n_items - 1)
    it, _ = torch.sort(it, dim=1)
    keep = torch.ones_like(it, dtype=torch.bool)
    keep[:, 1:] = it[:, 1:] != it[:, :-1]
    counts.append(keep.sum(dim=1))
    cols.append(it[keep].to(torch.int32))
  counts = torch.cat(counts)
  indptr = torch.zeros(n_users + 1, dtype=torch.int64, device=device)
  torch.cumsum(counts, 0, out=indptr[1:])
  return DeviceCSR.from_arrays((n_users, n_items), indptr, torch.cat(cols), None, device, check=False)

"""Input helpers with the reference's API (recoder/utils.py): DataFrame -> CSR
with id maps (utils.py:26-66), row normalisation, unzip.  Host-side, one-off."""
import numpy as np
from scipy.sparse import coo_matrix


def unzip(l):
  """Inverse of zip on a list (utils.py:5-12)."""
  return list(map(list, zip(*l)))


def normalize(x, axis=None):
  """x / ||x|| along axis (utils.py:15-23)."""
  return x / np.linalg.norm(x, axis=axis).reshape(-1, 1)


def dataframe_to_csr_matrix(dataframe, user_col, item_col, inter_col, item_id_map=None,
                            user_id_map=None):
  """(csr_matrix, item_id_map, user_id_map); ids are numbered in order of first
  appearance unless maps are given (utils.py:26-66)."""
  if user_id_map is None:
    user_id_map = {u: i for i, u in enumerate(dataframe[user_col].unique())}
  if item_id_map is None:
    item_id_map = {it: i for i, it in enumerate(dataframe[item_col].unique())}
  shape = (len(user_id_map), len(item_id_map))
  rows = dataframe[user_col].map(user_id_map)
  cols = dataframe[item_col].map(item_id_map)
  vals = dataframe[inter_col]
  return coo_matrix((vals, (rows, cols)), shape=shape).tocsr(), item_id_map, user_id_map

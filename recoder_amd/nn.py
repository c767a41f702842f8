"""Factorization models with the reference's public contract (recoder/nn.py).

``FactorizationModel`` (nn.py:12-65), ``DynamicAutoencoder`` (nn.py:68-253),
``LinearEmbedding`` (nn.py:256-280) and ``MatrixFactorization`` (nn.py:283-362)
keep their constructor arguments, the four-method model contract, the RNG
consumption order of ``init_model`` and -- because checkpoints are exchanged
with the reference (model.py:193-224) -- the exact ``state_dict`` key names,
including the name-mangled private sub-modules.

The modules are parameter containers: ``forward`` does not run torch ops, it
hands the tensors to the HIP kernels (recoder_amd.engine); training goes through
``Recoder`` which drives the fused step.
"""
import torch
from torch import nn

import torch.nn.functional as F

_ACTS = ("none", "tanh", "sigmoid", "relu", "selu", "elu")


def _activate(x, act):
  return x if act == "none" else getattr(torch, act)(x)


def _embedding_linear(table, bias, ids, x, input_based, sparse):
  """LinearEmbedding (nn.py:269-280) in torch ops: a linear layer whose weight is the
  row subset ``ids`` of an embedding table (all rows when ids is None)."""
  w = table.weight if ids is None else F.embedding(ids, table.weight, sparse=sparse)
  if input_based:
    return F.linear(x, w.t(), bias)
  b = bias if ids is None or bias is None else bias.index_select(0, ids)
  return F.linear(x, w, b)


class FactorizationModel(nn.Module):
  """Base class: subclasses implement the four methods below (nn.py:18-65)."""

  def init_model(self, num_items=None, num_users=None):
    raise NotImplementedError

  def model_params(self):
    raise NotImplementedError

  def load_model_params(self, model_params):
    raise NotImplementedError

  def forward(self, input, input_users=None, input_items=None, target_users=None,
              target_items=None):
    raise NotImplementedError


class LinearEmbedding(nn.Module):
  """A linear layer whose weight matrix is (a row subset of) an embedding
  table (nn.py:256-280).  Holds the bias; the table is shared."""

  def __init__(self, embedding_layer, input_based=True, bias=True):
    super().__init__()
    self.embedding_layer = embedding_layer
    self.input_based = input_based
    n, d = embedding_layer.num_embeddings, embedding_layer.embedding_dim
    self.in_features = n if input_based else d
    self.out_features = d if input_based else n
    self.bias = nn.Parameter(torch.zeros(self.out_features)) if bias else None


def _check_act(act):
  """The reference takes 'none' or the name of any ``torch.<name>`` function (nn.py:6-9).  The
  fused HIP kernels implement _ACTS; every other name trains through the generic torch-autograd
  path on the GPU (recoder_amd/generic.py) -- unknown names fail here, as early as possible."""
  if act not in _ACTS and not callable(getattr(torch, str(act), None)):
    raise AttributeError("module 'torch' has no attribute %r (activation_type)" % (act,))


def fused_supported(model):
  """True when the HIP kernels cover this model instance: a fused activation and 16-byte
  embedding rows (hidden_layers[0] / embedding_size a multiple of 4)."""
  h0 = model.hidden_layers[0] if isinstance(model, DynamicAutoencoder) else model.embedding_size
  return model.activation_type in _ACTS and h0 % 4 == 0


class DynamicAutoencoder(FactorizationModel):
  """Autoencoder over variable item subsets (nn.py:68-253).

  Args mirror the reference: hidden_layers, activation_type, is_constrained,
  dropout_prob, noise_prob, sparse.
  """

  def __init__(self, hidden_layers=None, activation_type="tanh", is_constrained=False,
               dropout_prob=0.0, noise_prob=0.0, sparse=False):
    super().__init__()
    self.activation_type = activation_type
    self.is_constrained = is_constrained
    self.hidden_layers = hidden_layers
    self.dropout_prob = dropout_prob
    self.noise_prob = noise_prob
    self.sparse = sparse
    self.num_items = None
    self.num_embeddings = None
    # kept for attribute compatibility (nn.py:141-143); the fused kernels own
    # the dropout arithmetic
    self.noise_layer = None
    self.dropout_layer = None

  # -- the four-method contract ------------------------------------------
  def init_model(self, num_items=None, num_users=None):
    _check_act(self.activation_type)
    self.num_items = num_items
    self.num_embeddings = num_items
    h = self.hidden_layers
    # encoder side first, then decoder side: same RNG consumption order as
    # nn.py:179-212 (Embedding N(0,1) draw, Linear default draws, xavier draws)
    self.en_embedding_layer = nn.Embedding(num_items, h[0], sparse=self.sparse)
    self.__en_linear_embedding_layer = LinearEmbedding(self.en_embedding_layer, input_based=True)
    self.encoding_layers = nn.Sequential(*self._coding_layers(h))
    nn.init.xavier_uniform_(self.en_embedding_layer.weight)
    nn.init.constant_(self.__en_linear_embedding_layer.bias, 0)

    dec = self._coding_layers(list(reversed(h)))
    if self.is_constrained:
      for layer in dec:
        del layer.weight          # only the biases stay registered (nn.py:192-196)
      self.de_embedding_layer = self.en_embedding_layer
    else:
      self.de_embedding_layer = nn.Embedding(num_items, h[0], sparse=self.sparse)
    self.decoding_layers = nn.Sequential(*dec)
    self.__de_linear_embedding_layer = LinearEmbedding(self.de_embedding_layer, input_based=False)
    nn.init.xavier_uniform_(self.de_embedding_layer.weight)
    nn.init.constant_(self.__de_linear_embedding_layer.bias, 0)

    self.noise_layer = nn.Dropout(p=self.noise_prob) if self.noise_prob > 0.0 else None
    self.dropout_layer = nn.Dropout(p=self.dropout_prob) if self.dropout_prob > 0.0 else None

  @staticmethod
  def _coding_layers(sizes):
    layers = []
    for i in range(1, len(sizes)):
      lin = nn.Linear(sizes[i - 1], sizes[i])
      nn.init.xavier_uniform_(lin.weight)
      nn.init.constant_(lin.bias, 0)
      layers.append(lin)
    return layers

  def model_params(self):
    return {
      "hidden_layers": self.hidden_layers,
      "activation_type": self.activation_type,
      "is_constrained": self.is_constrained,
      "dropout_prob": self.dropout_prob,
      "noise_prob": self.noise_prob,
    }

  def load_model_params(self, model_params):
    self.hidden_layers = model_params["hidden_layers"]
    self.activation_type = model_params["activation_type"]
    self.is_constrained = model_params["is_constrained"]
    self.dropout_prob = model_params["dropout_prob"]
    self.noise_prob = model_params["noise_prob"]

  # -- accessors used by the engine ---------------------------------------
  @property
  def en_bias(self):
    return self.__en_linear_embedding_layer.bias

  @property
  def de_bias(self):
    return self.__de_linear_embedding_layer.bias

  def forward(self, input, input_users=None, input_items=None, target_users=None,
              target_items=None):
    """Dense-input forward (nn.py:228-253) on the HIP kernels; no autograd."""
    if not fused_supported(self):
      with torch.no_grad():
        return self.torch_forward(input, input_users, input_items, target_users, target_items)
    from .engine import ae_dense_forward
    return ae_dense_forward(self, input, input_items, target_items)

  def torch_forward(self, input, input_users=None, input_items=None, target_users=None,
                    target_items=None):
    """The same forward in differentiable torch ops (on the GPU).  Only the generic
    path of Recoder uses it: nn.Module losses and the sgd/adagrad/rmsprop optimizers
    have no fused kernels and train through autograd (recoder_amd/generic.py)."""
    act, nl = self.activation_type, len(self.hidden_layers) - 1
    z = F.normalize(input, p=2, dim=1)
    if self.noise_layer is not None:
      z = self.noise_layer(z)
    z = _activate(_embedding_linear(self.en_embedding_layer, self.en_bias, input_items, z, True,
                                    self.sparse), act)
    for layer in self.encoding_layers:
      z = _activate(layer(z), act)
    if self.dropout_layer is not None:
      z = self.dropout_layer(z)
    for i, layer in enumerate(self.decoding_layers):
      w = self.encoding_layers[nl - 1 - i].weight.t() if self.is_constrained else layer.weight
      z = _activate(F.linear(z, w, layer.bias), act)
    return _embedding_linear(self.de_embedding_layer, self.de_bias, target_items, z, False,
                             self.sparse)


class MatrixFactorization(FactorizationModel):
  """Matrix factorization (nn.py:283-362): act(E_u[users]) . E_i[T]^T + b[T]."""

  def __init__(self, embedding_size, activation_type="none", dropout_prob=0, sparse=False):
    super().__init__()
    self.embedding_size = embedding_size
    self.activation_type = activation_type
    self.dropout_prob = dropout_prob
    self.sparse = sparse
    self.num_users = None
    self.num_items = None
    self.user_embedding_layer = None
    self.item_embedding_layer = None
    self.bias = None
    self.dropout_layer = None

  def init_model(self, num_items=None, num_users=None):
    _check_act(self.activation_type)
    self.num_users = num_users
    self.num_items = num_items
    self.user_embedding_layer = nn.Embedding(num_users, self.embedding_size, sparse=self.sparse)
    self.item_embedding_layer = nn.Embedding(num_items, self.embedding_size, sparse=self.sparse)
    self.bias = nn.Parameter(torch.zeros(num_items))
    self.dropout_layer = nn.Dropout(p=self.dropout_prob) if self.dropout_prob > 0.0 else None
    nn.init.xavier_uniform_(self.user_embedding_layer.weight)
    nn.init.xavier_uniform_(self.item_embedding_layer.weight)
    nn.init.constant_(self.bias, 0)

  def model_params(self):
    return {
      "embedding_size": self.embedding_size,
      "activation_type": self.activation_type,
      "dropout_prob": self.dropout_prob,
    }

  def load_model_params(self, model_params):
    self.embedding_size = model_params["embedding_size"]
    self.activation_type = model_params["activation_type"]
    self.dropout_prob = model_params["dropout_prob"]

  def forward(self, input, input_users=None, input_items=None, target_users=None,
              target_items=None):
    if not fused_supported(self):
      with torch.no_grad():
        return self.torch_forward(input, input_users, input_items, target_users, target_items)
    from .engine import mf_dense_forward
    return mf_dense_forward(self, input_users, target_items)

  def torch_forward(self, input, input_users=None, input_items=None, target_users=None,
                    target_items=None):
    """Differentiable torch-op forward for the generic path (see DynamicAutoencoder)."""
    u = _activate(self.user_embedding_layer(input_users), self.activation_type)
    if self.dropout_layer is not None:
      u = self.dropout_layer(u)
    if target_items is None:
      return F.linear(u, self.item_embedding_layer.weight, self.bias)
    return F.linear(u, self.item_embedding_layer(target_items),
                    self.bias.index_select(0, target_items))

"""Generic (non-fused) training path -- SURVEY section 8b.

The reference trains ANY ``FactorizationModel`` with ANY sum-reduction
``nn.Module`` loss and four optimizer kinds through torch autograd
(model.py:383-404, 454-485).  The fused HIP step covers what BASELINE names
(DynamicAutoencoder / MatrixFactorization x mse|logistic|logloss x Adam|SparseAdam).
Everything else -- user-defined models (tutorial.md "Your Own Factorization
Model"), custom loss modules, sgd / adagrad / rmsprop -- goes through this class:
the batch is still collated on the device (rk_collate), densified by rk_densify
(the reference's ``torch.sparse.FloatTensor(...).to_dense()``, model.py:457-458)
and handed to the model's own torch forward ON THE GPU; backward and the
optimizer are torch's.  Nothing here runs on the CPU.
"""
import ctypes

import torch

from . import _lib
from ._lib import check, ptr
from .device import require_gpu


class GenericEngine:
  generic = True

  def __init__(self, model, loss_module, device=None):
    self.model = model
    self.loss_module = loss_module
    self.device = device or require_gpu()
    self.lib = _lib.load()
    self.optimizer = None
    self.sparse_optimizer = None
    self.allreduce = None
    self._fwd = getattr(model, "torch_forward", None) or model.__call__

  # the fused engine's bookkeeping hooks
  def bind_optimizers(self, optimizer, sparse_optimizer):
    self.optimizer, self.sparse_optimizer = optimizer, sparse_optimizer

  def sync_optimizer_steps(self):
    pass

  # ------------------------------------------------------------------ helpers
  def _dense(self, blk, row_off, B):
    """(dense [B, n] fp32, items int64 [n] or None, users int64 [B])."""
    n = blk.host_n_b()
    out = torch.empty(B, n, dtype=torch.float32, device=self.device)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    check(self.lib.rk_densify(blk.ref, row_off, B, n, ptr(out), n, stream), "rk_densify")
    items = blk.items[:n].to(torch.int64) if blk.negative_sampling else None
    users = blk.users[row_off:row_off + B] if blk.users is not None else None
    return out, items, users

  def _loss(self, blk, row_off, B, tgt, denom_rows):
    x, items, users = self._dense(blk, row_off, B)
    if tgt is not None and tgt is not blk:
      t, t_items, t_users = self._dense(tgt, row_off, B)
    else:
      t, t_items, t_users = x, items, users
    out = self._fwd(x, input_users=users, input_items=items, target_users=t_users,
                    target_items=t_items)
    # model.py:483-484: the summed loss averaged over the rows of the batch
    return self.loss_module(out, t) / float(denom_rows)

  # -------------------------------------------------------------------- steps
  def train_step(self, blk, row_off, B, keep_noise=None, keep_drop=None, out=None,
                 global_rows=None, tgt=None):
    if global_rows is not None and global_rows != B:
      raise NotImplementedError("data-parallel training is implemented for the fused path only")
    if keep_noise is not None or keep_drop is not None:
      raise NotImplementedError("mask hooks drive the fused kernels; the generic path uses "
                                "torch's own dropout")
    for opt in (self.optimizer, self.sparse_optimizer):
      if opt is not None:
        opt.zero_grad()
    loss = self._loss(blk, row_off, B, tgt, B)
    loss.backward()
    for opt in (self.optimizer, self.sparse_optimizer):
      if opt is not None:
        opt.step()
    if out is not None:
      out.copy_(loss.detach().reshape(out.shape))
      return out
    return loss.detach()

  @torch.no_grad()
  def compute_loss(self, blk, row_off, B, tgt=None, out=None):
    loss = self._loss(blk, row_off, B, tgt, B)
    if out is not None:
      out.copy_(loss.reshape(out.shape))
      return out
    return loss

  @torch.no_grad()
  def predict_scores(self, blk, row_off, B, out, ld_out, tgt_items_blk):
    """model.py:487-511: ``model(input_dense, input_users=users)`` -> scores of every item."""
    x, _, users = self._dense(blk, row_off, B)
    y = self._fwd(x, input_users=users)
    out[:B, :y.shape[1]].copy_(y)
    return out

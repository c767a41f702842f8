"""Loss descriptors with the reference's names (recoder/losses.py).

``Recoder`` maps them to the fused decode+loss HIP epilogue
(``rk_decode_loss``): ``MSELoss`` -> RK_LOSS_MSE with ``confidence``
(losses.py:38-47), ``MultinomialNLLLoss`` -> RK_LOSS_MNLL (losses.py:64-71).
Calling an instance on dense tensors evaluates the same formula elementwise on
the tensors' device (a convenience for users; not on the training path).
"""
from torch import nn
import torch.nn.functional as F


def _reduce(x, reduction="elementwise_mean"):
  if reduction == "none":
    return x
  if reduction == "elementwise_mean":
    return x.mean()
  if reduction == "sum":
    return x.sum()
  raise ValueError("No such reduction {} defined".format(reduction))


class MSELoss(nn.Module):
  def __init__(self, confidence=0, reduction="elementwise_mean"):
    super().__init__()
    self.reduction = reduction
    self.confidence = confidence

  def forward(self, input, target):
    weights = 1 + self.confidence * (target > 0).float()
    return _reduce(weights * (input - target) ** 2, reduction=self.reduction)


class MultinomialNLLLoss(nn.Module):
  def __init__(self, reduction="elementwise_mean"):
    super().__init__()
    self.reduction = reduction

  def forward(self, input, target):
    return _reduce(-target * F.log_softmax(input, dim=1), reduction=self.reduction)

"""``Recoder``: the reference's trainer API (recoder/model.py:22-559) on the
MI355X hot path.

Same constructor, ``train`` signature, public attributes, checkpoint dict and
error behaviour as the reference; the body of the hot loop
(model.py:383-404) is ``FusedEngine.train_step`` (HIP kernels) over batches
collated on the device, with no per-step host synchronisation: the per-step
losses are left in a device buffer and read once per epoch.

Two hooks exist because a GPU cannot reproduce the reference's CPU RNG streams
(SURVEY section 7 "RNG parity"):
  ``user_order_hook(epoch, n_users) -> int64 array | None``  the user order of
      an epoch (default: the same draw ``RandomSampler`` makes from the global
      torch RNG, so noise-free runs reproduce the reference's order);
  ``mask_hook(step, users, nnz) -> (keep_noise | None, keep_drop | None)``
      explicit dropout keep-masks for a sampling group (default: counter RNG).
"""
import logging
import os

import numpy as np
import torch
import torch.optim as optim
from torch.optim.lr_scheduler import MultiStepLR

from . import __version__
from .data import (BatchCollator, RecommendationDataLoader, RecommendationDataset,
                   epoch_user_order)
from .device import Block, DeviceCSR, require_gpu
from .engine import FusedEngine
from .losses import MSELoss, MultinomialNLLLoss
from .metrics import RecommenderEvaluator
from .nn import DynamicAutoencoder, FactorizationModel, MatrixFactorization
from .recommender import InferenceRecommender

log = logging.getLogger("recoder_amd")


def _top_sum(degrees, k):
  d = np.asarray(degrees)
  if len(d) <= k:
    return int(d.sum())
  return int(np.partition(d, len(d) - k)[len(d) - k:].sum())


class Recoder(object):
  """Trains / evaluates a :class:`recoder_amd.nn.FactorizationModel`
  (arguments as model.py:26-47)."""

  def __init__(self, model: FactorizationModel, num_items=None, num_users=None,
               optimizer_type="sgd", loss="mse", loss_params=None, use_cuda=False,
               user_based=True, item_based=True):
    self.model = model
    self.num_items = num_items
    self.num_users = num_users
    self.optimizer_type = optimizer_type
    self.loss = loss
    self.loss_params = loss_params if loss_params else {}
    self.use_cuda = use_cuda
    self.user_based = user_based
    self.item_based = item_based
    # this build has exactly one device: the MI355X PyTorch-ROCm exposes as
    # 'cuda'.  use_cuda is kept for signature compatibility.
    self.device = torch.device("cuda")
    self.optimizer = None
    self.sparse_optimizer = None
    self.current_epoch = 1
    self.items = None
    self.users = None
    self.user_order_hook = None
    self.mask_hook = None
    # steps collated per side-stream hand-over (CollatePrefetcher)
    self.prefetch_group = 4
    # steps per replayed HIP graph (graph.py)
    self.graph_group = int(os.environ.get("RK_GRAPH_GROUP", "8"))
    # {global step index: callable}: called right before that step's collation is submitted and
    # its kernels are enqueued (the pipeline is cut there: nothing of the step is in flight yet);
    # returning True ends the training.  bench.py brackets its timed region with two of these.
    self.step_marks = {}
    self.last_epoch_losses = None
    self.loss_history = []      # per-epoch arrays of the per-step training losses
    self.__model_initialized = False
    self.__optimizer_state_dict = None
    self.__sparse_optimizer_state_dict = None
    self.__engine = None
    self._dist = None        # (rank, world_size) when data parallel

  # ------------------------------------------------------------------ init
  def __init_model(self):
    if self.__model_initialized:
      return
    require_gpu()
    self.model.init_model(self.num_items, self.num_users)
    self.model = self.model.to(device=self.device)
    self.__model_initialized = True

  def _fused_kind(self):
    if isinstance(self.model, DynamicAutoencoder):
      return "ae"
    if isinstance(self.model, MatrixFactorization):
      return "mf"
    return None

  def _use_generic(self):
    """True when the combination has no fused HIP step and trains through torch
    autograd on the GPU instead (recoder_amd/generic.py): user-defined
    FactorizationModel subclasses, arbitrary nn.Module losses, sgd/adagrad/rmsprop."""
    if self._fused_kind() is None or self.optimizer_type != "adam":
      return True
    if getattr(self, "_force_generic", None):
      return True
    from .nn import fused_supported
    if not fused_supported(self.model):
      if not getattr(self, "_warned_generic", False):
        self._warned_generic = True
        log.warning("activation %r / first hidden size are outside the fused HIP kernels "
                    "(activations none|tanh|sigmoid|relu|selu|elu, size %% 4 == 0): training through "
                    "torch autograd on the GPU instead", self.model.activation_type)
      return True
    if isinstance(self.loss, str):
      # named losses are built as <Loss>(reduction='sum', **loss_params) (model.py:87-99): the fused
      # epilogues implement the MSE confidence weight and nothing else a module can be given
      extra = set(self.loss_params) - ({"confidence"} if self.loss == "mse" else set())
      if extra and not getattr(self, "_warned_loss_params", False):
        self._warned_loss_params = True
        log.warning("loss_params %s are outside the fused loss epilogues: training through torch "
                    "autograd on the GPU instead", sorted(extra))
      return bool(extra)
    # loss MODULES: the fused epilogues compute the plain summed loss -- anything else a module
    # can be configured with (mean / none reduction, element or class weights) goes through torch
    if isinstance(self.loss, (MSELoss, MultinomialNLLLoss)):
      return self.loss.reduction != "sum"
    if isinstance(self.loss, torch.nn.BCEWithLogitsLoss):
      return self.loss.reduction != "sum" or self.loss.weight is not None or \
          self.loss.pos_weight is not None
    return True

  def __init_loss_module(self):
    """model.py:87-99 -- same names, same errors."""
    if issubclass(self.loss.__class__, torch.nn.Module):
      self.loss_module = self.loss
      if isinstance(self.loss, MSELoss):
        self._loss_name, self._loss_params = "mse", {"confidence": self.loss.confidence}
      elif isinstance(self.loss, MultinomialNLLLoss):
        self._loss_name, self._loss_params = "logloss", {}
      elif isinstance(self.loss, torch.nn.BCEWithLogitsLoss):
        self._loss_name, self._loss_params = "logistic", {}
      else:
        self._loss_name, self._loss_params = None, {}       # generic path only
    elif self.loss == "logistic":
      self._loss_name, self._loss_params = "logistic", dict(self.loss_params)
      self.loss_module = torch.nn.BCEWithLogitsLoss(reduction="sum", **self.loss_params)
    elif self.loss == "mse":
      self._loss_name, self._loss_params = "mse", dict(self.loss_params)
      self.loss_module = MSELoss(reduction="sum", **self.loss_params)
    elif self.loss == "logloss":
      self._loss_name, self._loss_params = "logloss", {}
      self.loss_module = MultinomialNLLLoss(reduction="sum")
    elif self.loss is None:
      raise ValueError("No loss function defined")
    else:
      raise ValueError("Unknown loss function {}".format(self.loss))

  def __init_optimizer(self, lr, weight_decay):
    """model.py:101-164: dense / sparse parameter split, one group per tensor,
    no weight decay on biases."""
    if self.optimizer is not None:
      self._engine().sync_optimizer_steps()
      self.__optimizer_state_dict = self.optimizer.state_dict()
    if self.sparse_optimizer is not None:
      self._engine().sync_optimizer_steps()
      self.__sparse_optimizer_state_dict = self.sparse_optimizer.state_dict()

    sparse_params_names = []
    sparse_modules = [torch.nn.Embedding, torch.nn.EmbeddingBag]
    for module_name, module in self.model.named_modules():
      if type(module) in sparse_modules and module.sparse:
        sparse_params_names.extend([module_name + "." + n for n, _ in module.named_parameters()])

    params, sparse_params = [], []
    for param_name, param in self.model.named_parameters():
      wd = 0 if "bias" in param_name else weight_decay
      group = {"params": param, "weight_decay": wd}
      (sparse_params if param_name in sparse_params_names else params).append(group)

    if self.optimizer_type == "adam":
      if len(params) > 0:
        self.optimizer = optim.Adam(params, lr=lr)
      if len(sparse_params) > 0:
        self.sparse_optimizer = optim.SparseAdam(sparse_params, lr=lr)
    elif self.optimizer_type in ("adagrad", "sgd", "rmsprop"):
      # model.py:140-154; these run through torch on the GPU (generic path)
      if len(sparse_params) > 0:
        raise ValueError("Sparse gradients optimization not supported with {}"
                         .format(self.optimizer_type))
      if self.optimizer_type == "adagrad":
        self.optimizer = optim.Adagrad(params, lr=lr)
      elif self.optimizer_type == "sgd":
        self.optimizer = optim.SGD(params, lr=lr, momentum=0.9)
      else:
        self.optimizer = optim.RMSprop(params, lr=lr, momentum=0.9)
    else:
      raise Exception("Unknown optimizer kind")

    if self.__optimizer_state_dict is not None and self.optimizer is not None:
      self.optimizer.load_state_dict(self.__optimizer_state_dict)
      self.__optimizer_state_dict = None
    if self.__sparse_optimizer_state_dict is not None and self.sparse_optimizer is not None:
      self.sparse_optimizer.load_state_dict(self.__sparse_optimizer_state_dict)
      self.__sparse_optimizer_state_dict = None
    self._engine().bind_optimizers(self.optimizer, self.sparse_optimizer)

  def _sync_ranges(self):
    """Decoder-weight bound of the split-fp16 GEMMs from the table itself (engine.ranges): at every
    public entry, because parameters may have been written from outside since the last call."""
    eng = self._engine()
    if hasattr(eng, "refresh_weight_range"):
      eng.refresh_weight_range()

  def _check_ranges(self):
    """The same bound, recomputed only when the table changed since it was last taken (its torch
    version counter / address: predict and recommend are called once per evaluation batch, and a
    full-table norm is ~800 MB of reads at 1 M items)."""
    eng = self._engine()
    if hasattr(eng, "_check_weight_range"):
      eng._check_weight_range()

  def _engine(self):
    if self.__engine is None:
      self.__init_loss_module()
      if self._use_generic():
        from .generic import GenericEngine
        self.__engine = GenericEngine(self.model, self.loss_module, self.device)
      else:
        self.__engine = FusedEngine(self.model, self._fused_kind(), self._loss_name,
                                    self._loss_params, self.device)
    return self.__engine

  # ------------------------------------------------------------ checkpoint
  def init_from_model_file(self, model_file):
    """model.py:166-191."""
    log.info("Loading model from: {}".format(model_file))
    if not os.path.isfile(model_file):
      raise Exception("No state file found in {}".format(model_file))
    st = torch.load(model_file, map_location="cpu", weights_only=False)
    model_params = st["model_params"]
    self.current_epoch = st["last_epoch"]
    self.loss = st.get("loss", self.loss)
    self.loss_params = st.get("loss_params", self.loss_params)
    self.optimizer_type = st["optimizer_type"]
    self.items = st.get("items", None)
    self.users = st.get("users", None)
    self.num_items = st.get("num_items", None)
    self.num_users = st.get("num_users", None)
    self.__optimizer_state_dict = st["optimizer"]
    self.__sparse_optimizer_state_dict = st.get("sparse_optimizer", None)
    self.model.load_model_params(model_params)
    self.__init_model()
    self.model.load_state_dict(st["model"])
    if self.__engine is not None and hasattr(self.__engine, "_w_range_stale"):
      self.__engine._w_range_stale = True

  def save_state(self, model_checkpoint_prefix):
    """model.py:193-224 (same dict; like the reference the sparse optimizer's
    state is not written)."""
    checkpoint_file = "{}_epoch_{}.model".format(model_checkpoint_prefix, self.current_epoch)
    log.info("Saving model to {}".format(checkpoint_file))
    if self.__engine is not None:
      self.__engine.sync_optimizer_steps()
    current_state = {
      "recoder_version": __version__,
      "model_params": self.model.model_params(),
      "last_epoch": self.current_epoch,
      "model": self.model.state_dict(),
      "optimizer_type": self.optimizer_type,
      "optimizer": self.optimizer.state_dict(),
      "items": self.items,
      "users": self.users,
      "num_items": self.num_items,
      "num_users": self.num_users,
    }
    if type(self.loss) is str:
      current_state["loss"] = self.loss
      current_state["loss_params"] = self.loss_params
    # (no collective in here: a caller may save from one rank only.  The checkpoint train() writes
    # under data / item parallelism has one writer and a barrier -- see _epoch_end)
    torch.save(current_state, checkpoint_file)
    return checkpoint_file

  # --------------------------------------------------------------- training
  def __init_training(self, train_dataset, lr, weight_decay):
    """model.py:226-254."""
    if self.items is None:
      self.items = train_dataset.items
    else:
      self.items = np.unique(np.append(self.items, train_dataset.items))
    if self.users is None:
      self.users = train_dataset.users
    else:
      self.users = np.unique(np.append(self.users, train_dataset.users))

    if self.item_based and self.num_items is None:
      self.num_items = int(np.max(self.items)) + 1
    elif self.item_based:
      assert self.num_items >= int(np.max(self.items)) + 1, \
        "The largest item id should be smaller than number of items." \
        "If your model is not based on items, set item_based to False in Recoder constructor."
    if self.user_based and self.num_users is None:
      self.num_users = int(np.max(self.users)) + 1
    elif self.user_based:
      assert self.num_users >= int(np.max(self.users)) + 1, \
        "The largest user id should be smaller than number of users." \
        "If your model is not based on users, set user_based to False in Recoder constructor."

    self.__init_model()
    self.__init_loss_module()
    self.__init_optimizer(lr=lr, weight_decay=weight_decay)

  def train(self, train_dataset, val_dataset=None, lr=0.001, weight_decay=0, num_epochs=1,
            iters_per_epoch=None, batch_size=64, lr_milestones=None, negative_sampling=False,
            num_sampling_users=0, num_data_workers=0, model_checkpoint_prefix=None,
            checkpoint_freq=0, eval_freq=0, eval_num_recommendations=None, eval_num_users=None,
            metrics=None, eval_batch_size=None):
    """Trains the model (arguments as model.py:256-289)."""
    log.info("GPU Mode (MI355X / HIP)")
    for k, v in self.model.model_params().items():
      log.info("Model {}: {}".format(k, v))
    log.info("lr {} wd {} batch {} optimizer {} milestones {} loss {}".format(
        lr, weight_decay, batch_size, self.optimizer_type, lr_milestones, self.loss))

    if num_sampling_users == 0:
      num_sampling_users = batch_size
    if eval_batch_size is None:
      eval_batch_size = batch_size
    assert num_sampling_users >= batch_size and num_sampling_users % batch_size == 0, \
      "number of sampling users should be a multiple of the batch size"

    self._pick_engine_for(train_dataset)
    self.__init_training(train_dataset=train_dataset, lr=lr, weight_decay=weight_decay)
    train_dataset = self._setup_data_parallel(train_dataset, negative_sampling, batch_size)
    if getattr(self, "_ip", None) is not None:
      # item parallel: every rank processes the whole global batch against its items
      batch_size *= self._ip.world
      num_sampling_users *= self._ip.world

    train_dataloader = RecommendationDataLoader(train_dataset, batch_size=batch_size,
                                                negative_sampling=negative_sampling,
                                                num_sampling_users=num_sampling_users,
                                                num_workers=num_data_workers)
    if val_dataset is not None:
      val_dataloader = RecommendationDataLoader(val_dataset, batch_size=batch_size,
                                                negative_sampling=negative_sampling,
                                                num_sampling_users=num_sampling_users,
                                                num_workers=num_data_workers)
    else:
      val_dataloader = None

    if lr_milestones is not None:
      _last_epoch = -1 if self.current_epoch == 1 else (self.current_epoch - 2)
      if _last_epoch != -1:
        for g in self.optimizer.param_groups:
          g.setdefault("initial_lr", lr)
      lr_scheduler = MultiStepLR(self.optimizer, milestones=lr_milestones, gamma=0.1,
                                 last_epoch=_last_epoch)
    else:
      lr_scheduler = None

    self._train(train_dataloader=train_dataloader, val_dataloader=val_dataloader,
                num_epochs=num_epochs, current_epoch=self.current_epoch,
                lr_scheduler=lr_scheduler, batch_size=batch_size,
                model_checkpoint_prefix=model_checkpoint_prefix,
                checkpoint_freq=checkpoint_freq, eval_freq=eval_freq, metrics=metrics,
                eval_num_recommendations=eval_num_recommendations,
                iters_per_epoch=iters_per_epoch, eval_num_users=eval_num_users,
                eval_batch_size=eval_batch_size)
    self._sync_user_rows()

  def _pick_engine_for(self, train_dataset):
    """Combinations the fused step does not cover train through the generic engine (torch autograd
    on the GPU, recoder_amd/generic.py) instead of raising -- the reference trains all of them
    (model.py:464-476, nn.py:191-202): tied weights with a separate target matrix."""
    has_target = getattr(train_dataset, "target_interactions_matrix", None) is not None or \
        (hasattr(train_dataset, "device_target_csr") and train_dataset.device_target_csr() is not None)
    tied = self._fused_kind() == "ae" and bool(getattr(self.model, "is_constrained", False))
    force = "tied weights with a separate target matrix" if (has_target and tied) else None
    if force != getattr(self, "_force_generic", None):
      self._force_generic = force
      if self.__engine is not None:
        # the engine is rebuilt for this call; the optimizers (and their state) stay
        if hasattr(self.__engine, "sync_optimizer_steps"):
          self.__engine.sync_optimizer_steps()
        self.__engine = None
      if force:
        log.warning("%s has no fused HIP step: training through torch autograd on the GPU", force)

  def _sync_user_rows(self):
    """Bring every replica up to date with the rows other ranks own -- parameters and
    Adam moments -- so that the result equals the single-process run with
    batch_size = N * B (called at the end of train(), before evaluation and before
    checkpoints).  Item parallel: item i's embedding rows / bias live on rank i % N.
    MatrixFactorization under data parallelism: a user's row only receives gradients
    on the rank that holds the user."""
    self._sync_owned_moments()
    ip = getattr(self, "_ip", None)
    if ip is not None:
      m = self.model
      if self._fused_kind() == "mf":
        params = [m.item_embedding_layer.weight, m.bias]     # user rows are replicated
      else:
        params = [m.en_embedding_layer.weight, m.de_bias]
        if not m.is_constrained:
          params.append(m.de_embedding_layer.weight)
      tensors = []
      for w in params:
        tensors.append(w.data)
        for opt in (self.optimizer, self.sparse_optimizer):
          st = opt.state.get(w) if opt is not None else None
          if st:
            tensors += [st["exp_avg"], st["exp_avg_sq"]]
      ip.sync_owned(tensors, self.num_items)
      self._weights_written()
      return
    if getattr(self, "_dp", None) is None or self._fused_kind() != "mf":
      return
    from .parallel import sync_owned_rows
    w = self.model.user_embedding_layer.weight
    tensors = [w.data]
    for opt in (self.optimizer, self.sparse_optimizer):
      st = opt.state.get(w) if opt is not None else None
      if st:
        tensors += [st["exp_avg"], st["exp_avg_sq"]]
    if self._dp_n_users < w.shape[0]:
      tensors = [t[:self._dp_n_users] for t in tensors]
    sync_owned_rows(tensors, self._dp_n_users, self._dp.group)
    self._weights_written()

  def _weights_written(self):
    """Parameters were written through `.data` (no torch version bump: owner-row syncs): the decoder
    weight bound of the split contractions and the evaluation images must be taken again."""
    eng = self._engine()
    if hasattr(eng, "_w_range_stale"):
      eng._w_range_stale = True
      eng._eval_img = None

  def _setup_data_parallel(self, train_dataset, negative_sampling=True, batch_size=None):
    """Under an initialised torch.distributed group (one process per GPU, backend
    'nccl' = RCCL) the USERS are sharded over the ranks and the gradients are
    all-reduced (recoder_amd/parallel.DataParallel) -- north_star's partitioning and the
    default.  RK_PARALLEL=items shards the item dimension instead (parallel.ItemParallel).
    Returns this rank's shard."""
    import torch.distributed as dist
    self._dp = None
    self._ip = None
    self._dp_nnz_cap = {}                  # (keyed on id(matrix): never survives the matrix it was taken for)
    # (a previous data-parallel train() may have left the engine on owned-row Adam: a run that does not
    # reach _setup_owned_rows -- no group, the replicated fallback -- must get the one-call step back)
    eng0 = getattr(self, "_Recoder__engine", None)
    if eng0 is not None:
      eng0.owned_rows = False
      eng0.zero_adam = False
    if getattr(self, "_ip_override", None) is not None:
      # tests: several virtual ranks in one process, collectives injected
      return self._enable_item_parallel(self._ip_override, train_dataset)
    dp = getattr(self, "_dp_override", None)
    if dp is None:
      if not (dist.is_available() and dist.is_initialized()):
        return train_dataset
      if dist.get_world_size() == 1 and os.environ.get("RK_FORCE_DP") != "1":
        return train_dataset
    if self._use_generic() or train_dataset.device_target_csr() is not None:
      # no sharded formulation for these: every rank trains the same replica on the whole data
      # (identical results on every rank, no speed-up) instead of refusing to run
      log.warning("multi-GPU training covers the fused DynamicAutoencoder / MatrixFactorization step "
                  "without a separate target matrix: running this configuration replicated on "
                  "every rank")
      if dp is None:
        for p_ in self.model.parameters():        # identical replicas whatever each process seeded
          dist.broadcast(p_.data, src=0)
        self._weights_written()
      return train_dataset
    if dp is None:
      for p_ in self.model.parameters():          # identical replicas: rank 0's initial weights
        dist.broadcast(p_.data, src=0)
    mode = os.environ.get("RK_PARALLEL", "users")
    if mode == "auto":
      mode = "users"
    if mode == "items" and not negative_sampling:
      # without sampling every rank's block spans the whole catalogue, but an item shard only
      # holds its own columns' targets: the loss over the other columns would be wrong
      log.warning("RK_PARALLEL=items needs negative_sampling=True; sharding the users instead")
      mode = "users"
    if mode == "items" and dp is None:
      from .parallel import ItemParallel
      return self._enable_item_parallel(ItemParallel(), train_dataset)
    from .parallel import DataParallel, shard_range
    if dp is None:
      dp = DataParallel().prepare(self.device)
    dp.attach(self._engine())
    self._dp = dp
    self._setup_owned_rows(dp, train_dataset, negative_sampling, batch_size)
    self._setup_local_sets(dp, negative_sampling)
    self._setup_zero_adam(dp)
    n = len(train_dataset)
    lo, hi = shard_range(n, dp.rank, dp.world)
    dp.user_offset = lo
    self._dp_n_users = n
    if hasattr(train_dataset, "row_shard"):           # data.DeviceDataset: sliced in HBM
      shard = train_dataset.row_shard(lo, hi)
    elif getattr(train_dataset, "_dev", None) is not None:
      # a host dataset whose matrix is already resident (dataset.device_csr() was called): this
      # rank's rows are a device-side slice of it -- no host slicing, no second upload
      from .data import DeviceDataset
      shard = DeviceDataset(train_dataset._dev.row_slice(lo, hi))
    else:
      shard = RecommendationDataset(train_dataset.interactions_matrix[lo:hi])
    # every rank runs the same number of equally sized steps (collectives in lockstep)
    self._dp_users_per_epoch = n // dp.world
    return shard

  def _setup_owned_rows(self, dp, train_dataset, negative_sampling, batch_size):
    """Owned-row Adam (parallel.DataParallel): with SparseAdam embedding tables every item's rows -- and
    their moments -- belong to one rank; RK_DP_OWNED=0 keeps the replicated update.  The item-id ranges
    are balanced by the items' expected presence in a global batch, from the training matrix's column
    counts (the same on every rank: it is taken before the users are sharded)."""
    eng = self._engine()
    eng.owned_rows = False
    dp.owner_bounds = None
    sparse = bool(getattr(self.model, "sparse", False)) and self.sparse_optimizer is not None
    # RK_DP_OWNED: 0 = replicated update; 1 = owned rows with more than one rank; force = also with one rank
    # (tests); auto (default) = owned rows only where their price is paid back.  The replicated update
    # replays as HIP graphs with its collectives captured (every engine, round 5); the owned-row exchange
    # moves row counts only the host knows -- a device -> host read, ~25 launches enqueued one by one and the
    # received rows scattered into the tables per step: one forced rank, C4 0.266 vs 0.129 ms per step,
    # C5-shaped 1.43 vs 0.79 (profiles/r05_bench_c*_dp1_*.json).  What it saves is (1 - 1/N) of the
    # SparseAdam sweep over the union rows.
    mode = os.environ.get("RK_DP_OWNED", "auto")
    if not (sparse and negative_sampling and batch_size and (dp.world > 1 or mode == "force") and mode != "0"):
      return
    if dp.virtual and dp._gather_fn is None:
      return                                # (injected collectives without a gather: replicated update)
    dev = getattr(train_dataset, "_dev", None)
    n_items = int(self.num_items)
    if dev is not None:
      freq = torch.bincount(dev.indices[:dev.nnz].long(), minlength=n_items).cpu().numpy()
    else:
      freq = np.bincount(train_dataset.interactions_matrix.indices, minlength=n_items)
    from .parallel import DataParallel
    if mode == "auto":
      est = DataParallel.owned_rows_estimate(freq[:n_items], len(train_dataset), dp.world * int(batch_size),
                                             dp.world, int(eng.h[0]),
                                             1 if (eng.kind != "ae" or bool(self.model.is_constrained)) else 2)
      self._dp_owned_estimate = est
      if est["saved_us"] < 2.0 * est["cost_us"]:
        return
    dp.set_owner_bounds(DataParallel.balanced_bounds(freq[:n_items], len(train_dataset),
                                                     dp.world * int(batch_size), dp.world))
    eng.owned_rows = True

  def _setup_local_sets(self, dp, negative_sampling):
    """RK_DP_ITEMSETS = union (default) | local.  `local` (opt-in, NOT the reference's single-process semantics):
    every rank samples its negatives from the items of ITS OWN B users -- what the reference's trainer would do
    under conventional DDP -- instead of the union over all ranks' users (data.py:216-223 read as one shared item
    set).  The contractions then stop growing with the number of ranks (C2: 7.8 k items per rank instead of 18.4 k
    at 8 ranks) and no stamp exchange precedes the collation; the ranks' compact columns differ, so the gradients
    travel laid out by item id (dense Adam tables of the one-call autoencoder step only).  Its oracle is
    oracle.recoder_oracle.OracleRecoder.train_step_ddp (gradient accumulation over the ranks' batches)."""
    eng = self._engine()
    dp.local_sets = False
    if os.environ.get("RK_DP_ITEMSETS", "union") != "local":
      return
    if getattr(eng, "generic", False) or getattr(eng, "owned_rows", False) or not eng.c_step_eligible():
      return
    if bool(getattr(self.model, "sparse", False)) or self.optimizer is None or not negative_sampling:
      return
    if dp.virtual and dp._gather_fn is None:
      return
    dp.local_sets = True

  def _setup_zero_adam(self, dp):
    """Sharded dense Adam (parallel.DataParallel, "ZeRO-1"): with optim.Adam on the embedding tables of the
    one-call autoencoder step every rank owns an equal range of table rows -- gradient rows reduce-scattered,
    the sweep over 1/N of the rows, the updated rows all-gathered; the moments of a row are kept up to date on
    its owner only and gathered before checkpoints / validation (_sync_owned_moments).  RK_DP_ZERO = 0 (default) |
    1 (with more than one rank) | force (also with one rank: tests, the one-rank bench line).  OPT-IN: the
    exchange then moves the DENSE layout (every row of the tables, = the capacity-sized exchange of a replayed
    step) where the replicated update's moves the union rows, and by tools/dp_model.py the saved (1 - 1/N) of
    the sweep only pays for that at 8 ranks, by 1-3 % (C2 282 vs 290, C3 624 vs 627 us per step at 1 TB/s per
    rank; 2 and 4 ranks lose 5-15 %) -- a model: no run on more than one GPU exists, and a path that has never
    met a second RCCL rank is not what the first 8-GPU run should take by default."""
    eng = self._engine()
    eng.zero_adam = False
    dp.zero = None
    mode = os.environ.get("RK_DP_ZERO", "0")
    if not ((mode == "1" and dp.world > 1) or mode == "force"):
      return
    if getattr(eng, "generic", False) or getattr(eng, "owned_rows", False) or not eng.c_step_eligible():
      return
    if bool(getattr(self.model, "sparse", False)) or self.optimizer is None:
      return                                # (SparseAdam tables: the replicated or the owned-row update)
    if dp.virtual and dp._gather_fn is None:
      return
    dp.setup_zero(int(self.num_items))
    eng.zero_adam = True

  def _sync_owned_moments(self):
    """The Adam moments of the owned rows live on their owners: every replica gets them (checkpoints,
    the end of train(), a later single-process continuation)."""
    dp = getattr(self, "_dp", None)
    eng = self._engine()
    if dp is not None and getattr(eng, "zero_adam", False) and dp.zero is not None:
      # (sharded dense Adam: the moments of rows [lo_r, hi_r) are current on rank r only)
      names = ["en_embedding_layer.weight"] + ([] if self.model.is_constrained else ["de_embedding_layer.weight"])
      for name in names:
        st = eng.states[name]
        dp.sync_owned_moments([st.m, st.v], bounds=dp.zero_bounds())
      return
    if dp is None or not getattr(eng, "owned_rows", False):
      return
    for name, _ in eng._sparse_tables():
      st = eng.states[name]
      dp.sync_owned_moments([st.m, st.v])

  def _enable_item_parallel(self, ip, train_dataset):
    from .parallel import ItemParallel
    full = train_dataset.interactions_matrix
    dev = getattr(train_dataset, "_dev", None)
    on_device = dev is not None and (full is None or dev.implicit)
    if on_device:
      # the matrix is resident in HBM (a DeviceDataset has no host copy at all): the row statistics
      # and this rank's column shard are taken there (implicit feedback: exactly the host's numbers)
      ip.user_norm_dev = ItemParallel.user_norms_dev(dev)
      ip.user_tsum_dev = ItemParallel.user_target_sums_dev(dev)
    else:
      ip.user_norm_dev = torch.from_numpy(ItemParallel.user_norms(full)).to(self.device)
      ip.user_tsum_dev = torch.from_numpy(ItemParallel.user_target_sums(full)).to(self.device)
    ip.prepare(self.device)
    self._engine().item_parallel = ip
    self._ip = ip
    if on_device:
      from .data import DeviceDataset
      return DeviceDataset(ip.shard_device_csr(dev))
    return RecommendationDataset(ip.shard_csr(full))

  def _make_block(self, dcsr, S, negative_sampling, train=False):
    """train: a block of the (sharded) training matrix; validation / target blocks are
    collated from unsharded matrices and get the plain capacity."""
    nnz_cap = max(1, _top_sum(dcsr.degrees, S))
    n_cap = nnz_cap
    if train and getattr(self, "_dp", None) is not None and not getattr(self._dp, "local_sets", False):
      # the union item set can exceed one rank's nnz bound; the SAME capacity on every rank (a
      # rank with lighter users would otherwise clamp n_b on its own and the gradient exchange
      # would disagree on its sizes): MAX over the ranks, once per (matrix, group size)
      cache = self.__dict__.setdefault("_dp_nnz_cap", {})
      key = (id(dcsr), int(S))
      if key not in cache:
        t = torch.tensor([nnz_cap], dtype=torch.int32, device=self.device)
        cache[key] = int(self._dp.union_marks(t).max().item())
      n_cap = min(dcsr.n_items, cache[key] * self._dp.world)
    if train and getattr(self, "_ip", None) is not None and negative_sampling:
      n_cap = min(nnz_cap, -(-dcsr.n_items // self._ip.world))   # at most the owned items
    return Block(S, nnz_cap, dcsr.n_items, self.device, negative_sampling=negative_sampling,
                 n_cap=n_cap)

  def _step_generator(self, dataloader):
    """Yields (blk, row_off, B, keep_noise, keep_drop, tgt_blk) for one pass over the
    dataset: the device-side equivalent of iterating the reference's
    RecommendationDataLoader (data.py:138-144).  tgt_blk: the same rows collated from the
    dataset's target matrix (its own item set, data.py:60-62), or None."""
    ds = dataloader.dataset
    dcsr = ds.device_csr()
    dcsr_t = ds.device_target_csr()
    tgt_blk = None
    if dcsr_t is not None:
      tgt_blk = getattr(self, "_train_tgt_blk", None)
      if tgt_blk is None or tgt_blk.S_cap < dataloader.num_sampling_users or \
          getattr(self, "_train_tgt_src", None) is not dcsr_t or \
          tgt_blk.negative_sampling != dataloader.negative_sampling:
        tgt_blk = self._make_block(dcsr_t, dataloader.num_sampling_users,
                                   dataloader.negative_sampling)
        self._train_tgt_blk, self._train_tgt_src = tgt_blk, dcsr_t
    B, S = dataloader.batch_size, dataloader.num_sampling_users
    pf = getattr(self, "_train_pf", None)
    dp = getattr(self, "_dp", None)
    if pf is None or pf.dcsr is not dcsr or pf.blocks[0][0].S_cap < S or \
        pf.blocks[0][0].negative_sampling != dataloader.negative_sampling or \
        getattr(pf, "owner_dp", None) is not dp:
      from .device import CollatePrefetcher
      ns = dataloader.negative_sampling
      pf = CollatePrefetcher(lambda: self._make_block(dcsr, S, ns, train=True), dcsr, self.device,
                             collate_fn=(dp.collate if dp is not None else None),
                             group=self.prefetch_group)
      pf.owner_dp = dp
      self._train_pf = pf
    pf.reset()
    n = len(ds)
    order, self._pending_order = getattr(self, "_pending_order", None), None
    if order is None and self.user_order_hook is not None:
      order = self.user_order_hook(self.current_epoch, n)
    if order is None:
      order = epoch_user_order(n)
    if getattr(self, "_ip", None) is not None:
      o = torch.from_numpy(np.ascontiguousarray(order, dtype=np.int64)).to(self.device)
      order = self._ip.broadcast(o).cpu().numpy()   # one user order for all item shards
    if getattr(self, "_dp", None) is not None:
      order = order[:self._dp_users_per_epoch]      # equal step counts on every rank
      n = len(order)
    order_dev = torch.from_numpy(np.ascontiguousarray(order, dtype=np.int64)).to(self.device)
    offs = [o for o in range(0, n, S) if order_dev[o:o + S].numel() > 0]
    G = pf.group
    # chunks of up to G sampling groups; a chunk never straddles a step mark
    marks = self.step_marks
    step0 = getattr(self, "_global_step", 0)
    first_step, k = {}, step0
    for o in offs:
      first_step[o] = k
      k += -(-min(S, n - o) // B)
    chunks = []
    for o in offs:
      if not chunks or len(chunks[-1]) == G or first_step[o] in marks:
        chunks.append([])
      chunks[-1].append(o)
    users_of = lambda chunk: [order_dev[o:o + S] for o in chunk]
    marked = lambda ci: first_step[chunks[ci][0]] in marks
    submitted = set()

    def submit(ci):
      if ci < len(chunks) and ci not in submitted:
        submitted.add(ci)
        pf.submit(ci % 2, users_of(chunks[ci]))
    # the chunk after the current one is collated on the prefetcher's side stream (unless it
    # starts at a mark: then nothing of it may be in flight before the mark's callback ran)
    if chunks and not marked(0):
      submit(0)
    for ci, chunk in enumerate(chunks):
      slot = ci % 2
      if marked(ci):
        if marks[first_step[chunk[0]]]():
          self._stop_training = True
          return
      submit(ci)
      if ci + 1 < len(chunks) and not marked(ci + 1):
        submit(ci + 1)
      for blk, off in zip(pf.acquire(slot), chunk):
        Sg = int(order_dev[off:off + S].numel())
        if tgt_blk is not None:
          tgt_blk.collate(dcsr_t, order_dev[off:off + S])
        keep_noise = keep_drop = None
        if self.mask_hook is not None:
          keep_noise, keep_drop = self.mask_hook(self._global_step, order[off:off + S])
        for r in range(0, Sg, B):
          rows = min(B, Sg - r)
          kd = None
          if keep_drop is not None:
            kd = keep_drop[r:r + rows].contiguous()
          yield blk, r, rows, keep_noise, kd, tgt_blk
      pf.release(slot)

  def _train(self, train_dataloader, val_dataloader, num_epochs, current_epoch, lr_scheduler,
             batch_size, model_checkpoint_prefix, checkpoint_freq, eval_freq, metrics,
             eval_num_recommendations, iters_per_epoch, eval_num_users, eval_batch_size):
    """model.py:349-437."""
    engine = self._engine()
    self._order_ahead = None             # (an order drawn ahead belongs to ONE train() call)
    num_batches = len(train_dataloader)
    iters_processed = 0
    if iters_per_epoch is None:
      iters_per_epoch = num_batches
    self._global_step = getattr(self, "_global_step", 0)
    loss_buf = torch.zeros(max(1, min(iters_per_epoch, num_batches)), dtype=torch.float32,
                           device=self.device)
    iterator = None
    self._stop_training = False
    for epoch in range(current_epoch, num_epochs + 1):
      if self._stop_training:
        break
      self.current_epoch = epoch
      self.model.train()
      self._sync_ranges()
      if lr_scheduler is not None:
        lr_scheduler.step()     # at epoch start, as model.py:364-366
      if iters_processed == 0 or iters_processed == num_batches:
        iters_processed = 0
        iterator = enumerate(self._step_generator(train_dataloader), 1)
      iters_to_process = min(iters_per_epoch, num_batches - iters_processed)
      iters_processed += iters_to_process

      if self._graph_ok(train_dataloader, iters_per_epoch, num_batches):
        # whole epochs of the one-call autoencoder step: replayed as HIP graphs (graph.py)
        iters_processed = num_batches
        # the NEXT epoch's user order is drawn while the GPU still works on this one (torch.randperm
        # of 10^5 users costs 2-11 ms of host time, a quarter of a C2 epoch) -- unless something
        # else would draw from the global RNG in between (validation), a hook supplies the order or
        # this is the last epoch: the SEQUENCE of draws stays the reference's
        ahead = (epoch < num_epochs and self.user_order_hook is None and
                 not (eval_freq > 0 and epoch % eval_freq == 0 and val_dataloader is not None))
        losses = self._train_epoch_graph(train_dataloader, draw_next_order=ahead)
        if losses is not None:
          self.last_epoch_losses = losses
          self._epoch_end(epoch, num_epochs, len(self.last_epoch_losses), val_dataloader, eval_freq,
                          metrics, eval_num_recommendations, eval_batch_size, eval_num_users,
                          model_checkpoint_prefix, checkpoint_freq)
          continue
        # (the hook's order is not one pass over the users: eagerly sequenced steps below)
        iterator = enumerate(self._step_generator(train_dataloader), 1)
        iters_to_process = num_batches

      n_done = 0
      for batch_itr, (blk, row_off, rows, keep_noise, keep_drop, tgt_blk) in iterator:
        dp = getattr(self, "_dp", None)
        engine.train_step(blk, row_off, rows, keep_noise, keep_drop,
                          out=loss_buf[n_done:n_done + 1],
                          global_rows=(rows * dp.world if dp is not None else None), tgt=tgt_blk)
        n_done += 1
        self._global_step += 1
        if batch_itr % iters_per_epoch == 0:
          break
      if getattr(self, "_ip", None) is not None and n_done:
        self._ip.allreduce_sum(loss_buf[:n_done])   # every rank holds its items' share
      # one device->host read per epoch instead of loss.item() per step (model.py:404)
      self.last_epoch_losses = loss_buf[:n_done].cpu().numpy().copy()
      self._epoch_end(epoch, num_epochs, n_done, val_dataloader, eval_freq, metrics,
                      eval_num_recommendations, eval_batch_size, eval_num_users,
                      model_checkpoint_prefix, checkpoint_freq)

  def _epoch_end(self, epoch, num_epochs, n_done, val_dataloader, eval_freq, metrics,
                 eval_num_recommendations, eval_batch_size, eval_num_users, model_checkpoint_prefix,
                 checkpoint_freq):
    """model.py:406-437: the epoch's log line, validation / evaluation, checkpoint."""
    # (the stream is drained here: the one place where the blocks' overflow flags are read back)
    for holder in (getattr(self, "_train_pf", None), getattr(self, "_graph_stepper", None)):
      if holder is not None:
        for blk in (b for row in holder.blocks for b in row):
          blk.check()
    self.loss_history.append(self.last_epoch_losses)
    if n_done and not np.all(np.isfinite(self.last_epoch_losses)):
      # the reference keeps training on a NaN loss too (the split-fp16 decoder GEMMs take their
      # operand scales from device-side maxima, recoder_amd/csrc/gemm.hip, so they add no way to
      # get one that fp32 does not have)
      log.warning("non-finite training loss in epoch %d", epoch)
    postfix = {"loss": float(self.last_epoch_losses[-1]) if n_done else float("nan")}
    if eval_freq > 0 and epoch % eval_freq == 0 and val_dataloader is not None:
      self._sync_user_rows()
      postfix["val_loss"] = self._validate(val_dataloader)
      if metrics is not None and eval_num_recommendations is not None:
        results = self._evaluate(val_dataloader.dataset,
                                 num_recommendations=eval_num_recommendations,
                                 metrics=metrics, batch_size=eval_batch_size,
                                 num_users=eval_num_users)
        for metric in results:
          postfix[str(metric)] = np.mean(results[metric])
    log.info("Epoch {}/{} {}".format(epoch, num_epochs, postfix))
    self.last_epoch_summary = postfix
    if model_checkpoint_prefix and \
        ((checkpoint_freq > 0 and epoch % checkpoint_freq == 0) or epoch == num_epochs):
      self._sync_user_rows()
      # multi-GPU training (every rank is here): the replicas are identical -- one writer
      # (concurrent writers of one path can truncate each other on a shared filesystem), the
      # others wait for the file
      import torch.distributed as dist
      # (also the replicated fallback of _setup_data_parallel -- generic engine, target matrix --
      # where neither _dp nor _ip is set: every process of an initialised group is here)
      multi = getattr(self, "_dp_override", None) is None and getattr(self, "_ip_override", None) is None and \
          dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
      if not multi or dist.get_rank() == 0:
        self.save_state(model_checkpoint_prefix)
      if multi:
        dist.barrier()

  # ------------------------------------------------------------ graph replay
  def _graph_ok(self, dataloader, iters_per_epoch, num_batches):
    if os.environ.get("RK_GRAPH", "1") == "0":
      return False
    eng = self._engine()
    if getattr(eng, "generic", False):
      return False
    if getattr(self, "_ip", None) is not None:
      return False
    dp = getattr(self, "_dp", None)
    if dp is not None:
      # users-DP replays too when its collectives are our own in-order RCCL calls (capturable) and
      # the step is the one-call autoencoder step; injected collectives (virtual ranks in tests),
      # torch.distributed / gloo and the entry-by-entry engines keep the eager sequencing
      # (round 5: so do the entry-by-entry engines -- MatrixFactorization, hidden stacks -- with the
      # replicated update; owned-row SparseAdam exchanges row counts only the host knows: eager)
      if not (dp.direct and not dp.virtual and not getattr(eng, "owned_rows", False) and
              self.graph_group <= 8):
        return False
    ds = dataloader.dataset
    return (self.mask_hook is None and dataloader.num_sampling_users == dataloader.batch_size and
            ds.device_target_csr() is None and iters_per_epoch == num_batches and
            len(ds) >= dataloader.batch_size)

  def _train_epoch_graph(self, dataloader, draw_next_order=False):
    """One pass over the dataset with the whole-batch steps replayed as HIP graphs
    (recoder_amd/graph.py); returns the per-step losses."""
    from .graph import GraphStepper
    eng = self._engine()
    ds = dataloader.dataset
    dcsr = ds.device_csr()
    B, ns, n = dataloader.batch_size, dataloader.negative_sampling, len(ds)
    gs = getattr(self, "_graph_stepper", None)
    G = max(1, min(self.graph_group, n // B))
    if gs is None or gs.eng is not eng or gs.dcsr is not dcsr or gs.B != B or gs.ns != ns or gs.G != G or \
        gs.dp is not getattr(eng, "allreduce", None) or \
        (not gs.c_step and set(gs.slots) != set(eng.states)):
      if gs is not None:
        gs.close()
      gs = GraphStepper(eng, dcsr, lambda: self._make_block(dcsr, B, ns, train=True), B, ns, G, n,
                        self.device)
      self._graph_stepper = gs
    order = None
    ahead, self._order_ahead = getattr(self, "_order_ahead", None), None
    if self.user_order_hook is not None:
      order = self.user_order_hook(self.current_epoch, n)
    elif ahead is not None and ahead[0] == n:
      order = ahead[1]                     # drawn at the end of the previous epoch
    if order is None:
      order = epoch_user_order(n)
    order = np.ascontiguousarray(order, dtype=np.int64)
    n_draw = n
    if getattr(self, "_dp", None) is not None and order.shape == (n,):
      n = self._dp_users_per_epoch         # equal step counts on every rank (as _step_generator)
      order = np.ascontiguousarray(order[:n])
    if order.shape != (n,):
      # a hook may hand over any list of users (a subset, repeats): the replayed graphs are laid out
      # for one pass over the n users -- this epoch takes the eagerly sequenced path with the SAME
      # order (the hook is not asked twice)
      self._pending_order = order
      return None
    caller = torch.cuda.current_stream()
    gs.main.wait_stream(caller)
    with torch.cuda.stream(gs.main):
      losses = self._run_epoch_graph(gs, eng, dcsr, order, n, B, draw_next_order, n_draw)
    caller.wait_stream(gs.main)
    return losses

  def _run_epoch_graph(self, gs, eng, dcsr, order, n, B, draw_next_order=False, n_draw=None):
    n_full = gs.begin_epoch(order, eng.rng_step)
    n_total = n_full + (1 if n % B else 0)
    g0 = self._global_step
    # (a mark is called BEFORE step m is enqueued: one that coincides with the end of this epoch
    # belongs to the first step of the next one -- as in the eager path)
    marks = sorted(m - g0 for m in self.step_marks if g0 <= m < g0 + n_total)
    # (looked up per group: a step mark may install / change the engine's time plan)
    plan = lambda i: eng.time_plan is not None and eng.time_plan(i) is not None
    pos, stopped = 0, False
    for m in marks + [None]:
      target = n_full if m is None else min(m, n_full)
      if target > pos:
        gs.run(target - pos, plan)
        self._global_step += target - pos
        pos = target
      if m is None or m > n_full:
        break                            # (a mark behind the ragged step is handled below)
      gs.cut()                           # nothing of the steps after the mark may be in flight
      if self.step_marks[g0 + m]():
        self._stop_training = stopped = True
        break
    losses = gs.losses(pos)
    if not stopped and n % B:
      # the ragged last batch: eager, host-provided arguments (its own block)
      users = torch.from_numpy(order[n_full * B:]).to(self.device)
      dp = getattr(self, "_dp", None)
      out = torch.zeros(1, dtype=torch.float32, device=self.device)
      if dp is not None:
        dp.collate(gs.tail_blk, dcsr, users)
        eng.train_step(gs.tail_blk, 0, int(users.numel()), out=out,
                       global_rows=int(users.numel()) * dp.world)
      else:
        gs.tail_blk.collate(dcsr, users)
        eng.train_step(gs.tail_blk, 0, int(users.numel()), out=out)
      self._global_step += 1
      losses = torch.cat([losses, out])
    if draw_next_order and not self._stop_training:
      n_draw = n if n_draw is None else n_draw
      self._order_ahead = (n_draw, epoch_user_order(n_draw))      # (everything of this epoch is enqueued)
    return losses.cpu().numpy().copy()

  def _validate(self, val_dataloader):
    """model.py:439-452: mean over batches of the eval-mode loss; input and
    target are collated independently (different item sets)."""
    self.model.eval()
    engine = self._engine()
    ds = val_dataloader.dataset
    dcsr = ds.device_csr()
    dcsr_t = ds.device_target_csr()
    B, S = val_dataloader.batch_size, val_dataloader.num_sampling_users
    ns = val_dataloader.negative_sampling
    blk = self._make_block(dcsr, S, ns)
    blk_t = self._make_block(dcsr_t, S, ns) if dcsr_t is not None else None
    n = len(ds)
    order = None
    if self.user_order_hook is not None:
      order = self.user_order_hook(-1, n)
    if order is None:
      order = epoch_user_order(n)
    order_dev = torch.from_numpy(np.ascontiguousarray(order, dtype=np.int64)).to(self.device)
    nb = int(np.ceil(n / B)) if n else 1
    buf = torch.zeros(max(1, nb), dtype=torch.float32, device=self.device)
    k = 0
    for off in range(0, n, S):
      users = order_dev[off:off + S]
      blk.collate(dcsr, users)
      if blk_t is not None:
        blk_t.collate(dcsr_t, users)
      Sg = int(users.numel())
      for r in range(0, Sg, B):
        rows = min(B, Sg - r)
        engine.compute_loss(blk, r, rows, tgt=blk_t, out=buf[k:k + 1])
        k += 1
    total = float(buf[:k].double().sum().item()) if k else 0.0
    return total / max(1, k)

  # -------------------------------------------------------------- inference
  def predict(self, users_interactions, return_input=False):
    """model.py:487-511.  Returns ``(output, input_dense)`` -- like the
    reference, always a tuple (its ``return output, input if .. else output``
    precedence quirk); when return_input is False the second element is the
    output again."""
    if self.model is None:
      raise Exception("Model not initialized.")
    self.model.eval()
    self._check_ranges()
    out, blk, B = self._predict_scores(users_interactions)
    if return_input:
      m = users_interactions.interactions_matrix
      dense = torch.from_numpy(np.asarray(m.todense(), dtype=np.float32)).to(self.device)
      return out, dense
    return out, out

  def _input_block(self, users_interactions):
    """The users' interactions as an unsampled device block (bitmap over the whole catalogue),
    in buffers that are kept and reused across predict / recommend calls."""
    m = users_interactions.interactions_matrix
    B = m.shape[0]
    n_items = self.num_items if self.num_items is not None else m.shape[1]
    dcsr = DeviceCSR(m, self.device)
    assert dcsr.n_items <= n_items
    if dcsr.n_items != n_items:
      dcsr.shape = (B, n_items)
    ws = getattr(self, "_eval_ws", None)
    if ws is None or ws["n_items"] != n_items:
      ws = self._eval_ws = dict(n_items=n_items, blk=None, strips={}, scores=None, cand=None)
    blk = ws["blk"]
    if blk is None or blk.S_cap < B or blk.nnz_cap < max(1, dcsr.nnz):
      blk = ws["blk"] = Block(max(B, blk.S_cap if blk else 0),
                              max(1, dcsr.nnz, blk.nnz_cap if blk else 0), n_items, self.device,
                              negative_sampling=False, need_bits_cr=False)
    rows = torch.arange(B, dtype=torch.int64, device=self.device)
    blk.collate(dcsr, rows, negative_sampling=False)
    # MF looks user rows up by their global ids
    blk.users = torch.as_tensor(np.asarray(users_interactions.users), dtype=torch.int64) \
        .to(self.device)
    return blk, B, n_items

  def _predict_scores(self, users_interactions):
    engine = self._engine()
    blk, B, n_items = self._input_block(users_interactions)
    ld = blk.ld_cap
    out = torch.empty(B, ld, dtype=torch.float32, device=self.device)
    engine.predict_scores(blk, 0, B, out, ld, blk)
    return out[:, :n_items], blk, B

  def _evaluate(self, eval_dataset, num_recommendations, metrics, batch_size=1, num_users=None):
    """model.py:513-523."""
    if self.model is None:
      raise Exception("Model not initialized")
    self.model.eval()
    recommender = InferenceRecommender(self, num_recommendations)
    evaluator = RecommenderEvaluator(recommender, metrics)
    return evaluator.evaluate(eval_dataset, batch_size=batch_size, num_users=num_users)

  # items decoded at a time by recommend(): [B, strip] fp32 scores stay cache-resident
  # (B = 500: 128 MB) instead of a [B, n_items] matrix in HBM (2 GB at C5's 1 M items)
  eval_strip_items = 65536

  def recommend(self, users_interactions, num_recommendations):
    """model.py:525-544: scores with the seen (positive) items at -inf, top-k sorted; a list of
    lists like the reference's (``recommend_array``: the same as one [users, k] int64 array)."""
    return self.recommend_array(users_interactions, num_recommendations).tolist()

  def recommend_array(self, users_interactions, num_recommendations):
    """recommend() as a numpy array (what the evaluator consumes: 50 k Python ints per batch of
    500 users cost more than scoring them).

    The fused engines never materialise the [B, n_items] score matrix: the catalogue is decoded
    in strips of ``eval_strip_items`` items, each strip's masked top k is kept
    (``rk_topk_masked`` with a column offset) and the per-strip winners are merged with one more top-k pass --
    ties resolve to the lower item id at both levels, as torch.topk on the full row would."""
    self.model.eval()
    self._check_ranges()
    k = int(num_recommendations)
    from . import _lib
    from .device import current_stream
    lib = _lib.load()
    engine = self._engine()
    if getattr(engine, "generic", False) or k > lib.rk_topk_max_k():
      return self._recommend_dense(users_interactions, k)
    blk, B, n_items = self._input_block(users_interactions)
    ws = self._eval_ws
    z = engine.encode_eval(blk, 0, B)
    strip = max(k, min(n_items, int(self.eval_strip_items)))
    bounds = [(lo, min(n_items, lo + strip)) for lo in range(0, n_items, strip)]
    if len(bounds) > 1 and bounds[-1][1] - bounds[-1][0] < k:      # a last strip shorter than k:
      lo0 = bounds[-2][0]                                         # merge it into the one before
      bounds = bounds[:-2] + [(lo0, n_items)]
    ns = len(bounds)
    if ns > 1 and os.environ.get("RK_EVAL_FUSED", "1") != "0" and hasattr(engine, "recommend_fused"):
      # catalogues of more than one strip: the top-k filter rides in the decode's epilogue (no score
      # matrix, no passes over it): a strided sample of the catalogue bounds every row's k-th best
      # score from below, the decode over all items keeps only what reaches the bound
      # (engine.recommend_fused / include/recoder_hip.h "The fused form") -- the same ids, ties included
      res = engine.recommend_fused(blk, B, k, n_items)
      if res is not None:
        out, status = res
        host = torch.cat([out.reshape(-1), status.to(torch.int64)]).cpu().numpy()
        if host[-1] == 0:
          self.eval_fused_batches = getattr(self, "eval_fused_batches", 0) + 1
          return host[:-1].reshape(B, k).copy()
        # (a candidate list overflowed, or a row has fewer than k unseen items: strip by strip)
    width = max(hi - lo for lo, hi in bounds)
    ld = -(-width // 32) * 32
    if ws["scores"] is None or ws["scores"].numel() < B * ld:
      ws["scores"] = torch.empty(B * ld, dtype=torch.float32, device=self.device)
    if ws["cand"] is None or ws["cand"][0].shape[0] < B or ws["cand"][0].shape[1] != ns * k:
      ws["cand"] = (torch.empty(B, ns * k, dtype=torch.int64, device=self.device),
                    torch.empty(B, ns * k, dtype=torch.float32, device=self.device))
    cand_idx, cand_val = ws["cand"][0][:B], ws["cand"][1][:B]
    scores = ws["scores"]
    for s, (lo, hi) in enumerate(bounds):
      key = (lo, hi, B)
      sblk = ws["strips"].get(key)
      if sblk is None:
        sblk = Block(B, 1, n_items, self.device, negative_sampling=True, need_bits_cr=False,
                     n_cap=hi - lo)
        sblk.set_items(torch.arange(lo, hi, dtype=torch.int32, device=self.device), hi - lo, B)
        if len(ws["strips"]) > 64:
          ws["strips"].clear()
        ws["strips"][key] = sblk
      engine.decode_scores(z, B, sblk, scores, ld)
      _lib.check(lib.rk_topk_masked(scores.data_ptr(), B, hi - lo, ld, blk.ref, 0, k, lo, 1,
                                          cand_idx[:, s * k:].data_ptr(), cand_val[:, s * k:].data_ptr(),
                                          ns * k, current_stream()), "rk_topk_masked")
    if ns == 1:
      return cand_idx[:, :k].cpu().numpy().copy()
    # merge: top k of the ns * k candidates (positions), then their item ids.  Candidates are laid
    # out strip by strip, each sorted by (score desc, id asc): equal scores keep ascending ids
    pos = torch.empty(B, k, dtype=torch.int64, device=self.device)
    _lib.check(lib.rk_topk_masked(cand_val.data_ptr(), B, ns * k, ns * k, None, 0, k, 0, 1, pos.data_ptr(),
                                  None, k, current_stream()), "rk_topk_masked")
    return torch.gather(cand_idx, 1, pos).cpu().numpy()

  def _recommend_dense(self, users_interactions, k):
    """Full score matrix + torch.topk: the generic (torch-autograd) engine and k above the
    top-k kernel's limit."""
    out, dense = self.predict(users_interactions, return_input=True)
    out = out.clone()
    out[dense > 0] = -float("inf")
    return torch.topk(out, k, dim=1, sorted=True)[1].cpu().numpy()

  def evaluate(self, eval_dataset, num_recommendations, metrics, batch_size=1, num_users=None):
    """model.py:546-559."""
    results = self._evaluate(eval_dataset, num_recommendations, metrics, batch_size=batch_size,
                             num_users=num_users)
    for metric in results:
      log.info("{}: {}".format(metric, np.mean(results[metric])))
    return results

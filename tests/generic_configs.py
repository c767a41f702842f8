"""Configurations of the generic-path golden vectors (shared by the generator
and the tests)."""
_DATA = dict(n_users=120, n_items=100, mean_deg=8)

GENERIC_CONFIGS = {
  "ae_sgd": dict(kind="ae", model=dict(hidden_layers=[16], activation_type="tanh"), loss="mse",
                 optimizer="sgd",
                 train=dict(batch_size=32, lr=1e-2, weight_decay=1e-5, num_epochs=2,
                            negative_sampling=True),
                 data=dict(seed=31, **_DATA)),
  "mf_rmsprop": dict(kind="mf", model=dict(embedding_size=12, activation_type="tanh"),
                     loss="logistic", optimizer="rmsprop",
                     train=dict(batch_size=32, lr=1e-3, weight_decay=1e-5, num_epochs=2,
                                negative_sampling=True),
                     data=dict(seed=32, **_DATA)),
  "ae2_smoothl1_adagrad": dict(kind="ae", model=dict(hidden_layers=[16, 8], activation_type="relu"),
                               loss="smooth_l1_sum", optimizer="adagrad",
                               train=dict(batch_size=32, lr=1e-2, weight_decay=0.0, num_epochs=2,
                                          negative_sampling=False),
                               data=dict(seed=33, **_DATA)),
  # an activation the fused kernels do not implement (any torch.<name> is legal, reference
  # nn.py:6-9) and a first hidden size that is not a multiple of 4: both route to the generic path
  "ae_erf_h30_adam": dict(kind="ae", model=dict(hidden_layers=[30], activation_type="erf", sparse=True),
                          loss="mse", optimizer="adam",
                          train=dict(batch_size=32, lr=1e-3, weight_decay=1e-5, num_epochs=2,
                                     negative_sampling=True),
                          data=dict(seed=35, **_DATA)),
  "twotower_adam": dict(kind="custom", model=dict(d=8), loss="logistic", optimizer="adam",
                        train=dict(batch_size=32, lr=1e-3, weight_decay=1e-5, num_epochs=2,
                                   negative_sampling=True),
                        data=dict(seed=34, **_DATA)),
}

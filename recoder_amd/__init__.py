"""recoder_amd -- MI355X-native hot path of amoussawi/recoder (reference v0.4.0).

Drop-in for the reference's training path: ``Recoder`` / ``FactorizationModel``
/ losses / sparse-batch data classes keep their API; underneath, a thin C ABI
(include/recoder_hip.h, recoder_amd/csrc) of hand-written gfx950 HIP kernels.
"""
# value of the reference's recoder.__version__ (recoder/__init__.py:1); stored
# in checkpoints as 'recoder_version' (model.py:207)
__version__ = "0.4.0"

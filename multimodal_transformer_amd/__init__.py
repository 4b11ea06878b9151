"""MI355X-native implementation of the attention hot path of frankaging/Multimodal-Transformer.

``multiTransformer`` mirrors the reference module of the same name; ``functional`` holds the
autograd bindings of the HIP entry points declared in ``include/mmt_hip.h``.
"""
from . import _lib, batching, functional, graphs, models, multiTransformer, optim  # noqa: F401
from .metrics import batched_ccc, eval_ccc  # noqa: F401

__all__ = ["functional", "multiTransformer", "models", "batching", "optim", "graphs", "eval_ccc", "batched_ccc"]

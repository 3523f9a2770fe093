"""MI355X-native drop-in for the batched prefix-scoring training path of open_knowledge_graph_embeddings.

Host code mirrors the reference's plugin interface (``model.Models``, ``RelationScorer`` /
``RelationEmbedder`` method names, ``trainer.AddLossModule``); the arithmetic runs in hand-written
HIP kernels for gfx950 behind the C ABI declared in ``include/okge.h``.
"""
from . import _native  # noqa: F401
from ._native import OkgeError, build_native  # noqa: F401

from . import model, token_pooled  # noqa: F401,E402  (token_pooled registers its classes in model.Models)
from . import optim  # noqa: F401,E402  (registers OkgeAdagrad in torch.optim for the reference's OptimRegime)

__all__ = ["OkgeError", "build_native"]

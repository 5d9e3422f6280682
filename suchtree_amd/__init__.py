"""suchtree_amd -- MI355X-native bulk patristic-distance / MRCA path behind
the SuchTree class surface (reference: /root/reference/SuchTree/__init__.py)."""
from .exceptions import (SuchTreeError, NodeNotFoundError, InvalidNodeError,
                         TreeStructureError, HipBackendError)

__version__ = "0.1.0"

__all__ = ["SuchTree", "SuchLinkedTrees", "SuchTreeError", "NodeNotFoundError",
           "InvalidNodeError", "TreeStructureError", "HipBackendError", "__version__"]


def __getattr__(name):
    # lazy: importing the package must not need the HIP library
    if name == "SuchTree":
        from .suchtree import SuchTree
        return SuchTree
    if name == "SuchLinkedTrees":
        from .linked import SuchLinkedTrees
        return SuchLinkedTrees
    raise AttributeError(name)

"""Exception types of the SuchTree class surface.

Same names, constructor arguments, attributes and messages as the reference
(/root/reference/SuchTree/exceptions.py:2-38) so that callers' ``except``
clauses and message checks keep working.
"""


class SuchTreeError(Exception):
    """Base class for every error raised by this package."""


class NodeNotFoundError(SuchTreeError):
    """A leaf name (or node) is not present in the tree."""

    def __init__(self, node, message=None):
        if message is None:
            if isinstance(node, str):
                message = "Leaf name not found: %s." % str(node)
            else:
                message = "Node not found: %s" % str(node)
        super().__init__(message)
        self.node = node


class InvalidNodeError(SuchTreeError):
    """A node id is out of bounds or of the wrong kind."""

    def __init__(self, node_id, tree_size=None, message=None):
        if message is None:
            if tree_size is not None:
                message = "Node ID %s out of bounds (tree size: %s)" % (str(node_id), str(tree_size))
            else:
                message = "Invalid node ID: %s" % str(node_id)
        super().__init__(message)
        self.node_id = node_id
        self.tree_size = tree_size


class TreeStructureError(SuchTreeError):
    """The tree structure is invalid or inconsistent."""


class HipBackendError(SuchTreeError):
    """The HIP library is missing, failed to load, or a HIP call failed.

    Not part of the reference surface: the reference has no device.  Raised
    loudly instead of falling back to any CPU path.
    """

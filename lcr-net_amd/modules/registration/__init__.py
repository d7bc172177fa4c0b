from .matching import get_node_correspondences  # noqa: F401

"""Operator API of the hot path — same names/signatures as the reference's experiments/lcrnet/modules/ops."""
from .grid_subsample import grid_subsample  # noqa: F401
from .radius_search import radius_search, radius_count  # noqa: F401

"""Operator API of the hot path — same names/signatures as the reference's experiments/lcrnet/modules/ops."""
from .grid_subsample import grid_subsample, grid_subsample_device  # noqa: F401
from .radius_search import radius_search, radius_search_deferred, finish_deferred, radius_count, SupportGrid  # noqa: F401

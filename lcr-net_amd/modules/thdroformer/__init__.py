from .thdroformer_linear import ThDRoFormer  # noqa: F401

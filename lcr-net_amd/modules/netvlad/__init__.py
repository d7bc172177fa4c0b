from .NetVlad import NetVLADLoupe2, GatingContext  # noqa: F401

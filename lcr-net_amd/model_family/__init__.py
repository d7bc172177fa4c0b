from .LCRNet_GlobalDescrition import LCRNet_GlobalDescrition, create_model  # noqa: F401
from .LCRNet import LCRNet  # noqa: F401
from . import LCRNet_Matching, LCRNet_Matching_infer  # noqa: F401  (both define a class named LCRNet_Matching, like the reference)

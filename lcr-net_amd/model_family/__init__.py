from .LCRNet_GlobalDescrition import LCRNet_GlobalDescrition, create_model  # noqa: F401
from .LCRNet import LCRNet  # noqa: F401

from .kpconv import KPConv  # noqa: F401
from .modules import ConvBlock, GroupNorm, ResidualBlock, UnaryBlock, LastUnaryBlock, StageContext  # noqa: F401
from .functional import maxpool, nearest_upsample  # noqa: F401

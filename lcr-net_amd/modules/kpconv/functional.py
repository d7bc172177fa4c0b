"""kpconv/functional.py of the reference (maxpool :54-67, nearest_upsample :6-22) on the HIP path."""
import torch

from ... import functional as F


def maxpool(x, neighbor_indices):
    """Max pooling from neighbours; the zero shadow row takes part (functional.py:54-67)."""
    return F.maxpool(x.contiguous(), neighbor_indices.contiguous())


def nearest_upsample(x, upsample_indices):
    """Closest-neighbour feature pull: only column 0 is used (functional.py:6-22); a 1-column maxpool is the same op
    except for negative features under a shadow index, so index explicitly."""
    x = torch.cat((x, torch.zeros_like(x[:1, :])), 0)
    return x.index_select(0, upsample_indices[:, 0].long())

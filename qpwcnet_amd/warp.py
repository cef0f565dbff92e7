"""Function-level mirror of qpwcnet/core/warp.py: ``tf_warp`` and
``dense_image_warp``, on the HIP warp kernel."""
import torch

from . import ops
from .backend import CHANNELS_LAST, image_data_format


def tf_warp(img, flow, data_format=None):
    """qpwcnet/core/warp.py:63-153.  Rank-3 (unbatched) input is an error in the
    reference too (unbound ``is_batch``, warp.py:75-79)."""
    if data_format is None:
        data_format = image_data_format()
    return ops.warp(img, flow, "tfwarp", data_format)


def dense_image_warp(image, flow, name=None):
    """In-tree copy with ``query = grid + flow`` (qpwcnet/core/warp.py:156-211,
    ``+`` at :201): flow[..., 0] displaces rows, flow[..., 1] columns.  NHWC."""
    if image.dim() != 4 or flow.dim() != 4:
        raise ValueError("image and flow must be rank 4")
    if image.shape[1] < 2 or image.shape[2] < 2:
        raise ValueError("Grid must be at least 2x2")  # warp.py:182-184
    return ops.warp(image, torch.flip(flow, dims=(-1,)), "clamp", CHANNELS_LAST)


def tfa_dense_image_warp(image, flow):
    """Upstream sign convention (``query = grid - flow``, docstring warp.py:161-162)."""
    return dense_image_warp(image, -flow)

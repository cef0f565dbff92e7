"""Occlusion map and inverse flow -- mirror of qpwcnet/core/occlusion.py
(``get_spatial_shape`` :9-24, ``estimate_occlusion_map`` :27-118) on the HIP kernels
``qpwc_occlusion_fwd`` / ``qpwc_invert_flow_fwd``.  ``invert_flow`` names the expression
``-tf_warp(flow, flow, data_format)`` that the reference spells inline
(occlusion.py:85; app/test/test_invert_flow.py:47).

Flow convention as everywhere in the reference: channel 0 = x, channel 1 = y,
``prv[i,j] = nxt[i+f[i,j,1], j+f[i,j,0]]`` (occlusion.py:33-34).
"""
from . import ops
from .backend import CHANNELS_FIRST, image_data_format


def get_spatial_shape(x, data_format=None):
    """occlusion.py:9-24 -> {'n','h','w'}.  The reference only handles batched (rank >= 4)
    input -- an unbatched tensor hits an unbound local there (:19-21); here it is a ValueError."""
    if data_format is None:
        data_format = image_data_format()
    if x.dim() < 4:
        raise ValueError("get_spatial_shape needs a batched (rank-4) tensor, got rank {}".format(x.dim()))
    if data_format == CHANNELS_FIRST:
        n, _, h, w = x.shape
    else:
        n, h, w, _ = x.shape
    return {"n": n, "h": h, "w": w}


def invert_flow(flow, data_format=None):
    """inv_flow = -tf_warp(flow, flow, data_format)."""
    if data_format is None:
        data_format = image_data_format()
    return ops.invert_flow(flow, data_format)


def estimate_occlusion_map(flow, data_format=None):
    """(B,H,W) float32: 1 where a pixel of the next frame cannot be determined from the flow
    (it leaves the image, or nothing maps onto it under the inverse flow), else 0."""
    if data_format is None:
        data_format = image_data_format()
    get_spatial_shape(flow, data_format)
    return ops.occlusion_map(flow, data_format)

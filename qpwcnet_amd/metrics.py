"""Per-level end-point error, the reduction the benchmark all-gathers.

EPE definition: ``epe_error`` (qpwcnet/app/optical_flow/train.py:247-253);
per-level ground truth as in ``FlowMseLoss.call`` (qpwcnet/train/loss.py:56-62):
bilinear resize of the full-resolution flow to (h, w), times h/H.
"""
import torch
import torch.nn.functional as F

from . import ops
from .backend import CHANNELS_LAST


def multiscale_ground_truth(flow_gt, shapes, data_format=CHANNELS_LAST):
    """flow_gt (B,H,W,2) -> list of (B,h,w,2), one per (h,w) in shapes."""
    x = flow_gt.permute(0, 3, 1, 2) if data_format == CHANNELS_LAST else flow_gt
    H = x.shape[2]
    out = []
    for (h, w) in shapes:
        y = F.interpolate(x, size=(h, w), mode="bilinear", align_corners=False) * (h / H)
        out.append(y.permute(0, 2, 3, 1).contiguous() if data_format == CHANNELS_LAST
                   else y.contiguous())
    return out


def per_level_epe(flows_true, flows_pred, data_format=CHANNELS_LAST, out=None):
    """-> float32 tensor [n_levels] on the flows' device (HIP reduction kernel); written into ``out``
    when given (the all-gather payload, see qpwcnet_amd.dist.EpeGather.payload_view)."""
    if len(flows_true) <= 8:
        return ops.epe_multi(flows_true, flows_pred, out=out, data_format=data_format)  # two launches
    res = torch.stack([ops.epe(t, p.float(), data_format) for t, p in zip(flows_true, flows_pred)])
    if out is not None:
        out.copy_(res)
        return out
    return res

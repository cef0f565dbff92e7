"""torch-CPU restatement of the reference hot path, op for op (test
infrastructure only; also the timed "reference-algorithm CPU restatement" of
bench.py's cpu_baseline leg, since TF2 itself cannot run offline).

PARITY UNPINNED -- see ``oracle/__init__.py``.  Works in float32 or float64
(dtype of the inputs).  NHWC ('channels_last') tensors.
"""
import torch
import torch.nn.functional as F


def cost_volume(prv, nxt, search_range=4):
    """``CostVolume.call`` -- qpwcnet/core/layers.py:72-100: zero-pad, 81 x
    (slice, multiply, mean over C), concat, LeakyReLU(0.1)."""
    r = search_range
    d = 2 * r + 1
    h, w = prv.shape[1], prv.shape[2]
    pad_nxt = F.pad(nxt, (0, 0, r, r, r, r))                        # layers.py:50-51,77
    cost_vol = []
    for i0 in range(d):                                              # layers.py:80
        for j0 in range(d):                                          # layers.py:81
            roi = pad_nxt[:, i0:i0 + h, j0:j0 + w, :]                # layers.py:86-88
            cost_vol.append(torch.mean(prv * roi, dim=3, keepdim=True))  # layers.py:94
    cost_vol = torch.cat(cost_vol, dim=3)                            # layers.py:96
    return F.leaky_relu(cost_vol, 0.1)                               # layers.py:99


def _interpolate_bilinear(grid, query):
    """tfa ``interpolate_bilinear`` (indexing 'ij'), called at warp.py:207."""
    B, H, W, C = grid.shape
    alphas, floors, ceils = [], [], []
    for dim, size in ((0, H), (1, W)):
        q = query[..., dim]
        floor = torch.clamp(torch.floor(q), 0.0, float(size - 2))
        int_floor = floor.to(torch.int64)
        floors.append(int_floor)
        ceils.append(int_floor + 1)
        alphas.append(torch.clamp(q - floor, 0.0, 1.0).unsqueeze(-1))
    flat = grid.reshape(B * H * W, C)
    boff = (torch.arange(B) * H * W).reshape(B, 1)

    def gather(y, x):
        return flat[(boff + y * W + x).reshape(-1)].reshape(B, -1, C)

    tl = gather(floors[0], floors[1])
    tr = gather(floors[0], ceils[1])
    bl = gather(ceils[0], floors[1])
    br = gather(ceils[0], ceils[1])
    top = alphas[1] * (tr - tl) + tl
    bot = alphas[1] * (br - bl) + bl
    return alphas[0] * (bot - top) + top


def warp_v2(img, flo):
    """``WarpV2.call`` -- qpwcnet/core/layers.py:177-186:
    tfa.image.dense_image_warp(img, -flo[..., ::-1]) (algorithm: warp.py:156-211)."""
    B, H, W, C = img.shape
    flow = -torch.flip(flo, dims=(-1,))
    gy, gx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    grid = torch.stack([gy, gx], dim=2).to(flow.dtype).unsqueeze(0)
    query = (grid - flow).expand(B, H, W, 2).reshape(B, H * W, 2)
    return _interpolate_bilinear(img, query).reshape(B, H, W, C)


def tf_warp(img, flow):
    """``tf_warp`` -- qpwcnet/core/warp.py:63-153."""
    B, H, W, C = img.shape
    gy, gx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    x = gx.to(flow.dtype).unsqueeze(0) + flow[..., 0]
    y = gy.to(flow.dtype).unsqueeze(0) + flow[..., 1]
    x, y = x.expand(B, H, W), y.expand(B, H, W)
    x0 = x.to(torch.int32).to(torch.int64)                           # truncation, warp.py:115
    y0 = y.to(torch.int32).to(torch.int64)
    x1, y1 = x0 + 1, y0 + 1
    x0, x1 = x0.clamp(0, W - 1), x1.clamp(0, W - 1)                  # warp.py:121-124
    y0, y1 = y0.clamp(0, H - 1), y1.clamp(0, H - 1)
    b = torch.arange(B).reshape(B, 1, 1)
    Ia, Ib, Ic, Id = img[b, y0, x0], img[b, y1, x0], img[b, y0, x1], img[b, y1, x1]
    x0f, x1f, y0f, y1f = (t.to(flow.dtype) for t in (x0, x1, y0, y1))
    wa = ((x1f - x) * (y1f - y)).unsqueeze(-1)                       # warp.py:139-142
    wb = ((x1f - x) * (y - y0f)).unsqueeze(-1)
    wc = ((x - x0f) * (y1f - y)).unsqueeze(-1)
    wd = ((x - x0f) * (y - y0f)).unsqueeze(-1)
    return wa * Ia + wb * Ib + wc * Ic + wd * Id                     # warp.py:151


def epe_error(y_true, y_pred):
    """qpwcnet/app/optical_flow/train.py:247-253 (channels_last)."""
    return torch.linalg.vector_norm(y_true - y_pred, ord=2, dim=-1).mean()


def mish(x):
    """qpwcnet/core/mish.py:27-28."""
    return x * torch.tanh(F.softplus(x))


def depthwise3x3(sources, weight, mish_on_load=False):
    """Depthwise half of SeparableConv2D(3x3,'same') (qpwcnet/core/non_layers.py:223-231)
    on concat(sources) (non_layers.py:336-338): NHWC in, NHWC out; weight (C,1,3,3)."""
    x = torch.cat(list(sources), dim=3)
    if mish_on_load:
        x = mish(x)
    y = F.conv2d(x.permute(0, 3, 1, 2), weight.reshape(-1, 1, 3, 3), None, padding=1, groups=x.shape[3])
    return y.permute(0, 2, 3, 1)


def flow_head(z, w1, b1, gamma, beta, mean, var, eps, wf, scale):
    """Tail of OptFlow.__call__ (qpwcnet/core/non_layers.py:238-254, 268-273) on the
    pre-activation output z (B,H,W,16) of the last pointwise conv."""
    x = mish(z).permute(0, 3, 1, 2)
    x = mish(F.conv2d(x, w1, b1))
    x = (x - mean.view(1, -1, 1, 1)) / torch.sqrt(var.view(1, -1, 1, 1) + eps) * gamma.view(1, -1, 1, 1) \
        + beta.view(1, -1, 1, 1)
    f = F.conv2d(x, wf, None, padding=1)
    return (scale * f).permute(0, 2, 3, 1)

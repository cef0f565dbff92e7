"""numpy restatement of the reference hot path (test infrastructure only).

Every function follows the reference op for op and cites the lines it
restates (paths relative to the reference checkout).  The arithmetic dtype is
the dtype of the inputs: feed float32 for a "what TF2-CPU computes" answer,
float64 for a higher-precision answer.

PARITY UNPINNED: see ``oracle/__init__.py`` -- the reference holds no golden
vectors and TensorFlow is not installable offline; these functions are pinned by
the reference's in-tree source and by analytic known answers only.
"""
import numpy as np

CHANNELS_LAST = "channels_last"
CHANNELS_FIRST = "channels_first"


def get_axis(data_format):
    """``_get_axis`` -- qpwcnet/core/layers.py:19-29."""
    if data_format == CHANNELS_FIRST:
        return 1
    if data_format == CHANNELS_LAST:
        return 3
    raise ValueError("Unsupported data format : {}".format(data_format))


def leaky_relu(x, alpha=0.1):
    """``tf.nn.leaky_relu(x, 0.1)`` -- qpwcnet/core/layers.py:15-16,99."""
    return np.where(x > 0, x, x * np.asarray(alpha, dtype=x.dtype))


def cost_volume(prv, nxt, search_range=4, data_format=CHANNELS_LAST, activation=True):
    """``CostVolume.call`` -- qpwcnet/core/layers.py:72-100
    (twin: qpwcnet/core/non_layers.py:72-104).

    ``CostVolumeV2`` (layers.py:128-132) is the same function by the reference's
    own invariant (app/test/test_cvol_equal.py:25); its own arithmetic (the tfa
    CorrelationCost op) is restated independently in ``oracle/tfa_ref.py``.
    ``activation=False`` stops before the LeakyReLU of layers.py:99 (what the tfa op
    itself returns, for comparing the two restatements).
    """
    prv = np.asarray(prv)
    nxt = np.asarray(nxt)
    axis = get_axis(data_format)
    r = search_range
    d = r * 2 + 1
    if data_format == CHANNELS_FIRST:
        h, w = prv.shape[2], prv.shape[3]
        pad_nxt = np.pad(nxt, ((0, 0), (0, 0), (r, r), (r, r)))  # layers.py:50-51,77
    else:
        h, w = prv.shape[1], prv.shape[2]
        pad_nxt = np.pad(nxt, ((0, 0), (r, r), (r, r), (0, 0)))
    cost_vol = []
    for i0 in range(0, d):          # layers.py:80  (row offset, outer)
        for j0 in range(0, d):      # layers.py:81  (col offset, inner)
            if data_format == CHANNELS_FIRST:
                roi = pad_nxt[:, :, i0:i0 + h, j0:j0 + w]      # layers.py:83-85
            else:
                roi = pad_nxt[:, i0:i0 + h, j0:j0 + w, :]      # layers.py:86-88
            cost = np.mean(prv * roi, axis=axis, keepdims=True, dtype=prv.dtype)  # :94
            cost_vol.append(cost)
    cost_vol = np.concatenate(cost_vol, axis=axis)              # layers.py:96
    if not activation:
        return cost_vol
    return leaky_relu(cost_vol, 0.1)                            # layers.py:99


def _to_nhwc(x, data_format):
    if data_format == CHANNELS_FIRST:
        return np.transpose(x, (0, 2, 3, 1))
    return x


def _from_nhwc(x, data_format):
    if data_format == CHANNELS_FIRST:
        return np.transpose(x, (0, 3, 1, 2))
    return x


def _gather_yx(img_nhwc, y, x):
    """``get_pixel_value`` -- qpwcnet/core/warp.py:8-47 (gather_nd, batch_dims=1)."""
    b = np.arange(img_nhwc.shape[0]).reshape(-1, 1, 1)
    return img_nhwc[b, y, x]


def tf_warp(img, flow, data_format=CHANNELS_LAST):
    """``tf_warp`` -- qpwcnet/core/warp.py:63-153 (``Warp.call``, layers.py:166-168).

    Unbatched (rank-3) input is an error in the reference (unbound ``is_batch``,
    warp.py:75-79); restated as ValueError.
    """
    img = np.asarray(img)
    flow = np.asarray(flow)
    if flow.ndim < 4 or img.ndim < 4:
        raise ValueError("tf_warp requires batched rank-4 inputs (warp.py:75-79)")
    dt = flow.dtype
    img_l = _to_nhwc(img, data_format)
    flow_l = _to_nhwc(flow, data_format)
    B, H, W, _ = img_l.shape
    gx, gy = np.meshgrid(np.arange(W), np.arange(H))            # warp.py:87
    gx = gx.astype(dt)[None]
    gy = gy.astype(dt)[None]
    flow_l = np.broadcast_to(flow_l, (B, H, W, 2))
    x = gx + flow_l[..., 0]                                     # warp.py:100-111
    y = gy + flow_l[..., 1]
    max_y = H - 1
    max_x = W - 1
    x0 = x.astype(np.int32)                                     # warp.py:115 (truncation)
    x1 = x0 + 1
    y0 = y.astype(np.int32)                                     # warp.py:117
    y1 = y0 + 1
    x0 = np.clip(x0, 0, max_x)                                  # warp.py:121-124
    x1 = np.clip(x1, 0, max_x)
    y0 = np.clip(y0, 0, max_y)
    y1 = np.clip(y1, 0, max_y)
    Ia = _gather_yx(img_l, y0, x0)                              # warp.py:127-130
    Ib = _gather_yx(img_l, y1, x0)
    Ic = _gather_yx(img_l, y0, x1)
    Id = _gather_yx(img_l, y1, x1)
    x0f = x0.astype(dt)                                         # warp.py:133-136
    x1f = x1.astype(dt)
    y0f = y0.astype(dt)
    y1f = y1.astype(dt)
    wa = (x1f - x) * (y1f - y)                                  # warp.py:139-142
    wb = (x1f - x) * (y - y0f)
    wc = (x - x0f) * (y1f - y)
    wd = (x - x0f) * (y - y0f)
    wa, wb, wc, wd = (w[..., None].astype(img_l.dtype) for w in (wa, wb, wc, wd))
    out = wa * Ia + wb * Ib + wc * Ic + wd * Id                 # warp.py:151 (add_n)
    return _from_nhwc(out, data_format)


def interpolate_bilinear(grid, query_points):
    """Published algorithm of ``tfa.image.interpolate_bilinear`` (indexing='ij')
    as called from qpwcnet/core/warp.py:207; behaviour documented in-tree at
    warp.py:157-185 (clamp-to-border).  grid (B,H,W,C), query (B,N,2)=(y,x).
    """
    grid = np.asarray(grid)
    q = np.asarray(query_points)
    B, H, W, C = grid.shape
    if H < 2 or W < 2:
        raise ValueError("Grid must be at least 2x2 (warp.py:182-184)")
    dt = q.dtype
    alphas, floors, ceils = [], [], []
    for dim, size in ((0, H), (1, W)):
        queries = q[..., dim]
        max_floor = np.asarray(size - 2, dtype=dt)
        min_floor = np.asarray(0.0, dtype=dt)
        floor = np.minimum(np.maximum(min_floor, np.floor(queries)), max_floor)
        int_floor = floor.astype(np.int32)
        floors.append(int_floor)
        ceils.append(int_floor + 1)
        alpha = (queries - floor).astype(grid.dtype)
        alpha = np.minimum(np.maximum(np.asarray(0.0, grid.dtype), alpha),
                           np.asarray(1.0, grid.dtype))
        alphas.append(alpha[..., None])
    flat = grid.reshape(B * H * W, C)
    boff = (np.arange(B) * H * W).reshape(B, 1)

    def gather(y, x):
        return flat[boff + y * W + x]

    top_left = gather(floors[0], floors[1])
    top_right = gather(floors[0], ceils[1])
    bottom_left = gather(ceils[0], floors[1])
    bottom_right = gather(ceils[0], ceils[1])
    interp_top = alphas[1] * (top_right - top_left) + top_left
    interp_bottom = alphas[1] * (bottom_right - bottom_left) + bottom_left
    return alphas[0] * (interp_bottom - interp_top) + interp_top


def tfa_dense_image_warp(image, flow):
    """Upstream ``tfa.image.dense_image_warp``: query = grid - flow
    (docstring kept in-tree at qpwcnet/core/warp.py:157-185)."""
    image = np.asarray(image)
    flow = np.asarray(flow)
    B, H, W, C = image.shape
    gx, gy = np.meshgrid(np.arange(W), np.arange(H))
    stacked = np.stack([gy, gx], axis=2).astype(flow.dtype)[None]
    query = stacked - flow
    query = np.broadcast_to(query, (B, H, W, 2)).reshape(B, H * W, 2)
    return interpolate_bilinear(image, query).reshape(B, H, W, C)


def dense_image_warp(image, flow):
    """In-tree ``dense_image_warp`` with the sign flipped (query = grid + flow)
    -- qpwcnet/core/warp.py:156-211, ``+`` at :201."""
    return tfa_dense_image_warp(image, -np.asarray(flow))


def warp_v2(img, flo, data_format=CHANNELS_LAST):
    """``WarpV2.call`` -- qpwcnet/core/layers.py:177-186
    (twin: qpwcnet/core/non_layers.py:147-158):
    ``tfa.image.dense_image_warp(img, -flo[..., ::-1])``."""
    img_l = _to_nhwc(np.asarray(img), data_format)
    flo_l = _to_nhwc(np.asarray(flo), data_format)
    out = tfa_dense_image_warp(img_l, -flo_l[..., ::-1])
    return _from_nhwc(out, data_format)


def epe_error(y_true, y_pred, data_format=CHANNELS_LAST):
    """``epe_error`` -- qpwcnet/app/optical_flow/train.py:247-253."""
    axis = -1 if data_format == CHANNELS_LAST else 1
    err = np.sqrt(np.sum((np.asarray(y_true) - np.asarray(y_pred)) ** 2, axis=axis))
    return err.mean()


def cost_volume_to_flow(cvol, data_format=CHANNELS_LAST):
    """``cost_volume_to_flow`` -- qpwcnet/core/vis.py:9-34 (pins channel order)."""
    axis = -1 if data_format == CHANNELS_LAST else -3
    dims = cvol.shape[axis]
    imax = np.argmax(cvol, axis=axis).astype(np.float32)
    q = np.sqrt(np.float32(dims))
    di = np.floor(imax / q)
    dj = imax - di * q
    di = di - (q - 1) / 2
    dj = dj - (q - 1) / 2
    return np.stack([di, dj], axis=axis)


def invert_flow(flow, data_format=CHANNELS_LAST):
    """``inv_flow = -tf_warp(flow, flow, data_format)`` -- qpwcnet/core/occlusion.py:85,
    qpwcnet/app/test/test_invert_flow.py:47."""
    return -tf_warp(flow, flow, data_format)


def estimate_occlusion_map(flow, data_format=CHANNELS_LAST):
    """``estimate_occlusion_map`` -- qpwcnet/core/occlusion.py:27-118, op for op; (B,H,W) f32.

    Unbatched input is an error in the reference (unbound ``is_batch``, occlusion.py:19-21)."""
    flow = np.asarray(flow, dtype=np.float32)
    if flow.ndim < 4:
        raise ValueError("estimate_occlusion_map requires batched rank-4 input (occlusion.py:19-21)")
    axis = -3 if data_format == CHANNELS_FIRST else -1          # occlusion.py:43-46
    if data_format == CHANNELS_FIRST:
        n, _, h, w = flow.shape
    else:
        n, h, w, _ = flow.shape
    i, j = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")  # occlusion.py:50
    i = i.astype(np.float32)
    j = j.astype(np.float32)
    dj, di = np.moveaxis(flow, axis, 0)                         # occlusion.py:59 (unstack)
    i2, j2 = i + di, j + dj                                     # occlusion.py:60
    oob = np.any([i2 < 0, i2 >= h, j2 < 0, j2 >= w], axis=0)    # occlusion.py:74
    oob = oob.astype(np.float32)
    inv_flow = invert_flow(flow, data_format)                   # occlusion.py:85
    dj, di = np.moveaxis(inv_flow, axis, 0)
    i2, j2 = i + di, j + dj
    idx3 = np.stack([i2, j2], axis=-1).astype(np.int32)         # occlusion.py:88 (truncation)
    idx3 = np.clip(idx3, [0, 0], [h - 1, w - 1])                # occlusion.py:89-92
    b = np.broadcast_to(np.arange(n).reshape(n, 1, 1), (n, h, w))
    map3 = np.ones_like(oob)                                    # occlusion.py:95 scatter_nd_min
    np.minimum.at(map3, (b, idx3[..., 0], idx3[..., 1]), np.float32(0.0))
    return np.maximum(oob, map3)                                # occlusion.py:98

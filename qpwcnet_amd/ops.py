"""Functional entry points of the hot path: torch tensors in, HIP kernels underneath.

Each function validates shapes/dtypes/devices the way the reference's graph
construction would, then calls the C ABI (``include/qpwc.h``) on the caller's
current HIP stream.  There is no CPU path: a non-GPU tensor is an error.
"""
import ctypes

import torch

from . import _hip
from .backend import CHANNELS_FIRST, CHANNELS_LAST, get_axis

_DTYPES = {torch.float32: _hip.F32, torch.float16: _hip.F16}


class KernelTimer:
    """HIP-event timing of the hot-path launches, on the stream they are launched
    on (the caller's current stream).  Used by bench.py for the live roofline:

        with ops.kernel_timing() as kt:
            model(x)
        kt.summary()  # {(op, B, H, W, C): (launches, mean_ms)}
    """

    def __init__(self):
        self.records = []

    def __enter__(self):
        global _TIMER
        self._prev, _TIMER = _TIMER, self
        return self

    def __exit__(self, *exc):
        global _TIMER
        _TIMER = self._prev
        return False

    def summary(self):
        torch.cuda.synchronize()
        acc = {}
        for key, e0, e1 in self.records:
            n, t = acc.get(key, (0, 0.0))
            acc[key] = (n + 1, t + e0.elapsed_time(e1))
        return {k: (n, t / n) for k, (n, t) in acc.items()}


_TIMER = None


def kernel_timing():
    return KernelTimer()


class _timed:
    __slots__ = ("key", "e0")

    def __init__(self, op, dims):
        self.key = (op,) + tuple(dims)

    def __enter__(self):
        if _TIMER is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if _TIMER is not None and exc[0] is None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            _TIMER.records.append((self.key, self.e0, e1))
        return False


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def cost_volume_kernel(B, H, W, C, dtype=torch.float32, fused=False, search_range=4, layout=None,
                       out_pixel_stride=0):
    """Name of the kernel ``qpwc_cost_volume_fwd`` (``fused=False``) / ``qpwc_warp_cost_volume_fwd`` (``fused=True``)
    launches for this shape: the C side's own selection rules, run without launching (``qpwc_cost_volume_kernel``;
    host only).  '' for arguments the entry point refuses."""
    dt = {"f32": _hip.F32, "f16": _hip.F16}.get(dtype) if isinstance(dtype, str) else _DTYPES.get(dtype)
    if dt is None:
        return ""
    name = _hip.lib().qpwc_cost_volume_kernel(int(B), int(H), int(W), int(C), int(search_range),
                                              _hip.NHWC if layout is None else layout, dt, int(out_pixel_stride),
                                              1 if fused else 0)
    return name.decode() if name else ""


def _check_tensor(name, t):
    if not isinstance(t, torch.Tensor):
        raise TypeError("{} must be a torch.Tensor".format(name))
    if not t.is_cuda:
        raise RuntimeError(
            "qpwcnet_amd: {} is on '{}'; the hot path runs on a HIP device only "
            "(no CPU fallback)".format(name, t.device))
    if t.dtype not in _DTYPES:
        raise ValueError("{}: unsupported dtype {}".format(name, t.dtype))
    if t.dim() != 4:
        # the reference's tf_warp breaks on unbatched input too (warp.py:75-79)
        raise ValueError("{} must be rank 4 (batched), got shape {}".format(name, tuple(t.shape)))


def _physical(t, data_format):
    """-> (dense tensor, layout code, (B,H,W,C), nhwc_view_of_nchw).

    A ``channels_first`` tensor stored in torch's channels_last memory format is
    physically NHWC: hand it to the NHWC kernels through a permuted view."""
    get_axis(data_format)  # raises ValueError('Unsupported data format')
    if data_format == CHANNELS_LAST:
        t = t.contiguous()
        B, H, W, C = t.shape
        return t, _hip.NHWC, (B, H, W, C), False
    B, C, H, W = t.shape
    if C > 1 and t.is_contiguous(memory_format=torch.channels_last) and not t.is_contiguous():
        return t.permute(0, 2, 3, 1), _hip.NHWC, (B, H, W, C), True
    return t.contiguous(), _hip.NCHW, (B, H, W, C), False


def _empty_like_layout(ref, dims, channels, layout, as_nchw_view):
    B, H, W, _ = dims
    if layout == _hip.NHWC:
        buf = torch.empty((B, H, W, channels), dtype=ref.dtype, device=ref.device)
        return buf, (buf.permute(0, 3, 1, 2) if as_nchw_view else buf)
    buf = torch.empty((B, channels, H, W), dtype=ref.dtype, device=ref.device)
    return buf, buf


def layout_transpose(x, to_format):
    """Dense (B,C,H,W) -> dense (B,H,W,C) for to_format 'channels_last', the reverse for
    'channels_first' (qpwc_layout_transpose_fwd): how a 'channels_first' tensor crosses the boundary
    of the channels-last kernels (the reference transposes in and out the same way, layers.py:179-183)."""
    _check_tensor("x", x)
    get_axis(to_format)
    x = x.contiguous()
    if to_format == CHANNELS_LAST:
        B, C, H, W = x.shape
        out = torch.empty((B, H, W, C), dtype=x.dtype, device=x.device)
        code = _hip.NHWC
    else:
        B, H, W, C = x.shape
        out = torch.empty((B, C, H, W), dtype=x.dtype, device=x.device)
        code = _hip.NCHW
    with torch.cuda.device(x.device), _timed("layout_transpose", (B, H, W, C)):
        rc = _hip.lib().qpwc_layout_transpose_fwd(x.data_ptr(), out.data_ptr(), B, H, W, C, code,
                                                   _DTYPES[x.dtype], _stream(x))
    _hip.check(rc)
    return out


def copy_pixels_ok(src, dst):
    """Whether copy_pixels() can take this pair of (B,H,W,C) views: same shape and dtype, channels contiguous, whole
    16-byte units everywhere (qpwc_copy_pixels_fwd's contract)."""
    if src.dim() != 4 or src.shape != dst.shape or src.dtype != dst.dtype or src.dtype not in _DTYPES or \
            not (src.is_cuda and dst.is_cuda) or src.stride(3) != 1 or dst.stride(3) != 1:
        return False
    es = src.element_size()
    if (src.shape[3] * es) % 16 or src.data_ptr() % 16 or dst.data_ptr() % 16:
        return False
    return all((t.stride(i) * es) % 16 == 0 and t.stride(i) >= 0 for t in (src, dst) for i in range(3))


def copy_pixels(src, dst):
    """dst[...] = src for two channels-last (B,H,W,C) views with their own batch / row / pixel strides
    (qpwc_copy_pixels_fwd): the skip half of the decoder's concat([UpConv(x), skip]) (pwcnet.py:186-195)."""
    _check_tensor("src", src)
    _check_tensor("dst", dst)
    if not copy_pixels_ok(src, dst):
        raise ValueError("copy_pixels needs two (B,H,W,C) views of one shape and dtype with contiguous channels in "
                         "whole 16-byte units")
    B, H, W, C = src.shape
    ss = (ctypes.c_int64 * 3)(src.stride(0), src.stride(1), src.stride(2))
    ds = (ctypes.c_int64 * 3)(dst.stride(0), dst.stride(1), dst.stride(2))
    with torch.cuda.device(src.device), _timed("copy_pixels", (B, H, W, C)):
        rc = _hip.lib().qpwc_copy_pixels_fwd(src.data_ptr(), dst.data_ptr(), B, H, W, C, ss, ds, _DTYPES[src.dtype],
                                             _stream(src))
    _hip.check(rc)
    return dst


def _nchw_fast_path(C, search_range=4):
    """Dense NCHW operands of the hot-path layers go through a transposition to the channels-last
    kernels when those have a vector path for the shape (else: the generic any-layout kernels)."""
    return C % 4 == 0 and search_range == 4


def cost_volume(prv, nxt, search_range=4, data_format=CHANNELS_LAST, lrelu_slope=0.1):
    """CostVolume / CostVolumeV2 forward (reference: qpwcnet/core/layers.py:72-100,128-132)."""
    _check_tensor("prv", prv)
    _check_tensor("nxt", nxt)
    if prv.shape != nxt.shape:
        raise ValueError("prv and nxt must have the same shape, got {} and {}".format(
            tuple(prv.shape), tuple(nxt.shape)))
    if prv.dtype != nxt.dtype or prv.device != nxt.device:
        raise ValueError("prv and nxt must share dtype and device")
    p, layout, dims, as_view = _physical(prv, data_format)
    n, layout_n, _, _ = _physical(nxt, data_format)
    if layout_n != layout:  # mixed memory formats: fall back to the declared layout
        p, n = prv.contiguous(), nxt.contiguous()
        layout, as_view = (_hip.NCHW if data_format == CHANNELS_FIRST else _hip.NHWC), False
    B, H, W, C = dims
    d = 2 * int(search_range) + 1
    if layout == _hip.NCHW and _nchw_fast_path(C, int(search_range)):
        # dense (B,C,H,W): transpose in, matrix-core / LDS-tiled channels-last kernel, transpose out
        out_l = cost_volume(layout_transpose(p, CHANNELS_LAST), layout_transpose(n, CHANNELS_LAST), search_range,
                            CHANNELS_LAST, lrelu_slope)
        return layout_transpose(out_l, CHANNELS_FIRST)
    buf, out = _empty_like_layout(p, dims, d * d, layout, as_view)
    with torch.cuda.device(p.device), _timed("cost_volume", dims):
        rc = _hip.lib().qpwc_cost_volume_fwd(
            p.data_ptr(), n.data_ptr(), buf.data_ptr(), B, H, W, C, int(search_range), layout,
            _DTYPES[p.dtype], float(lrelu_slope), _stream(p))
    _hip.check(rc)
    return out


def cost_volume_to_flow(cvol, data_format=CHANNELS_LAST):
    """Displacement of the strongest correlation per pixel, (di, dj) = (row, column)
    (qpwcnet/core/vis.py:9-34): fp32 (..., H, W, 2) / (..., 2, H, W).  Rank 4 or unbatched rank 3 (the
    reference's channels_first branch only parses rank 3, vis.py:19-20; both ranks work here for both
    layouts).  A channels-last view with a wider pixel stride (the 84-channel padded volume's [..., :81])
    is read in place."""
    if not isinstance(cvol, torch.Tensor) or cvol.dim() not in (3, 4):
        raise ValueError("cvol must be a rank 3 or 4 tensor, got {}".format(
            tuple(cvol.shape) if isinstance(cvol, torch.Tensor) else type(cvol)))
    x = cvol if cvol.dim() == 4 else cvol.unsqueeze(0)
    _check_tensor("cvol", x)
    if data_format == CHANNELS_LAST:
        B, H, W, D = x.shape
        if x.stride(3) != 1 or x.stride(1) != W * x.stride(2) or x.stride(0) != H * x.stride(1) or x.stride(2) < D:
            x = x.contiguous()
        layout, stride = _hip.NHWC, x.stride(2)
        out = torch.empty((B, H, W, 2), dtype=torch.float32, device=x.device)
    elif data_format == CHANNELS_FIRST:
        B, D, H, W = x.shape
        x = x.contiguous()
        layout, stride = _hip.NCHW, D
        out = torch.empty((B, 2, H, W), dtype=torch.float32, device=x.device)
    else:
        raise ValueError("Unsupported data format : {}".format(data_format))
    with torch.cuda.device(x.device), _timed("cost_volume_to_flow", (B, H, W, D)):
        rc = _hip.lib().qpwc_cost_volume_to_flow_fwd(x.data_ptr(), out.data_ptr(), B, H, W, D, stride, layout,
                                                      _DTYPES[x.dtype], _stream(x))
    _hip.check(rc)
    return out if cvol.dim() == 4 else out[0]


def _flow_physical(flo, dims, data_format, layout):
    """fp32 flow, dense over its non-broadcast dims, in the layout the image uses."""
    B, H, W, _ = dims
    if flo.dim() != 4:
        raise ValueError("flo must be rank 4, got shape {}".format(tuple(flo.shape)))
    if data_format == CHANNELS_LAST:
        fb, fh, fw, fc = flo.shape
    else:
        fb, fc, fh, fw = flo.shape
    if fc != 2:
        raise ValueError("flo must have 2 channels (x, y), got {}".format(fc))
    mask = 0
    for ext, full, bit in ((fb, B, _hip.BCAST_B), (fh, H, _hip.BCAST_H), (fw, W, _hip.BCAST_W)):
        if ext == full:
            continue
        if ext != 1:
            raise ValueError("flo shape {} is not broadcastable to the image".format(
                tuple(flo.shape)))
        mask |= bit
    f = getattr(flo, "_qpwc_f32", None) if flo.dtype == torch.float16 else None   # written beside it by flow_head_up()
    if f is None:
        f = flo.to(torch.float32)
    if data_format == CHANNELS_FIRST and layout == _hip.NHWC:
        f = f.permute(0, 2, 3, 1)  # image is physically NHWC: give the flow the same layout
    return f.contiguous(), mask


def warp(img, flo, mode="clamp", data_format=CHANNELS_LAST):
    """Warp (mode 'tfwarp', qpwcnet/core/warp.py:63-153) / WarpV2 (mode 'clamp',
    qpwcnet/core/layers.py:177-186): sample img at (y + flo[...,1], x + flo[...,0])."""
    _check_tensor("img", img)
    if not isinstance(flo, torch.Tensor) or not flo.is_cuda or flo.device != img.device:
        raise RuntimeError("flo must be a tensor on the same HIP device as img")
    if mode not in ("clamp", "tfwarp"):
        raise ValueError("unknown warp mode '{}'".format(mode))
    i, layout, dims, as_view = _physical(img, data_format)
    B, H, W, C = dims
    if layout == _hip.NCHW and C % 4 == 0 and (H >= 2 and W >= 2 or mode == "tfwarp"):
        # dense (B,C,H,W): the 16-byte gather kernel on a channels-last copy (flow: 2 channels, permuted)
        out_l = warp(layout_transpose(i, CHANNELS_LAST), flo.permute(0, 2, 3, 1), mode, CHANNELS_LAST)
        return layout_transpose(out_l, CHANNELS_FIRST)
    f, mask = _flow_physical(flo, dims, data_format, layout)
    buf, out = _empty_like_layout(i, dims, C, layout, as_view)
    with torch.cuda.device(i.device), _timed("warp_" + mode, dims):
        rc = _hip.lib().qpwc_warp_fwd(
            i.data_ptr(), f.data_ptr(), buf.data_ptr(), B, H, W, C, mask, layout,
            _DTYPES[i.dtype], _hip.WARP_CLAMP if mode == "clamp" else _hip.WARP_TFWARP,
            _stream(i))
    _hip.check(rc)
    return out


def cost_volume_into(prv, nxt, out, channel_offset=0, search_range=4, lrelu_slope=0.1, flo=None):
    """NHWC cost volume written into channels [offset, offset+d*d) of the wider
    channels-last buffer ``out`` (B,H,W,Ctot) -- Flow/UpFlow's concat target
    (qpwcnet/core/non_layers.py:332-338, 381-385).  With ``flo`` the WarpV2 of
    ``nxt`` is fused in (non_layers.py:377-380) and ``nxt_w`` never exists."""
    _check_tensor("prv", prv)
    _check_tensor("nxt", nxt)
    _check_tensor("out", out)
    if prv.shape != nxt.shape or prv.dtype != nxt.dtype or out.dtype != prv.dtype:
        raise ValueError("prv, nxt (and out's dtype) must match")
    if not (prv.is_contiguous() and nxt.is_contiguous() and out.is_contiguous()):
        raise ValueError("cost_volume_into needs dense NHWC tensors")
    B, H, W, C = prv.shape
    if out.shape[:3] != prv.shape[:3]:
        raise ValueError("out must be (B,H,W,Ctot) with the image's B,H,W")
    L = _hip.lib()
    with torch.cuda.device(prv.device), \
            _timed("cost_volume" if flo is None else "warp_cost_volume", (B, H, W, C)):
        if flo is None:
            rc = L.qpwc_cost_volume_fwd_strided(
                prv.data_ptr(), nxt.data_ptr(), out.data_ptr(), B, H, W, C, int(search_range),
                _DTYPES[prv.dtype], float(lrelu_slope), out.shape[3], int(channel_offset),
                _stream(prv))
        else:
            if tuple(flo.shape) != (B, H, W, 2) or flo.dtype != torch.float32 or \
                    not flo.is_contiguous() or flo.device != prv.device:
                raise ValueError("flo must be a dense fp32 (B,H,W,2) tensor on the same device")
            rc = L.qpwc_warp_cost_volume_fwd(
                prv.data_ptr(), nxt.data_ptr(), flo.data_ptr(), out.data_ptr(), B, H, W, C,
                int(search_range), _DTYPES[prv.dtype], float(lrelu_slope), out.shape[3],
                int(channel_offset), _stream(prv))
    _hip.check(rc)
    return out


def warp_cost_volume(prv, nxt, flo, search_range=4, lrelu_slope=0.1):
    """cost_volume(prv, WarpV2(nxt, flo)) in one launch; NHWC, dense result."""
    d = 2 * int(search_range) + 1
    out = torch.empty(prv.shape[:3] + (d * d,), dtype=prv.dtype, device=prv.device)
    return cost_volume_into(prv, nxt, out, 0, search_range, lrelu_slope, flo=flo)


def epe(y_true, y_pred, data_format=CHANNELS_LAST):
    """End-point error, qpwcnet/app/optical_flow/train.py:247-253 -> 0-dim tensor."""
    for name, t in (("y_true", y_true), ("y_pred", y_pred)):
        _check_tensor(name, t)
        if t.dtype != torch.float32:
            raise ValueError("{} must be float32".format(name))
    if y_true.shape != y_pred.shape:
        raise ValueError("y_true and y_pred must have the same shape")
    get_axis(data_format)
    a, b = y_true.contiguous(), y_pred.contiguous()
    if data_format == CHANNELS_LAST:
        B, H, W, C = a.shape
        layout = _hip.NHWC
    else:
        B, C, H, W = a.shape
        layout = _hip.NCHW
    if C != 2:
        raise ValueError("flows must have 2 channels")
    L = _hip.lib()
    # per-call scratch: stream-ordered reuse across concurrent streams would race
    ws = torch.empty(L.qpwc_epe_workspace_floats(), dtype=torch.float32, device=a.device)
    out = torch.empty((), dtype=torch.float32, device=a.device)
    with torch.cuda.device(a.device):
        rc = L.qpwc_epe_fwd(a.data_ptr(), b.data_ptr(), out.data_ptr(), ws.data_ptr(), B, H, W,
                            layout, _stream(a))
    _hip.check(rc)
    return out


def dwconv3x3(sources, weight, mish_on_load=False):
    """Depthwise 3x3 'same' convolution over the channel-wise concatenation of 1..3
    channels-last fp32/fp16 sources (each (B,H,W,Ci), last dim contiguous; fp32 weights) -- the
    depthwise half of OptFlow's SeparableConv2D (qpwcnet/core/non_layers.py:223-231) without ever
    building Flow/UpFlow's concat (non_layers.py:336-338, 381-385).
    weight: (C,1,3,3) or (C,3,3) with C = sum(Ci).  -> (B,H,W,C)."""
    keep, c_ptrs, c_ch, c_st, B, H, W, C = _dw_sources(sources)
    w = weight.reshape(-1, 9)
    if w.shape[0] != C or w.dtype != torch.float32 or not w.is_cuda:
        raise ValueError("weight must be fp32 (C,3,3) on the device with C = {}".format(C))
    w = w.contiguous()
    out = torch.empty((B, H, W, C), dtype=keep[0].dtype, device=keep[0].device)
    with torch.cuda.device(out.device), _timed("dwconv3x3", (B, H, W, C)):
        rc = _hip.lib().qpwc_dwconv3x3_fwd(c_ptrs, c_ch, c_st, len(keep), int(bool(mish_on_load)),
                                            w.data_ptr(), out.data_ptr(), B, H, W, _DTYPES[out.dtype],
                                            _stream(out))
    _hip.check(rc)
    return out


def flow_head(z, params, scale, out_format=CHANNELS_LAST):
    """Tail of OptFlow (non_layers.py:238-254, 268-273) on the pre-activation 16-channel
    tensor z (B,H,W,16): scale * conv3x3(BN(Mish(W1 Mish(z) + b1))) -> (B,H,W,2), or (B,2,H,W) for
    out_format 'channels_first' (a channels_first model's output, written by the kernel itself).
    params: packed fp32 vector, see include/qpwc.h / non_layers.pack_flow_head."""
    _check_tensor("z", z)
    if z.shape[3] != 16 or not z.is_contiguous():
        raise ValueError("z must be a dense (B,H,W,16) tensor")
    L = _hip.lib()
    if params.numel() != L.qpwc_flow_head_param_floats() or params.dtype != torch.float32 or \
            not params.is_cuda or not params.is_contiguous():
        raise ValueError("params must be a dense fp32 device vector of {} floats".format(
            L.qpwc_flow_head_param_floats()))
    B, H, W, _ = z.shape
    get_axis(out_format)
    cf = out_format == CHANNELS_FIRST
    out = torch.empty((B, 2, H, W) if cf else (B, H, W, 2), dtype=z.dtype, device=z.device)
    with torch.cuda.device(z.device), _timed("flow_head", (B, H, W, 16)):
        rc = L.qpwc_flow_head_fwd(z.data_ptr(), params.data_ptr(), out.data_ptr(), B, H, W,
                                  float(scale), _DTYPES[z.dtype], _hip.NCHW if cf else _hip.NHWC, _stream(z))
    _hip.check(rc)
    return out


def pointwise_bias(y, pw_padded, bias):
    """Pointwise 1x1 + bias of a split SeparableConv2D on the matrix cores (qpwc_pointwise_bias_fwd): y (..., C) fp32 dense
    -> (..., F) = y . W^T + bias, W = pw_padded (F, ceil(C/32)*32) from pad_pointwise().  The own replacement of the library
    GEMM behind dwconv3x3() on the few-pixel levels."""
    if not isinstance(y, torch.Tensor) or not y.is_cuda:
        raise RuntimeError("qpwcnet_amd: y must be a tensor on a HIP device (no CPU fallback)")
    C = y.shape[-1]
    F_, cpad = pw_padded.shape
    if y.dtype != torch.float32 or not y.is_contiguous() or pw_padded.dtype != torch.float32 or not pw_padded.is_contiguous() \
            or cpad != (C + 31) // 32 * 32 or bias.numel() != F_ or bias.dtype != torch.float32 or not bias.is_contiguous():
        raise ValueError("pointwise_bias takes dense fp32 y (..., C), weights (F, ceil(C/32)*32) from pad_pointwise, bias (F)")
    M = y.numel() // C
    out = torch.empty(y.shape[:-1] + (F_,), dtype=torch.float32, device=y.device)
    with torch.cuda.device(y.device), _timed("pointwise_bias", (M, C, F_)):
        rc = _hip.lib().qpwc_pointwise_bias_fwd(y.data_ptr(), pw_padded.data_ptr(), bias.data_ptr(), out.data_ptr(), M, C, F_,
                                                _stream(y))
    _hip.check(rc)
    return out


def flow_head_up(z, params, scale, up_scale=2.0):
    """flow_head() (channels-last) and the Upsample(x2, * up_scale) that follows it in the flow chain (pwcnet.py:55,60) in one
    launch (qpwc_flow_head_up_fwd) -> (flow (B,H,W,2), up_scale * bilinear x2 of it (B,2H,2W,2)); the second equals
    upsample2x_flow(flow, up_scale) bit for bit."""
    _check_tensor("z", z)
    if z.shape[3] != 16 or not z.is_contiguous():
        raise ValueError("z must be a dense (B,H,W,16) tensor")
    L = _hip.lib()
    if params.numel() != L.qpwc_flow_head_param_floats() or params.dtype != torch.float32 or \
            not params.is_cuda or not params.is_contiguous():
        raise ValueError("params must be a dense fp32 device vector of {} floats".format(
            L.qpwc_flow_head_param_floats()))
    B, H, W, _ = z.shape
    out = torch.empty((B, H, W, 2), dtype=z.dtype, device=z.device)
    up = torch.empty((B, 2 * H, 2 * W, 2), dtype=z.dtype, device=z.device)
    # fp16 storage: the upsampled flow once more as fp32, written by the same launch (== up.float(); found by its consumers
    # as up._qpwc_f32: the next level's WarpV2 takes fp32 coordinates, and the cast was a launch of its own per level)
    up32 = torch.empty((B, 2 * H, 2 * W, 2), dtype=torch.float32, device=z.device) if z.dtype == torch.float16 else None
    with torch.cuda.device(z.device), _timed("flow_head", (B, H, W, 16)):
        rc = L.qpwc_flow_head_up_fwd(z.data_ptr(), params.data_ptr(), out.data_ptr(), up.data_ptr(),
                                     up32.data_ptr() if up32 is not None else None, B, H, W,
                                     float(scale), float(up_scale), _DTYPES[z.dtype], _stream(z))
    _hip.check(rc)
    if up32 is not None:
        up._qpwc_f32 = up32
    return out, up


def optflow_tail(z2, dw3, pw3, b3, dw4, pw4, b4, head_params, scale, mish_on_load=False, out_format=CHANNELS_LAST):
    """Last two SeparableConv2D (64 -> 32 -> 16) + flow head of OptFlow (non_layers.py:223-231, 238-254,
    268-273) in one launch (qpwc_optflow_tail_fwd): z2 (B,H,W,64) fp32, the second layer's output (activated
    unless mish_on_load) -> flow (B,H,W,2) / (B,2,H,W).  For the coarse pyramid levels."""
    _check_tensor("z2", z2)
    if z2.dtype != torch.float32 or z2.shape[3] != 64 or not z2.is_contiguous():
        raise ValueError("z2 must be a dense fp32 (B,H,W,64) tensor")
    for name, t, shape in (("dw3", dw3, (64, 9)), ("pw3", pw3, (32, 64)), ("b3", b3, (32,)), ("dw4", dw4, (32, 9)),
                           ("pw4", pw4, (16, 32)), ("b4", b4, (16,))):
        if tuple(t.shape) != shape or t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous():
            raise ValueError("{} must be a dense fp32 device tensor of shape {}".format(name, shape))
    L = _hip.lib()
    if head_params.numel() != L.qpwc_flow_head_param_floats() or head_params.dtype != torch.float32:
        raise ValueError("head_params must hold {} fp32 values".format(L.qpwc_flow_head_param_floats()))
    get_axis(out_format)
    cf = out_format == CHANNELS_FIRST
    B, H, W, _ = z2.shape
    out = torch.empty((B, 2, H, W) if cf else (B, H, W, 2), dtype=torch.float32, device=z2.device)
    with torch.cuda.device(z2.device), _timed("optflow_tail", (B, H, W, 64)):
        rc = L.qpwc_optflow_tail_fwd(z2.data_ptr(), dw3.data_ptr(), pw3.data_ptr(), b3.data_ptr(), dw4.data_ptr(),
                                     pw4.data_ptr(), b4.data_ptr(), head_params.data_ptr(), out.data_ptr(), B, H, W,
                                     float(scale), int(bool(mish_on_load)), _hip.NCHW if cf else _hip.NHWC,
                                     _stream(z2))
    _hip.check(rc)
    return out


def bias_mish_(x_nhwc, bias=None):
    """In place x = Mish(x + bias) on a dense channels-last fp32 tensor (..., C), C % 4 == 0 --
    the `activation='Mish'` epilogue of the reference's conv blocks (non_layers.py:196-210,
    390-449).  Returns x."""
    _check_tensor("x", x_nhwc)
    if not x_nhwc.is_contiguous():
        raise ValueError("bias_mish_ needs a dense channels-last tensor")
    C = x_nhwc.shape[-1]
    if bias is not None and (bias.numel() != C or bias.dtype != torch.float32 or not bias.is_cuda):
        raise ValueError("bias must be a fp32 device vector of {} elements".format(C))
    n = x_nhwc.numel() // C
    with torch.cuda.device(x_nhwc.device), _timed("bias_mish", tuple(x_nhwc.shape)):
        rc = _hip.lib().qpwc_bias_mish_fwd(x_nhwc.data_ptr(), 0 if bias is None else bias.data_ptr(),
                                            n, C, _DTYPES[x_nhwc.dtype], _stream(x_nhwc))
    _hip.check(rc)
    return x_nhwc


def upsample2x_flow(flo, scale=1.0, in_format=CHANNELS_LAST, out_format=CHANNELS_LAST):
    """scale * bilinear x2 upsampling of a flow (B,h,w,2) [(B,2,h,w) for in_format 'channels_first'] --
    the reference's Upsample functor (non_layers.py:183-193) as used on flows (pwcnet.py:55,60);
    out (B,2h,2w,2) or (B,2,2h,2w) by out_format."""
    _check_tensor("flo", flo)
    get_axis(in_format)
    get_axis(out_format)
    if flo.shape[1 if in_format == CHANNELS_FIRST else 3] != 2:
        raise ValueError("upsample2x_flow takes a 2-channel flow, got shape {}".format(tuple(flo.shape)))
    f = flo.contiguous()
    if in_format == CHANNELS_FIRST:
        B, _, h, w = f.shape
    else:
        B, h, w, _ = f.shape
    ocf = out_format == CHANNELS_FIRST
    out = torch.empty((B, 2, 2 * h, 2 * w) if ocf else (B, 2 * h, 2 * w, 2), dtype=f.dtype, device=f.device)
    with torch.cuda.device(f.device), _timed("upsample2x_flow", (B, h, w, 2)):
        rc = _hip.lib().qpwc_upsample2x_flow_fwd(f.data_ptr(), out.data_ptr(), B, h, w, float(scale),
                                                  _DTYPES[f.dtype],
                                                  _hip.NCHW if in_format == CHANNELS_FIRST else _hip.NHWC,
                                                  _hip.NCHW if ocf else _hip.NHWC, _stream(f))
    _hip.check(rc)
    return out


def _flow_dims(flow, data_format):
    get_axis(data_format)
    if flow.dim() != 4:
        raise ValueError("flow must be a batched rank-4 tensor, got rank {}".format(flow.dim()))
    if data_format == CHANNELS_LAST:
        B, H, W, C = flow.shape
        layout = _hip.NHWC
    else:
        B, C, H, W = flow.shape
        layout = _hip.NCHW
    if C != 2:
        raise ValueError("flow must have 2 channels, got {}".format(C))
    return B, H, W, layout


def invert_flow(flow, data_format=CHANNELS_LAST):
    """-tf_warp(flow, flow) (qpwcnet/core/occlusion.py:85; app/test/test_invert_flow.py:47)."""
    _check_tensor("flow", flow)
    B, H, W, layout = _flow_dims(flow, data_format)
    f = flow.contiguous()
    out = torch.empty_like(f)
    with torch.cuda.device(f.device), _timed("invert_flow", (B, H, W, 2)):
        rc = _hip.lib().qpwc_invert_flow_fwd(f.data_ptr(), out.data_ptr(), B, H, W, layout,
                                             _DTYPES[f.dtype], _stream(f))
    _hip.check(rc)
    return out


def occlusion_map(flow, data_format=CHANNELS_LAST):
    """estimate_occlusion_map (qpwcnet/core/occlusion.py:27-118) -> (B,H,W) float32."""
    _check_tensor("flow", flow)
    B, H, W, layout = _flow_dims(flow, data_format)
    f = flow.contiguous()
    out = torch.empty((B, H, W), dtype=torch.float32, device=f.device)
    with torch.cuda.device(f.device), _timed("occlusion", (B, H, W, 2)):
        rc = _hip.lib().qpwc_occlusion_fwd(f.data_ptr(), out.data_ptr(), B, H, W, layout,
                                           _DTYPES[f.dtype], _stream(f))
    _hip.check(rc)
    return out


def epe_multi(flows_true, flows_pred, out=None, data_format=CHANNELS_LAST):
    """Per-level EPE of up to 8 fp32 flow pairs ((B,h,w,2), or (B,2,h,w) for 'channels_first') in two
    launches (FlowMseLoss, qpwcnet/train/loss.py:56-67) -> float32 tensor [n_levels].
    ``out``: dense fp32 device vector [n_levels] to write into (e.g. the all-gather payload of
    qpwcnet_amd.dist.EpeGather, so that no copy stands between the reduction and the collective)."""
    import ctypes
    n = len(flows_true)
    if n != len(flows_pred) or not 1 <= n <= 8:
        raise ValueError("epe_multi takes 1..8 (true, pred) pairs")
    cf = data_format == CHANNELS_FIRST
    get_axis(data_format)
    keep, pa, pb, pdt, npix, plane = [], [], [], [], [], []
    for i, (a, b) in enumerate(zip(flows_true, flows_pred)):
        _check_tensor("y_true[%d]" % i, a)
        _check_tensor("y_pred[%d]" % i, b)
        if a.shape != b.shape or a.shape[1 if cf else 3] != 2:
            raise ValueError("level {}: shapes {} / {}".format(i, tuple(a.shape), tuple(b.shape)))
        a = a.float().contiguous()
        b = b.contiguous() if b.dtype == torch.float16 else b.float().contiguous()   # fp16 predictions as they are
        keep += [a, b]
        pa.append(a.data_ptr())
        pb.append(b.data_ptr())
        pdt.append(_DTYPES[b.dtype])
        npix.append(a.numel() // 2)
        plane.append(a.shape[2] * a.shape[3] if cf else 0)
    dev = keep[0].device
    L = _hip.lib()
    ws = torch.empty(L.qpwc_epe_multi_workspace_floats(), dtype=torch.float32, device=dev)
    if out is None:
        out = torch.empty(n, dtype=torch.float32, device=dev)
    elif out.dtype != torch.float32 or out.numel() != n or not out.is_contiguous() or out.device != dev:
        raise ValueError("out must be a dense fp32 vector of {} elements on {}".format(n, dev))
    with torch.cuda.device(dev):
        rc = L.qpwc_epe_multi_mixed_fwd((ctypes.c_void_p * n)(*pa), (ctypes.c_void_p * n)(*pb),
                                        (ctypes.c_int64 * n)(*npix), (ctypes.c_int64 * n)(*plane),
                                        (ctypes.c_int * n)(*pdt), n, out.data_ptr(), ws.data_ptr(), _stream(out))
    _hip.check(rc)
    return out


def _dw_sources(sources):
    """-> (kept tensors, ctypes pointer/channel/stride arrays, B, H, W, C) for 1..3 sources."""
    import ctypes
    if not 1 <= len(sources) <= 3:
        raise ValueError("takes 1..3 sources")
    B, H, W = sources[0].shape[:3]
    chans, strides, ptrs, keep = [], [], [], []
    for i, t in enumerate(sources):
        _check_tensor("source %d" % i, t)
        if t.dtype != sources[0].dtype or tuple(t.shape[:3]) != (B, H, W):
            raise ValueError("sources must share dtype and B,H,W")
        if t.stride(3) != 1 or t.stride(1) != W * t.stride(2) or t.stride(0) != H * t.stride(1):
            t = t.contiguous()
        keep.append(t)
        chans.append(t.shape[3])
        strides.append(t.stride(2))
        ptrs.append(t.data_ptr())
    n = len(keep)
    return keep, (ctypes.c_void_p * n)(*ptrs), (ctypes.c_int * n)(*chans), (ctypes.c_int64 * n)(*strides), \
        B, H, W, sum(chans)


def pad_pointwise(pw, dtype=torch.float32):
    """(F, C[,1,1]) pointwise kernel -> dense (F, ceil(C/32)*32), zero padded: the layout
    qpwc_sepconv3x3_fwd (fp32) / qpwc_sepconv3x3_f16_fwd (fp16) takes."""
    pw = pw.reshape(pw.shape[0], -1).to(dtype)
    F_, C = pw.shape
    cpad = (C + 31) // 32 * 32
    out = torch.zeros((F_, cpad), dtype=dtype, device=pw.device)
    out[:, :C] = pw
    return out


def sepconv3x3(sources, dw, pw_padded, bias, mish_on_load=False, mish_on_store=False):
    """SeparableConv2D(3x3,'same'), fused (fp32): depthwise 3x3 over the virtual concat of 1..3
    channels-last sources, pointwise 1x1 + bias on the matrix cores
    (qpwcnet/core/non_layers.py:223-231).  pw_padded from pad_pointwise().
    -> (B,H,W,F): the pre-activation output, or Mish of it with mish_on_store (the layer's own
    `activation='Mish'` applied once per element; the consumer then loads without Mish)."""
    keep, c_ptrs, c_ch, c_st, B, H, W, C = _dw_sources(sources)
    F_ = pw_padded.shape[-2]
    w = dw.reshape(-1, 9)
    if w.shape[0] != C or pw_padded.shape[-1] != (C + 31) // 32 * 32 or bias.numel() != F_:
        raise ValueError("weight shapes do not match C = {}".format(C))
    flags = int(bool(mish_on_load)) | (2 if mish_on_store else 0)
    if keep[0].dtype == torch.float16:
        return _sepconv3x3_f16(keep, c_ptrs, c_ch, c_st, w, pw_padded, bias, flags, B, H, W, C, F_)
    if keep[0].dtype != torch.float32:
        raise ValueError("sepconv3x3 takes fp32 or fp16 sources")
    if pw_padded.dtype == torch.bfloat16:
        return _sepconv3x3_x3(keep, c_ptrs, c_ch, c_st, w, pw_padded, bias, flags, B, H, W, C)
    for t in (w, pw_padded, bias):
        if t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous():
            raise ValueError("weights must be dense fp32 device tensors")
    out = torch.empty((B, H, W, F_), dtype=torch.float32, device=keep[0].device)
    with torch.cuda.device(out.device), _timed("sepconv3x3", (B, H, W, C, F_)):
        rc = _hip.lib().qpwc_sepconv3x3_fwd(c_ptrs, c_ch, c_st, len(keep), flags,
                                             w.data_ptr(), pw_padded.data_ptr(), bias.data_ptr(),
                                             out.data_ptr(), B, H, W, F_, _stream(out))
    _hip.check(rc)
    return out


def sepconv3x3_x3_applies(sources):
    """True when qpwc_sepconv3x3_x3_fwd takes these fp32 sources: every source but the last a multiple of 4 channels
    in 16-byte aligned pixels (a last source of fewer than 4 channels is allowed), H * W < 2^24."""
    n = len(sources)
    for i, t in enumerate(sources):
        aligned = t.shape[3] % 4 == 0 and t.stride(2) % 4 == 0 and t.data_ptr() % 16 == 0
        if not aligned and (i + 1 < n or t.shape[3] >= 4 or t.shape[0] * t.shape[1] * t.shape[2] * t.stride(2) < 8):
            return False
        if t.shape[1] * t.shape[2] >= 1 << 24 or t.stride(2) >= 1 << 24 or t.shape[1] * t.shape[2] * t.stride(2) >= 1 << 31:
            return False
    return all(t.dtype == torch.float32 for t in sources)


def _sepconv3x3_x3(keep, c_ptrs, c_ch, c_st, w, pw3, bias, flags, B, H, W, C):
    """fp32 with the pointwise products as bf16x3 splits (qpwc_sepconv3x3_x3_fwd): pw3 = split_bf16x3(pad_pointwise(..))
    = (3, F, Cpad) bfloat16, dw / bias fp32."""
    F_ = pw3.shape[1]
    if pw3.dim() != 3 or pw3.shape[0] != 3 or not pw3.is_cuda or not pw3.is_contiguous():
        raise ValueError("pw3 must be a dense (3, F, Cpad) bfloat16 device tensor")
    for t in (w, bias):
        if t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous():
            raise ValueError("dw / bias must be dense fp32 device tensors")
    out = torch.empty((B, H, W, F_), dtype=torch.float32, device=keep[0].device)
    with torch.cuda.device(out.device), _timed("sepconv3x3_x3", (B, H, W, C, F_)):
        rc = _hip.lib().qpwc_sepconv3x3_x3_fwd(c_ptrs, c_ch, c_st, len(keep), flags, w.data_ptr(), pw3.data_ptr(),
                                                bias.data_ptr(), out.data_ptr(), B, H, W, F_, _stream(out))
    _hip.check(rc)
    return out


def _sepconv3x3_f16(keep, c_ptrs, c_ch, c_st, w, pw_padded, bias, flags, B, H, W, C, F_):
    """fp16-storage form (qpwc_sepconv3x3_f16_fwd): pw_padded fp16, dw / bias fp32."""
    for i, t in enumerate(keep):
        tail = i + 1 == len(keep) and t.shape[3] < 4
        if not tail and (t.shape[3] % 4 or t.stride(2) % 4 or t.data_ptr() % 8):
            raise ValueError("fp16 sepconv3x3: sources must hold multiples of 4 channels in 8-byte aligned pixels")
    for t, dt in ((w, torch.float32), (pw_padded, torch.float16), (bias, torch.float32)):
        if t.dtype != dt or not t.is_cuda or not t.is_contiguous():
            raise ValueError("fp16 sepconv3x3: dw/bias dense fp32, pw dense fp16 device tensors")
    out = torch.empty((B, H, W, F_), dtype=torch.float16, device=keep[0].device)
    with torch.cuda.device(out.device), _timed("sepconv3x3_f16", (B, H, W, C, F_)):
        rc = _hip.lib().qpwc_sepconv3x3_f16_fwd(c_ptrs, c_ch, c_st, len(keep), flags, w.data_ptr(),
                                                 pw_padded.data_ptr(), bias.data_ptr(), out.data_ptr(),
                                                 B, H, W, F_, _stream(out))
    _hip.check(rc)
    return out


def conv3x3_taps(weight, dtype=torch.float32):
    """torch Conv2d weight (C_out, C_in, 3, 3) -> the (9, C_out, C_in) tap-major layout of
    qpwc_conv3x3_mish_fwd (fp32) / qpwc_conv3x3_mish_f16_fwd (dtype=torch.float16)."""
    return weight.to(dtype).permute(2, 3, 0, 1).reshape(9, weight.shape[0], weight.shape[1]).contiguous()


def conv3x3_mish(x_nhwc, taps, bias, pad_h=0, pad_w=0):
    """Mish(conv3x3_same(x) + bias) for C_in = C_out in {16, 32, 64, 128, 256}, channels-last fp32 (the encoder's
    conv_aa / conv_b, non_layers.py:410-449), written into a (B, H+pad_h, W+pad_w, C) tensor whose
    border is zero (the 'SAME' padding of a following stride-2 conv).  taps from conv3x3_taps()."""
    _check_tensor("x", x_nhwc)
    if x_nhwc.dtype not in (torch.float32, torch.float16) or not x_nhwc.is_contiguous():
        raise ValueError("conv3x3_mish needs a dense fp32 / fp16 channels-last tensor")
    B, H, W, C = x_nhwc.shape
    if tuple(taps.shape) != (9, C, C) or taps.dtype != x_nhwc.dtype or not taps.is_cuda or \
            not taps.is_contiguous() or bias.numel() != C or bias.dtype != torch.float32 or not bias.is_cuda:
        raise ValueError("taps must be a dense (9,{0},{0}) device tensor of the input's dtype, bias fp32 ({0})".format(C))
    out = torch.empty((B, H + pad_h, W + pad_w, C), dtype=x_nhwc.dtype, device=x_nhwc.device)
    fn = _hip.lib().qpwc_conv3x3_mish_fwd if x_nhwc.dtype == torch.float32 else _hip.lib().qpwc_conv3x3_mish_f16_fwd
    with torch.cuda.device(out.device), _timed("conv3x3_mish" if x_nhwc.dtype == torch.float32 else "conv3x3_mish_f16",
                                               (B, H, W, C)):
        rc = fn(x_nhwc.data_ptr(), taps.data_ptr(), bias.data_ptr(), out.data_ptr(), B, H, W, C, int(pad_h), int(pad_w),
                _stream(out))
    _hip.check(rc)
    return out


def split_bf16x3(t):
    """fp32 device tensor -> (3, *t.shape) bfloat16: the three-way split of csrc/split_bf16.h (qpwc_split_bf16x3_fwd);
    out[0] + out[1] + out[2] == t exactly (parts below the smallest normal fp32 flush to zero).  The weight operands of the *_x3 kernels."""
    if not isinstance(t, torch.Tensor) or not t.is_cuda or t.dtype != torch.float32 or t.numel() == 0:
        raise ValueError("split_bf16x3 takes a non-empty fp32 device tensor")
    t = t.contiguous()
    out = torch.empty((3,) + tuple(t.shape), dtype=torch.bfloat16, device=t.device)
    with torch.cuda.device(t.device):
        rc = _hip.lib().qpwc_split_bf16x3_fwd(t.data_ptr(), out.data_ptr(), t.numel(), _stream(t))
    _hip.check(rc)
    return out


def conv3x3_mish_x3(x_nhwc, taps3, bias, pad_h=0, pad_w=0):
    """conv3x3_mish() for fp32 tensors with the products on the bf16 matrix instructions (bf16x3 split, six partial
    products, fp32 accumulation: qpwc_conv3x3_mish_x3_fwd).  taps3 = split_bf16x3(conv3x3_taps(weight))."""
    _check_tensor("x", x_nhwc)
    if x_nhwc.dtype != torch.float32 or not x_nhwc.is_contiguous():
        raise ValueError("conv3x3_mish_x3 needs a dense fp32 channels-last tensor")
    B, H, W, C = x_nhwc.shape
    if tuple(taps3.shape) != (3, 9, C, C) or taps3.dtype != torch.bfloat16 or not taps3.is_cuda or \
            not taps3.is_contiguous() or bias.numel() != C or bias.dtype != torch.float32 or not bias.is_cuda:
        raise ValueError("taps3 must be a dense (3,9,{0},{0}) bfloat16 device tensor, bias fp32 ({0})".format(C))
    out = torch.empty((B, H + pad_h, W + pad_w, C), dtype=torch.float32, device=x_nhwc.device)
    with torch.cuda.device(out.device), _timed("conv3x3_mish_x3", (B, H, W, C)):
        rc = _hip.lib().qpwc_conv3x3_mish_x3_fwd(x_nhwc.data_ptr(), taps3.data_ptr(), bias.data_ptr(), out.data_ptr(),
                                                  B, H, W, C, int(pad_h), int(pad_w), _stream(out))
    _hip.check(rc)
    return out


def first_conv_taps(weight):
    """torch Conv2d weight (16, 3, 3, 3) of enc.0.conv_a -> (9, 16, 4) fp32 [tap][out][in, slot 3 = 0]."""
    w = weight.float().permute(2, 3, 0, 1).reshape(9, 16, 3)
    out = torch.zeros((9, 16, 4), dtype=torch.float32, device=w.device)
    out[..., :3] = w
    return out


def first_conv_mish(pairs, taps, bias, data_format=CHANNELS_LAST):
    """Split(2) + frame stacking + Conv2D(3->16, 3x3, stride 2, 'same') + bias + Mish of the first
    encoder layer (pwcnet.py:229, non_layers.py:402-409) on the raw (B,H,W,6) fp32 input -- or (B,6,H,W)
    for 'channels_first' --, H and W even -> (2B, H/2, W/2, 16) channels-last.  taps from first_conv_taps()."""
    _check_tensor("pairs", pairs)
    cf = data_format == CHANNELS_FIRST
    get_axis(data_format)
    if pairs.shape[1 if cf else 3] != 6 or pairs.dtype not in (torch.float32, torch.float16) or not pairs.is_contiguous():
        raise ValueError("pairs must be a dense fp32 / fp16 (B,H,W,6) / (B,6,H,W) tensor")
    if cf:
        B, _, H, W = pairs.shape
    else:
        B, H, W, _ = pairs.shape
    if tuple(taps.shape) != (9, 16, 4) or taps.dtype != torch.float32 or not taps.is_contiguous() or \
            bias.numel() != 16 or bias.dtype != torch.float32:
        raise ValueError("taps must be fp32 (9,16,4), bias fp32 (16)")
    out = torch.empty((2 * B, H // 2, W // 2, 16), dtype=pairs.dtype, device=pairs.device)
    fn = _hip.lib().qpwc_first_conv_mish_fwd if pairs.dtype == torch.float32 else _hip.lib().qpwc_first_conv_mish_f16_fwd
    with torch.cuda.device(out.device), _timed("first_conv_mish", (B, H, W, 6)):
        rc = fn(pairs.data_ptr(), taps.data_ptr(), bias.data_ptr(), out.data_ptr(), B, H, W,
                _hip.NCHW if cf else _hip.NHWC, _stream(out))
    _hip.check(rc)
    return out


def conv3x3s2_mish(x_padded, taps, bias):
    """Mish(conv3x3 stride 2 'same' (x) + bias), C_in in {16, 32, 64, 128} -> 2 C_in channels (conv_a of encoder
    levels 2..5, non_layers.py:402-409) on the zero-bordered (B, H+1, W+1, C_in) fp32 tensor
    conv3x3_mish(pad 1, 1) writes, H and W even -> (B, H/2, W/2, 2 C_in).  taps = conv3x3_taps(weight) of
    shape (9, 2 C_in, C_in)."""
    _check_tensor("x", x_padded)
    ci = x_padded.shape[3]
    if x_padded.dtype not in (torch.float32, torch.float16) or not x_padded.is_contiguous() or ci not in (16, 32, 64, 128):
        raise ValueError("conv3x3s2_mish needs a dense fp32 / fp16 (B,H+1,W+1,C) tensor, C in {16,32,64,128}")
    B, Hp, Wp, _ = x_padded.shape
    H, W = Hp - 1, Wp - 1
    if tuple(taps.shape) != (9, 2 * ci, ci) or taps.dtype != x_padded.dtype or not taps.is_contiguous() or \
            bias.numel() != 2 * ci or bias.dtype != torch.float32:
        raise ValueError("taps must be (9,{},{}) of the input's dtype, bias fp32 ({})".format(2 * ci, ci, 2 * ci))
    out = torch.empty((B, H // 2, W // 2, 2 * ci), dtype=x_padded.dtype, device=x_padded.device)
    f16 = x_padded.dtype == torch.float16
    fn = _hip.lib().qpwc_conv3x3s2_mish_f16_fwd if f16 else _hip.lib().qpwc_conv3x3s2_mish_c_fwd
    with torch.cuda.device(out.device), _timed("conv3x3s2_mish_f16" if f16 else "conv3x3s2_mish", (B, H, W, ci)):
        rc = fn(x_padded.data_ptr(), taps.data_ptr(), bias.data_ptr(), out.data_ptr(), B, H, W, ci, _stream(out))
    _hip.check(rc)
    return out


def conv3x3s2_mish_x3(x_padded, taps3, bias):
    """conv3x3s2_mish() for fp32 tensors, C_in in {32, 64, 128}, with the products on the bf16 matrix instructions
    (qpwc_conv3x3s2_mish_x3_fwd).  taps3 = split_bf16x3(conv3x3_taps(weight)) of shape (3, 9, 2 C_in, C_in)."""
    _check_tensor("x", x_padded)
    ci = x_padded.shape[3]
    if x_padded.dtype != torch.float32 or not x_padded.is_contiguous() or ci not in (32, 64, 128):
        raise ValueError("conv3x3s2_mish_x3 needs a dense fp32 (B,H+1,W+1,C) tensor, C in {32,64,128}")
    B, Hp, Wp, _ = x_padded.shape
    H, W = Hp - 1, Wp - 1
    if tuple(taps3.shape) != (3, 9, 2 * ci, ci) or taps3.dtype != torch.bfloat16 or not taps3.is_contiguous() or \
            not taps3.is_cuda or bias.numel() != 2 * ci or bias.dtype != torch.float32:
        raise ValueError("taps3 must be a dense (3,9,{},{}) bfloat16 device tensor, bias fp32 ({})".format(2 * ci, ci, 2 * ci))
    out = torch.empty((B, H // 2, W // 2, 2 * ci), dtype=torch.float32, device=x_padded.device)
    with torch.cuda.device(out.device), _timed("conv3x3s2_mish_x3", (B, H, W, ci)):
        rc = _hip.lib().qpwc_conv3x3s2_mish_x3_fwd(x_padded.data_ptr(), taps3.data_ptr(), bias.data_ptr(), out.data_ptr(),
                                                    B, H, W, ci, _stream(out))
    _hip.check(rc)
    return out


def upconv_taps(weight, dtype=torch.float32):
    """torch ConvTranspose2d weight (C_in, F, 4, 4) -> the (16, F, C_in) layout of qpwc_upconv4x4s2_mish_fwd (fp32) /
    qpwc_upconv4x4s2_mish_f16_fwd (dtype=torch.float16)."""
    return weight.to(dtype).permute(2, 3, 1, 0).reshape(16, weight.shape[1], weight.shape[0]).contiguous()


def upconv4x4s2_mish_into(x_nhwc, taps, bias, dst):
    """Mish(Conv2DTranspose(4x4, stride 2, 'same')(x) + bias) (the decoder's UpConv, non_layers.py:196-210)
    written into channels [0, F) of the dense channels-last buffer dst (B, 2H, 2W, Ctot >= F): the `up` half of
    concat([up, skip]) (pwcnet.py:186-195).  taps from upconv_taps().  Returns dst."""
    _check_tensor("x", x_nhwc)
    _check_tensor("dst", dst)
    if x_nhwc.dtype not in (torch.float32, torch.float16) or dst.dtype != x_nhwc.dtype or not x_nhwc.is_contiguous() or \
            not dst.is_contiguous():
        raise ValueError("upconv4x4s2_mish_into needs dense fp32 / fp16 channels-last tensors of one dtype")
    B, H, W, C = x_nhwc.shape
    x3 = taps.dtype == torch.bfloat16 and taps.dim() == 4     # split_bf16x3(upconv_taps(w)): the bf16x3 arithmetic
    F_ = taps.shape[-2]
    if x3:
        if tuple(taps.shape) != (3, 16, F_, C) or x_nhwc.dtype != torch.float32 or not taps.is_contiguous() or \
                bias.numel() != F_ or bias.dtype != torch.float32:
            raise ValueError("split taps must be (3,16,F,{}) bfloat16 with fp32 tensors, bias fp32 (F)".format(C))
    elif tuple(taps.shape) != (16, F_, C) or taps.dtype != x_nhwc.dtype or not taps.is_contiguous() or \
            bias.numel() != F_ or bias.dtype != torch.float32:
        raise ValueError("taps must be (16,F,{}) of the input's dtype, bias fp32 (F)".format(C))
    if tuple(dst.shape[:3]) != (B, 2 * H, 2 * W) or dst.shape[3] < F_:
        raise ValueError("dst must be (B,2H,2W,Ctot) with Ctot >= F")
    f16 = x_nhwc.dtype == torch.float16
    fn = _hip.lib().qpwc_upconv4x4s2_mish_x3_fwd if x3 else (
        _hip.lib().qpwc_upconv4x4s2_mish_f16_fwd if f16 else _hip.lib().qpwc_upconv4x4s2_mish_fwd)
    with torch.cuda.device(dst.device), _timed("upconv4x4s2_mish_f16" if f16 else "upconv4x4s2_mish", (B, H, W, C, F_)):
        rc = fn(x_nhwc.data_ptr(), taps.data_ptr(), bias.data_ptr(), dst.data_ptr(), B, H, W, C, F_, dst.shape[3],
                _stream(dst))
    _hip.check(rc)
    return dst


def upconv_cat_ok(x_nhwc, taps, skip, dst):
    """Can upconv4x4s2_mish_cat_into() take these?  (F skip channels, 4-element-aligned strides, plain fp32 / fp16 taps)"""
    if taps.dim() != 3 or skip.dim() != 4 or dst.dim() != 4:
        return False
    F_ = taps.shape[1]
    st = skip.stride()
    esz = 16 // skip.element_size()           # elements per 16 bytes (fp32: 4) -- pointers: 16 B (fp32) / 8 B (fp16)
    return (skip.is_cuda and skip.dtype == x_nhwc.dtype == dst.dtype == taps.dtype and skip.shape[3] == F_ and
            dst.shape[3] >= 2 * F_ and st[3] == 1 and st[2] % 4 == 0 and st[1] % 4 == 0 and st[0] % 4 == 0 and
            st[2] >= F_ and st[1] >= skip.shape[2] * st[2] and st[0] >= skip.shape[1] * st[1] and
            skip.data_ptr() % (4 * skip.element_size()) == 0 and esz in (4, 8))


def upconv4x4s2_mish_cat_into(x_nhwc, taps, bias, skip, dst):
    """upconv4x4s2_mish_into() AND the skip half of the decoder's concat in the same launch: channels [0, F) of dst =
    Mish(Conv2DTranspose(x) + bias), channels [F, 2F) = skip (B, 2H, 2W, F; any 4-element-aligned strides, e.g. the interior
    of a zero-bordered encoder buffer) -- concat([UpConv(x), skip]) of pwcnet.py:186-195 without a separate copy launch."""
    _check_tensor("x", x_nhwc)
    _check_tensor("dst", dst)
    _check_tensor("skip", skip)
    B, H, W, C = x_nhwc.shape
    F_ = taps.shape[-2]
    if x_nhwc.dtype not in (torch.float32, torch.float16) or not x_nhwc.is_contiguous() or not dst.is_contiguous():
        raise ValueError("upconv4x4s2_mish_cat_into needs dense fp32 / fp16 channels-last x and dst")
    if tuple(taps.shape) != (16, F_, C) or not taps.is_contiguous() or bias.numel() != F_ or bias.dtype != torch.float32:
        raise ValueError("taps must be (16,F,{}) of the input's dtype, bias fp32 (F)".format(C))
    if tuple(dst.shape[:3]) != (B, 2 * H, 2 * W) or tuple(skip.shape) != (B, 2 * H, 2 * W, F_) or not upconv_cat_ok(x_nhwc, taps, skip, dst):
        raise ValueError("dst must be (B,2H,2W,>=2F), skip (B,2H,2W,F) of the same dtype with 4-element-aligned strides")
    f16 = x_nhwc.dtype == torch.float16
    fn = _hip.lib().qpwc_upconv4x4s2_mish_cat_f16_fwd if f16 else _hip.lib().qpwc_upconv4x4s2_mish_cat_fwd
    st = skip.stride()
    with torch.cuda.device(dst.device), _timed("upconv4x4s2_mish_f16" if f16 else "upconv4x4s2_mish", (B, H, W, C, F_)):
        rc = fn(x_nhwc.data_ptr(), taps.data_ptr(), bias.data_ptr(), skip.data_ptr(), st[0], st[1], st[2], dst.data_ptr(),
                B, H, W, C, F_, dst.shape[3], _stream(dst))
    _hip.check(rc)
    return dst


def bias_mish_pad(x_nhwc, bias, pad_h, pad_w):
    """Mish(x + bias) written into a new (B, H+pad_h, W+pad_w, C) tensor whose border is zero:
    the activation epilogue and TensorFlow's 'SAME' padding of the following stride-2 conv
    (non_layers.py:402-409) in one pass.  Returns the padded tensor."""
    _check_tensor("x", x_nhwc)
    if not x_nhwc.is_contiguous():
        raise ValueError("bias_mish_pad needs a dense channels-last tensor")
    B, H, W, C = x_nhwc.shape
    out = torch.empty((B, H + pad_h, W + pad_w, C), dtype=x_nhwc.dtype, device=x_nhwc.device)
    with torch.cuda.device(x_nhwc.device), _timed("bias_mish_pad", (B, H, W, C)):
        rc = _hip.lib().qpwc_bias_mish_pad_fwd(x_nhwc.data_ptr(), 0 if bias is None else bias.data_ptr(),
                                                out.data_ptr(), B, H, W, C, int(pad_h), int(pad_w), C,
                                                _DTYPES[x_nhwc.dtype], _stream(x_nhwc))
    _hip.check(rc)
    return out


def bias_mish_into(x_nhwc, bias, dst, channel_offset=0):
    """Mish(x + bias) written into channels [offset, offset+C) of the wider dense channels-last
    buffer `dst` (B,H,W,Ctot): one half of the decoder's concat([up, skip]) (pwcnet.py:186-195)
    without the concat copy of that half.  Returns dst."""
    _check_tensor("x", x_nhwc)
    _check_tensor("dst", dst)
    if not (x_nhwc.is_contiguous() and dst.is_contiguous()) or dst.dtype != x_nhwc.dtype:
        raise ValueError("bias_mish_into needs dense channels-last tensors of one dtype")
    B, H, W, C = x_nhwc.shape
    if tuple(dst.shape[:3]) != (B, H, W) or channel_offset % 4 or channel_offset + C > dst.shape[3]:
        raise ValueError("dst must be (B,H,W,Ctot) with room for C channels at a 4-aligned offset")
    es = dst.element_size()
    with torch.cuda.device(dst.device), _timed("bias_mish_into", (B, H, W, C)):
        rc = _hip.lib().qpwc_bias_mish_pad_fwd(x_nhwc.data_ptr(), 0 if bias is None else bias.data_ptr(),
                                                dst.data_ptr() + channel_offset * es, B, H, W, C, 0, 0,
                                                dst.shape[3], _DTYPES[dst.dtype], _stream(dst))
    _hip.check(rc)
    return dst


def split_frames_pad(pairs, pad_h=0, pad_w=0):
    """(B,H,W,6) channels-last input pair -> (2B, H+pad_h, W+pad_w, 3): Split(2) (pwcnet.py:229),
    both frames stacked on the batch axis, far edges zero-padded for the first stride-2 conv."""
    _check_tensor("pairs", pairs)
    if pairs.shape[3] != 6:
        raise ValueError("pairs must be (B,H,W,6)")
    x = pairs.contiguous()
    B, H, W, _ = x.shape
    out = torch.empty((2 * B, H + pad_h, W + pad_w, 3), dtype=x.dtype, device=x.device)
    with torch.cuda.device(x.device), _timed("split_frames_pad", (B, H, W, 6)):
        rc = _hip.lib().qpwc_split_frames_pad_fwd(x.data_ptr(), out.data_ptr(), B, H, W, int(pad_h),
                                                   int(pad_w), _DTYPES[x.dtype], _stream(x))
    _hip.check(rc)
    return out

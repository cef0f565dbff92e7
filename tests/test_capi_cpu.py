"""CPU suite: the C-ABI library loads and exports every symbol include/qpwc.h
declares; argument validation (which happens before any HIP call) returns the
documented codes.  No compute calls without a GPU."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "qpwc.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(qpwc_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_all_exported(hip_lib):
    from qpwcnet_amd import _hip
    declared = _declared_symbols()
    assert declared, "no declarations parsed from include/qpwc.h"
    assert sorted(_hip.SYMBOLS) == declared
    for name in declared:
        assert hasattr(hip_lib, name), name


def test_version_and_strerror(hip_lib):
    assert hip_lib.qpwc_version() == 200
    assert b"product" in hip_lib.qpwc_build_info() and b"EXPERIMENTAL" not in hip_lib.qpwc_build_info()
    assert hip_lib.qpwc_strerror(0) == b"ok"
    assert b"data format" in hip_lib.qpwc_strerror(-2)


def test_argument_validation_needs_no_gpu(hip_lib):
    """Every check below fails before the first HIP call, so it runs on CPU."""
    from qpwcnet_amd import _hip
    buf = (ctypes.c_float * 64)()
    p = ctypes.cast(buf, ctypes.c_void_p).value
    q = p + 4 * 32
    L = hip_lib
    assert L.qpwc_cost_volume_fwd(None, p, q, 1, 2, 2, 4, 4, 0, 0, 0.1, None) == _hip.E_NULL
    assert L.qpwc_cost_volume_fwd(p, p, q, 1, 2, 2, 4, 4, 7, 0, 0.1, None) == _hip.E_LAYOUT
    assert b"Unsupported data format" in L.qpwc_last_error()
    assert L.qpwc_cost_volume_fwd(p, p, q, 1, 2, 2, 4, 4, 0, 9, 0.1, None) == _hip.E_DTYPE
    assert L.qpwc_cost_volume_fwd(p, p, q, 1, 0, 2, 4, 4, 0, 0, 0.1, None) == _hip.E_SHAPE
    assert L.qpwc_cost_volume_fwd(p, p, q, 1, 2, 2, 4, -1, 0, 0, 0.1, None) == _hip.E_RANGE
    assert L.qpwc_cost_volume_fwd(p, p, p, 1, 2, 2, 4, 0, 0, 0, 0.1, None) == _hip.E_ALIAS
    assert L.qpwc_cost_volume_fwd(p + 2, p, q, 1, 2, 2, 4, 0, 0, 0, 0.1, None) == _hip.E_ALIGN
    assert L.qpwc_cost_volume_fwd_strided(p, p, q, 1, 1, 1, 4, 0, 0, 0.1, 1, 1, None) == _hip.E_STRIDE
    assert L.qpwc_warp_fwd(p, p, q, 1, 1, 4, 2, 0, 0, 0, 0, None) == _hip.E_SHAPE   # H < 2, clamp
    assert b"at least 2x2" in L.qpwc_last_error()
    assert L.qpwc_warp_fwd(p, p, q, 1, 2, 2, 2, 0, 0, 0, 5, None) == _hip.E_MODE
    assert L.qpwc_warp_fwd(p, p, q, 1, 2, 2, 2, 64, 0, 0, 0, None) == _hip.E_SHAPE
    assert L.qpwc_warp_cost_volume_fwd(p, p, None, q, 1, 2, 2, 4, 4, 0, 0.1, 81, 0, None) == _hip.E_NULL
    assert L.qpwc_epe_fwd(p, p, None, q, 1, 2, 2, 0, None) == _hip.E_NULL
    assert L.qpwc_epe_workspace_floats() > 0
    assert L.qpwc_device_copy(p, q, 24, None) == _hip.E_SHAPE
    assert L.qpwc_device_copy(p, p, 64, None) == _hip.E_ALIAS
    assert L.qpwc_device_copy(None, q, 64, None) == _hip.E_NULL


def test_argument_validation_of_the_next_rows_needs_no_gpu(hip_lib):
    """The SURVEY 8(f) entry points (OptFlow pieces, encoder convolutions, layout change, multi-level EPE) reject bad
    arguments before any HIP call, with the same codes as the hot-path pair."""
    from qpwcnet_amd import _hip
    buf = (ctypes.c_float * 4096)()
    base = ctypes.cast(buf, ctypes.c_void_p).value
    base += (-base) % 16
    p, q, r = base, base + 4096, base + 8192
    L = hip_lib
    NHWC, NCHW, F32 = _hip.NHWC, _hip.NCHW, 0
    # layout change
    assert L.qpwc_layout_transpose_fwd(None, q, 1, 2, 2, 4, NCHW, F32, None) == _hip.E_NULL
    assert L.qpwc_layout_transpose_fwd(p, q, 1, 2, 2, 4, 7, F32, None) == _hip.E_LAYOUT
    assert L.qpwc_layout_transpose_fwd(p, q, 1, 2, 2, 4, NCHW, 9, None) == _hip.E_DTYPE
    assert L.qpwc_layout_transpose_fwd(p, q, 1, 0, 2, 4, NCHW, F32, None) == _hip.E_SHAPE
    assert L.qpwc_layout_transpose_fwd(p, p, 1, 2, 2, 4, NCHW, F32, None) == _hip.E_ALIAS
    # strided view copy (decoder concat)
    st = (ctypes.c_int64 * 3)(64, 16, 4)
    assert L.qpwc_copy_pixels_fwd(p, q, 1, 2, 2, 4, st, None, F32, None) == _hip.E_NULL
    assert L.qpwc_copy_pixels_fwd(p, q, 1, 2, 2, 3, st, st, F32, None) == _hip.E_SHAPE      # 12-byte pixels
    assert L.qpwc_copy_pixels_fwd(p, q, 1, 2, 2, 4, (ctypes.c_int64 * 3)(64, 16, 2), st, F32, None) == _hip.E_STRIDE
    assert L.qpwc_copy_pixels_fwd(p, q, 1, 2, 2, 4, (ctypes.c_int64 * 3)(64, 18, 4), st, F32, None) == _hip.E_STRIDE
    assert L.qpwc_copy_pixels_fwd(p, p + 16, 1, 2, 2, 4, st, st, F32, None) == _hip.E_ALIAS
    assert L.qpwc_copy_pixels_fwd(p + 4, q, 1, 2, 2, 4, st, st, F32, None) == _hip.E_ALIGN
    # flow head / upsample / one-launch tail
    assert L.qpwc_flow_head_fwd(p, q, None, 1, 2, 2, 1.0, F32, NHWC, None) == _hip.E_NULL
    assert L.qpwc_flow_head_fwd(p, q, r, 1, 2, 2, 1.0, F32, 5, None) == _hip.E_LAYOUT
    assert L.qpwc_flow_head_fwd(p, q, r, 1, 2, 2, 1.0, 9, NHWC, None) == _hip.E_DTYPE
    assert L.qpwc_flow_head_fwd(p, q, r, 1, 0, 2, 1.0, F32, NHWC, None) == _hip.E_SHAPE
    assert L.qpwc_upsample2x_flow_fwd(p, q, 1, 2, 2, 2.0, F32, 5, NHWC, None) == _hip.E_LAYOUT
    assert L.qpwc_upsample2x_flow_fwd(p, q, 1, 2, 0, 2.0, F32, NHWC, NHWC, None) == _hip.E_SHAPE
    assert L.qpwc_upsample2x_flow_fwd(p, p, 1, 2, 2, 2.0, F32, NHWC, NHWC, None) == _hip.E_ALIAS
    tail_ptrs = [p, q, q, q, q, q, q, q, r]
    assert L.qpwc_optflow_tail_fwd(*(tail_ptrs[:8] + [None]), 1, 2, 2, 1.0, 0, NHWC, None) == _hip.E_NULL
    assert L.qpwc_optflow_tail_fwd(*tail_ptrs, 1, 2, 2, 1.0, 0, 5, None) == _hip.E_LAYOUT
    assert L.qpwc_optflow_tail_fwd(*tail_ptrs, 1, 2, 0, 1.0, 0, NHWC, None) == _hip.E_SHAPE
    assert L.qpwc_optflow_tail_fwd(*([p + 8] + tail_ptrs[1:]), 1, 2, 2, 1.0, 0, NHWC, None) == _hip.E_ALIGN
    # fused SeparableConv2D
    vp, ci, i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
    srcs, chans, strides = (vp * 3)(p, 0, 0), (ci * 3)(32, 0, 0), (i64 * 3)(32, 0, 0)
    assert L.qpwc_sepconv3x3_fwd(srcs, chans, strides, 4, 0, q, q, q, r, 1, 2, 2, 16, None) == _hip.E_SHAPE   # n_src
    assert L.qpwc_sepconv3x3_fwd(srcs, chans, strides, 1, 0, q, q, q, r, 1, 2, 2, 24, None) == _hip.E_SHAPE   # F
    assert L.qpwc_sepconv3x3_fwd(srcs, chans, strides, 1, 7, q, q, q, r, 1, 2, 2, 16, None) == _hip.E_SHAPE   # mish flags
    assert L.qpwc_sepconv3x3_fwd(srcs, chans, (i64 * 3)(16, 0, 0), 1, 0, q, q, q, r, 1, 2, 2, 16, None) == _hip.E_STRIDE
    assert L.qpwc_sepconv3x3_fwd(srcs, chans, strides, 1, 0, q, q, q, p, 1, 2, 2, 16, None) == _hip.E_ALIAS
    assert L.qpwc_sepconv3x3_fwd(srcs, chans, strides, 1, 0, q, q + 4, q, r, 1, 2, 2, 16, None) == _hip.E_ALIGN
    # encoder convolutions
    assert L.qpwc_conv3x3_mish_fwd(p, q, q, r, 1, 2, 2, 24, 0, 0, None) == _hip.E_SHAPE
    assert b"not in {16,32,64,128,256}" in L.qpwc_last_error()
    assert L.qpwc_conv3x3_mish_fwd(p, q, q, p, 1, 2, 2, 16, 0, 0, None) == _hip.E_ALIAS
    assert L.qpwc_first_conv_mish_fwd(p, q, q, r, 1, 3, 4, NHWC, None) == _hip.E_SHAPE     # odd height
    assert L.qpwc_first_conv_mish_fwd(p, q, q, r, 1, 2, 4, 5, None) == _hip.E_LAYOUT
    # multi-level EPE
    yt, yp, npix, planes = (vp * 2)(p, q), (vp * 2)(q, 0), (i64 * 2)(4, 4), (i64 * 2)(0, 0)
    assert L.qpwc_epe_multi_fwd(yt, yp, npix, planes, 2, r, r, None) == _hip.E_NULL           # level 1 prediction missing
    yp = (vp * 2)(q, q)
    assert L.qpwc_epe_multi_fwd(yt, yp, npix, planes, 9, r, r, None) == _hip.E_SHAPE
    assert L.qpwc_epe_multi_fwd(yt, yp, (i64 * 2)(4, 0), planes, 2, r, r, None) == _hip.E_SHAPE
    assert L.qpwc_epe_multi_fwd(yt, yp, npix, (i64 * 2)(3, 0), 2, r, r, None) == _hip.E_SHAPE   # 4 pixels are not planes of 3
    assert L.qpwc_epe_multi_mixed_fwd(yt, yp, npix, planes, None, 2, r, r, None) == _hip.E_NULL
    assert L.qpwc_epe_multi_mixed_fwd(yt, yp, npix, planes, (ci * 2)(0, 9), 2, r, r, None) == _hip.E_DTYPE
    assert L.qpwc_epe_multi_mixed_fwd(yt, (vp * 2)(q, q + 2), npix, planes, (ci * 2)(0, 1), 2, r, r, None) == _hip.E_ALIGN


def test_argument_validation_of_the_round4_entry_points_needs_no_gpu(hip_lib):
    """qpwc_upconv4x4s2_mish_cat_fwd / _f16, qpwc_flow_head_up_fwd, qpwc_pointwise_bias_fwd and the two query / probe entry points
    reject bad arguments before any HIP call."""
    from qpwcnet_amd import _hip
    buf = (ctypes.c_float * 65536)()
    base = ctypes.cast(buf, ctypes.c_void_p).value
    base += (-base) % 16
    p, q, r, t = base, base + 16384, base + 32768, base + 131072
    L = hip_lib
    F32, F16 = 0, 1
    # UpConv + skip half of the concat: x (1,2,2,64) -> out (1,4,4,>=32), skip (1,4,4,16)
    ok = dict(B=1, H=2, W=2, C=64, F=16)
    def cat(x=p, w=q, b=q, skip=r, bs=256, rs=64, ps=16, out=t, ops=32, **kw):
        a = dict(ok, **kw)
        return L.qpwc_upconv4x4s2_mish_cat_fwd(x, w, b, skip, bs, rs, ps, out, a["B"], a["H"], a["W"], a["C"], a["F"], ops, None)
    assert cat(skip=None) == _hip.E_NULL
    assert cat(C=48) == _hip.E_SHAPE
    assert cat(F=24) == _hip.E_SHAPE
    assert cat(ops=16) == _hip.E_STRIDE            # no room for the skip half
    assert cat(ps=8) == _hip.E_STRIDE              # skip pixels narrower than F
    assert cat(rs=48) == _hip.E_STRIDE             # rows overlap
    assert cat(ps=18, rs=72, bs=288) == _hip.E_STRIDE   # not a multiple of 4 elements
    assert cat(skip=r + 4) == _hip.E_ALIGN
    assert cat(out=r) == _hip.E_ALIAS              # out overlaps skip
    assert L.qpwc_upconv4x4s2_mish_cat_f16_fwd(p, q, q, r + 2, 256, 64, 16, t, 1, 2, 2, 64, 16, 32, None) == _hip.E_ALIGN
    # flow head + upsampling
    assert L.qpwc_flow_head_up_fwd(p, q, r, None, None, 1, 2, 2, 1.0, 2.0, F32, None) == _hip.E_NULL
    assert L.qpwc_flow_head_up_fwd(p, q, r, t, None, 1, 2, 2, 1.0, 2.0, 9, None) == _hip.E_DTYPE
    assert L.qpwc_flow_head_up_fwd(p, q, r, t, None, 1, 0, 2, 1.0, 2.0, F32, None) == _hip.E_SHAPE
    assert L.qpwc_flow_head_up_fwd(p, q, r, r, None, 1, 2, 2, 1.0, 2.0, F32, None) == _hip.E_ALIAS
    assert L.qpwc_flow_head_up_fwd(p, q, r, t + 4, None, 1, 2, 2, 1.0, 2.0, F32, None) == _hip.E_ALIGN
    assert L.qpwc_flow_head_up_fwd(p, q, r, t, t + 4096, 1, 2, 2, 1.0, 2.0, F32, None) == _hip.E_DTYPE   # fp32 copy of an fp32 flow
    assert L.qpwc_flow_head_up_fwd(p, q, r, t, t, 1, 2, 2, 1.0, 2.0, F16, None) == _hip.E_ALIAS
    # pointwise half of a split SeparableConv2D
    assert L.qpwc_pointwise_bias_fwd(p, q, q, None, 16, 32, 16, None) == _hip.E_NULL
    assert L.qpwc_pointwise_bias_fwd(p, q, q, r, 16, 32, 48, None) == _hip.E_SHAPE
    assert L.qpwc_pointwise_bias_fwd(p, q, q, r, 0, 32, 16, None) == _hip.E_SHAPE
    assert L.qpwc_pointwise_bias_fwd(p + 4, q, q, r, 16, 32, 16, None) == _hip.E_ALIGN
    assert L.qpwc_pointwise_bias_fwd(p, q, q, p, 16, 32, 16, None) == _hip.E_ALIAS
    # the kernel query answers without a device: refused arguments give the empty name
    L.qpwc_cost_volume_kernel.restype = ctypes.c_char_p
    assert L.qpwc_cost_volume_kernel(8, 128, 256, 32, 4, _hip.NHWC, F32, 84, 1).startswith(b"cost_volume_mfma_lds")
    assert L.qpwc_cost_volume_kernel(8, 1, 32, 256, 4, _hip.NHWC, F32, 81, 1) == b""
    assert L.qpwc_clock_probe(None, 4, 4, None) != 0


def test_check_maps_codes_to_reference_exceptions(hip_lib):
    from qpwcnet_amd import _hip
    with pytest.raises(ValueError):
        _hip.check(_hip.E_LAYOUT)
    with pytest.raises(RuntimeError):
        _hip.check(_hip.E_LAUNCH)
    _hip.check(0)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from qpwcnet_amd import _hip
    monkeypatch.setattr(_hip, "_lib", None)
    monkeypatch.setattr(_hip, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback|missing"):
        _hip.lib()

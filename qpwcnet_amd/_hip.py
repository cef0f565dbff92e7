"""ctypes binding of ``libqpwc_hip.so`` (C ABI declared in ``include/qpwc.h``).

There is deliberately no fallback: if the library is missing or a call fails
the error propagates.  PyTorch is used only for device memory and streams; the
ABI itself sees raw device pointers.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# QPWC_HIP_LIB: another build of the same library (kernel A/B experiments); default = in-tree build
LIB_PATH = os.environ.get("QPWC_HIP_LIB") or os.path.join(_HERE, "csrc", "libqpwc_hip.so")

NHWC, NCHW = 0, 1
F32, F16 = 0, 1
WARP_CLAMP, WARP_TFWARP = 0, 1
BCAST_B, BCAST_H, BCAST_W = 1, 2, 4

E_NULL, E_LAYOUT, E_DTYPE, E_SHAPE, E_RANGE, E_MODE, E_ALIAS, E_LAUNCH, E_ALIGN, E_STRIDE, \
    E_NODEVICE = range(-1, -12, -1)
_ARGUMENT_ERRORS = {E_NULL, E_LAYOUT, E_DTYPE, E_SHAPE, E_RANGE, E_MODE, E_ALIAS, E_ALIGN, E_STRIDE}

# every symbol include/qpwc.h declares
SYMBOLS = (
    "qpwc_version", "qpwc_last_error", "qpwc_strerror", "qpwc_build_info", "qpwc_device_copy", "qpwc_clock_probe", "qpwc_layout_transpose_fwd", "qpwc_copy_pixels_fwd",
    "qpwc_cost_volume_fwd", "qpwc_cost_volume_fwd_strided", "qpwc_warp_fwd",
    "qpwc_warp_cost_volume_fwd", "qpwc_cost_volume_kernel", "qpwc_epe_workspace_floats", "qpwc_epe_fwd",
    "qpwc_dwconv3x3_fwd", "qpwc_flow_head_param_floats", "qpwc_flow_head_fwd", "qpwc_flow_head_up_fwd", "qpwc_pointwise_bias_fwd", "qpwc_optflow_tail_fwd", "qpwc_bias_mish_fwd",
    "qpwc_upsample2x_flow_fwd", "qpwc_epe_multi_workspace_floats", "qpwc_epe_multi_fwd", "qpwc_epe_multi_mixed_fwd",
    "qpwc_cost_volume_to_flow_fwd", "qpwc_sepconv3x3_fwd", "qpwc_sepconv3x3_f16_fwd", "qpwc_bias_mish_pad_fwd", "qpwc_split_frames_pad_fwd",
    "qpwc_invert_flow_fwd", "qpwc_occlusion_fwd", "qpwc_conv3x3_mish_fwd", "qpwc_conv3x3_mish_f16_fwd", "qpwc_conv3x3_mish_x3_fwd", "qpwc_split_bf16x3_fwd", "qpwc_sepconv3x3_x3_fwd", "qpwc_conv3x3s2_mish_x3_fwd", "qpwc_upconv4x4s2_mish_x3_fwd",
    "qpwc_first_conv_mish_fwd", "qpwc_first_conv_mish_f16_fwd", "qpwc_conv3x3s2_mish_fwd", "qpwc_conv3x3s2_mish_c_fwd", "qpwc_conv3x3s2_mish_f16_fwd", "qpwc_upconv4x4s2_mish_fwd", "qpwc_upconv4x4s2_mish_f16_fwd", "qpwc_upconv4x4s2_mish_cat_fwd", "qpwc_upconv4x4s2_mish_cat_f16_fwd",
)

_lib = None


def build(verbose=False):
    """Compile the HIP library in-tree for gfx950 (hipcc cross-compiles on CPU)."""
    import subprocess
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j4"]
    if not verbose:
        cmd.append("-s")
    subprocess.check_call(cmd)
    return LIB_PATH


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "qpwcnet_amd: {} is missing -- build it with `make -C qpwcnet_amd/csrc` or "
            "`python -c 'import __graft_entry__ as g; g.build()'`. There is no CPU "
            "fallback.".format(LIB_PATH))
    # PyTorch-ROCm bundles its own libamdhip64 (SONAME libamdhip64.so.7).  Load it
    # FIRST so that our NEEDED entry binds to that same runtime instance: a second
    # HIP runtime in the process cannot see torch's streams or allocations
    # ("no ROCm-capable device is detected" at the first launch).
    import torch
    hip_rt = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    if os.path.exists(hip_rt):
        ctypes.CDLL(hip_rt, mode=ctypes.RTLD_GLOBAL)
    L = ctypes.CDLL(LIB_PATH)
    vp, ci, cf, i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_int64
    L.qpwc_version.argtypes = []
    L.qpwc_version.restype = ci
    L.qpwc_last_error.argtypes = []
    L.qpwc_last_error.restype = ctypes.c_char_p
    L.qpwc_strerror.argtypes = [ci]
    L.qpwc_strerror.restype = ctypes.c_char_p
    L.qpwc_build_info.argtypes = []
    L.qpwc_build_info.restype = ctypes.c_char_p
    L.qpwc_layout_transpose_fwd.argtypes = [vp, vp, ci, ci, ci, ci, ci, ci, vp]
    L.qpwc_layout_transpose_fwd.restype = ci
    L.qpwc_copy_pixels_fwd.argtypes = [vp, vp, ci, ci, ci, ci, ctypes.POINTER(i64), ctypes.POINTER(i64), ci, vp]
    L.qpwc_copy_pixels_fwd.restype = ci
    L.qpwc_clock_probe.argtypes = [vp, ci, ci, vp]
    L.qpwc_clock_probe.restype = ci
    L.qpwc_device_copy.argtypes = [vp, vp, i64, vp]
    L.qpwc_device_copy.restype = ci
    L.qpwc_cost_volume_fwd.argtypes = [vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, cf, vp]
    L.qpwc_cost_volume_fwd.restype = ci
    L.qpwc_cost_volume_kernel.argtypes = [ci, ci, ci, ci, ci, ci, ci, i64, ci]
    L.qpwc_cost_volume_kernel.restype = ctypes.c_char_p
    L.qpwc_cost_volume_fwd_strided.argtypes = [vp, vp, vp, ci, ci, ci, ci, ci, ci, cf, i64, i64, vp]
    L.qpwc_cost_volume_fwd_strided.restype = ci
    L.qpwc_warp_fwd.argtypes = [vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, ci, vp]
    L.qpwc_warp_fwd.restype = ci
    L.qpwc_warp_cost_volume_fwd.argtypes = [vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, cf, i64, i64, vp]
    L.qpwc_warp_cost_volume_fwd.restype = ci
    L.qpwc_epe_workspace_floats.argtypes = []
    L.qpwc_epe_workspace_floats.restype = ci
    L.qpwc_epe_fwd.argtypes = [vp, vp, vp, vp, ci, ci, ci, ci, vp]
    L.qpwc_epe_fwd.restype = ci
    L.qpwc_dwconv3x3_fwd.argtypes = [ctypes.POINTER(vp), ctypes.POINTER(ci), ctypes.POINTER(i64),
                                     ci, ci, vp, vp, ci, ci, ci, ci, vp]
    L.qpwc_dwconv3x3_fwd.restype = ci
    L.qpwc_flow_head_param_floats.argtypes = []
    L.qpwc_flow_head_param_floats.restype = ci
    L.qpwc_flow_head_fwd.argtypes = [vp, vp, vp, ci, ci, ci, cf, ci, ci, vp]
    L.qpwc_flow_head_fwd.restype = ci
    L.qpwc_flow_head_up_fwd.argtypes = [vp, vp, vp, vp, vp, ci, ci, ci, cf, cf, ci, vp]
    L.qpwc_flow_head_up_fwd.restype = ci
    L.qpwc_pointwise_bias_fwd.argtypes = [vp, vp, vp, vp, i64, ci, ci, vp]
    L.qpwc_pointwise_bias_fwd.restype = ci
    L.qpwc_optflow_tail_fwd.argtypes = [vp] * 9 + [ci, ci, ci, cf, ci, ci, vp]
    L.qpwc_optflow_tail_fwd.restype = ci
    L.qpwc_bias_mish_fwd.argtypes = [vp, vp, i64, ci, ci, vp]
    L.qpwc_bias_mish_fwd.restype = ci
    L.qpwc_upsample2x_flow_fwd.argtypes = [vp, vp, ci, ci, ci, cf, ci, ci, ci, vp]
    L.qpwc_upsample2x_flow_fwd.restype = ci
    L.qpwc_epe_multi_workspace_floats.argtypes = []
    L.qpwc_epe_multi_workspace_floats.restype = ci
    L.qpwc_epe_multi_fwd.argtypes = [ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(i64),
                                     ctypes.POINTER(i64), ci, vp, vp, vp]
    L.qpwc_epe_multi_fwd.restype = ci
    L.qpwc_epe_multi_mixed_fwd.argtypes = [ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(i64),
                                           ctypes.POINTER(i64), ctypes.POINTER(ci), ci, vp, vp, vp]
    L.qpwc_epe_multi_mixed_fwd.restype = ci
    L.qpwc_sepconv3x3_fwd.argtypes = [ctypes.POINTER(vp), ctypes.POINTER(ci), ctypes.POINTER(i64),
                                      ci, ci, vp, vp, vp, vp, ci, ci, ci, ci, vp]
    L.qpwc_sepconv3x3_fwd.restype = ci
    L.qpwc_sepconv3x3_f16_fwd.argtypes = L.qpwc_sepconv3x3_fwd.argtypes
    L.qpwc_sepconv3x3_x3_fwd.argtypes = L.qpwc_sepconv3x3_fwd.argtypes
    L.qpwc_sepconv3x3_x3_fwd.restype = ci
    L.qpwc_cost_volume_to_flow_fwd.argtypes = [vp, vp, ci, ci, ci, ci, i64, ci, ci, vp]
    L.qpwc_cost_volume_to_flow_fwd.restype = ci
    L.qpwc_sepconv3x3_f16_fwd.restype = ci
    L.qpwc_bias_mish_pad_fwd.argtypes = [vp, vp, vp, ci, ci, ci, ci, ci, ci, i64, ci, vp]
    L.qpwc_bias_mish_pad_fwd.restype = ci
    L.qpwc_split_frames_pad_fwd.argtypes = [vp, vp, ci, ci, ci, ci, ci, ci, vp]
    L.qpwc_split_frames_pad_fwd.restype = ci
    L.qpwc_invert_flow_fwd.argtypes = [vp, vp, ci, ci, ci, ci, ci, vp]
    L.qpwc_invert_flow_fwd.restype = ci
    L.qpwc_occlusion_fwd.argtypes = [vp, vp, ci, ci, ci, ci, ci, vp]
    L.qpwc_occlusion_fwd.restype = ci
    L.qpwc_conv3x3_mish_f16_fwd.argtypes = [vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, vp]
    L.qpwc_conv3x3_mish_f16_fwd.restype = ci
    L.qpwc_conv3x3_mish_fwd.argtypes = [vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, vp]
    L.qpwc_conv3x3_mish_fwd.restype = ci
    L.qpwc_conv3x3_mish_x3_fwd.argtypes = [vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, vp]
    L.qpwc_conv3x3_mish_x3_fwd.restype = ci
    L.qpwc_split_bf16x3_fwd.argtypes = [vp, vp, i64, vp]
    L.qpwc_split_bf16x3_fwd.restype = ci
    L.qpwc_first_conv_mish_fwd.argtypes = [vp, vp, vp, vp, ci, ci, ci, ci, vp]
    L.qpwc_first_conv_mish_fwd.restype = ci
    L.qpwc_first_conv_mish_f16_fwd.argtypes = L.qpwc_first_conv_mish_fwd.argtypes
    L.qpwc_first_conv_mish_f16_fwd.restype = ci
    L.qpwc_conv3x3s2_mish_fwd.argtypes = [vp, vp, vp, vp, ci, ci, ci, vp]
    L.qpwc_conv3x3s2_mish_fwd.restype = ci
    L.qpwc_conv3x3s2_mish_c_fwd.argtypes = [vp, vp, vp, vp, ci, ci, ci, ci, vp]
    L.qpwc_conv3x3s2_mish_c_fwd.restype = ci
    L.qpwc_conv3x3s2_mish_f16_fwd.argtypes = L.qpwc_conv3x3s2_mish_c_fwd.argtypes
    L.qpwc_conv3x3s2_mish_x3_fwd.argtypes = L.qpwc_conv3x3s2_mish_c_fwd.argtypes
    L.qpwc_conv3x3s2_mish_x3_fwd.restype = ci
    L.qpwc_conv3x3s2_mish_f16_fwd.restype = ci
    L.qpwc_upconv4x4s2_mish_fwd.argtypes = [vp, vp, vp, vp, ci, ci, ci, ci, ci, i64, vp]
    L.qpwc_upconv4x4s2_mish_fwd.restype = ci
    L.qpwc_upconv4x4s2_mish_f16_fwd.argtypes = L.qpwc_upconv4x4s2_mish_fwd.argtypes
    L.qpwc_upconv4x4s2_mish_x3_fwd.argtypes = L.qpwc_upconv4x4s2_mish_fwd.argtypes
    L.qpwc_upconv4x4s2_mish_x3_fwd.restype = ci
    L.qpwc_upconv4x4s2_mish_f16_fwd.restype = ci
    L.qpwc_upconv4x4s2_mish_cat_fwd.argtypes = [vp, vp, vp, vp, i64, i64, i64, vp, ci, ci, ci, ci, ci, i64, vp]
    L.qpwc_upconv4x4s2_mish_cat_fwd.restype = ci
    L.qpwc_upconv4x4s2_mish_cat_f16_fwd.argtypes = L.qpwc_upconv4x4s2_mish_cat_fwd.argtypes
    L.qpwc_upconv4x4s2_mish_cat_f16_fwd.restype = ci
    _lib = L
    return L


def build_info():
    """{'path', 'version', 'build', 'product'} of the loaded library: bench.py prints it, so that a number
    from another build (QPWC_HIP_LIB -> `make experimental`) can never pass for the product's."""
    L = lib()
    info = L.qpwc_build_info().decode()
    return {"path": LIB_PATH, "version": int(L.qpwc_version()), "build": info,
            "product": "EXPERIMENTAL" not in info and not os.environ.get("QPWC_HIP_LIB")}


def source_sha256(names=("cost_volume_mfma.hip", "cost_volume.hip", "warp.hip", "common.h")):
    """sha256 over the named kernel sources (qpwcnet_amd/csrc): stamps profiles/traffic.json, and
    bench.py drops `roofline.traffic` when the kernels have changed since the counters were read."""
    import hashlib
    h = hashlib.sha256()
    for n in names:
        with open(os.path.join(_HERE, "csrc", n), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def check(rc):
    """Map a QPWC_E_* code to the exception the reference would raise."""
    if rc == 0:
        return
    L = lib()
    msg = "{} ({})".format(L.qpwc_last_error().decode(), L.qpwc_strerror(rc).decode())
    if rc in _ARGUMENT_ERRORS:
        raise ValueError(msg)
    raise RuntimeError(msg)

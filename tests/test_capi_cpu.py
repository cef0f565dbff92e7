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

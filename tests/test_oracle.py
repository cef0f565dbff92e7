"""CPU suite: the oracle against the committed golden vectors and against the
analytic known answers derivable from the reference's own test scripts
(SURVEY.md 8(c)).  No GPU, no HIP library calls."""
import os

import numpy as np
import pytest

import cases
from oracle import np_ref, tfa_ref, torch_ref

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _np_oracle(name, dtype):
    op, fmt, shape, seed, extra = cases.CASES[name]
    a, b = cases.make_inputs(name, dtype)
    if op == "cost_volume":
        return np_ref.cost_volume(a, b, extra.get("search_range", 4), fmt)
    if op == "warp_v2":
        return np_ref.warp_v2(a, b, fmt)
    return np_ref.tf_warp(a, b, fmt)


def _c_oracle(c_ref, name):
    op, fmt, shape, seed, extra = cases.CASES[name]
    a, b = cases.make_inputs(name)
    if op == "cost_volume":
        return c_ref.cost_volume(a, b, extra.get("search_range", 4), fmt)
    return c_ref.warp(a, b, fmt, "clamp" if op == "warp_v2" else "tfwarp")


@pytest.mark.parametrize("name", sorted(cases.CASES))
def test_numpy_oracle_f32_matches_golden(name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    out = _np_oracle(name, np.float32)
    assert tuple(out.shape) == tuple(g["shape"])
    flat = out.reshape(-1).astype(np.float64)
    # golden = float64 restatement; fp32 evaluation of the same ops stays within 1e-5
    np.testing.assert_allclose(flat[g["idx"]], g["val"], rtol=0, atol=2e-5)
    assert abs(flat.sum() - g["total"]) <= 1e-6 * max(1.0, g["abs_total"])


@pytest.mark.parametrize("name", sorted(cases.CASES))
def test_c_oracle_matches_golden(c_oracle, name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    out = _c_oracle(c_oracle, name)
    flat = out.reshape(-1).astype(np.float64)
    np.testing.assert_allclose(flat[g["idx"]], g["val"], rtol=0, atol=2e-5)
    assert abs(flat.sum() - g["total"]) <= 1e-6 * max(1.0, g["abs_total"])


def test_known_answer_3x3_warp():
    """qpwcnet/app/optical_flow/test_warp.py:28-33: one-hot at (1,1), flow (x=1,y=0)
    broadcast from (1,1,1,2) -> WarpV2 output one-hot at (row 1, col 0)."""
    g = np.load(os.path.join(GOLDEN, "known_3x3.npz"))
    out = np_ref.warp_v2(g["nxt"], g["flo"])
    expect = np.zeros((3, 3), np.float32)
    expect[1, 0] = 1.0
    np.testing.assert_array_equal(out[0, ..., 0], expect)
    np.testing.assert_array_equal(g["warp_v2"][0, ..., 0], expect)
    # tf_warp agrees here: the sampled column x+1 <= W-1 only reaches W-1 at x = 1..2
    out1 = np_ref.tf_warp(g["nxt"], g["flo"])
    np.testing.assert_array_equal(out1, g["tf_warp"])


def test_cost_volume_channel_order_via_vis():
    """nxt = prv moved by (+dy rows, +dx cols) -> argmax channel (dy+4)*9+(dx+4) and
    cost_volume_to_flow returns (dy, dx) in the interior (qpwcnet/core/vis.py:22-34)."""
    rng = np.random.default_rng(5)
    prv = rng.standard_normal((1, 24, 28, 128)).astype(np.float32)
    for dy, dx in ((0, 0), (2, -3), (-4, 4), (1, 0)):
        nxt = np.roll(prv, (dy, dx), axis=(1, 2))
        cv = np_ref.cost_volume(prv, nxt)
        k = np.argmax(cv, axis=-1)[0, 8:16, 8:20]
        assert np.all(k == (dy + 4) * 9 + (dx + 4))
        flow = np_ref.cost_volume_to_flow(cv)[0, 8:16, 8:20]
        assert np.all(flow[..., 0] == dy) and np.all(flow[..., 1] == dx)


def test_cost_volume_constant_input():
    """prv == nxt == c: centre channel lrelu(c*c); border channels lose the padded part."""
    c = np.float32(1.5)
    x = np.full((1, 10, 12, 4), c, np.float32)
    cv = np_ref.cost_volume(x, x)
    assert np.allclose(cv[..., 40], c * c)
    assert np.allclose(cv[0, 0, 0, 0], 0.0)        # (dy,dx)=(-4,-4) at the corner: all padding
    assert np.allclose(cv[0, 5, 6], c * c)         # interior pixel sees no padding


def test_cost_volume_negative_goes_through_lrelu():
    x = np.ones((1, 10, 10, 2), np.float32)
    cv = np_ref.cost_volume(x, -x)
    assert np.allclose(cv[0, 5, 5], -0.1)


def test_warp_zero_flow():
    """WarpV2 = identity; Warp (tf_warp) = identity except last row and last column = 0
    (SURVEY 8(c) known answer 4)."""
    rng = np.random.default_rng(6)
    img = rng.random((2, 7, 9, 3)).astype(np.float32)
    flo = np.zeros((2, 7, 9, 2), np.float32)
    v2 = np_ref.warp_v2(img, flo)
    np.testing.assert_array_equal(v2[:, :-1, :-1], img[:, :-1, :-1])
    # last row/column: floor clamps to size-2 and alpha = 1 -> 1*(b-a)+a, one rounding off
    np.testing.assert_allclose(v2, img, rtol=0, atol=1.2e-7)
    out = np_ref.tf_warp(img, flo)
    np.testing.assert_array_equal(out[:, :-1, :-1], img[:, :-1, :-1])
    assert np.all(out[:, -1] == 0) and np.all(out[:, :, -1] == 0)


def test_warp_integer_flow_is_shift():
    rng = np.random.default_rng(7)
    img = rng.random((1, 8, 10, 2)).astype(np.float32)
    flo = np.zeros((1, 8, 10, 2), np.float32)
    flo[..., 0] = 2.0   # x + 2
    flo[..., 1] = -1.0  # y - 1
    ys = np.clip(np.arange(8) - 1, 0, 7)
    xs = np.clip(np.arange(10) + 2, 0, 9)
    np.testing.assert_allclose(np_ref.warp_v2(img, flo), img[:, ys][:, :, xs], atol=1e-7)
    out = np_ref.tf_warp(img, flo)
    np.testing.assert_allclose(out[:, 1:, :7], img[:, ys][:, :, xs][:, 1:, :7], atol=1e-7)
    assert np.all(out[:, :, 7:] == 0)  # x + 2 >= W-1 -> both x weights vanish


def test_warp_v1_v2_agree_in_the_interior():
    """Identical for 0 <= q < size-1: the only cross-check the
    reference offers for WarpV2 (test/test_warp.py:26-28)."""
    rng = np.random.default_rng(8)
    img = rng.random((2, 32, 40, 3)).astype(np.float32)
    flo = rng.standard_normal((2, 32, 40, 2)).astype(np.float32)
    a, b = np_ref.tf_warp(img, flo), np_ref.warp_v2(img, flo)
    yy, xx = np.meshgrid(np.arange(32), np.arange(40), indexing="ij")
    qx, qy = xx + flo[..., 0], yy + flo[..., 1]
    inside = (qx >= 0) & (qx < 39) & (qy >= 0) & (qy < 31)
    assert inside.mean() > 0.8
    np.testing.assert_allclose(a[inside], b[inside], atol=2e-6)


def test_dense_image_warp_sign_flip():
    """In-tree dense_image_warp uses grid + flow (warp.py:201); upstream grid - flow."""
    rng = np.random.default_rng(9)
    img = rng.random((1, 6, 7, 2)).astype(np.float32)
    flo = rng.standard_normal((1, 6, 7, 2)).astype(np.float32)
    np.testing.assert_array_equal(np_ref.dense_image_warp(img, flo),
                                  np_ref.tfa_dense_image_warp(img, -flo))
    np.testing.assert_array_equal(np_ref.warp_v2(img, flo),
                                  np_ref.dense_image_warp(img, flo[..., ::-1]))


def test_layouts_agree():
    rng = np.random.default_rng(10)
    a = rng.standard_normal((2, 9, 11, 5)).astype(np.float32)
    b = rng.standard_normal((2, 9, 11, 5)).astype(np.float32)
    f = rng.standard_normal((2, 9, 11, 2)).astype(np.float32)
    t = lambda x: np.transpose(x, (0, 3, 1, 2))
    np.testing.assert_allclose(t(np_ref.cost_volume(a, b)), np_ref.cost_volume(t(a), t(b), 4, "channels_first"),
                               atol=1e-6)
    np.testing.assert_array_equal(t(np_ref.warp_v2(a, f)), np_ref.warp_v2(t(a), t(f), "channels_first"))
    np.testing.assert_array_equal(t(np_ref.tf_warp(a, f)), np_ref.tf_warp(t(a), t(f), "channels_first"))


def test_errors():
    x = np.zeros((1, 4, 4, 2), np.float32)
    with pytest.raises(ValueError, match="Unsupported data format"):
        np_ref.cost_volume(x, x, 4, "nhwc")
    with pytest.raises(ValueError):
        np_ref.tf_warp(x[0], x[0])                 # unbatched: error in the reference too
    with pytest.raises(ValueError):
        np_ref.warp_v2(np.zeros((1, 1, 4, 2), np.float32), np.zeros((1, 1, 4, 2), np.float32))


def test_torch_restatement_matches_numpy():
    import torch
    rng = np.random.default_rng(11)
    a = rng.standard_normal((2, 12, 20, 8)).astype(np.float32)
    b = rng.standard_normal((2, 12, 20, 8)).astype(np.float32)
    f = (rng.standard_normal((2, 12, 20, 2)) * 3).astype(np.float32)
    ta, tb, tf_ = map(torch.from_numpy, (a, b, f))
    np.testing.assert_allclose(torch_ref.cost_volume(ta, tb).numpy(), np_ref.cost_volume(a, b), atol=1e-6)
    np.testing.assert_array_equal(torch_ref.warp_v2(ta, tf_).numpy(), np_ref.warp_v2(a, f))
    np.testing.assert_array_equal(torch_ref.tf_warp(ta, tf_).numpy(), np_ref.tf_warp(a, f))
    assert abs(float(torch_ref.epe_error(ta[..., :2], tb[..., :2])) -
               np_ref.epe_error(a[..., :2], b[..., :2])) < 1e-6


# ---- inverse flow / occlusion map ------------------------------------------
@pytest.mark.parametrize("name", sorted(cases.OCC_CASES))
def test_occlusion_oracle_matches_golden(name):
    fmt = cases.OCC_CASES[name][0]
    flow = cases.make_flow(name)
    gold = np.load(os.path.join(GOLDEN, name + ".npz"))
    assert np.array_equal(np_ref.estimate_occlusion_map(flow, fmt), gold["occ"].astype(np.float32))
    assert np.array_equal(np_ref.invert_flow(flow, fmt), gold["inv"])


def _occlusion_loops(flow):
    """Scalar restatement of occlusion.py:50-98 (channels_last), pixel by pixel."""
    n, h, w, _ = flow.shape
    f32 = np.float32
    inv = np_ref.invert_flow(flow)
    out = np.ones((n, h, w), np.float32)
    for b in range(n):
        for i in range(h):
            for j in range(w):
                ti = min(max(int(f32(i) + inv[b, i, j, 1]), 0), h - 1)   # int(): truncation
                tj = min(max(int(f32(j) + inv[b, i, j, 0]), 0), w - 1)
                out[b, ti, tj] = 0.0
    for b in range(n):
        for i in range(h):
            for j in range(w):
                i2, j2 = f32(i) + flow[b, i, j, 1], f32(j) + flow[b, i, j, 0]
                if i2 < 0 or i2 >= h or j2 < 0 or j2 >= w:
                    out[b, i, j] = 1.0
    return out


def test_occlusion_oracle_against_scalar_loops():
    flow = cases.make_flow("occ_noise")
    assert np.array_equal(np_ref.estimate_occlusion_map(flow), _occlusion_loops(flow))


def test_occlusion_known_answers_and_layouts():
    z = np.zeros((2, 5, 7, 2), np.float32)
    assert not np_ref.estimate_occlusion_map(z).any()          # zero flow: nothing occluded
    f = np.zeros((1, 6, 12, 2), np.float32)
    f[..., 0] = 2.0                                            # everything moves 2 px right
    occ = np_ref.estimate_occlusion_map(f)
    assert (occ[0, :, -2:] == 1).all()                         # the last two columns leave the image
    # inverse of a uniform shift is the opposite shift, except where tf_warp's far-border
    # rule (weights vanish at x >= W-1 and on the last row) zeroes the sample
    inv = np_ref.invert_flow(f)
    assert (inv[0, :-1, :9, 0] == -2).all() and (inv[0, :, 9:, 0] == 0).all() and (inv[0, -1] == 0).all()
    assert (inv[..., 1] == 0).all()
    flow = cases.make_flow("occ_nhwc")
    nchw = np.ascontiguousarray(np.moveaxis(flow, 3, 1))
    assert np.array_equal(np_ref.estimate_occlusion_map(flow),
                          np_ref.estimate_occlusion_map(nchw, "channels_first"))
    with pytest.raises(ValueError):
        np_ref.estimate_occlusion_map(flow[0])


# ---- A2: tfa CorrelationCost restated on its own (oracle/tfa_ref.py) ---------------------------
# The reference's claim (qpwcnet/app/test/test_cvol_equal.py:9-25, test/test_cost_volume.py:16-24):
# CostVolume (in-tree pure TF) and CostVolumeV2 (tfa CorrelationCost(1, 4, 1, 1, 4)) print a summed
# difference of 0.  Here: the two independent RESTATEMENTS agree, at the reference's own shapes.
_CVEQ_SHAPES = [((4, 32, 64, 3), "channels_last"), ((4, 3, 32, 64), "channels_first"),
                ((1, 128, 256, 3), "channels_last"), ((1, 3, 128, 256), "channels_first")]


@pytest.mark.parametrize("shape,fmt", _CVEQ_SHAPES)
def test_tfa_correlation_cost_equals_in_tree_cost_volume_before_lrelu(c_oracle, shape, fmt):
    rng = np.random.default_rng(77)
    prv = rng.standard_normal(shape).astype(np.float32)   # tf.random.normal, test_cost_volume.py:20-21
    nxt = rng.standard_normal(shape).astype(np.float32)
    v1 = np_ref.cost_volume(prv, nxt, 4, fmt, activation=False)
    v2 = tfa_ref.correlation_cost(prv, nxt, 1, 4, 1, 1, 4, fmt)
    assert v1.shape == v2.shape
    np.testing.assert_allclose(v2, v1, rtol=0, atol=1e-6)
    assert abs(float((v1 - v2).sum())) <= 1e-5            # what test_cvol_equal.py:25 prints
    # float64: the two op orders differ only by rounding
    v1d = np_ref.cost_volume(prv.astype(np.float64), nxt.astype(np.float64), 4, fmt, activation=False)
    v2d = tfa_ref.correlation_cost(prv.astype(np.float64), nxt.astype(np.float64), 1, 4, 1, 1, 4, fmt)
    np.testing.assert_allclose(v2d, v1d, rtol=0, atol=1e-14)
    # C scalar loop == numpy form of the same published algorithm; and with the layer's LeakyReLU
    np.testing.assert_allclose(c_oracle.correlation_cost(prv, nxt, data_format=fmt), v2, rtol=0, atol=1e-6)
    np.testing.assert_allclose(c_oracle.cost_volume_v2(prv, nxt, 4, fmt), np_ref.cost_volume(prv, nxt, 4, fmt),
                               rtol=0, atol=1e-6)
    np.testing.assert_allclose(tfa_ref.cost_volume_v2(prv, nxt, 4, fmt), np_ref.cost_volume(prv, nxt, 4, fmt),
                               rtol=0, atol=1e-6)


@pytest.mark.parametrize("fmt", ["channels_last", "channels_first"])
@pytest.mark.parametrize("ks,md,s1,s2,pad", [(1, 4, 1, 1, 4), (3, 2, 1, 1, 2), (1, 4, 1, 2, 4), (3, 4, 2, 2, 4),
                                             (1, 2, 1, 1, 0)])
def test_tfa_correlation_cost_three_forms_agree(c_oracle, fmt, ks, md, s1, s2, pad):
    """Scalar 7-deep loop (the published functor, literally) == vectorised numpy form == C loop, also
    away from the reference's arguments (kernel window, strides, no padding)."""
    rng = np.random.default_rng(ks * 100 + md * 10 + s2)
    shape = (2, 6, 7, 3) if fmt == "channels_last" else (2, 3, 6, 7)
    a = rng.standard_normal(shape).astype(np.float32)
    b = rng.standard_normal(shape).astype(np.float32)
    loops = tfa_ref.correlation_cost_loops(a, b, ks, md, s1, s2, pad, fmt)
    np.testing.assert_allclose(tfa_ref.correlation_cost(a, b, ks, md, s1, s2, pad, fmt), loops, rtol=0, atol=1e-6)
    np.testing.assert_allclose(c_oracle.correlation_cost(a, b, ks, md, s1, s2, pad, fmt), loops, rtol=0, atol=1e-6)


def test_tfa_correlation_cost_known_answers():
    """Channel (tj+4)*9 + (ti+4) with tj the ROW displacement: a one-hot pair and the
    shifted-noise argmax of vis.cost_volume_to_flow (qpwcnet/core/vis.py:22-32)."""
    a = np.zeros((1, 12, 12, 1), np.float32)
    b = np.zeros((1, 12, 12, 1), np.float32)
    a[0, 5, 6, 0] = 2.0
    b[0, 5 + 3, 6 - 2, 0] = 4.0          # b is a moved by (+3 rows, -2 cols)
    out = tfa_ref.correlation_cost(a, b)
    expect = np.zeros_like(out)
    expect[0, 5, 6, (3 + 4) * 9 + (-2 + 4)] = 8.0
    np.testing.assert_array_equal(out, expect)
    # constant images: every in-image displacement gives const^2, out-of-image ones 0 (zero padding)
    c = np.full((1, 10, 10, 4), 0.5, np.float32)
    out = tfa_ref.correlation_cost(c, c)
    assert out[0, 5, 5].min() == 0.25 and out[0, 0, 0, 40] == 0.25
    assert out[0, 0, 0, 0] == 0.0 and out[0, 9, 9, 80] == 0.0
    rng = np.random.default_rng(6)
    prv = rng.standard_normal((1, 24, 28, 64)).astype(np.float32)
    for dy, dx in ((2, -3), (-4, 4)):
        nxt = np.roll(prv, (dy, dx), axis=(1, 2))
        k = np.argmax(tfa_ref.cost_volume_v2(prv, nxt), axis=-1)[0, 8:16, 8:20]
        assert np.all(k == (dy + 4) * 9 + (dx + 4))

"""GPU suite (-m gpu): the HIP path, called through the layer facades and the C
ABI, against the committed golden vectors and the CPU oracle on the same seeded
inputs.  Tolerance: BASELINE.json north_star -- 1e-4 absolute in fp32."""
import os

import numpy as np
import pytest
import torch

import cases
from oracle import c_ref, np_ref, net_ref, tfa_ref, torch_ref
from qpwcnet_amd import layers, non_layers, ops, synth, warp as warp_mod
from qpwcnet_amd.backend import image_data_format, set_image_data_format
from qpwcnet_amd.pwcnet import build_flower

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 1e-4      # north_star: "within 1e-4 fp32 on identical inputs"
DEV = "cuda:0"


def gpu(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(DEV)


def _facade(mod, name):
    op, fmt, shape, seed, extra = cases.CASES[name]
    if op == "cost_volume":
        return mod.CostVolumeV2(search_range=extra.get("search_range", 4), data_format=fmt)
    if op == "warp_v2":
        return mod.WarpV2(data_format=fmt)
    return mod.Warp(data_format=fmt)


def _oracle_f32(name):
    op, fmt, shape, seed, extra = cases.CASES[name]
    a, b = cases.make_inputs(name)
    if op == "cost_volume":
        return c_ref.cost_volume(a, b, extra.get("search_range", 4), fmt)
    return c_ref.warp(a, b, fmt, "clamp" if op == "warp_v2" else "tfwarp")


@pytest.fixture(scope="module", autouse=True)
def _loaded():
    from qpwcnet_amd import _hip
    assert os.path.exists(_hip.LIB_PATH), "libqpwc_hip.so must be built in-tree"
    _hip.lib()
    c_ref.build()


@pytest.mark.parametrize("mod", [layers, non_layers], ids=["layers", "non_layers"])
@pytest.mark.parametrize("name", sorted(cases.CASES))
def test_golden_and_oracle(name, mod):
    a, b = cases.make_inputs(name)
    out = _facade(mod, name)((gpu(a), gpu(b))).cpu().numpy()
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    assert tuple(out.shape) == tuple(g["shape"])
    flat = out.reshape(-1).astype(np.float64)
    np.testing.assert_allclose(flat[g["idx"]], g["val"], rtol=0, atol=TOL)
    assert abs(flat.sum() - g["total"]) <= 1e-5 * max(1.0, g["abs_total"])
    np.testing.assert_allclose(out, _oracle_f32(name), rtol=0, atol=TOL)


def test_warp_is_bit_exact_against_the_unfused_c_oracle():
    """Same fp32 op sequence, no FMA contraction on either side."""
    for name in ("warp2_test_nhwc", "warp1_test_nhwc", "warp2_border", "warp1_border", "warp2_L4"):
        a, b = cases.make_inputs(name)
        out = _facade(non_layers, name)((gpu(a), gpu(b))).cpu().numpy()
        np.testing.assert_array_equal(out, _oracle_f32(name), err_msg=name)


@pytest.mark.parametrize("mod", [layers, non_layers], ids=["layers", "non_layers"])
@pytest.mark.parametrize("cn,shape", [("channels_last", (4, 32, 64, 3)), ("channels_first", (4, 3, 32, 64)),
                                      ("channels_last", (1, 128, 256, 3)), ("channels_first", (1, 3, 128, 256))])
def test_cost_volume_v2_against_the_tfa_correlation_cost_restatement(mod, cn, shape):
    """A2: CostVolumeV2 (the variant the network instantiates, use_tfa=True) against the INDEPENDENT
    restatement of tfa CorrelationCost(1, 4, 1, 1, 4) + leaky_relu (oracle/tfa_ref.py; layers.py:124-132)
    at the reference's own shapes (test/test_cost_volume.py:16-21, app/test/test_cvol_equal.py:11), and
    CostVolume (V1) against the in-tree restatement; the reference's printed sum of differences V1 - V2
    (test_cvol_equal.py:25) is checked between the two ORACLES in tests/test_oracle.py, not between two
    names of one kernel."""
    fmt0 = image_data_format()
    try:
        set_image_data_format(cn)   # the layers read the global at construction (layers.py:41)
        rng = np.random.default_rng(99)
        prv, nxt = rng.standard_normal(shape).astype(np.float32), rng.standard_normal(shape).astype(np.float32)
        v2 = mod.CostVolumeV2(4)((gpu(prv), gpu(nxt))).cpu().numpy()
        v1 = mod.CostVolume(4)((gpu(prv), gpu(nxt))).cpu().numpy()
    finally:
        set_image_data_format(fmt0)
    np.testing.assert_allclose(v2, tfa_ref.cost_volume_v2(prv, nxt, 4, cn), rtol=0, atol=TOL)
    np.testing.assert_allclose(v2, c_ref.cost_volume_v2(prv, nxt, 4, cn), rtol=0, atol=TOL)
    np.testing.assert_allclose(v1, np_ref.cost_volume(prv, nxt, 4, cn), rtol=0, atol=TOL)


@pytest.mark.parametrize("name", ["cv_L0", "cv_L1", "cv_L2", "cv_L3", "cv_L4", "cv_ragged", "cv_r2"])
def test_cost_volume_v2_level_shapes_against_tfa_restatement(name):
    """The same at the five 256x512 level shapes (every matrix-core kernel path), a ragged shape and
    search_range 2 -> CorrelationCost(1, 2, 1, 1, 2): scalar C loop of the published tfa algorithm."""
    op, fmt, shape, seed, extra = cases.CASES[name]
    r = extra.get("search_range", 4)
    a, b = cases.make_inputs(name)
    out = non_layers.CostVolumeV2(search_range=r, data_format=fmt)((gpu(a), gpu(b))).cpu().numpy()
    np.testing.assert_allclose(out, c_ref.cost_volume_v2(a, b, r, fmt), rtol=0, atol=TOL)


def test_known_answer_3x3_and_broadcast_flow():
    """qpwcnet/app/optical_flow/test_warp.py:28-33, flow broadcast from (1,1,1,2)."""
    g = np.load(os.path.join(GOLDEN, "known_3x3.npz"))
    out = layers.WarpV2(data_format="channels_last")((gpu(g["nxt"]), gpu(g["flo"]))).cpu().numpy()
    expect = np.zeros((3, 3), np.float32)
    expect[1, 0] = 1.0
    np.testing.assert_array_equal(out[0, ..., 0], expect)
    out1 = layers.Warp(data_format="channels_last")((gpu(g["nxt"]), gpu(g["flo"]))).cpu().numpy()
    np.testing.assert_array_equal(out1, g["tf_warp"])
    # partial broadcasts
    rng = np.random.default_rng(3)
    img = rng.random((3, 9, 11, 8)).astype(np.float32)
    for fshape in ((1, 9, 11, 2), (3, 1, 11, 2), (3, 9, 1, 2), (1, 1, 1, 2)):
        flo = rng.standard_normal(fshape).astype(np.float32) * 2
        for mode, ref in (("clamp", np_ref.warp_v2), ("tfwarp", np_ref.tf_warp)):
            o = ops.warp(gpu(img), gpu(flo), mode).cpu().numpy()
            np.testing.assert_allclose(o, ref(img, flo), atol=1e-6, err_msg=str(fshape))


def test_zero_flow_known_answers():
    rng = np.random.default_rng(4)
    img = rng.random((2, 16, 20, 12)).astype(np.float32)
    flo = np.zeros((2, 16, 20, 2), np.float32)
    v2 = ops.warp(gpu(img), gpu(flo), "clamp").cpu().numpy()
    np.testing.assert_allclose(v2, img, rtol=0, atol=1.2e-7)
    v1 = ops.warp(gpu(img), gpu(flo), "tfwarp").cpu().numpy()
    np.testing.assert_array_equal(v1[:, :-1, :-1], img[:, :-1, :-1])
    assert np.all(v1[:, -1] == 0) and np.all(v1[:, :, -1] == 0)


def test_function_level_mirrors():
    rng = np.random.default_rng(5)
    img = rng.random((2, 10, 12, 4)).astype(np.float32)
    flo = rng.standard_normal((2, 10, 12, 2)).astype(np.float32)
    np.testing.assert_array_equal(warp_mod.tf_warp(gpu(img), gpu(flo), "channels_last").cpu().numpy(),
                                  np_ref.tf_warp(img, flo))
    np.testing.assert_array_equal(warp_mod.dense_image_warp(gpu(img), gpu(flo)).cpu().numpy(),
                                  np_ref.dense_image_warp(img, flo))


def test_channels_first_stored_channels_last():
    """A logical NCHW tensor in torch channels_last memory takes the NHWC kernels."""
    rng = np.random.default_rng(6)
    a = rng.standard_normal((2, 16, 24, 32)).astype(np.float32)
    b = rng.standard_normal((2, 16, 24, 32)).astype(np.float32)
    f = rng.standard_normal((2, 16, 24, 2)).astype(np.float32) * 3
    an, bn, fn = (gpu(x).permute(0, 3, 1, 2) for x in (a, b, f))    # NCHW views of NHWC memory
    cv = ops.cost_volume(an, bn, 4, "channels_first")
    assert cv.shape == (2, 81, 16, 24) and cv.is_contiguous(memory_format=torch.channels_last)
    np.testing.assert_allclose(cv.permute(0, 2, 3, 1).cpu().numpy(), c_ref.cost_volume(a, b), atol=TOL)
    w = ops.warp(an, fn, "clamp", "channels_first")
    np.testing.assert_array_equal(w.permute(0, 2, 3, 1).cpu().numpy(), c_ref.warp(a, f))


def test_strided_output_into_concat_buffer():
    rng = np.random.default_rng(7)
    B, H, W, C = 2, 20, 28, 16
    prv = rng.standard_normal((B, H, W, C)).astype(np.float32)
    nxt = rng.standard_normal((B, H, W, C)).astype(np.float32)
    feat = torch.full((B, H, W, 81 + C + 2), 7.0, device=DEV)
    ops.cost_volume_into(gpu(prv), gpu(nxt), feat, 0)
    ref = c_ref.cost_volume(prv, nxt)
    np.testing.assert_allclose(feat[..., :81].cpu().numpy(), ref, atol=TOL)
    assert bool((feat[..., 81:] == 7.0).all())          # nothing else touched
    feat2 = torch.full((B, H, W, 100), 7.0, device=DEV)
    ops.cost_volume_into(gpu(prv), gpu(nxt), feat2, 10)
    np.testing.assert_allclose(feat2[..., 10:91].cpu().numpy(), ref, atol=TOL)
    assert bool((feat2[..., :10] == 7.0).all()) and bool((feat2[..., 91:] == 7.0).all())
    with pytest.raises(ValueError):
        ops.cost_volume_into(gpu(prv), gpu(nxt), feat2, 30)


@pytest.mark.parametrize("shape", [(2, 16, 32, 256), (1, 64, 128, 64), (2, 19, 37, 8), (8, 8, 16, 32)])
def test_fused_warp_cost_volume(shape):
    """UpFlow front end (non_layers.py:377-380) in one launch == warp then cost volume."""
    rng = np.random.default_rng(8)
    prv = rng.standard_normal(shape).astype(np.float32)
    nxt = rng.standard_normal(shape).astype(np.float32)
    flo = (rng.standard_normal(shape[:3] + (2,)) * 3).astype(np.float32)
    fused = ops.warp_cost_volume(gpu(prv), gpu(nxt), gpu(flo)).cpu().numpy()
    ref = c_ref.cost_volume(prv, c_ref.warp(nxt, flo))
    np.testing.assert_allclose(fused, ref, rtol=0, atol=TOL)
    unfused = ops.cost_volume(gpu(prv), ops.warp(gpu(nxt), gpu(flo), "clamp")).cpu().numpy()
    np.testing.assert_allclose(fused, unfused, rtol=0, atol=1e-5)


def test_epe():
    rng = np.random.default_rng(9)
    a = rng.standard_normal((3, 33, 47, 2)).astype(np.float32)
    b = rng.standard_normal((3, 33, 47, 2)).astype(np.float32)
    ref = c_ref.epe(a, b)
    assert abs(float(ops.epe(gpu(a), gpu(b))) - ref) < 1e-5
    an, bn = np.transpose(a, (0, 3, 1, 2)), np.transpose(b, (0, 3, 1, 2))
    assert abs(float(ops.epe(gpu(an), gpu(bn), "channels_first")) - ref) < 1e-5


def test_fp16_storage_path():
    """Config 5: fp16 storage, fp32 accumulate.  The reference has no fp16 path; the
    bound is build-defined: oracle on the fp16-rounded inputs, output rounding only."""
    rng = np.random.default_rng(10)
    for shape in ((2, 16, 32, 64), (2, 9, 13, 6)):
        a = rng.standard_normal(shape).astype(np.float16)
        b = rng.standard_normal(shape).astype(np.float16)
        f = (rng.standard_normal(shape[:3] + (2,)) * 2).astype(np.float32)
        cv = ops.cost_volume(gpu(a), gpu(b)).float().cpu().numpy()
        ref = c_ref.cost_volume(a.astype(np.float32), b.astype(np.float32))
        np.testing.assert_allclose(cv, ref, rtol=1e-3, atol=1e-3)
        w = ops.warp(gpu(a), gpu(f), "clamp").float().cpu().numpy()
        np.testing.assert_allclose(w, c_ref.warp(a.astype(np.float32), f), rtol=1e-3, atol=1e-3)


def test_error_behaviour_on_device():
    x = torch.zeros(1, 8, 8, 4, device=DEV)
    with pytest.raises(ValueError, match="Unsupported data format"):
        ops.cost_volume(x, x, 4, "nhwc")
    with pytest.raises(ValueError):
        ops.cost_volume(x, torch.zeros(1, 8, 9, 4, device=DEV))
    with pytest.raises(ValueError):
        ops.warp(x[0], torch.zeros(8, 8, 2, device=DEV))                 # unbatched
    with pytest.raises(ValueError, match="at least 2x2"):
        ops.warp(torch.zeros(1, 1, 8, 4, device=DEV), torch.zeros(1, 1, 8, 2, device=DEV))
    with pytest.raises(ValueError):
        ops.warp(x, torch.zeros(1, 8, 8, 3, device=DEV))
    with pytest.raises(ValueError):
        ops.cost_volume(x, x, search_range=-1)
    # tf_warp has no 2x2 restriction
    ops.warp(torch.zeros(1, 1, 8, 4, device=DEV), torch.zeros(1, 1, 8, 2, device=DEV), "tfwarp")


# ---- BASELINE.json full sizes: oracle where it finishes in seconds, plus
# ---- size-independent properties ------------------------------------------
LEVELS_256 = [(8, 16, 256), (16, 32, 256), (32, 64, 128), (64, 128, 64), (128, 256, 32)]


@pytest.mark.parametrize("hwc", LEVELS_256)
def test_config2_level_shapes_batch8(hwc):
    H, W, C = hwc
    rng = np.random.default_rng(20 + H)
    shape = (8, H, W, C)
    prv = rng.standard_normal(shape).astype(np.float32)
    nxt = rng.standard_normal(shape).astype(np.float32)
    flo = (rng.standard_normal((8, H, W, 2)) * 4).astype(np.float32)
    cv = ops.cost_volume(gpu(prv), gpu(nxt)).cpu().numpy()
    np.testing.assert_allclose(cv, c_ref.cost_volume(prv, nxt), rtol=0, atol=TOL)
    w = ops.warp(gpu(nxt), gpu(flo), "clamp").cpu().numpy()
    np.testing.assert_array_equal(w, c_ref.warp(nxt, flo))
    fused = ops.warp_cost_volume(gpu(prv), gpu(nxt), gpu(flo)).cpu().numpy()
    np.testing.assert_allclose(fused, c_ref.cost_volume(prv, w), rtol=0, atol=TOL)


def test_config4_sintel_full_level_properties():
    """1024x2048 finest level (512,1024,32), B=2: properties that need no oracle."""
    g = torch.Generator(device=DEV).manual_seed(0)
    B, H, W, C = 2, 512, 1024, 32
    prv = torch.randn(B, H, W, C, device=DEV, generator=g)
    # (1) shifted copy: the matching displacement channel equals mean(prv^2) in the interior
    dy, dx = 3, -2
    nxt = torch.roll(prv, (dy, dx), dims=(1, 2))
    cv = ops.cost_volume(prv, nxt)
    k = (dy + 4) * 9 + (dx + 4)
    auto = (prv * prv).mean(dim=3)
    assert torch.allclose(cv[:, 8:-8, 8:-8, k], auto[:, 8:-8, 8:-8], atol=1e-5)
    assert bool((cv[:, 8:-8, 8:-8].argmax(dim=3) == k).float().mean() > 0.999)
    # (2) positive homogeneity: cv(a*prv, nxt) == a*cv(prv, nxt), a > 0 (lrelu is homogeneous)
    cv2 = ops.cost_volume(2.0 * prv, nxt)
    assert torch.allclose(cv2, 2.0 * cv, atol=1e-5)
    # (3) zero padding: displacement (-4,-4) at the top-left corner is all padding
    assert float(cv[:, 0, 0, 0].abs().max()) == 0.0
    # (4) integer flow == roll in the interior, and fused == unfused
    flo = torch.zeros(B, H, W, 2, device=DEV)
    flo[..., 0], flo[..., 1] = 2.0, -1.0
    w = ops.warp(prv, flo, "clamp")
    assert torch.equal(w[:, 4:-4, 4:-4], torch.roll(prv, (1, -2), dims=(1, 2))[:, 4:-4, 4:-4])
    fused = ops.warp_cost_volume(prv, nxt, flo)
    assert torch.allclose(fused, ops.cost_volume(prv, ops.warp(nxt, flo, "clamp")), atol=1e-5)
    # (5) spot check against the numpy oracle on a crop far from the borders
    ys, xs = slice(200, 232), slice(500, 532)
    crop = np_ref.cost_volume(prv[:1, 192:240, 492:540].cpu().numpy(), nxt[:1, 192:240, 492:540].cpu().numpy())
    np.testing.assert_allclose(cv[:1, ys, xs].cpu().numpy(), crop[:, 8:40, 8:40], atol=TOL)


# ---- whole network: GPU (HIP hot path + PyTorch-ROCm convs) vs CPU oracle ---
@pytest.mark.parametrize("fused", [False, True], ids=["unfused", "fused"])
@pytest.mark.parametrize("hw,batch", [((64, 128), 2), ((256, 512), 1)])
def test_full_network_per_level_epe(hw, batch, fused):
    weights = synth.make_weights(42, hw)
    pairs, _ = synth.make_frames(batch, hw[0], hw[1], seed=1234)
    model = build_flower(True, hw, "channels_last", weights=weights, device=DEV, fused=fused)
    flows = model.predict(pairs)
    ref = net_ref.RefNet(weights)(pairs)
    assert len(flows) == 6
    for lvl, (a, b) in enumerate(zip(flows, ref)):
        assert tuple(a.shape) == tuple(b.shape)
        e = float(torch_ref.epe_error(a.cpu(), b))
        assert e < TOL, "level {} EPE vs oracle {:.3e}".format(lvl, e)
    # inference graph (train=False) returns only the full-resolution flow
    last = build_flower(False, hw, "channels_last", weights=weights, device=DEV, fused=fused).predict(pairs)
    assert torch.allclose(last, flows[-1], rtol=0, atol=1e-5)  # conv solvers may differ run to run


def test_full_network_channels_first():
    hw = (64, 128)
    weights = synth.make_weights(42, hw)
    pairs, _ = synth.make_frames(1, hw[0], hw[1], seed=1234)
    model = build_flower(True, hw, "channels_first", weights=weights, device=DEV)
    flows = model.predict(np.ascontiguousarray(np.transpose(pairs, (0, 3, 1, 2))))
    ref = net_ref.RefNet(weights)(pairs)
    for a, b in zip(flows, ref):
        assert float(torch_ref.epe_error(a.permute(0, 2, 3, 1).cpu(), b)) < TOL


def test_full_network_fp16_config5():
    """BASELINE configs[4]: fp16 storage/convs, fp32 accumulation in the hot path.  The
    reference has no fp16 path; the bound is build-defined: per-level EPE vs the fp32 oracle
    below 5 % of the flow magnitude at that level."""
    hw = (64, 128)
    weights = synth.make_weights(42, hw)
    pairs, _ = synth.make_frames(2, hw[0], hw[1], seed=1234)
    model = build_flower(True, hw, "channels_last", weights=weights, device=DEV, dtype=torch.float16)
    flows = model.predict(pairs)
    ref = net_ref.RefNet(weights)(pairs)
    for lvl, (a, b) in enumerate(zip(flows, ref)):
        assert a.dtype == torch.float16
        e = float(torch_ref.epe_error(a.float().cpu(), b))
        mag = float(torch.linalg.vector_norm(b, dim=-1).mean())
        assert e < 0.05 * max(mag, 0.1), "level {}: EPE {:.3e} vs |flow| {:.3e}".format(lvl, e, mag)


@pytest.mark.parametrize("shape", [(32, 36, 52, 32), (40, 30, 44, 64), (3, 19, 37, 32), (5, 7, 9, 64),
                                   (2, 12, 20, 16), (1, 4, 4, 256), (2, 70, 290, 32)])   # last: 2 strips + a remainder
def test_matrix_core_kernels_ragged_edges(shape):
    """H, W not multiples of the 4x4 tile / 8x8 region: workgroup-shared kernel (many
    regions) and per-wave split-K kernel (few tiles), zero padding at every border."""
    rng = np.random.default_rng(shape[1] * 100 + shape[2])
    prv = rng.standard_normal(shape).astype(np.float32)
    nxt = rng.standard_normal(shape).astype(np.float32)
    out = ops.cost_volume(gpu(prv), gpu(nxt)).cpu().numpy()
    np.testing.assert_allclose(out, c_ref.cost_volume(prv, nxt), rtol=0, atol=TOL)
    # fp16 storage on the same shapes (fp16 matrix cores where eligible)
    ph, nh = prv.astype(np.float16), nxt.astype(np.float16)
    outh = ops.cost_volume(gpu(ph), gpu(nh)).float().cpu().numpy()
    refh = c_ref.cost_volume(ph.astype(np.float32), nh.astype(np.float32))
    np.testing.assert_allclose(outh, refh, rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("shape", [(32, 36, 52, 32), (40, 30, 44, 64), (20, 33, 47, 128), (8, 128, 256, 32),
                                   (32, 36, 52, 64), (30, 33, 47, 128), (2, 150, 290, 32), (8, 40, 290, 64),
                                   (32, 50, 70, 64), (44, 33, 60, 128), (48, 33, 47, 96)])   # >= 512 16 x 16 regions; 8 x 16 at C = 96
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16], ids=["f32", "f16"])
def test_fused_front_end_on_the_matrix_cores_ragged_edges_and_far_flows(shape, dtype):
    """qpwc_warp_cost_volume_fwd where the workgroup-shared matrix-core kernel takes it (>= 256 regions, C % 32 == 0):
    H, W not multiples of the 8 x 8 region, flows that leave the image on every side (clamp-to-border taps) and
    land exactly on the last row / column; every output form (dense 81, 84 with zero pads, strided into a concat
    buffer).  The gather blends with the WarpV2 kernel's own code: the result must equal warp -> cost volume bit for
    bit, and the C oracle within the fp32 / fp16-rounding bound."""
    assert non_layers.fused_kernel_applies(torch.empty(shape, device=DEV, dtype=dtype))
    B, H, W, C = shape
    g = torch.Generator(device=DEV).manual_seed(H * W + C)
    prv = torch.randn(*shape, device=DEV, generator=g).to(dtype)
    nxt = torch.randn(*shape, device=DEV, generator=g).to(dtype)
    flo = torch.randn(B, H, W, 2, device=DEV, generator=g) * 6
    flo[:, :2] = 40.0                      # far below / right of the image
    flo[:, -2:] = -40.0                    # far above / left
    flo[:, 5, :, 0] = (W - 1) - torch.arange(W, device=DEV, dtype=torch.float32)   # exactly the last column
    flo[:, 5, :, 1] = float(H - 1 - 5)                                             # ... of the last row
    flo[:, 7] = 0.0
    unf = ops.cost_volume(prv, ops.warp(nxt, flo, "clamp"))
    dense = ops.warp_cost_volume(prv, nxt, flo)
    assert torch.equal(dense, unf)
    pad = torch.full((B, H, W, 84), float("nan"), device=DEV, dtype=dtype)
    ops.cost_volume_into(prv, nxt, pad, 0, flo=flo)
    assert torch.equal(pad[..., :81], unf) and float(pad[..., 81:].abs().max()) == 0.0
    wide = torch.full((B, H, W, 81 + C + 2), 7.0, device=DEV, dtype=dtype)
    ops.cost_volume_into(prv, nxt, wide, 0, flo=flo)
    assert torch.equal(wide[..., :81], unf) and bool((wide[..., 81:] == 7.0).all())
    sub = [0, B - 1]
    p32, n32, f32 = (t[sub].float().cpu().numpy() for t in (prv, nxt, flo))
    ref = c_ref.cost_volume(p32, c_ref.warp(n32, f32).astype(np.float16 if dtype == torch.float16 else np.float32)
                            .astype(np.float32))
    got = dense[sub].float().cpu().numpy()
    tol = TOL if dtype == torch.float32 else TOL + 2.0 ** -11 * np.abs(ref)
    assert np.all(np.abs(got - ref) <= tol)


@pytest.mark.parametrize("shape", [(16, 64, 64, 32), (2, 16, 32, 64)])
def test_matrix_core_kernels_strided_output(shape):
    """81 channels written at a channel offset of a wider buffer (the concat target)."""
    rng = np.random.default_rng(7)
    prv = rng.standard_normal(shape).astype(np.float32)
    nxt = rng.standard_normal(shape).astype(np.float32)
    B, H, W, C = shape
    feat = torch.full((B, H, W, 81 + C + 2), 7.0, device=DEV)
    ops.cost_volume_into(gpu(prv), gpu(nxt), feat, 0)
    np.testing.assert_allclose(feat[..., :81].cpu().numpy(), c_ref.cost_volume(prv, nxt), atol=TOL)
    assert bool((feat[..., 81:] == 7.0).all())


def test_epe_multi_level():
    from qpwcnet_amd import metrics
    rng = np.random.default_rng(12)
    shapes = [(3, 8, 16), (3, 16, 32), (3, 33, 47), (3, 128, 256)]
    a = [rng.standard_normal(s + (2,)).astype(np.float32) for s in shapes]
    b = [rng.standard_normal(s + (2,)).astype(np.float32) for s in shapes]
    out = ops.epe_multi([gpu(x) for x in a], [gpu(x) for x in b]).cpu().numpy()
    ref = np.asarray([c_ref.epe(x, y) for x, y in zip(a, b)])
    np.testing.assert_allclose(out, ref, rtol=1e-5)
    out2 = metrics.per_level_epe([gpu(x) for x in a], [gpu(x) for x in b]).cpu().numpy()
    np.testing.assert_allclose(out2, ref, rtol=1e-5)


@pytest.mark.parametrize("fmt", ["channels_last", "channels_first"])
def test_epe_multi_level_takes_fp16_predictions_as_they_are(fmt):
    """qpwc_epe_multi_mixed_fwd: fp16 predictions (the fp16-storage network's flows) are converted inside the
    reduction -- same result as converting them first; mixed fp16 / fp32 levels in one call."""
    rng = np.random.default_rng(13)
    shapes = [(3, 8, 16), (3, 16, 32), (3, 33, 47), (2, 128, 256)]
    a = [rng.standard_normal(s + (2,)).astype(np.float32) for s in shapes]
    b = [rng.standard_normal(s + (2,)).astype(np.float16) for s in shapes]
    if fmt == "channels_first":
        a = [np.ascontiguousarray(x.transpose(0, 3, 1, 2)) for x in a]
        b = [np.ascontiguousarray(x.transpose(0, 3, 1, 2)) for x in b]
    ta, tb = [gpu(x) for x in a], [gpu(x) for x in b]
    tb[1] = tb[1].float()                                            # one fp32 level among fp16 ones
    out = ops.epe_multi(ta, tb, data_format=fmt).cpu().numpy()
    ref = ops.epe_multi(ta, [t.float() for t in tb], data_format=fmt).cpu().numpy()
    np.testing.assert_allclose(out, ref, rtol=1e-6)
    axis = 1 if fmt == "channels_first" else 3
    ref64 = np.asarray([np.sqrt(((x.astype(np.float64) - y.astype(np.float64)) ** 2).sum(axis=axis)).mean() for x, y in zip(a, b)])
    np.testing.assert_allclose(out, ref64, rtol=1e-5)


def test_randomised_shapes_every_kernel_path():
    """Seeded random shapes through every dispatch path (vector kernel, per-wave split-K,
    workgroup-shared, generic) against the C oracle."""
    rng = np.random.default_rng(2026)
    for _ in range(24):
        B = int(rng.integers(1, 5))
        H, W = int(rng.integers(2, 41)), int(rng.integers(2, 41))
        C = int(rng.choice([1, 3, 4, 8, 12, 16, 32, 48, 64, 96]))
        prv = rng.standard_normal((B, H, W, C)).astype(np.float32)
        nxt = rng.standard_normal((B, H, W, C)).astype(np.float32)
        flo = (rng.standard_normal((B, H, W, 2)) * 3).astype(np.float32)
        msg = "shape {}".format((B, H, W, C))
        np.testing.assert_allclose(ops.cost_volume(gpu(prv), gpu(nxt)).cpu().numpy(),
                                   c_ref.cost_volume(prv, nxt), rtol=0, atol=TOL, err_msg=msg)
        for mode in ("clamp", "tfwarp"):
            np.testing.assert_array_equal(ops.warp(gpu(nxt), gpu(flo), mode).cpu().numpy(),
                                          c_ref.warp(nxt, flo, mode=mode), err_msg=msg + " " + mode)
        if C % 4 == 0:
            fused = ops.warp_cost_volume(gpu(prv), gpu(nxt), gpu(flo)).cpu().numpy()
            np.testing.assert_allclose(fused, c_ref.cost_volume(prv, c_ref.warp(nxt, flo)), rtol=0,
                                       atol=TOL, err_msg=msg + " fused")


def test_graphed_forward_replays_match_eager():
    from qpwcnet_amd.pwcnet import GraphedForward
    hw = (64, 128)
    weights = synth.make_weights(42, hw)
    pairs, _ = synth.make_frames(2, hw[0], hw[1], seed=1)
    pairs2, _ = synth.make_frames(2, hw[0], hw[1], seed=2)
    model = build_flower(True, hw, "channels_last", weights=weights, device=DEV)
    g = GraphedForward(model, gpu(pairs))
    for p in (pairs2, pairs):
        outs, _ = g.replay(gpu(p))
        eager = model.predict(p)
        for a, b in zip(outs, eager):
            assert torch.allclose(a, b, rtol=0, atol=1e-5)


def test_config4_full_network_one_pair_1024x2048():
    """BASELINE configs[3] resolution, one pair: per-level EPE vs the CPU oracle."""
    hw = (1024, 2048)
    weights = synth.make_weights(42, hw)
    pairs, _ = synth.make_frames(1, hw[0], hw[1], seed=3)
    flows = build_flower(True, hw, "channels_last", weights=weights, device=DEV).predict(pairs)
    ref = net_ref.RefNet(weights)(pairs)
    for lvl, (a, b) in enumerate(zip(flows, ref)):
        e = float(torch_ref.epe_error(a.cpu(), b))
        assert e < TOL, "level {} EPE vs oracle {:.3e}".format(lvl, e)


def test_cost_volume_output_only_element_aligned():
    """The C ABI asks for element alignment only: an output that is 4-byte but not 16-byte
    aligned must take the scalar-store path of the tile epilogue."""
    from qpwcnet_amd import _hip
    rng = np.random.default_rng(13)
    shape = (16, 64, 64, 32)
    prv = rng.standard_normal(shape).astype(np.float32)
    nxt = rng.standard_normal(shape).astype(np.float32)
    p, n = gpu(prv), gpu(nxt)
    buf = torch.zeros(16 * 64 * 64 * 81 + 4, device=DEV)
    out = buf[1:1 + 16 * 64 * 64 * 81]
    assert out.data_ptr() % 16 == 4
    rc = _hip.lib().qpwc_cost_volume_fwd(p.data_ptr(), n.data_ptr(), out.data_ptr(), 16, 64, 64, 32, 4,
                                         _hip.NHWC, _hip.F32, 0.1, torch.cuda.current_stream().cuda_stream)
    _hip.check(rc)
    np.testing.assert_allclose(out.view(16, 64, 64, 81).cpu().numpy(), c_ref.cost_volume(prv, nxt), atol=TOL)


@pytest.mark.parametrize("shape", [(8, 128, 256, 32), (2, 64, 128, 64), (2, 16, 32, 256), (2, 8, 16, 256),
                                   (2, 19, 37, 32), (1, 12, 20, 8)])
def test_cost_volume_84_channel_padded_layout(shape):
    """out_pixel_stride 84 at offset 0: channels 0..80 = the cost volume, 81..83 = 0 (16-byte aligned
    pixels for vector consumers), whatever kernel path the shape takes and whatever the buffer held."""
    g = torch.Generator(device=DEV).manual_seed(shape[1])
    prv = torch.randn(*shape, device=DEV, generator=g)
    nxt = torch.randn(*shape, device=DEV, generator=g)
    dense = ops.cost_volume(prv, nxt)
    buf = torch.full(shape[:3] + (84,), float("nan"), device=DEV)
    ops.cost_volume_into(prv, nxt, buf, 0)
    assert torch.equal(buf[..., :81], dense)
    assert float(buf[..., 81:].abs().max()) == 0.0
    # an 84-wide destination at a non-zero offset is an ordinary strided write: neighbours untouched
    wide = torch.full(shape[:3] + (88,), 7.0, device=DEV)
    ops.cost_volume_into(prv, nxt, wide, 4)
    assert torch.equal(wide[..., 4:85], dense)
    assert bool((wide[..., :4] == 7).all()) and bool((wide[..., 85:] == 7).all())
    h = ops.cost_volume(prv.half(), nxt.half())
    bh = torch.full(shape[:3] + (84,), float("nan"), device=DEV, dtype=torch.float16)
    ops.cost_volume_into(prv.half(), nxt.half(), bh, 0)
    assert torch.equal(bh[..., :81], h) and float(bh[..., 81:].abs().max()) == 0.0


@pytest.mark.parametrize("shape,dtype", [((2, 24, 28, 81), torch.float32), ((1, 7, 5, 81), torch.float16),
                                         ((3, 9, 11, 25), torch.float32), ((8, 128, 256, 81), torch.float32)])
def test_cost_volume_to_flow_decode(shape, dtype):
    """cost_volume_to_flow (qpwcnet/core/vis.py:9-34) on the device: bit-exact against the numpy
    restatement, both layouts, ties resolved to the first maximum like tf.argmax, unbatched input, and the
    84-channel padded volume decoded in place."""
    from qpwcnet_amd import vis
    g = torch.Generator(device="cpu").manual_seed(shape[1])
    cv = torch.randn(*shape, generator=g).to(dtype)
    cv[0, 1, 1, :] = 0.25            # a pixel whose channels all tie: argmax = 0
    cv[0, 2, 3, 7] = cv[0, 2, 3, shape[3] - 3] = 9.0   # two equal maxima: the first one
    ref = np_ref.cost_volume_to_flow(cv.float().numpy())
    out = vis.cost_volume_to_flow(cv.to(DEV), "channels_last")
    assert out.dtype == torch.float32
    np.testing.assert_array_equal(out.cpu().numpy(), ref)
    out_cf = ops.cost_volume_to_flow(cv.permute(0, 3, 1, 2).contiguous().to(DEV), "channels_first")
    np.testing.assert_array_equal(out_cf.permute(0, 2, 3, 1).cpu().numpy(), ref)
    np.testing.assert_array_equal(ops.cost_volume_to_flow(cv[0].to(DEV)).cpu().numpy(), ref[0])
    if shape[3] == 81:
        wide = torch.full(shape[:3] + (84,), 100.0, dtype=dtype)     # pads must not be read
        wide[..., :81] = cv
        np.testing.assert_array_equal(ops.cost_volume_to_flow(wide.to(DEV)[..., :81]).cpu().numpy(), ref)


def test_cost_volume_to_flow_recovers_a_shift_on_the_device():
    """nxt = prv moved by (dy, dx): the decoded flow of the HIP cost volume is (dy, dx) in the interior."""
    rng = np.random.default_rng(5)
    prv = rng.standard_normal((2, 24, 32, 32)).astype(np.float32)
    for dy, dx in ((0, 0), (2, -3), (-4, 4)):
        nxt = np.roll(prv, (dy, dx), axis=(1, 2))
        cv = ops.cost_volume(torch.from_numpy(prv).to(DEV), torch.from_numpy(nxt).to(DEV))
        flow = ops.cost_volume_to_flow(cv).cpu().numpy()[:, 8:16, 8:24]
        assert np.all(flow[..., 0] == dy) and np.all(flow[..., 1] == dx)
    with pytest.raises(ValueError):
        ops.cost_volume_to_flow(torch.zeros(4, 4, device=DEV))

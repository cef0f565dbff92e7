"""GPU suite: the OptFlow pieces -- multi-source depthwise 3x3 and
the fused flow head -- against the torch-CPU restatement of qpwcnet/core/non_layers.py:213-273."""
import numpy as np
import pytest
import torch

from oracle import net_ref, torch_ref
from qpwcnet_amd import non_layers, ops, synth
from qpwcnet_amd.pwcnet import build_flower

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-4


def _rand(rng, *shape):
    return torch.from_numpy(rng.standard_normal(shape).astype(np.float32))


@pytest.mark.parametrize("chans", [(81, 32, 2), (81, 256, 256), (128,), (7, 3)])
@pytest.mark.parametrize("hw", [(8, 16), (19, 37)])
@pytest.mark.parametrize("act", [False, True])
def test_dwconv3x3_multi_source(chans, hw, act):
    rng = np.random.default_rng(sum(chans) + hw[0])
    H, W = hw
    srcs = [_rand(rng, 2, H, W, c) for c in chans]
    w = _rand(rng, sum(chans), 1, 3, 3)
    ref = torch_ref.depthwise3x3(srcs, w, act)
    out = ops.dwconv3x3([s.to(DEV) for s in srcs], w.to(DEV), mish_on_load=act).cpu()
    assert out.shape == ref.shape
    torch.testing.assert_close(out, ref, rtol=0, atol=2e-5)


def test_dwconv3x3_strided_source_view():
    """A source may be a channel slice of a wider channels-last buffer (pixel stride > channels)."""
    rng = np.random.default_rng(3)
    big = _rand(rng, 2, 12, 10, 40)
    w = _rand(rng, 16, 1, 3, 3)
    ref = torch_ref.depthwise3x3([big[..., 8:24]], w)
    out = ops.dwconv3x3([big.to(DEV)[..., 8:24]], w.to(DEV)).cpu()
    torch.testing.assert_close(out, ref, rtol=0, atol=2e-5)


def test_dwconv3x3_errors():
    x = torch.zeros(1, 4, 4, 8, device=DEV)
    with pytest.raises(ValueError):
        ops.dwconv3x3([x], torch.zeros(7, 3, 3, device=DEV))
    with pytest.raises(ValueError):
        ops.dwconv3x3([x, x, x, x], torch.zeros(32, 3, 3, device=DEV))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.dwconv3x3([x.cpu()], torch.zeros(8, 3, 3))


@pytest.mark.parametrize("hw", [(8, 16), (33, 50), (128, 256)])
def test_flow_head(hw):
    rng = np.random.default_rng(hw[0])
    H, W = hw
    z = _rand(rng, 2, H, W, 16)
    w1, b1 = _rand(rng, 16, 16, 1, 1) * 0.3, _rand(rng, 16) * 0.1
    gamma, beta = 1 + 0.1 * _rand(rng, 16), 0.1 * _rand(rng, 16)
    mean, var = 0.1 * _rand(rng, 16), 1 + 0.2 * torch.rand(16)
    wf = _rand(rng, 2, 16, 3, 3) * 0.2
    scale = float(H * H + W * W) ** 0.5
    ref = torch_ref.flow_head(z, w1, b1, gamma, beta, mean, var, 1e-3, wf, scale)
    params = non_layers.pack_flow_head(*(t.to(DEV) for t in (w1, b1, gamma, beta, mean, var)), 1e-3,
                                       wf.to(DEV))
    out = ops.flow_head(z.to(DEV), params, scale).cpu()
    # outputs are O(scale): compare relative to the scale factor
    torch.testing.assert_close(out / scale, ref / scale, rtol=0, atol=2e-5)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("hw", [(16, 32), (33, 47), (5, 9), (128, 256), (1, 1), (17, 16)])
def test_flow_head_up_is_flow_head_plus_upsample(hw, dtype):
    """qpwc_flow_head_up_fwd (round 4): flow head and the x2 upsampling that follows it in ONE launch -- both outputs bit for
    bit what the two launches write (tiles with the rim recomputed, ragged sizes, image edges, both storage types)."""
    rng = np.random.default_rng(hw[0] * 131 + hw[1])
    H, W = hw
    B = 3
    z = _rand(rng, B, H, W, 16).to(dtype).to(DEV)
    w1, b1 = _rand(rng, 16, 16, 1, 1) * 0.3, _rand(rng, 16) * 0.1
    gamma, beta = 1 + 0.1 * _rand(rng, 16), 0.1 * _rand(rng, 16)
    mean, var = 0.1 * _rand(rng, 16), 1 + 0.2 * torch.rand(16)
    wf = _rand(rng, 2, 16, 3, 3) * 0.2
    scale = float(H * H + W * W) ** 0.5
    params = non_layers.pack_flow_head(*(t.to(DEV) for t in (w1, b1, gamma, beta, mean, var)), 1e-3, wf.to(DEV))
    flow = ops.flow_head(z, params, scale)
    up = ops.upsample2x_flow(flow, 2.0)
    flow1, up1 = ops.flow_head_up(z, params, scale, 2.0)
    assert torch.equal(flow1, flow)
    assert torch.equal(up1, up)
    if dtype == torch.float16:   # the fp32 coordinates of the next level's warp, written by the same launch
        assert torch.equal(up1._qpwc_f32, up.float())
    else:
        assert not hasattr(up1, "_qpwc_f32")


@pytest.mark.parametrize("M,C,F", [(1024, 596, 128), (4096, 342, 128), (16384, 211, 128), (1000, 37, 16), (17, 64, 32),
                                   (5, 596, 256), (4096, 128, 64)])
def test_pointwise_bias_against_float64(M, C, F):
    """qpwc_pointwise_bias_fwd (round 4): the pointwise half of a split SeparableConv2D, y . W^T + b on the fp32 matrix
    instructions, against float64 -- the step's two shapes (L0 / L1 first layers), odd channel counts (rows that straddle
    the 16-byte loads), a last tile with fewer than 16 rows, every output width; and run-to-run bit stability."""
    rng = np.random.default_rng(M + C + F)
    y = _rand(rng, M, C).to(DEV)
    w = (_rand(rng, F, C) / np.sqrt(C)).to(DEV)
    b = _rand(rng, F).to(DEV)
    out = ops.pointwise_bias(y, ops.pad_pointwise(w), b)
    ref = (y.double() @ w.double().t() + b.double())
    torch.testing.assert_close(out.double(), ref, rtol=0, atol=2e-5)
    lib = torch.addmm(b, y, w.t())
    torch.testing.assert_close(out, lib, rtol=0, atol=2e-5)
    for _ in range(5):
        assert torch.equal(ops.pointwise_bias(y, ops.pad_pointwise(w), b), out)
    with pytest.raises(ValueError):
        ops.pointwise_bias(y, ops.pad_pointwise(w)[:, :-32].contiguous() if C > 32 else w, b)


def test_optflow_block_hip_equals_torch_path():
    """OptFlow.from_sources (HIP) == OptFlow(concat) (PyTorch convs) == oracle."""
    hw = (32, 64)
    weights = synth.make_weights(42, (256, 512))
    rng = np.random.default_rng(5)
    cost, prv, flo = _rand(rng, 2, *hw, 81), _rand(rng, 2, *hw, 128), _rand(rng, 2, *hw, 2)
    params = {k: torch.as_tensor(v).to(DEV) for k, v in weights.items()}
    of = non_layers.OptFlow(params, "upflow.1.flow.", data_format="channels_last")
    srcs = [t.to(DEV) for t in (cost, prv, flo)]
    a = of.from_sources(srcs).cpu()
    b = of(torch.cat(srcs, dim=3)).cpu()
    ref = net_ref.RefNet(weights).opt_flow("upflow.1.flow.", torch.cat([cost, prv, flo], dim=3))
    scale = float(hw[0] ** 2 + hw[1] ** 2) ** 0.5
    torch.testing.assert_close(a / scale, ref / scale, rtol=0, atol=2e-5)
    torch.testing.assert_close(b / scale, ref / scale, rtol=0, atol=2e-5)


@pytest.mark.parametrize("hip_optflow", [False, True])
def test_full_network_both_optflow_paths(hip_optflow):
    hw = (64, 128)
    weights = synth.make_weights(42, hw)
    pairs, _ = synth.make_frames(2, hw[0], hw[1], seed=1234)
    model = build_flower(True, hw, "channels_last", weights=weights, device=DEV, hip_optflow=hip_optflow)
    flows = model.predict(pairs)
    ref = net_ref.RefNet(weights)(pairs)
    for lvl, (x, y) in enumerate(zip(flows, ref)):
        e = float(torch_ref.epe_error(x.cpu(), y))
        assert e < TOL, "level {} EPE vs oracle {:.3e}".format(lvl, e)


def test_bias_mish_inplace():
    """`activation='Mish'` epilogue (qpwcnet/core/mish.py:27-28) vs the oracle's definition."""
    rng = np.random.default_rng(1)
    x = torch.from_numpy((rng.standard_normal((3, 17, 19, 32)) * 6).astype(np.float32))
    x[0, 0, 0, :4] = torch.tensor([25.0, -30.0, 0.0, 88.0])     # softplus threshold / tails
    b = torch.from_numpy(rng.standard_normal(32).astype(np.float32))
    ref = torch_ref.mish(x.double() + b.double()).float()
    out = ops.bias_mish_(x.to(DEV).clone(), b.to(DEV)).cpu()
    torch.testing.assert_close(out, ref, rtol=2e-6, atol=2e-6)
    out2 = ops.bias_mish_(x.to(DEV).clone()).cpu()
    torch.testing.assert_close(out2, torch_ref.mish(x.double()).float(), rtol=2e-6, atol=2e-6)
    with pytest.raises(ValueError):
        ops.bias_mish_(torch.zeros(2, 2, 2, 6, device=DEV))


@pytest.mark.parametrize("hw", [(8, 16), (5, 7), (128, 256)])
def test_upsample2x_flow(hw):
    """Upsample functor (non_layers.py:183-193) vs the oracle's F.interpolate restatement."""
    rng = np.random.default_rng(hw[0])
    f = torch.from_numpy(rng.standard_normal((2, hw[0], hw[1], 2)).astype(np.float32))
    ref = net_ref.RefNet.upsample(f, 2.0)
    out = ops.upsample2x_flow(f.to(DEV), 2.0).cpu()
    torch.testing.assert_close(out, ref, rtol=0, atol=2e-6)
    out2 = non_layers.Upsample(scale=2.0, data_format="channels_last")(f.to(DEV)).cpu()
    torch.testing.assert_close(out2, ref, rtol=0, atol=2e-6)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16], ids=["f32", "f16"])
@pytest.mark.parametrize("hw", [(1, 1), (1, 5), (7, 1), (9, 13), (64, 128)])
def test_upsample2x_flow_pair_kernel_equals_the_pixel_kernel(hw, dtype):
    """Channels-last calls take the two-pixels-per-thread kernel (vector loads, one 16- / 8-byte store per pair); the
    channels_first layout still runs the one-pixel kernel: same expressions, so the two must agree bit for bit."""
    rng = np.random.default_rng(hw[0] * 31 + hw[1])
    f = torch.from_numpy(rng.standard_normal((3, hw[0], hw[1], 2)).astype(np.float32)).to(DEV, dtype)
    pair = ops.upsample2x_flow(f, 2.0)
    pixel = ops.upsample2x_flow(f.permute(0, 3, 1, 2).contiguous(), 2.0, in_format="channels_first",
                                out_format="channels_first")
    assert tuple(pair.shape) == (3, 2 * hw[0], 2 * hw[1], 2)
    assert torch.equal(pair, pixel.permute(0, 2, 3, 1))


def test_optflow_pieces_fp16_storage():
    """fp16 storage / fp32 arithmetic variants of the OptFlow kernels (BASELINE configs[4]);
    bound: the fp32 oracle on the fp16-rounded inputs, output rounding only."""
    rng = np.random.default_rng(11)
    srcs = [_rand(rng, 2, 12, 20, c).half() for c in (81, 32, 2)]
    w = _rand(rng, 115, 1, 3, 3)
    for act in (False, True):
        ref = torch_ref.depthwise3x3([s.float() for s in srcs], w, act)
        out = ops.dwconv3x3([s.to(DEV) for s in srcs], w.to(DEV), mish_on_load=act)
        assert out.dtype == torch.float16
        torch.testing.assert_close(out.float().cpu(), ref, rtol=2e-3, atol=2e-3)
    # every source a multiple of 4 channels in aligned pixels (config 5's level-0 layer: 84 + 256 + 256): the kernel with
    # 4 channels per lane over a virtual concat (round 4) -- same bound, and the same bits as one dense source
    srcs4 = [_rand(rng, 3, 9, 17, c).half() for c in (84, 256, 256)]
    w4 = _rand(rng, 596, 1, 3, 3)
    for act in (False, True):
        ref = torch_ref.depthwise3x3([s.float() for s in srcs4], w4, act)
        out = ops.dwconv3x3([s.to(DEV) for s in srcs4], w4.to(DEV), mish_on_load=act)
        torch.testing.assert_close(out.float().cpu(), ref, rtol=2e-3, atol=2e-3)
        one = ops.dwconv3x3([torch.cat(srcs4, dim=3).to(DEV)], w4.to(DEV), mish_on_load=act)
        assert torch.equal(out, one)
    x = (_rand(rng, 2, 9, 11, 16) * 3).half()
    b = _rand(rng, 16)
    out = ops.bias_mish_(x.to(DEV).clone(), b.to(DEV)).float().cpu()
    torch.testing.assert_close(out, torch_ref.mish(x.float() + b), rtol=2e-3, atol=2e-3)
    f = _rand(rng, 2, 6, 9, 2).half()
    up = ops.upsample2x_flow(f.to(DEV), 2.0).float().cpu()
    torch.testing.assert_close(up, net_ref.RefNet.upsample(f.float(), 2.0), rtol=2e-3, atol=2e-3)


@pytest.mark.parametrize("shape", [(8, 128, 256, 32), (2, 64, 128, 64), (2, 19, 37, 32)])
def test_cost_volume_fp16_matrix_core_kernel(shape):
    """v_mfma_f32_16x16x32_f16 path: exact fp16 products, fp32 accumulation."""
    from oracle import c_ref
    rng = np.random.default_rng(shape[1])
    a = rng.standard_normal(shape).astype(np.float16)
    b = rng.standard_normal(shape).astype(np.float16)
    out = ops.cost_volume(torch.from_numpy(a).to(DEV), torch.from_numpy(b).to(DEV)).float().cpu().numpy()
    ref = c_ref.cost_volume(a.astype(np.float32), b.astype(np.float32))
    np.testing.assert_allclose(out, ref, rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("chans,F", [((81, 32, 2), 128), ((81, 256, 256), 128), ((128,), 64), ((64,), 32),
                                     ((32,), 16), ((7, 3), 16),
                                     # three steps and more on the narrow kernels (two staging sets, round 3):
                                     # 3 / 4 / 5 steps, and a 3-source layer whose outputs are split over workgroups
                                     ((96,), 32), ((128,), 32), ((160,), 16), ((84, 64, 2), 32)])
@pytest.mark.parametrize("hw", [(8, 16), (19, 37)])
@pytest.mark.parametrize("act", [False, True])
def test_sepconv3x3_fused(chans, F, hw, act):
    """Fused SeparableConv2D (depthwise on chip -> matrix-core pointwise + bias) vs the oracle."""
    rng = np.random.default_rng(sum(chans) + F + hw[0])
    H, W = hw
    C = sum(chans)
    srcs = [_rand(rng, 2, H, W, c) for c in chans]
    dw = _rand(rng, C, 1, 3, 3)
    pw = _rand(rng, F, C, 1, 1) / np.sqrt(C)
    bias = _rand(rng, F)
    y = torch_ref.depthwise3x3(srcs, dw, act)
    ref = torch.nn.functional.conv2d(y.permute(0, 3, 1, 2), pw, bias).permute(0, 2, 3, 1)
    out = ops.sepconv3x3([s.to(DEV) for s in srcs], dw.to(DEV), ops.pad_pointwise(pw.to(DEV)),
                         bias.to(DEV), mish_on_load=act).cpu()
    assert out.shape == ref.shape
    torch.testing.assert_close(out, ref, rtol=0, atol=5e-5)
    # the layer's own activation applied at the store (what a following fused layer consumes)
    out_act = ops.sepconv3x3([s.to(DEV) for s in srcs], dw.to(DEV), ops.pad_pointwise(pw.to(DEV)),
                             bias.to(DEV), mish_on_load=act, mish_on_store=True).cpu()
    torch.testing.assert_close(out_act, torch_ref.mish(ref), rtol=0, atol=5e-5)


@pytest.mark.parametrize("C,F", [(64, 32), (32, 16), (40, 16)])
@pytest.mark.parametrize("act", [False, True])
def test_sepconv3x3_fused_resident_workgroups(C, F, act):
    """Round 3: launches of more than 512 tiles with F <= 32 run as 512 RESIDENT workgroups that walk the tiles and
    request the next tile's first step early.  676 ragged tiles (100 x 200 pixels are 12.5 x 12.5
    tiles), so workgroups own one or two tiles and the last tiles of a row / image are partial; against the oracle."""
    rng = np.random.default_rng(C + F)
    B, H, W = 4, 100, 200
    assert B * ((H + 7) // 8) * ((W + 15) // 16) > 512
    x = _rand(rng, B, H, W, C)
    dw = _rand(rng, C, 1, 3, 3)
    pw = _rand(rng, F, C, 1, 1) / np.sqrt(C)
    bias = _rand(rng, F)
    y = torch_ref.depthwise3x3([x], dw, act)
    ref = torch.nn.functional.conv2d(y.permute(0, 3, 1, 2), pw, bias).permute(0, 2, 3, 1)
    out = ops.sepconv3x3([x.to(DEV)], dw.to(DEV), ops.pad_pointwise(pw.to(DEV)), bias.to(DEV), mish_on_load=act,
                         mish_on_store=True).cpu()
    torch.testing.assert_close(out, torch_ref.mish(ref), rtol=0, atol=5e-5)


@pytest.mark.parametrize("chans,F", [((84, 32, 2), 128), ((128,), 64), ((40,), 128), ((32,), 64), ((160,), 64)])
@pytest.mark.parametrize("act", [False, True])
def test_sepconv3x3_wide_layers_many_tiles(chans, F, act):
    """The wide layers (F = 64 / 128) on a launch of 1183 ragged tiles (7 images of 100 x 200 pixels = 12.5 x 12.5 tiles:
    several rounds of workgroups per CU, partial tiles at the right / bottom edges), 1 / 2 / 4 / 5 steps per tile, a
    three-source first layer with the 2-channel flow tail: against the oracle, and BIT-IDENTICAL to the same images
    launched one at a time (169 tiles).  Written for round 4's flat-pipeline kernel (csrc/experimental/sepconv_flat.inc,
    not in the product build: it failed the bit-identity half of this test through a hardware hazard)."""
    rng = np.random.default_rng(sum(chans) + F)
    B, H, W = 7, 100, 200
    assert B * ((H + 7) // 8) * ((W + 15) // 16) >= 1024 > ((H + 7) // 8) * ((W + 15) // 16)
    C = sum(chans)
    srcs = [_rand(rng, B, H, W, c) for c in chans]
    dw = _rand(rng, C, 1, 3, 3)
    pw = _rand(rng, F, C, 1, 1) / np.sqrt(C)
    bias = _rand(rng, F)
    y = torch_ref.depthwise3x3(srcs, dw, act)
    ref = torch.nn.functional.conv2d(y.permute(0, 3, 1, 2), pw, bias).permute(0, 2, 3, 1)
    d_srcs, d_dw, d_pw, d_b = [s.to(DEV) for s in srcs], dw.to(DEV), ops.pad_pointwise(pw.to(DEV)), bias.to(DEV)
    for store_act in (False, True):
        out = ops.sepconv3x3(d_srcs, d_dw, d_pw, d_b, mish_on_load=act, mish_on_store=store_act)
        torch.testing.assert_close(out.cpu(), torch_ref.mish(ref) if store_act else ref, rtol=0, atol=5e-5)
        for b in (0, B - 1):
            one = ops.sepconv3x3([s[b:b + 1].contiguous() for s in d_srcs], d_dw, d_pw, d_b, mish_on_load=act,
                                 mish_on_store=store_act)
            assert torch.equal(out[b:b + 1], one), "image %d differs between the batched and the single-image launch" % b


@pytest.mark.parametrize("tail,stride", [(1, 1), (1, 2), (2, 2), (3, 3), (3, 4), (1, 5)])
def test_sepconv3x3_fused_short_tail_source(tail, stride):
    """The 16-byte path reads a short last source (1..3 channels) with a load that ENDS at its last channel;
    the first pixels of the tensor, which have fewer floats in front of them than that shift, read forward
    instead.  The tail tensor is placed at the very start of its own allocation (a 1-channel source
    of pixel stride 1 or 2 used to read in front of the tensor's base for pixels 1 and 2 of image 0)."""
    rng = np.random.default_rng(10 * tail + stride)
    B, H, W, F = 2, 9, 13, 16
    a = _rand(rng, B, H, W, 8)
    wide = _rand(rng, B, H, W, stride)
    t = wide[..., :tail]                       # pixel stride `stride`, data pointer = start of the allocation
    dw = _rand(rng, 8 + tail, 1, 3, 3)
    pw = _rand(rng, F, 8 + tail, 1, 1) / 3
    bias = _rand(rng, F)
    y = torch_ref.depthwise3x3([a, t.contiguous()], dw, False)
    ref = torch.nn.functional.conv2d(y.permute(0, 3, 1, 2), pw, bias).permute(0, 2, 3, 1)
    wide_d = wide.to(DEV)
    out = ops.sepconv3x3([a.to(DEV), wide_d[..., :tail]], dw.to(DEV), ops.pad_pointwise(pw.to(DEV)),
                         bias.to(DEV)).cpu()
    torch.testing.assert_close(out, ref, rtol=0, atol=5e-5)


@pytest.mark.parametrize("chans,F", [((128,), 64), ((64,), 32), ((32,), 16), ((40,), 128), ((8,), 16), ((12,), 16),
                                     ((84, 32, 2), 128), ((84, 256, 256), 128), ((84, 64, 2), 64)])
@pytest.mark.parametrize("hw", [(8, 16), (19, 37)])
@pytest.mark.parametrize("act", [False, True])
def test_sepconv3x3_fused_fp16_storage(chans, F, hw, act):
    """qpwc_sepconv3x3_f16_fwd: fp16 in/out, fp32 depthwise rounded to fp16 once, f16 matrix cores with
    fp32 accumulation; one dense source (16-byte loads) or the virtual concat of up to three (8-byte loads,
    short last source element-wise).  Bound: the same rounding points restated on the oracle's ops (input
    and pointwise weights fp16, depthwise result fp16) -> differences are accumulation order + the output
    rounding."""
    C = sum(chans)
    rng = np.random.default_rng(C + F + hw[0])
    H, W = hw
    srcs = [_rand(rng, 2, H, W, c).half() for c in chans]
    dw = _rand(rng, C, 1, 3, 3)
    pw = (_rand(rng, F, C, 1, 1) / np.sqrt(C)).half()
    bias = _rand(rng, F)
    y = torch_ref.depthwise3x3([x.float() for x in srcs], dw, act).half().float()
    ref = torch.nn.functional.conv2d(y.permute(0, 3, 1, 2), pw.float(), bias).permute(0, 2, 3, 1)
    pw_pad = ops.pad_pointwise(pw.to(DEV), torch.float16)
    dsrcs = [x.to(DEV) for x in srcs]
    out = ops.sepconv3x3(dsrcs, dw.to(DEV), pw_pad, bias.to(DEV), mish_on_load=act)
    assert out.dtype == torch.float16 and out.shape == ref.shape
    # depthwise sums that land on an fp16 rounding boundary may round the other way (fma order)
    torch.testing.assert_close(out.float().cpu(), ref, rtol=2e-3, atol=4e-3)
    out_act = ops.sepconv3x3(dsrcs, dw.to(DEV), pw_pad, bias.to(DEV), mish_on_load=act, mish_on_store=True)
    torch.testing.assert_close(out_act.float().cpu(), torch_ref.mish(ref), rtol=2e-3, atol=4e-3)


def test_sepconv3x3_fp16_rejects_unaligned_sources():
    x = torch.zeros((1, 8, 16, 6), dtype=torch.float16, device=DEV)
    dw = torch.zeros((6, 9), device=DEV)
    pw = torch.zeros((16, 32), dtype=torch.float16, device=DEV)
    with pytest.raises(ValueError):   # 6 channels: neither a multiple of 4 nor a short tail
        ops.sepconv3x3([x], dw, pw, torch.zeros(16, device=DEV))
    with pytest.raises(ValueError):   # an odd-width source in front of another one (the dense 81-channel volume)
        ops.sepconv3x3([torch.zeros((1, 8, 16, 81), dtype=torch.float16, device=DEV), x[..., :4].contiguous()],
                       torch.zeros((85, 9), device=DEV), torch.zeros((16, 96), dtype=torch.float16, device=DEV),
                       torch.zeros(16, device=DEV))
    with pytest.raises(ValueError):   # fp32 pointwise weights with fp16 sources
        ops.sepconv3x3([x[..., :4].contiguous()], dw[:4], pw.float(), torch.zeros(16, device=DEV))


def test_optflow_fp16_fused_and_unfused_sepconv_agree():
    hw = (32, 64)
    weights = synth.make_weights(42, (256, 512))
    rng = np.random.default_rng(7)
    srcs = [t.half().to(DEV) for t in (_rand(rng, 2, *hw, 81), _rand(rng, 2, *hw, 128), _rand(rng, 2, *hw, 2))]
    params = {k: torch.as_tensor(v).half().to(DEV) for k, v in weights.items()}
    of = non_layers.OptFlow(params, "upflow.1.flow.", data_format="channels_last")
    default = non_layers.OptFlow.fused_sepconv
    try:
        non_layers.OptFlow.fused_sepconv = True
        a = of.from_sources(srcs).float().cpu()
        # the zero-padded 84-channel cost volume lets the first layer fuse as well
        cost84 = torch.cat([srcs[0], torch.zeros_like(srcs[0][..., :3])], dim=3)
        c = of.from_sources([cost84] + srcs[1:]).float().cpu()
        non_layers.OptFlow.fused_sepconv = False
        b = of.from_sources(srcs).float().cpu()
    finally:
        non_layers.OptFlow.fused_sepconv = default
    scale = float(hw[0] ** 2 + hw[1] ** 2) ** 0.5
    assert a.abs().max() > 0
    torch.testing.assert_close(a / scale, b / scale, rtol=0, atol=3e-3)
    torch.testing.assert_close(c / scale, b / scale, rtol=0, atol=3e-3)


def test_optflow_fused_and_unfused_sepconv_agree():
    hw = (32, 64)
    weights = synth.make_weights(42, (256, 512))
    rng = np.random.default_rng(6)
    srcs = [t.to(DEV) for t in (_rand(rng, 2, *hw, 81), _rand(rng, 2, *hw, 128), _rand(rng, 2, *hw, 2))]
    params = {k: torch.as_tensor(v).to(DEV) for k, v in weights.items()}
    of = non_layers.OptFlow(params, "upflow.1.flow.", data_format="channels_last")
    default = non_layers.OptFlow.fused_sepconv
    try:
        non_layers.OptFlow.fused_sepconv = True
        a = of.from_sources(srcs).cpu()
        non_layers.OptFlow.fused_sepconv = False
        b = of.from_sources(srcs).cpu()
        non_layers.OptFlow.fused_sepconv = None      # per layer by size (the default)
        c = of.from_sources(srcs).cpu()
    finally:
        non_layers.OptFlow.fused_sepconv = default
    scale = float(hw[0] ** 2 + hw[1] ** 2) ** 0.5
    torch.testing.assert_close(a / scale, b / scale, rtol=0, atol=2e-5)
    torch.testing.assert_close(c / scale, b / scale, rtol=0, atol=2e-5)


def test_bias_mish_pad():
    rng = np.random.default_rng(2)
    x = _rand(rng, 3, 10, 14, 16)
    b = _rand(rng, 16)
    out = ops.bias_mish_pad(x.to(DEV), b.to(DEV), 1, 1).cpu()
    assert out.shape == (3, 11, 15, 16)
    torch.testing.assert_close(out[:, :10, :14], torch_ref.mish(x + b), rtol=2e-6, atol=2e-6)
    assert float(out[:, 10].abs().max()) == 0.0 and float(out[:, :, 14].abs().max()) == 0.0


def test_bias_mish_into_concat_buffer():
    rng = np.random.default_rng(4)
    x, b = _rand(rng, 2, 6, 10, 16), _rand(rng, 16)
    dst = torch.full((2, 6, 10, 40), 7.0, device=DEV)
    ops.bias_mish_into(x.to(DEV), b.to(DEV), dst, 8)
    torch.testing.assert_close(dst[..., 8:24].cpu(), torch_ref.mish(x + b), rtol=2e-6, atol=2e-6)
    assert bool((dst[..., :8] == 7.0).all()) and bool((dst[..., 24:] == 7.0).all())
    with pytest.raises(ValueError):
        ops.bias_mish_into(x.to(DEV), b.to(DEV), dst, 30)


@pytest.mark.parametrize("C,F", [(64, 16), (128, 32), (256, 64), (256, 128)])
@pytest.mark.parametrize("hw", [(8, 16), (19, 37), (5, 9)])
def test_upconv4x4s2_mish_into_concat_buffer(C, F, hw):
    """Decoder UpConv (Conv2DTranspose 4x4 stride 2 'same' + bias + Mish, non_layers.py:196-210) written into
    the `up` half of the concat buffer vs torch's transposed convolution; the skip half stays untouched."""
    rng = np.random.default_rng(C + F + hw[0])
    H, W = hw
    x = _rand(rng, 2, H, W, C)
    w = _rand(rng, C, F, 4, 4) / np.sqrt(4 * C)
    b = _rand(rng, F)
    ref = torch_ref.mish(torch.nn.functional.conv_transpose2d(x.permute(0, 3, 1, 2), w, b, stride=2, padding=1))
    ref = ref.permute(0, 2, 3, 1)
    dst = torch.full((2, 2 * H, 2 * W, F + 24), 7.0, device=DEV)
    ops.upconv4x4s2_mish_into(x.to(DEV), ops.upconv_taps(w.to(DEV)), b.to(DEV), dst)
    torch.testing.assert_close(dst[..., :F].cpu(), ref, rtol=0, atol=5e-5)
    assert bool((dst[..., F:] == 7.0).all())
    with pytest.raises(ValueError):
        ops.upconv4x4s2_mish_into(x.to(DEV), ops.upconv_taps(w.to(DEV)), b.to(DEV), dst[:, :-1].contiguous())


@pytest.mark.parametrize("C,F", [(256, 256), (256, 128), (128, 64), (64, 32)])
@pytest.mark.parametrize("hw", [(8, 16), (5, 19), (16, 32)])
def test_upconv4x4s2_mish_fp16_storage(C, F, hw):
    """The fp16-storage twin (qpwc_upconv4x4s2_mish_f16_fwd, BASELINE configs[4]) against fp32 torch on the same
    fp16-rounded operands: the stored value's rounding (2^-11 relative) plus accumulation-order noise."""
    rng = np.random.default_rng(C + F + hw[0] + 3)
    H, W = hw
    x = _rand(rng, 2, H, W, C).half()
    w = (_rand(rng, C, F, 4, 4) / np.sqrt(4 * C)).half()
    b = _rand(rng, F)
    ref = torch_ref.mish(torch.nn.functional.conv_transpose2d(x.float().permute(0, 3, 1, 2), w.float(), b, stride=2, padding=1))
    ref = ref.permute(0, 2, 3, 1)
    dst = torch.full((2, 2 * H, 2 * W, F + 24), 7.0, device=DEV, dtype=torch.float16)
    ops.upconv4x4s2_mish_into(x.to(DEV), ops.upconv_taps(w.to(DEV), torch.float16), b.to(DEV), dst)
    err = (dst[..., :F].cpu().float() - ref).abs()
    assert float((err - (2.0 ** -11) * ref.abs()).max()) <= 2e-5
    assert bool((dst[..., F:] == 7.0).all())


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("C,F", [(64, 16), (128, 32), (256, 64), (256, 128)])
@pytest.mark.parametrize("hw", [(8, 16), (19, 37), (5, 9)])
def test_upconv4x4s2_mish_cat_is_upconv_plus_skip_copy(C, F, hw, dtype):
    """qpwc_upconv4x4s2_mish_cat_fwd (round 4): concat([UpConv(x), skip]) in ONE launch -- bit for bit the transposed
    convolution's own launch plus a copy of the skip, with the skip read through strides (the interior of a zero-bordered
    encoder buffer, as the network hands it over), ragged sizes, both storage types; channels past 2F stay untouched."""
    rng = np.random.default_rng(C + F + hw[0] + 11)
    H, W = hw
    B = 3
    x = _rand(rng, B, H, W, C).to(dtype).to(DEV)
    w = (_rand(rng, C, F, 4, 4) / np.sqrt(4 * C)).to(DEV)
    b = _rand(rng, F).to(DEV)
    taps = ops.upconv_taps(w, dtype)
    padded = torch.zeros(B, 2 * H + 1, 2 * W + 1, F, device=DEV, dtype=dtype)
    padded[:, :2 * H, :2 * W] = _rand(rng, B, 2 * H, 2 * W, F).to(dtype).to(DEV)
    skip = padded[:, :2 * H, :2 * W, :]                       # strides ((2H+1)(2W+1)F, (2W+1)F, F, 1)
    want = torch.full((B, 2 * H, 2 * W, 2 * F + 8), 7.0, device=DEV, dtype=dtype)
    ops.upconv4x4s2_mish_into(x, taps, b, want)
    want[..., F:2 * F] = skip
    got = torch.full_like(want, 7.0)
    assert ops.upconv_cat_ok(x, taps, skip, got)
    ops.upconv4x4s2_mish_cat_into(x, taps, b, skip, got)
    assert torch.equal(got, want)
    # a slice of the batch (the decoder's chunked launches) into the matching slice of the buffer
    got2 = torch.full_like(want, 7.0)
    for sl in (slice(0, 1), slice(1, 3)):
        ops.upconv4x4s2_mish_cat_into(x[sl], taps, b, skip[sl], got2[sl])
    assert torch.equal(got2, want)
    # refused: a skip with another channel count, a destination too narrow for both halves
    assert not ops.upconv_cat_ok(x, taps, skip[..., :F - 4], got)
    with pytest.raises(ValueError):
        ops.upconv4x4s2_mish_cat_into(x, taps, b, skip, got[..., :2 * F - 4].contiguous())


def test_split_frames_pad():
    rng = np.random.default_rng(8)
    x = _rand(rng, 3, 10, 12, 6)
    out = ops.split_frames_pad(x.to(DEV), 1, 1).cpu()
    assert out.shape == (6, 11, 13, 3)
    assert torch.equal(out[:3, :10, :12], x[..., :3]) and torch.equal(out[3:, :10, :12], x[..., 3:])
    assert float(out[:, 10].abs().max()) == 0.0 and float(out[:, :, 12].abs().max()) == 0.0
    assert torch.equal(ops.split_frames_pad(x.to(DEV)).cpu(), torch.cat([x[..., :3], x[..., 3:]], dim=0))


@pytest.mark.parametrize("C", [16, 32, 64, 128, 256])
@pytest.mark.parametrize("hw,pad", [((16, 32), 0), ((19, 37), 0), ((24, 48), 1), ((9, 17), 1), ((8, 16), 0)])
def test_conv3x3_mish_encoder_kernel(C, hw, pad):
    """Encoder conv_aa / conv_b (3x3 'same' + bias + Mish, non_layers.py:410-449) on the matrix cores vs
    torch's convolution; optional zero border = the next stride-2 conv's 'SAME' padding."""
    rng = np.random.default_rng(C + hw[0] + pad)
    H, W = hw
    x = _rand(rng, 3, H, W, C)
    w = _rand(rng, C, C, 3, 3) / np.sqrt(9 * C)
    b = _rand(rng, C)
    ref = torch_ref.mish(torch.nn.functional.conv2d(x.permute(0, 3, 1, 2), w, b, padding=1)).permute(0, 2, 3, 1)
    out = ops.conv3x3_mish(x.to(DEV), ops.conv3x3_taps(w.to(DEV)), b.to(DEV), pad, pad).cpu()
    assert tuple(out.shape) == (3, H + pad, W + pad, C)
    torch.testing.assert_close(out[:, :H, :W], ref, rtol=0, atol=2e-5 if C <= 32 else 5e-5)
    if pad:
        assert float(out[:, H:].abs().max()) == 0.0 and float(out[:, :, W:].abs().max()) == 0.0


@pytest.mark.parametrize("pad", [0, 1])
@pytest.mark.parametrize("hw", [(8, 16), (21, 37), (64, 128)])
@pytest.mark.parametrize("C", [16, 32, 64, 128, 256])
def test_conv3x3_mish_encoder_kernel_fp16_storage(C, hw, pad):
    """The fp16-storage twin (qpwc_conv3x3_mish_f16_fwd, BASELINE configs[4]): fp16 operands, fp32 accumulation,
    ONE rounding at the store -- against fp32 torch on the same fp16-rounded operands; the bound is the rounding
    of the stored value (2^-11 relative) plus accumulation-order noise."""
    rng = np.random.default_rng(C + hw[0] + pad + 7)
    H, W = hw
    x = _rand(rng, 2, H, W, C).half()
    w = (_rand(rng, C, C, 3, 3) / np.sqrt(9 * C)).half()
    b = _rand(rng, C)
    ref = torch_ref.mish(torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2), w.float(), b, padding=1)).permute(0, 2, 3, 1)
    out = ops.conv3x3_mish(x.to(DEV), ops.conv3x3_taps(w.to(DEV), torch.float16), b.to(DEV), pad, pad).cpu()
    assert out.dtype == torch.float16 and tuple(out.shape) == (2, H + pad, W + pad, C)
    err = (out[:, :H, :W].float() - ref).abs()
    assert float((err - (2.0 ** -11) * ref.abs()).max()) <= 2e-5
    if pad:
        assert float(out[:, H:].abs().max()) == 0.0 and float(out[:, :, W:].abs().max()) == 0.0


@pytest.mark.parametrize("hw", [(16, 32), (34, 50), (64, 128)])
def test_first_conv_mish_on_raw_pairs(hw):
    """enc.0.conv_a fused with Split(2), frame stacking and TF 'SAME' stride-2 padding vs torch."""
    rng = np.random.default_rng(hw[0])
    H, W = hw
    pairs = _rand(rng, 3, H, W, 6)
    w = _rand(rng, 16, 3, 3, 3) / np.sqrt(27)
    b = _rand(rng, 16)
    frames = torch.cat([pairs[..., :3], pairs[..., 3:]], dim=0).permute(0, 3, 1, 2)
    ref = torch_ref.mish(torch.nn.functional.conv2d(torch.nn.functional.pad(frames, (0, 1, 0, 1)), w, b,
                                                    stride=2)).permute(0, 2, 3, 1)
    out = ops.first_conv_mish(pairs.to(DEV), ops.first_conv_taps(w.to(DEV)), b.to(DEV)).cpu()
    assert tuple(out.shape) == (6, H // 2, W // 2, 16)
    torch.testing.assert_close(out, ref, rtol=0, atol=2e-5)


@pytest.mark.parametrize("fmt", ["channels_last", "channels_first"])
@pytest.mark.parametrize("hw", [(16, 32), (34, 50)])
def test_first_conv_mish_fp16_storage(hw, fmt):
    """enc.0.conv_a on fp16 pairs (qpwc_first_conv_mish_f16_fwd): exact fp32 products of the fp16 inputs, one rounding
    at the store; both input layouts."""
    rng = np.random.default_rng(hw[0] + 11)
    H, W = hw
    pairs = _rand(rng, 3, H, W, 6).half()
    w = (_rand(rng, 16, 3, 3, 3) / np.sqrt(27)).half().float()
    b = _rand(rng, 16)
    frames = torch.cat([pairs[..., :3], pairs[..., 3:]], dim=0).float().permute(0, 3, 1, 2)
    ref = torch_ref.mish(torch.nn.functional.conv2d(torch.nn.functional.pad(frames, (0, 1, 0, 1)), w, b,
                                                    stride=2)).permute(0, 2, 3, 1)
    x = pairs if fmt == "channels_last" else pairs.permute(0, 3, 1, 2).contiguous()
    out = ops.first_conv_mish(x.to(DEV), ops.first_conv_taps(w.to(DEV)), b.to(DEV), fmt).cpu()
    assert out.dtype == torch.float16 and tuple(out.shape) == (6, H // 2, W // 2, 16)
    err = (out.float() - ref).abs()
    assert float((err - (2.0 ** -11) * ref.abs()).max()) <= 2e-5


@pytest.mark.parametrize("ci", [16, 32, 64, 128])
@pytest.mark.parametrize("hw", [(16, 32), (34, 50), (8, 16), (2, 2)])
def test_conv3x3s2_mish_fp16_storage(hw, ci):
    """conv_a of encoder levels 2..5 for fp16 storage (qpwc_conv3x3s2_mish_f16_fwd) on a zero-bordered fp16 input vs
    fp32 torch on the same rounded operands."""
    rng = np.random.default_rng(hw[1] + ci + 5)
    H, W = hw
    x = _rand(rng, 2, H, W, ci).half()
    w = (_rand(rng, 2 * ci, ci, 3, 3) / np.sqrt(9 * ci)).half()
    b = _rand(rng, 2 * ci)
    xp = torch.zeros(2, H + 1, W + 1, ci, dtype=torch.float16)
    xp[:, :H, :W] = x
    ref = torch_ref.mish(torch.nn.functional.conv2d(xp.float().permute(0, 3, 1, 2), w.float(), b, stride=2)).permute(0, 2, 3, 1)
    out = ops.conv3x3s2_mish(xp.to(DEV), ops.conv3x3_taps(w.to(DEV), torch.float16), b.to(DEV)).cpu()
    assert out.dtype == torch.float16 and tuple(out.shape) == (2, H // 2, W // 2, 2 * ci)
    err = (out.float() - ref).abs()
    assert float((err - (2.0 ** -11) * ref.abs()).max()) <= 2e-5


@pytest.mark.parametrize("ci", [16, 32, 64, 128])
@pytest.mark.parametrize("hw", [(16, 32), (34, 50), (64, 128), (8, 16)])
def test_conv3x3s2_mish_stride2_levels(hw, ci):
    """conv_a of encoder levels 2..5 (C_in -> 2 C_in, stride 2, TF 'SAME') on the zero-bordered output of
    conv3x3_mish vs torch."""
    rng = np.random.default_rng(hw[1] + ci)
    H, W = hw
    x = _rand(rng, 3, H, W, ci)
    w = _rand(rng, 2 * ci, ci, 3, 3) / np.sqrt(9 * ci)
    b = _rand(rng, 2 * ci)
    xp = torch.nn.functional.pad(x.permute(0, 3, 1, 2), (0, 1, 0, 1))
    ref = torch_ref.mish(torch.nn.functional.conv2d(xp, w, b, stride=2)).permute(0, 2, 3, 1)
    out = ops.conv3x3s2_mish(xp.permute(0, 2, 3, 1).contiguous().to(DEV), ops.conv3x3_taps(w.to(DEV)), b.to(DEV)).cpu()
    assert tuple(out.shape) == (3, H // 2, W // 2, 2 * ci)
    torch.testing.assert_close(out, ref, rtol=0, atol=2e-5 if ci <= 32 else 5e-5)


@pytest.mark.parametrize("hw,batch", [((8, 16), 8), ((16, 32), 8), ((32, 64), 2), ((19, 37), 3), ((5, 7), 1), ((64, 128), 1)])
@pytest.mark.parametrize("act_in", [False, True], ids=["activated-input", "mish-on-load"])
@pytest.mark.parametrize("out_format", ["channels_last", "channels_first"])
def test_optflow_tail_one_launch(hw, batch, act_in, out_format):
    """qpwc_optflow_tail_fwd (SeparableConv2D 64 -> 32 -> 16 + flow head in one launch, 8 x 8 tiles with
    recomputed halos, every intermediate zero outside the image) against the three launches it replaces and
    against the torch-CPU restatement of non_layers.py:223-231, 238-254, 268-273; ragged image sizes included."""
    H, W = hw
    rng = np.random.default_rng(H * 100 + W + batch)
    weights = synth.make_weights(42, (256, 512))
    params = {k: torch.as_tensor(v).to(DEV) for k, v in weights.items()}
    of = non_layers.OptFlow(params, "upflow.2.flow.", data_format="channels_last")
    of._prepare_hip()
    z2 = _rand(rng, batch, H, W, 64)
    scale = float(H * H + W * W) ** 0.5
    out = ops.optflow_tail(z2.to(DEV), of._dw[2], of._pw3, of._pw_b32[2], of._dw[3], of._pw4, of._pw_b32[3], of._head,
                           scale, mish_on_load=act_in, out_format=out_format)
    if out_format == "channels_first":
        assert tuple(out.shape) == (batch, 2, H, W)
        out = out.permute(0, 2, 3, 1)
    # the launches it replaces
    z3 = ops.sepconv3x3([z2.to(DEV)], of._dw[2], of._pw_pad[2], of._pw_b32[2], mish_on_load=act_in, mish_on_store=True)
    z4 = ops.sepconv3x3([z3], of._dw[3], of._pw_pad[3], of._pw_b32[3])
    three = ops.flow_head(z4, of._head, scale)
    torch.testing.assert_close(out / scale, three / scale, rtol=0, atol=2e-6)
    # the oracle's ops
    def sep(x, i, act):
        y = torch_ref.depthwise3x3([x], torch.as_tensor(weights["upflow.2.flow.feat.%d.depthwise.weight" % i]), act)
        pw = torch.as_tensor(weights["upflow.2.flow.feat.%d.pointwise.weight" % i])
        return torch.nn.functional.conv2d(y.permute(0, 3, 1, 2), pw, torch.as_tensor(weights["upflow.2.flow.feat.%d.bias" % i])).permute(0, 2, 3, 1)
    w = lambda k: torch.as_tensor(weights["upflow.2.flow." + k])
    r3 = torch_ref.mish(sep(z2, 2, act_in))
    r4 = sep(r3, 3, False)
    ref = torch_ref.flow_head(r4, w("conv.weight"), w("conv.bias"), w("norm.gamma"), w("norm.beta"), w("norm.mean"),
                              w("norm.var"), 1e-3, w("flow.weight"), scale)
    torch.testing.assert_close(out.cpu() / scale, ref / scale, rtol=0, atol=2e-5)

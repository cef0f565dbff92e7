"""GPU suite (-m gpu): 'channels_first' -- the reference's default for inference
(qpwcnet/app/optical_flow/test_infer.py:52; layers.py:83-88,179-183) -- on the channels-last kernels:
the boundary transposition, the planar variants of the first encoder layer / flow head / Upsample / EPE,
the hot-path layers on dense (B,C,H,W) operands against the C oracle, and the whole network in both
layouts (same kernels, same arithmetic: bit-identical flows)."""
import numpy as np
import pytest
import torch

from oracle import c_ref, net_ref, torch_ref
from qpwcnet_amd import layers, metrics, non_layers, ops, synth
from qpwcnet_amd.pwcnet import GraphedForward, build_flower

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-4


def gpu(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(DEV)


@pytest.mark.parametrize("shape", [(2, 6, 256, 512), (3, 32, 19, 37), (1, 2, 8, 16), (2, 81, 33, 50), (1, 256, 8, 16),
                                   (2, 3, 5, 7)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_layout_transpose_both_ways(shape, dtype):
    g = torch.Generator(device=DEV).manual_seed(shape[1])
    x = torch.randn(*shape, device=DEV, generator=g).to(dtype)           # (B,C,H,W)
    nhwc = ops.layout_transpose(x, "channels_last")
    assert nhwc.is_contiguous() and torch.equal(nhwc, x.permute(0, 2, 3, 1))
    back = ops.layout_transpose(nhwc, "channels_first")
    assert back.is_contiguous() and torch.equal(back, x)
    with pytest.raises(ValueError, match="Unsupported data format"):
        ops.layout_transpose(x, "nhwc")


@pytest.mark.parametrize("shape", [(2, 32, 36, 52), (1, 64, 19, 37), (2, 8, 12, 20), (8, 32, 128, 256), (2, 256, 8, 16)])
def test_hot_path_layers_on_dense_channels_first_operands(shape):
    """CostVolumeV2 / WarpV2 / Warp given dense (B,C,H,W) tensors with C % 4 == 0: transposed onto the
    matrix-core / 16-byte-gather kernels instead of the one-thread-per-element generic ones; same bound
    against the C oracle (which reads NCHW natively)."""
    rng = np.random.default_rng(shape[1] + shape[2])
    prv = rng.standard_normal(shape).astype(np.float32)
    nxt = rng.standard_normal(shape).astype(np.float32)
    flo = (rng.standard_normal((shape[0], 2) + shape[2:]) * 3).astype(np.float32)
    cv = non_layers.CostVolumeV2(data_format="channels_first")((gpu(prv), gpu(nxt)))
    assert tuple(cv.shape) == (shape[0], 81) + shape[2:] and cv.is_contiguous()
    np.testing.assert_allclose(cv.cpu().numpy(), c_ref.cost_volume_v2(prv, nxt, 4, "channels_first"), rtol=0, atol=TOL)
    w2 = layers.WarpV2(data_format="channels_first")((gpu(nxt), gpu(flo)))
    np.testing.assert_array_equal(w2.cpu().numpy(), c_ref.warp(nxt, flo, "channels_first", "clamp"))
    w1 = layers.Warp(data_format="channels_first")((gpu(nxt), gpu(flo)))
    np.testing.assert_array_equal(w1.cpu().numpy(), c_ref.warp(nxt, flo, "channels_first", "tfwarp"))
    # broadcast flow (1,2,1,1), as app/optical_flow/test_warp.py:32 passes it
    fb = np.asarray([1.5, -2.25], np.float32).reshape(1, 2, 1, 1)
    wb = layers.WarpV2(data_format="channels_first")((gpu(nxt), gpu(fb)))
    np.testing.assert_array_equal(wb.cpu().numpy(), c_ref.warp(nxt, np.broadcast_to(fb, flo.shape).copy(),
                                                               "channels_first", "clamp"))


def test_planar_variants_of_the_model_boundary_kernels():
    """first_conv_mish on (B,6,H,W), flow_head writing (B,2,H,W), Upsample reading / writing planes and the
    planar multi-level EPE: bit-identical to their channels-last forms."""
    g = torch.Generator(device=DEV).manual_seed(3)
    weights = synth.make_weights(42, (64, 128))
    params = {k: torch.as_tensor(v).to(DEV) for k, v in weights.items()}
    pairs = torch.rand(3, 64, 128, 6, device=DEV, generator=g) - 0.5
    taps = ops.first_conv_taps(params["enc.0.conv_a.weight"])
    bias = params["enc.0.conv_a.bias"].float()
    a = ops.first_conv_mish(pairs, taps, bias)
    b = ops.first_conv_mish(pairs.permute(0, 3, 1, 2).contiguous(), taps, bias, "channels_first")
    assert torch.equal(a, b)
    z = torch.randn(3, 33, 50, 16, device=DEV, generator=g)
    of = non_layers.OptFlow(params, "upflow.1.flow.", data_format="channels_last")
    of._prepare_hip()
    f_l = ops.flow_head(z, of._head, 7.5)
    f_f = ops.flow_head(z, of._head, 7.5, "channels_first")
    assert tuple(f_f.shape) == (3, 2, 33, 50) and torch.equal(f_f.permute(0, 2, 3, 1), f_l)
    u_ll = ops.upsample2x_flow(f_l, 2.0)
    assert torch.equal(ops.upsample2x_flow(f_f, 2.0, "channels_first", "channels_last"), u_ll)
    assert torch.equal(ops.upsample2x_flow(f_f, 2.0, "channels_first", "channels_first").permute(0, 2, 3, 1), u_ll)
    assert torch.equal(ops.upsample2x_flow(f_l, 2.0, "channels_last", "channels_first").permute(0, 2, 3, 1), u_ll)
    ta = [torch.randn(3, h, w, 2, device=DEV, generator=g) for h, w in ((8, 16), (33, 47), (128, 256))]
    tb = [torch.randn(3, h, w, 2, device=DEV, generator=g) for h, w in ((8, 16), (33, 47), (128, 256))]
    ref = np.asarray([c_ref.epe(x.cpu().numpy(), y.cpu().numpy()) for x, y in zip(ta, tb)])
    cf = lambda ts: [t.permute(0, 3, 1, 2).contiguous() for t in ts]
    out = metrics.per_level_epe(cf(ta), cf(tb), data_format="channels_first")
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=1e-5)


@pytest.mark.parametrize("hw,batch", [((64, 128), 3), ((256, 512), 2)])
def test_full_network_channels_first_equals_channels_last(hw, batch):
    """build_flower(..., 'channels_first') runs the same kernels as 'channels_last' between a planar first
    layer and planar flow outputs: flows equal the channels-last model's bit for bit (train and inference
    graphs, eager and hipGraph replay), and the CPU oracle within 1e-4."""
    weights = synth.make_weights(42, hw)
    pairs, _ = synth.make_frames(batch, hw[0], hw[1], seed=77)
    x_l = gpu(pairs)
    x_f = x_l.permute(0, 3, 1, 2).contiguous()
    m_l = build_flower(True, hw, "channels_last", weights=weights, device=DEV)
    m_f = build_flower(True, hw, "channels_first", weights=weights, device=DEV)
    with torch.no_grad():
        f_l, f_f = m_l(x_l), m_f(x_f)
    assert len(f_f) == 6
    for a, b in zip(f_l, f_f):
        assert b.is_contiguous() and tuple(b.shape) == (batch, 2) + tuple(a.shape[1:3])
        assert torch.equal(b.permute(0, 2, 3, 1), a)
    ref = net_ref.RefNet(weights)(pairs[:1])
    for lvl, (b, r) in enumerate(zip(f_f, ref)):
        e = float(torch_ref.epe_error(b[:1].permute(0, 2, 3, 1).cpu(), r))
        assert e < TOL, "level {} EPE vs oracle {:.3e}".format(lvl, e)
    last = build_flower(False, hw, "channels_first", weights=weights, device=DEV).predict(x_f)
    assert tuple(last.shape) == (batch, 2) + hw and torch.equal(last, f_f[-1])
    g = GraphedForward(m_f, x_f)
    outs, _ = g.replay(x_f)
    for a, b in zip(outs, f_f):
        assert torch.equal(a, b)
    with pytest.raises(ValueError):
        m_f(x_l)            # wrong layout for the declared format


def test_channels_first_model_without_first_layer_kernel_and_with_library_optflow():
    """a channels_first model on the channels-last blocks (a) when enc.0's first-layer kernel does not
    apply -- the pairs are transposed and split on the axis of the INTERNAL layout -- and (b) when an OptFlow block
    falls back to the library composition, whose (B,h,w,2) result must still come out as (B,2,h,w)."""
    hw, batch = (64, 128), 2
    weights = synth.make_weights(42, hw)
    pairs, _ = synth.make_frames(batch, hw[0], hw[1], seed=79)
    x_f = gpu(pairs).permute(0, 3, 1, 2).contiguous()
    ref = build_flower(True, hw, "channels_first", weights=weights, device=DEV)
    with torch.no_grad():
        want = ref(x_f)
    m = build_flower(True, hw, "channels_first", weights=weights, device=DEV)
    m.enc[0].hip_conv = False                      # (a): first_layer() -> None
    m.flow.flow.can_use_hip = lambda sources: False   # (b): coarsest block on the library path
    with torch.no_grad():
        got = m(x_f)
    for lvl, (a, b) in enumerate(zip(want, got)):
        assert tuple(a.shape) == tuple(b.shape)
        e = float(torch_ref.epe_error(a.permute(0, 2, 3, 1).cpu(), b.permute(0, 2, 3, 1).cpu()))
        assert e < TOL, "level {} EPE between the two paths {:.3e}".format(lvl, e)


def test_fp16_network_channels_first_equals_channels_last():
    """The fp16-storage network (config 5's kernels: fp16 first layer on planar pairs, fp16 encoder / decoder / OptFlow
    kernels) gives the same flows for both declared layouts, bit for bit."""
    hw, batch = (64, 128), 3
    weights = synth.make_weights(42, hw)
    pairs, _ = synth.make_frames(batch, hw[0], hw[1], seed=78)
    x_l = gpu(pairs).half()
    x_f = x_l.permute(0, 3, 1, 2).contiguous()
    m_l = build_flower(True, hw, "channels_last", weights=weights, device=DEV, dtype=torch.float16)
    m_f = build_flower(True, hw, "channels_first", weights=weights, device=DEV, dtype=torch.float16)
    with torch.no_grad():
        f_l, f_f = m_l(x_l), m_f(x_f)
    for a, b in zip(f_l, f_f):
        assert b.dtype == a.dtype and torch.equal(b.permute(0, 2, 3, 1), a)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("shape", [(2, 5, 7, 16), (3, 8, 16, 32), (1, 33, 19, 8), (16, 16, 32, 64)])
def test_copy_pixels_between_strided_views(shape, dtype):
    """qpwc_copy_pixels_fwd: the skip half of the decoder's concat -- source = the un-padded view of a zero-bordered
    tensor, destination = the upper channels of a wider buffer; bit-exact, nothing outside the view touched."""
    B, H, W, C = shape
    if (C * torch.empty((), dtype=dtype).element_size()) % 16:
        pytest.skip("needs whole 16-byte chunks")
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(5)
    padded = torch.randn(B, H + 1, W + 1, C, device=dev, generator=g).to(dtype)
    src = padded[:, :H, :W, :]
    extra = 16
    buf = torch.full((B, H, W, extra + C), 7.0, device=dev, dtype=dtype)
    dst = buf[..., extra:]
    assert ops.copy_pixels_ok(src, dst)
    ops.copy_pixels(src, dst)
    torch.cuda.synchronize()
    assert torch.equal(dst, src)
    assert torch.equal(buf[..., :extra], torch.full((B, H, W, extra), 7.0, device=dev, dtype=dtype))
    # what the C side refuses: overlapping views, misaligned channel counts
    with pytest.raises(ValueError):
        ops.copy_pixels(padded[:, :H, :W, :], padded[:, 1:, 1:, :])
    assert not ops.copy_pixels_ok(src[..., 1:], dst[..., 1:])


def test_decoder_concat_uses_the_copy_kernel_and_matches_tensor_copy():
    hw, B = (64, 128), 2
    weights = synth.make_weights(3, hw)
    pairs_np, _ = synth.make_frames(B, hw[0], hw[1], seed=9)
    pairs = torch.from_numpy(pairs_np).to("cuda:0")
    model = build_flower(True, hw, "channels_last", weights=weights, device="cuda:0")
    assert all(d.skip_copy_hip for d in model.dec)      # round 3: the own copy kernel is the default
    with torch.no_grad():
        a = model(pairs)
        for d in model.dec:
            d.skip_copy_hip = False                     # the library's tensor.copy_
        b = model(pairs)
    torch.cuda.synchronize()
    assert all(torch.equal(x, y) for x, y in zip(a, b))


"""Run-to-run bit stability of the fp32 matrix-core kernels (round 4).

hipcc does not guard a VALU write into the C registers of a just-issued v_mfma_f32_16x16x4_f32 whose result it renamed
(tools/mfma_war_lint.py); under matrix-pipe contention gfx950 then reads the overwritten value.  A kernel that carries
such a pair in a hot spot returns different bits from launch to launch (the round-4 flat-pipeline SeparableConv2D did:
300 of 300 launches).  Every product kernel that issues fp32 matrix instructions is launched repeatedly on the same
inputs at the shapes of the bench step (two / three waves per SIMD contending for the matrix pipe) and must return the
first launch's bits every time."""
import pytest
import torch

from qpwcnet_amd import ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
N = 40


def _same_bits_every_launch(fn):
    ref = fn().clone()
    for i in range(N):
        out = fn()
        assert torch.equal(out, ref), "launch {} differs from the first (max |diff| {:.3g})".format(
            i + 1, float((out.float() - ref.float()).abs().max()))


def _rnd(g, *shape):
    return torch.randn(*shape, device=DEV, generator=g)


@pytest.mark.parametrize("shape,fused", [((8, 128, 256, 32), False), ((8, 128, 256, 32), True), ((8, 64, 128, 64), False),
                                         ((8, 64, 128, 64), True), ((8, 32, 64, 128), True), ((8, 16, 32, 256), False)],
                         ids=["L4", "L4-fused", "L3", "L3-fused", "L2-fused", "L1-splitK"])
def test_cost_volume_kernels_are_run_to_run_stable(shape, fused):
    g = torch.Generator(device=DEV).manual_seed(1)
    prv, nxt = _rnd(g, *shape), _rnd(g, *shape)
    flo = _rnd(g, *shape[:3], 2) * 3
    _same_bits_every_launch((lambda: ops.warp_cost_volume(prv, nxt, flo)) if fused else (lambda: ops.cost_volume(prv, nxt)))


@pytest.mark.parametrize("chans,F", [((84, 32, 2), 128), ((128,), 64), ((64,), 32), ((32,), 16)])
@pytest.mark.parametrize("hw", [(128, 256), (64, 128)], ids=["L4", "L3"])
def test_fused_separable_conv_is_run_to_run_stable(chans, F, hw):
    g = torch.Generator(device=DEV).manual_seed(2)
    C = sum(chans)
    srcs = [_rnd(g, 8, hw[0], hw[1], c) for c in chans]
    dw, pw, bias = _rnd(g, C, 9), _rnd(g, F, C) / C ** 0.5, _rnd(g, F)
    pwp = ops.pad_pointwise(pw)
    _same_bits_every_launch(lambda: ops.sepconv3x3(srcs, dw, pwp, bias, mish_on_store=True))


@pytest.mark.parametrize("C", [16, 32, 64, 128, 256])
def test_encoder_convolutions_are_run_to_run_stable(C):
    g = torch.Generator(device=DEV).manual_seed(3)
    hw = {16: (128, 256), 32: (64, 128), 64: (32, 64), 128: (16, 32), 256: (8, 16)}[C]
    x = _rnd(g, 16, hw[0], hw[1], C)
    taps = ops.conv3x3_taps(_rnd(g, C, C, 3, 3) / (3 * C ** 0.5))
    bias = _rnd(g, C)
    _same_bits_every_launch(lambda: ops.conv3x3_mish(x, taps, bias))


@pytest.mark.parametrize("shape", [(256, 8, 16, 128), (256, 16, 32, 64), (128, 32, 64, 32), (64, 64, 128, 16)],
                         ids=["dec0", "dec1", "dec2", "dec3"])
def test_decoder_upconv_with_skip_is_run_to_run_stable(shape):
    """The decoder's transposed convolution + skip copy (qpwc_upconv4x4s2_mish_cat_fwd) at the step's four shapes."""
    C, H, W, F = shape
    g = torch.Generator(device=DEV).manual_seed(4)
    x = _rnd(g, 16, H, W, C)
    taps = ops.upconv_taps(_rnd(g, C, F, 4, 4) / (2 * C ** 0.5))
    bias = _rnd(g, F)
    skip = _rnd(g, 16, 2 * H, 2 * W, F)
    dst = torch.empty(16, 2 * H, 2 * W, 2 * F, device=DEV)
    _same_bits_every_launch(lambda: ops.upconv4x4s2_mish_cat_into(x, taps, bias, skip, dst))


@pytest.mark.parametrize("hw", [(128, 256), (64, 128)], ids=["L4", "L3"])
def test_flow_head_with_upsampling_is_run_to_run_stable(hw):
    """qpwc_flow_head_up_fwd (its 1x1 stage runs on the fp32 matrix instructions) at the two levels that use it."""
    from qpwcnet_amd import non_layers
    g = torch.Generator(device=DEV).manual_seed(5)
    z = _rnd(g, 8, hw[0], hw[1], 16)
    params = non_layers.pack_flow_head(_rnd(g, 16, 16, 1, 1) * 0.3, _rnd(g, 16) * 0.1, 1 + 0.1 * _rnd(g, 16), 0.1 * _rnd(g, 16),
                                       0.1 * _rnd(g, 16), 1 + 0.2 * torch.rand(16, device=DEV, generator=g), 1e-3,
                                       _rnd(g, 2, 16, 3, 3) * 0.2)
    _same_bits_every_launch(lambda: torch.cat([t.reshape(-1) for t in ops.flow_head_up(z, params, 100.0, 2.0)]))

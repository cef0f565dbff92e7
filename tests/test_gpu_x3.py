"""GPU suite: the opt-in "bf16x3" arithmetic of the convolution kernels (csrc/split_bf16.h) -- fp32 products computed
as six bf16 x bf16 partial products of three-way splits, fp32 accumulation.  Every test compares the bf16x3 kernel AND
the fp32-matrix-instruction kernel with a float64 evaluation of the same layer
(qpwcnet/core/non_layers.py:223-231, 410-449) on the same inputs: the split form must be within the same tolerance
(1e-4, `north_star`) and no worse than twice the fp32 instructions' own error (+ one fp32 rounding of the output)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import net_ref, torch_ref
from qpwcnet_amd import ops, synth
from qpwcnet_amd.pwcnet import build_flower

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-4


def _rand(rng, *shape):
    return torch.from_numpy(rng.standard_normal(shape).astype(np.float32))


def _no_worse(e_x3, e_f32, ref):
    """the split products' error budget: twice what the fp32 instructions commit + one rounding of the largest output"""
    return e_x3 <= 2.0 * e_f32 + 2.0 ** -23 * float(ref.abs().max())


def test_split_is_exact():
    rng = np.random.default_rng(0)
    x = _rand(rng, 1 << 16)
    x = torch.cat([x, x * 1e30, x * 1e-30, torch.tensor([0.0, -0.0, 1.0, -1.0, 3.3895314e38, 1e-35, 65504.0])])
    parts = ops.split_bf16x3(x.to(DEV))
    assert parts.dtype == torch.bfloat16 and tuple(parts.shape) == (3, x.numel())
    back = parts.double().sum(0).cpu()
    err = (back - x.double()).abs()
    # exact wherever all three parts are normal numbers; parts below the smallest normal fp32 flush to zero (absolute
    # error < 2^-126 there, split_bf16.h)
    big = x.double().abs() >= 1e-25
    assert bool((err[big] == 0).all()), float((err[big] / x.double().abs()[big]).max())
    assert bool((err <= 2.0 ** -126).all())
    # the first part is plain round-to-nearest-even bf16
    assert torch.equal(parts[0].cpu(), x.to(torch.bfloat16))


@pytest.mark.parametrize("C,hw", [(16, (32, 48)), (16, (21, 19)), (32, (16, 32)), (32, (37, 50)), (64, (16, 32)),
                                  (64, (19, 23)), (128, (8, 16)), (128, (13, 33)), (256, (8, 16)), (256, (5, 7))])
@pytest.mark.parametrize("pad", [0, 1])
def test_conv3x3_mish_x3_against_float64(C, hw, pad):
    rng = np.random.default_rng(C + hw[0] + pad)
    H, W = hw
    x = (_rand(rng, 3, H, W, C) * 3).to(DEV)
    w = (_rand(rng, C, C, 3, 3) / (9 * C) ** 0.5).to(DEV)
    b = _rand(rng, C).to(DEV)
    ref = F.mish(F.conv2d(x.double().permute(0, 3, 1, 2), w.double(), b.double(), padding=1)).permute(0, 2, 3, 1)
    taps = ops.conv3x3_taps(w)
    y32 = ops.conv3x3_mish(x, taps, b, pad, pad)
    y3 = ops.conv3x3_mish_x3(x, ops.split_bf16x3(taps), b, pad, pad)
    assert tuple(y3.shape) == (3, H + pad, W + pad, C)
    e32 = float((y32[:, :H, :W].double() - ref).abs().max())
    e3 = float((y3[:, :H, :W].double() - ref).abs().max())
    assert e3 < TOL and _no_worse(e3, e32, ref), (e3, e32)
    if pad:   # the 'SAME' border the next stride-2 layer reads
        assert float(y3[:, H:].abs().max()) == 0.0 and float(y3[:, :, W:].abs().max()) == 0.0


@pytest.mark.parametrize("CI,hw", [(32, (16, 32)), (32, (22, 38)), (64, (8, 16)), (64, (14, 50)), (128, (4, 32)), (128, (10, 6))])
def test_conv3x3s2_mish_x3_against_float64(CI, hw):
    """conv_a of encoder levels 3..5 (stride 2, TF 'SAME' = the zero border of the padded input) in both arithmetics."""
    rng = np.random.default_rng(CI + hw[0])
    H, W = hw
    xp = torch.zeros(3, H + 1, W + 1, CI)
    xp[:, :H, :W] = _rand(rng, 3, H, W, CI) * 3
    xp = xp.to(DEV)
    w = (_rand(rng, 2 * CI, CI, 3, 3) / (9 * CI) ** 0.5).to(DEV)
    b = _rand(rng, 2 * CI).to(DEV)
    ref = F.mish(F.conv2d(xp.double().permute(0, 3, 1, 2), w.double(), b.double(), stride=2)).permute(0, 2, 3, 1)
    taps = ops.conv3x3_taps(w)
    y32 = ops.conv3x3s2_mish(xp, taps, b)
    y3 = ops.conv3x3s2_mish_x3(xp, ops.split_bf16x3(taps), b)
    assert tuple(y3.shape) == (3, H // 2, W // 2, 2 * CI) == tuple(ref.shape)
    e32, e3 = float((y32.double() - ref).abs().max()), float((y3.double() - ref).abs().max())
    assert e3 < TOL and _no_worse(e3, e32, ref), (e3, e32)


@pytest.mark.parametrize("C,Fo,hw", [(64, 16, (16, 32)), (64, 32, (11, 19)), (128, 32, (8, 16)), (128, 64, (5, 21)),
                                      (256, 64, (4, 16)), (256, 128, (3, 5))])
def test_upconv4x4s2_mish_x3_against_float64(C, Fo, hw):
    """The decoder's UpConv (Conv2DTranspose 4x4, stride 2, 'same', bias, Mish) written into the concat buffer, in both
    arithmetics; the skip half of the buffer stays untouched."""
    rng = np.random.default_rng(C + Fo + hw[0])
    H, W = hw
    x = (_rand(rng, 2, H, W, C) * 2).to(DEV)
    w = (_rand(rng, C, Fo, 4, 4) / (4 * C) ** 0.5).to(DEV)
    b = _rand(rng, Fo).to(DEV)
    ref = F.mish(F.conv_transpose2d(x.double().permute(0, 3, 1, 2), w.double(), b.double(), stride=2, padding=1)
                 ).permute(0, 2, 3, 1)
    taps = ops.upconv_taps(w)
    d32 = torch.full((2, 2 * H, 2 * W, Fo + 8), 7.0, device=DEV)
    d3 = torch.full((2, 2 * H, 2 * W, Fo + 8), 7.0, device=DEV)
    ops.upconv4x4s2_mish_into(x, taps, b, d32)
    ops.upconv4x4s2_mish_into(x, ops.split_bf16x3(taps), b, d3)
    assert bool((d3[..., Fo:] == 7.0).all())
    e32 = float((d32[..., :Fo].double() - ref).abs().max())
    e3 = float((d3[..., :Fo].double() - ref).abs().max())
    assert e3 < TOL and _no_worse(e3, e32, ref), (e3, e32)


def test_conv3x3_mish_x3_rejects_bad_operands():
    x = torch.zeros(1, 8, 16, 32, device=DEV)
    taps3 = torch.zeros(3, 9, 32, 32, device=DEV, dtype=torch.bfloat16)
    b = torch.zeros(32, device=DEV)
    with pytest.raises(ValueError):
        ops.conv3x3_mish_x3(x, taps3.float(), b)            # not split
    with pytest.raises(ValueError):
        ops.conv3x3_mish_x3(x.half(), taps3, b)             # fp16 storage has its own kernel
    with pytest.raises(ValueError):
        ops.conv3x3_mish_x3(torch.zeros(1, 8, 16, 48, device=DEV), torch.zeros(3, 9, 48, 48, device=DEV, dtype=torch.bfloat16),
                            torch.zeros(48, device=DEV))    # C outside {16, 32, 64, 128, 256} -> the C-ABI's error


@pytest.mark.parametrize("chans,Fo", [((84, 32, 2), 128), ((84, 64, 2), 128), ((128,), 64), ((64,), 32), ((32,), 16),
                                      ((84, 256, 1), 64), ((96,), 128)])
@pytest.mark.parametrize("hw", [(16, 32), (19, 37)])
@pytest.mark.parametrize("act", [(False, False), (True, False), (False, True)])
def test_sepconv3x3_x3_against_float64(chans, Fo, hw, act):
    rng = np.random.default_rng(sum(chans) + Fo + hw[0] + 2 * act[0] + act[1])
    H, W = hw
    C = sum(chans)
    srcs = [_rand(rng, 2, H, W, c).to(DEV) for c in chans]
    dw = (_rand(rng, C, 9) / 3).to(DEV)
    pw = (_rand(rng, Fo, C) / C ** 0.5).to(DEV)
    bias = _rand(rng, Fo).to(DEV)
    x = torch.cat(srcs, dim=3).double().permute(0, 3, 1, 2)
    if act[0]:
        x = F.mish(x)
    y = F.conv2d(x, dw.double().view(C, 1, 3, 3), None, padding=1, groups=C)
    ref = F.conv2d(y, pw.double().view(Fo, C, 1, 1), bias.double()).permute(0, 2, 3, 1)
    if act[1]:
        ref = F.mish(ref)
    pwp = ops.pad_pointwise(pw)
    assert ops.sepconv3x3_x3_applies(srcs)
    z32 = ops.sepconv3x3(srcs, dw, pwp, bias, mish_on_load=act[0], mish_on_store=act[1])
    z3 = ops.sepconv3x3(srcs, dw, ops.split_bf16x3(pwp), bias, mish_on_load=act[0], mish_on_store=act[1])
    assert tuple(z3.shape) == (2, H, W, Fo)
    e32, e3 = float((z32.double() - ref).abs().max()), float((z3.double() - ref).abs().max())
    assert e3 < TOL and _no_worse(e3, e32, ref), (e3, e32)


def test_sepconv3x3_x3_source_rule():
    """Sources off the 16-byte staging path are refused by the C-ABI (the model keeps the fp32-instruction kernel)."""
    a = torch.zeros(1, 8, 16, 7, device=DEV)
    assert not ops.sepconv3x3_x3_applies([a, a])
    pw3 = torch.zeros(3, 16, 32, device=DEV, dtype=torch.bfloat16)
    with pytest.raises((RuntimeError, ValueError), match="16-byte aligned"):
        ops.sepconv3x3([a, a], torch.zeros(14, 9, device=DEV), pw3, torch.zeros(16, device=DEV))


@pytest.mark.parametrize("hw,batch", [((64, 128), 2), ((256, 512), 1)])
def test_full_network_bf16x3_per_level_epe(hw, batch):
    """QpwcNet(matmul='bf16x3') against the CPU oracle at the bound of the fp32 network (1e-4), and against the fp32
    instruction path of the same build."""
    weights = synth.make_weights(42, hw)
    pairs, _ = synth.make_frames(batch, hw[0], hw[1], seed=1234)
    model = build_flower(True, hw, "channels_last", weights=weights, device=DEV)
    f32 = model.predict(pairs)
    model.matmul = "bf16x3"
    assert model.enc[0].matmul == "bf16x3" and model.upflows[-1].flow.matmul == "bf16x3"
    x3 = model.predict(pairs)
    ref = net_ref.RefNet(weights)(pairs)
    for lvl, (a, b, r) in enumerate(zip(x3, f32, ref)):
        e = float(torch_ref.epe_error(a.cpu(), r))
        assert e < TOL, "level {} EPE vs oracle {:.3e}".format(lvl, e)
        assert float((a - b).abs().max()) < 2e-5 * max(1.0, float(b.abs().max())), lvl
    with pytest.raises(ValueError):
        model.matmul = "bf16"

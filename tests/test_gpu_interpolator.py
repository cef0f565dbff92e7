"""GPU parity of the frame-interpolation model: ``build_interpolator``
(qpwcnet/core/pwcnet.py:70-131,247-281) and its ``FrameInterpolate`` block
(non_layers.py:276-312) on the HIP kernels (WarpV2, cost volume, depthwise, bias+Mish) vs the
torch-CPU oracle ``oracle/net_ref.RefInterpolator`` on the same seeded weights and frames.
fp32 tolerance as for the flow network: max |difference| of every image below 1e-4."""
import numpy as np
import pytest
import torch

from oracle import net_ref
from qpwcnet_amd import synth
from qpwcnet_amd.non_layers import FrameInterpolate
from qpwcnet_amd.pwcnet import build_interpolator

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-4


@pytest.mark.parametrize("hw,batch", [((64, 128), 2), ((256, 512), 1)])
def test_interpolator_images_match_oracle(hw, batch):
    weights = synth.make_interpolator_weights(42, hw)
    pairs, _ = synth.make_frames(batch, hw[0], hw[1], seed=1234)
    model = build_interpolator(hw, "channels_last", weights=weights, device=DEV)
    imgs = model.predict(pairs)
    ref = net_ref.RefInterpolator(weights)(pairs)
    assert len(imgs) == 6 and tuple(imgs[-1].shape) == (batch, hw[0], hw[1], 3)
    for lvl, (a, b) in enumerate(zip(imgs, ref)):
        assert tuple(a.shape) == tuple(b.shape)
        err = float((a.cpu() - b).abs().max())
        assert err < TOL, "image {}: max abs err {:.3e}".format(lvl, err)
    last = build_interpolator(hw, "channels_last", weights=weights, device=DEV,
                              output_multiscale=False).predict(pairs)
    assert torch.allclose(last, imgs[-1], rtol=0, atol=1e-5)
    # the reference's two separate Flower passes (pwcnet.py:271-278) instead of one batched pass
    two = build_interpolator(hw, "channels_last", weights=weights, device=DEV,
                             batch_directions=False).predict(pairs)
    for a, b in zip(two, ref):
        assert float((a.cpu() - b).abs().max()) < TOL


def test_interpolator_channels_first():
    hw = (64, 128)
    weights = synth.make_interpolator_weights(42, hw)
    pairs, _ = synth.make_frames(1, hw[0], hw[1], seed=1234)
    model = build_interpolator(hw, "channels_first", weights=weights, device=DEV)
    imgs = model.predict(np.ascontiguousarray(np.transpose(pairs, (0, 3, 1, 2))))
    ref = net_ref.RefInterpolator(weights)(pairs)
    for a, b in zip(imgs, ref):
        assert float((a.permute(0, 2, 3, 1).cpu() - b).abs().max()) < TOL


@pytest.mark.parametrize("k,shape", [(0, (2, 8, 16, 3)), (2, (2, 16, 32, 128)), (4, (1, 32, 64, 32))])
def test_frame_interpolate_block(k, shape):
    """One block in isolation with flows that cross the borders (half-flow WarpV2 clamps)."""
    weights = synth.make_interpolator_weights(42, (64, 128))
    rng = np.random.default_rng(k)
    B, H, W, C = shape
    prv = rng.standard_normal(shape).astype(np.float32)
    nxt = rng.standard_normal(shape).astype(np.float32)
    f01 = (rng.standard_normal((B, H, W, 2)) * 6).astype(np.float32)
    f10 = (rng.standard_normal((B, H, W, 2)) * 6).astype(np.float32)
    img_u = rng.standard_normal((B, H, W, 3)).astype(np.float32)
    params = {n: torch.from_numpy(v).to(DEV) for n, v in weights.items()}
    blk = FrameInterpolate(params, "img.{}.".format(k), up=k > 0, data_format="channels_last")
    g = [torch.from_numpy(t).to(DEV) for t in (prv, nxt, f01, f10)]
    t = [torch.from_numpy(t) for t in (prv, nxt, f01, f10)]
    if k > 0:
        g.append(torch.from_numpy(img_u).to(DEV))
        t.append(torch.from_numpy(img_u))
    got = blk(tuple(g))
    ref = net_ref.RefInterpolator(weights).frame_interpolate(k, *t)
    assert tuple(got.shape) == (B, H, W, 3)
    assert float((got.cpu() - ref).abs().max()) < TOL

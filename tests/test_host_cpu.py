"""CPU suite: host-side logic of the layer facades (construction, config round
trip, data-format handling, error behaviour) -- everything up to the device call."""
import numpy as np
import pytest
import torch

import qpwcnet_amd
from qpwcnet_amd import layers, non_layers, ops, synth
from qpwcnet_amd.backend import get_axis, image_data_format, set_image_data_format


@pytest.fixture(autouse=True)
def _restore_format():
    fmt = image_data_format()
    yield
    set_image_data_format(fmt)


def test_global_data_format_is_read_at_construction():
    """layers.py:41,119,146,173 -- tf.keras.backend.image_data_format() at __init__."""
    assert image_data_format() == "channels_last"
    a = layers.CostVolume(4)
    set_image_data_format("channels_first")
    b = layers.CostVolume(4)
    c = non_layers.WarpV2()
    set_image_data_format("channels_last")
    assert (a.data_format, a.axis) == ("channels_last", 3)
    assert (b.data_format, b.axis) == ("channels_first", 1)
    assert c.data_format == "channels_first"
    with pytest.raises(ValueError):
        set_image_data_format("nhwc")


def test_explicit_data_format_kwarg():
    """What test/test_cost_volume.py:10-11 and test/test_warp.py:14-15 try to pass."""
    assert layers.CostVolumeV2(search_range=4, data_format="channels_first").axis == 1
    assert layers.Warp(data_format="channels_first").data_format == "channels_first"
    with pytest.raises(ValueError, match="Unsupported data format"):
        layers.WarpV2(data_format="NHWC")
    with pytest.raises(ValueError, match="Unsupported data format"):
        get_axis("bogus")


def test_unknown_kwargs_rejected_like_keras():
    with pytest.raises(TypeError):
        layers.CostVolume(4, bogus=1)


def test_config_round_trip():
    """layers.py:102-109."""
    l = layers.CostVolume(search_range=3, name="cv")
    cfg = l.get_config()
    assert cfg["search_range"] == 3 and cfg["name"] == "cv"
    l2 = layers.CostVolume.from_config(cfg)
    assert l2.search_range == 3
    assert layers.CostVolumeV2.from_config(layers.CostVolumeV2(2).get_config()).search_range == 2


def test_build_captures_hw():
    """layers.py:57-70."""
    l = layers.CostVolume(data_format="channels_first")
    l.build(((2, 5, 7, 9), (2, 5, 7, 9)))
    assert (l.h, l.w) == (7, 9)
    l = layers.Warp(data_format="channels_last")
    l.build(((2, 7, 9, 5), (2, 7, 9, 2)))
    assert (l.h, l.w) == (7, 9)


def test_cpu_tensor_is_an_error_not_a_fallback():
    x = torch.zeros(1, 8, 8, 4)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        layers.CostVolume()((x, x))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        non_layers.WarpV2()((x, torch.zeros(1, 8, 8, 2)))
    with pytest.raises(RuntimeError):
        ops.epe(torch.zeros(1, 4, 4, 2), torch.zeros(1, 4, 4, 2))


def test_same_padding_rule():
    """TF 'SAME': stride-2 3x3 on an even size pads (0, 1); on odd sizes (1, 1)."""
    assert non_layers._same_pad(256, 3, 2) == (0, 1)
    assert non_layers._same_pad(255, 3, 2) == (1, 1)
    assert non_layers._same_pad(16, 3, 1) == (1, 1)
    assert non_layers._same_pad(16, 1, 1) == (0, 0)
    x = torch.arange(36, dtype=torch.float32).reshape(1, 1, 6, 6)
    w = torch.ones(1, 1, 3, 3)
    y = non_layers.conv2d_same(x, w, None, 2)
    assert y.shape == (1, 1, 3, 3)
    assert float(y[0, 0, 0, 0]) == float(x[0, 0, :3, :3].sum())          # no padding before
    assert float(y[0, 0, 2, 2]) == float(x[0, 0, 4:, 4:].sum())          # padded after


def test_synthetic_weights_shape_and_determinism():
    w1, w2 = synth.make_weights(42), synth.make_weights(42)
    assert sum(v.size for v in w1.values()) == 3094165     # 3.09 M parameters
    assert all(np.array_equal(w1[k], w2[k]) for k in w1)
    assert synth.level_channels() == [256, 256, 128, 64, 32]
    assert w1["flow.flow.feat.0.depthwise.weight"].shape == (593, 1, 3, 3)
    assert w1["upflow.3.flow.feat.0.depthwise.weight"].shape == (115, 1, 3, 3)


def test_synthetic_frames():
    p, f = synth.make_frames(2, 32, 64, seed=1234)
    p2, _ = synth.make_frames(2, 32, 64, seed=1234)
    assert p.shape == (2, 32, 64, 6) and f.shape == (2, 32, 64, 2)
    assert p.dtype == np.float32 and np.array_equal(p, p2)
    assert p.min() >= -0.5 and p.max() <= 0.5
    assert np.abs(f).max() <= 8.0


def test_version():
    assert qpwcnet_amd.__version__


def test_keras_layout_round_trip(tmp_path):
    from qpwcnet_amd import weights as W
    w = synth.make_weights(42)
    k = W.to_keras_layout(w)
    assert k["enc.0.conv_a.weight"].shape == (3, 3, 3, 16)              # (kh, kw, in, out)
    assert k["flow.flow.feat.0.depthwise.weight"].shape == (3, 3, 593, 1)
    assert k["flow.flow.feat.0.pointwise.weight"].shape == (1, 1, 593, 128)
    assert k["dec.0.conv_up.weight"].shape == (4, 4, 128, 256)          # (kh, kw, out, in)
    back = W.from_keras_layout(k)
    assert all(np.array_equal(back[n], w[n]) for n in w)
    # one tap, checked by hand: Keras kernel[ky,kx,i,o] == torch weight[o,i,ky,kx]
    assert k["enc.1.conv_b.weight"][2, 0, 5, 7] == w["enc.1.conv_b.weight"][7, 5, 2, 0]
    assert k["dec.1.conv_up.weight"][1, 3, 9, 4] == w["dec.1.conv_up.weight"][4, 9, 1, 3]
    p = tmp_path / "w.npz"
    W.save_npz(str(p), w)
    again = W.load_npz(str(p))
    assert all(np.array_equal(again[n], w[n]) for n in w)


# ---- occlusion / frame interpolation host logic --------------------------
def test_downsample_is_tf_same_average_pooling():
    from oracle.net_ref import RefInterpolator
    from qpwcnet_amd.non_layers import Downsample
    x = torch.arange(2 * 5 * 7 * 3, dtype=torch.float32).reshape(2, 5, 7, 3)
    y = Downsample(data_format="channels_last")(x)
    assert tuple(y.shape) == (2, 3, 4, 3)
    assert torch.allclose(y, RefInterpolator.downsample(x))
    # last window of an odd extent averages only the cells inside the image
    assert torch.allclose(y[:, 2, 3], x[:, 4, 6])
    yc = Downsample(data_format="channels_first")(x.permute(0, 3, 1, 2))
    assert torch.equal(yc.permute(0, 2, 3, 1), y)


def test_flower_keeps_the_reference_typo():
    from qpwcnet_amd.non_layers import Flower
    fl = Flower({}, 4, data_format="channels_last")
    assert [u.scale for u in fl.upsamples] == [2.0, 2.0, 2.0, 2.0, 1.0]   # non_layers.py:464-468
    assert len(fl.upflows) == 4


def test_interpolator_weights_and_no_cpu_fallback():
    from qpwcnet_amd.non_layers import FrameInterpolate
    from qpwcnet_amd.synth import make_interpolator_weights, make_weights
    w = make_interpolator_weights(42, (64, 128))
    base = make_weights(42, (64, 128))
    assert all(np.array_equal(w[k], base[k]) for k in base)       # build_flower part unchanged
    cins = [w["img.{}.conv1.depthwise.weight".format(k)].shape[0] for k in range(5)]
    assert cins == [10, 519, 263, 135, 71]                        # pwcnet.py:101-121
    assert w["img.0.conv2.weight"].shape == (3, 64, 1, 1)
    params = {k: torch.from_numpy(v) for k, v in w.items()}
    blk = FrameInterpolate(params, "img.0.", up=False, name="img_0", data_format="channels_last")
    assert blk.get_config() == {"up": False}
    z = torch.zeros((1, 4, 6, 3))
    f = torch.zeros((1, 4, 6, 2))
    with pytest.raises(RuntimeError):                             # WarpV2 is HIP-only
        blk((z, z, f, f))


def test_occlusion_host_errors():
    from qpwcnet_amd import occlusion
    assert occlusion.get_spatial_shape(torch.zeros((2, 5, 7, 2)), "channels_last") == {"n": 2, "h": 5, "w": 7}
    assert occlusion.get_spatial_shape(torch.zeros((2, 2, 5, 7)), "channels_first") == {"n": 2, "h": 5, "w": 7}
    with pytest.raises(ValueError):
        occlusion.get_spatial_shape(torch.zeros((5, 7, 2)), "channels_last")
    with pytest.raises(RuntimeError):
        occlusion.estimate_occlusion_map(torch.zeros((1, 5, 7, 2)), "channels_last")


def test_keras_by_name_round_trip():
    """by-name Keras layout <-> the flat torch-layout dict of build_flower."""
    from qpwcnet_amd import weights as W
    w = synth.make_weights(7, (64, 128))
    names = W.keras_variable_names()
    assert set(names) == set(w)                                   # every parameter has a Keras name
    assert len(set(names.values())) == len(names)                 # ... and a distinct one
    assert names["enc.0.conv_a.weight"] == "conv2d/kernel:0"
    assert names["enc.4.conv_b.bias"] == "conv2d_14/bias:0"
    assert names["dec.3.conv_up.weight"] == "conv2d_transpose_3/kernel:0"
    assert names["flow.flow.feat.0.depthwise.weight"] == "separable_conv2d/depthwise_kernel:0"
    assert names["flow.flow.conv.weight"] == "conv2d_15/kernel:0"
    assert names["flow.flow.flow.weight"] == "conv2d_16/kernel:0"
    assert names["upflow.3.flow.norm.var"] == "batch_normalization_4/moving_variance:0"
    assert names["upflow.3.flow.flow.weight"] == "conv2d_24/kernel:0"
    named = W.to_keras_named(w)
    assert named["conv2d/kernel:0"].shape == (3, 3, 3, 16)        # (kh, kw, in, out)
    assert named["conv2d_transpose/kernel:0"].shape == (4, 4, 128, 256)   # (kh, kw, out, in)
    assert named["separable_conv2d/depthwise_kernel:0"].shape == (3, 3, 593, 1)
    back = W.from_keras_named({"model_weights/" + k.split("/")[0] + "/" + k: v for k, v in named.items()})
    assert set(back) == set(w) and all(np.array_equal(back[k], w[k]) for k in w)
    del named["conv2d_3/bias:0"]
    with pytest.raises(KeyError, match="conv2d_3/bias:0"):
        W.from_keras_named(named)


# ---- round 2: dispatch rules and bench labels (pure host logic) ---------------------------------------
def test_fused_front_end_rule_mirrors_the_c_side():
    """UpFlow fuses WarpV2 + cost volume exactly where cost_volume_mfma_launch takes the fused launch:
    channels-last fp32 / fp16, C % 32 == 0, >= 256 regions of 8 x 8 pixels, H, W >= 2, search range 4."""
    class T:   # shape / dtype / device carrier: the rule never touches data
        def __init__(self, shape, dtype=torch.float32, cuda=True):
            self.shape, self.dtype, self.is_cuda = shape, dtype, cuda
        def dim(self):
            return len(self.shape)
    ok = non_layers.fused_front_end_applies
    assert ok(T((8, 128, 256, 32))) and ok(T((8, 64, 128, 64))) and ok(T((8, 32, 64, 128)))   # L4, L3, L2 at B=8
    assert not ok(T((8, 16, 32, 256)))                  # L1 at B=8: 64 regions
    assert ok(T((32, 16, 32, 256), torch.float16))      # L1 at B=32: 256 regions
    assert not ok(T((8, 128, 256, 48)))                 # C % 32
    assert not ok(T((8, 128, 256, 32)), search_range=3)
    assert not ok(T((8, 128, 256, 32), cuda=False))
    assert not ok(T((8, 128, 256, 32), torch.float64))
    # round 3: a measured size cut on top of the kernel's own rule -- BASELINE config 4 (B=16, 1024x2048)
    kern = non_layers.fused_kernel_applies
    assert kern(T((16, 512, 1024, 32))) and not ok(T((16, 512, 1024, 32)))     # L4: 1.07 GB, the pair wins
    assert kern(T((16, 128, 256, 128))) and not ok(T((16, 128, 256, 128)))     # L2: 268 MB, the pair wins by 4 %
    assert ok(T((16, 64, 128, 256)))                                            # L1: 134 MB, fused wins
    assert kern(T((32, 128, 256, 32), torch.float16)) and not ok(T((32, 128, 256, 32), torch.float16))   # config 5 L4: pair
    assert ok(T((32, 64, 128, 64), torch.float16))                              # config 5 L3: fused


def test_optflow_layer_fusion_rule():
    f = non_layers.OptFlow._fuse_layer
    assert f(128, 8) and f(64, 1) and f(32, 1)          # layers 2-4: fused at every level (outputs split over workgroups)
    assert f(211, 128) and not f(211, 64)               # first layer of L2 at B=8 / fewer tiles
    assert not f(339, 32) and not f(593, 8)             # wide first layers of L1 / L0 stay depthwise + GEMM
    assert f(147, 512) and f(115, 2048) and f(593, 256)
    assert f(339, 128, fp16=True) and not f(339, 128) and not f(593, 32, fp16=True)   # config 5's L1 fused, its L0 split
    assert non_layers.OptFlow.tail_max_pixels == 16384  # one-launch tail at L0-L2 for B=8, 256x512


def test_bench_labels_follow_the_arguments():
    import bench
    assert bench.baseline_config_name(8, (256, 512), "f32", 1) == "BASELINE configs[1]"
    assert bench.baseline_config_name(8, (256, 512), "f32", 8) == "BASELINE configs[2]"
    assert bench.baseline_config_name(16, (1024, 2048), "f32", 1) == "BASELINE configs[3]"
    assert bench.baseline_config_name(32, (256, 512), "f16", 1) == "BASELINE configs[4]"
    assert bench.baseline_config_name(4, (256, 512), "f32", 1) == "non-BASELINE workload"
    assert bench.metric_name((1024, 2048), "f32") == "image-pairs/sec at 1024x2048 fp32"
    assert bench.metric_name((256, 512), "f16") == "image-pairs/sec at 256x512 fp16"
    # SURVEY 8(d) byte counts per launch
    assert bench.cost_volume_bytes(8, 128, 256, 32) == 152043520
    assert bench.warp_bytes(8, 128, 256, 32) == 69206016
    assert bench.fused_front_bytes(8, 128, 256, 32) == 154140672
    assert bench.sepconv_flops(8, 128, 256, 115, 128) == 8260157440
    # the symbol the launcher picks
    assert bench.cost_volume_symbol(8, 128, 256, 32, "f32") == "cost_volume_mfma_lds_kernel"
    assert bench.cost_volume_symbol(32, 128, 256, 32, "f16") == "cost_volume_mfma_lds_f16_kernel"
    assert bench.cost_volume_symbol(8, 16, 32, 256, "f32") == "cost_volume_mfma_kernel"
    assert bench.cost_volume_symbol(1, 128, 256, 3, "f32") == "cost_volume_generic_kernel"   # C % 4 != 0
    assert bench.cost_volume_symbol(1, 128, 256, 12, "f32") == "cost_volume_tiled_kernel"
    # ... and the fused front end's: the library's OWN rules (qpwc_cost_volume_kernel = a dry run of the launchers),
    # e.g. config 2's L3 (C = 64, 256 regions of 16 x 16) runs on 8 x 16 regions, config 4's L1-L3 on 16 x 16
    assert bench.fused_front_symbol(8, 128, 256, 32, "f32", 84) == "cost_volume_mfma_lds8x16_warp_kernel"
    assert bench.fused_front_symbol(8, 64, 128, 64, "f32") == "cost_volume_mfma_lds8x16_warp_kernel"
    assert bench.fused_front_symbol(8, 32, 64, 128, "f32") == "cost_volume_mfma_lds_kernel<true>"
    assert bench.fused_front_symbol(16, 256, 512, 64, "f32") == "cost_volume_mfma_lds16_kernel<true>"
    assert bench.fused_front_symbol(32, 64, 128, 64, "f16") == "cost_volume_mfma_lds_f16_kernel<true>"
    assert bench.fused_front_symbol(8, 16, 32, 256, "f32") == "cost_volume_tiled_kernel<fused>"   # 7 x slower than the pair
    assert bench.fused_front_symbol(8, 1, 32, 256, "f32") == ""                                    # refused: H < 2
    assert bench.baseline_config_name(8, (256, 512), "f32", 8, 64).startswith("BASELINE configs[2]")
    assert bench.parse_args(["--global-batch", "64"]).global_batch == 64
    a = bench.parse_args([])
    assert (a.gpus, a.batch, a.height, a.width, a.dtype, a.data_format, a.fused) == (1, 8, 256, 512, "f32", "channels_last", None)
    assert bench.parse_args(["--no-fused"]).fused is False and bench.parse_args(["--fused"]).fused is True


def test_model_layout_plan_without_a_device():
    """A channels_first model on a HIP device runs channels-last inside; on the CPU (no kernels) it keeps the
    declared layout.  Launch order / decoder granularity attributes exist with their defaults."""
    from qpwcnet_amd.pwcnet import build_flower
    m = build_flower(True, (64, 128), "channels_first", weights=synth.make_weights(42, (64, 128)), device="cpu")
    assert m.data_format == "channels_first" and m._df == "channels_first"
    assert m.capture_order == ("F0", "D0", "D1", "D2", "D3", "F1", "F2", "F3", "F4") and m.dec_chunks == (2, 4, 4, 1)
    assert all(u.fused for u in build_flower(True, (64, 128), "channels_last", weights=synth.make_weights(42, (64, 128)),
                                             device="cpu").upflows)
    with pytest.raises(ValueError):
        m(torch.zeros(1, 64, 128, 6))      # declared channels_first: (B,6,H,W) expected


def test_source_hash_and_traffic_file_agree():
    """profiles/traffic.json is stamped with the sha256 of the hot-path kernel sources; bench.py reports its
    numbers only while the sources still hash to it."""
    import json
    import os
    from qpwcnet_amd import _hip
    h = _hip.source_sha256()
    assert len(h) == 64 and h == _hip.source_sha256()
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic.json")
    t = json.load(open(path))
    assert set(("kernel_source_sha256", "tag", "cost_volume_L4_bytes_per_launch", "warp_clamp_L4_bytes_per_launch",
                "warp_cost_volume_L4_bytes_per_launch")) <= set(t)
    import bench
    val, src = bench.load_traffic("cost_volume_L4_bytes_per_launch")
    assert (val is not None) == (t["kernel_source_sha256"] == h), src


def test_bench_line_fits_the_drivers_window():
    """the one JSON line must stay < 4 KB and its last 2000 characters (what the driver's
    record keeps) must hold every extra config's value, ms_per_step and roofline."""
    import importlib.util
    import json
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(root, "bench.py"))
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        bench = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(bench)
    finally:
        sys.argv = argv
    roof = {"bound": "hbm", "achieved": 4422.6165565062165, "peak": 8000.0, "unit": "GB/s", "frac": 0.5528270695632771,
            "traffic": 156810294.15384614, "traffic_source": "x" * 80, "copy_ceiling_GBs": 6524.7,
            "frac_of_copy_ceiling": 0.6778251608555776, "kernel": "fused WarpV2+cost volume L4 16x512x1024x32",
            "kernel_symbol": "cost_volume_mfma_lds_f16_kernel<true>", "algorithmic_bytes_per_launch": 7079985152,
            "avg_launch_ms": 0.050026841163635254, "unfused_pair_ms": 0.0552796983718872,
            "frac_vs_fused_bytes": 0.38514492524076654, "method": "y" * 400}
    ws = {"gflop_per_step": 68.72, "achieved": 56.9012345, "peak": 157.3, "unit": "TFLOP/s", "frac": 0.361789}

    def leg(name, value):
        return {"metric": "image-pairs/sec at 1024x2048 fp32", "value": value, "unit": "pairs/s", "n_gpus": 1,
                "steps": 10, "warmup": 3, "ms_per_step": 29.987678, "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": name + ": full 6-level PWC-Net (qpwcnet build_flower) inference, batch 16 per "
                           "GPU, 1024x2048 fp32, channels_last, d=4 cost volume + WarpV2", "global_batch": 16,
                           "batch_per_gpu": 16, "parallelism": "dp1 (pairs sharded, RCCL all-gather of the 6 "
                           "per-level EPE)", "hipgraph": True, "fused_upflow": [False, True, True, True],
                           "epe_payload": "z" * 120},
                "roofline": dict(roof), "rooflines_other": {"a": dict(roof), "b": dict(roof)},
                "hot_path": {"kernels_ms": {"k%d" % i: 0.01 * i for i in range(120)}}, "whole_step": dict(ws)}
    full = leg("BASELINE configs[1]", 6624.861032851855)
    full.update(cpu_baseline={"value": 1.2473222377051585, "unit": "pairs/s", "cores": 128, "kind": "port",
                              "sample": "s" * 300, "sample_short": "2 pairs, full net, median of 3 passes"},
                per_level_epe_vs_oracle=[2.69e-07] * 6, per_level_epe_vs_ground_truth=[1.0] * 6,
                serving_throughput={"value": 7232.9, "batches_in_flight": 2, "note": "n" * 100},
                matmul_bf16x3={"value": 6967.123456, "unit": "pairs/s", "ms_per_step": 1.14823456,
                               "max_abs_flow_diff_vs_headline_px": 1.3113e-06,
                               "mean_abs_flow_diff_vs_headline_px": 1.8912e-07, "note": "n" * 200},
                library={"path": "/p", "version": 200, "build": "libqpwc_hip gfx950 product (no environment switches)",
                         "product": True},
                detail_file="bench_detail.json",
                extra_configs=[leg("BASELINE configs[3]", 533.5524), leg("BASELINE configs[4]", 18117.76)])
    line = json.dumps(bench.compact_line(full), separators=(",", ":"))
    assert len(line) < 4096, len(line)
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "whole_step"):
        assert k in d
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in d["roofline"]
    tail = line[-2000:]
    for v in ("533.5524", "18117.76", '"extra_configs"', '"whole_step"'):
        assert v in tail
    assert tail.count('"ms_per_step"') >= 2 and tail.count('"roofline"') >= 2
    # the opt-in arithmetic's leg travels beside the headline, never as `value`
    assert d["matmul_bf16x3"]["value"] == 6967.1 and d["value"] == 6624.861 and "note" not in d["matmul_bf16x3"]


def test_decoder_side_stream_mapping_must_not_return_to_an_earlier_stream():
    """Round 3: a two-stream forward whose decoder levels alternate between two side streams makes the runtime crash
    during hipGraph capture (the round-1 multi-stream crash, reproduced once in its own process by
    tools/dec_streams_ab.py); the model refuses the mapping before any stream is touched."""
    import numpy as np
    from qpwcnet_amd.pwcnet import QpwcNet

    class Enc:   # stands in for the stacked encoder outputs: only .shape / .device are read before the check
        shape = (16, 8, 16, 256)
        device = torch.device("cpu")
    m = QpwcNet.__new__(QpwcNet)
    m._side, m._sides, m.dec_stream_of = object(), [], (0, 1, 0, 1)
    m.allow_returning_dec_streams = False
    import unittest.mock as mock
    with mock.patch("torch.cuda.current_stream", return_value=None):
        with pytest.raises(ValueError, match="non-decreasing"):
            m._forward_two_streams([Enc()], 8)


def test_matmul_switch_reaches_every_convolution_block_and_rejects_unknown_modes():
    """Round 3: QpwcNet.matmul = "bf16x3" (csrc/split_bf16.h) is a property that sets the mode on the encoder, decoder
    and OptFlow blocks; the default is the fp32 matrix instructions and nothing else is accepted."""
    from qpwcnet_amd.pwcnet import QpwcNet

    class Blk:
        matmul = "f32"

    class Up:
        def __init__(self):
            self.flow = Blk()
    m = QpwcNet.__new__(QpwcNet)
    m._matmul, m.enc, m.dec, m.flow, m.upflows = "f32", [Blk(), Blk()], [Blk()], Up(), [Up(), Up()]
    assert m.matmul == "f32" and non_layers.DownConv.matmul == "f32" and non_layers.OptFlow.matmul == "f32"
    m.matmul = "bf16x3"
    assert all(b.matmul == "bf16x3" for b in m.enc + m.dec) and all(u.flow.matmul == "bf16x3" for u in m.upflows + [m.flow])
    with pytest.raises(ValueError, match="matmul"):
        m.matmul = "bf16"
    assert m.matmul == "bf16x3"


def test_sepconv_x3_source_rule_mirrors_the_c_side():
    """qpwc_sepconv3x3_x3_fwd takes the 16-byte staging path only: every source but the last a multiple of 4 channels in
    16-byte aligned pixels, a last source of fewer than 4 channels allowed (sepconv_x3_sources_ok in csrc/sepconv_x3.inc)."""
    z = lambda c: torch.zeros(2, 8, 16, c)
    assert ops.sepconv3x3_x3_applies([z(84), z(32), z(2)])
    assert ops.sepconv3x3_x3_applies([z(128)])
    assert not ops.sepconv3x3_x3_applies([z(7), z(8)])          # an odd source in front
    assert not ops.sepconv3x3_x3_applies([z(8), z(6)])          # a last source of >= 4 channels must be aligned too
    assert not ops.sepconv3x3_x3_applies([z(8).half()])         # fp16 storage has its own kernel
    wide = torch.zeros(2, 8, 16, 40)
    assert ops.sepconv3x3_x3_applies([wide[..., 8:24]])         # a 16-byte aligned channel slice of a wider buffer
    assert not ops.sepconv3x3_x3_applies([wide[..., 2:18]])


def test_bf16x3_split_error_budget_on_the_cpu():
    """The arithmetic behind csrc/split_bf16.h, restated with torch's bfloat16 (round to nearest even, as v_cvt_pk_bf16_f32):
    the three parts reproduce an fp32 value exactly, every partial product of two parts is exact in fp32, and the six
    partial products the kernels accumulate differ from the exact product by less than one fp32 rounding (2^-24) on these
    samples (worst case 2^-23; 2^-28 on average)."""
    g = torch.Generator().manual_seed(7)
    a = torch.randn(1 << 16, generator=g) * torch.exp2(torch.randint(-20, 20, (1 << 16,), generator=g).float())
    b = torch.randn(1 << 16, generator=g) * torch.exp2(torch.randint(-20, 20, (1 << 16,), generator=g).float())

    def split(x):
        p1 = x.bfloat16().float()
        r1 = x - p1
        p2 = r1.bfloat16().float()
        r2 = r1 - p2
        assert torch.equal(r1.double(), x.double() - p1.double()) and torch.equal(r2.double(), r1.double() - p2.double())
        return p1, p2, r2.bfloat16().float()

    a1, a2, a3 = split(a)
    b1, b2, b3 = split(b)
    assert torch.equal(a1.double() + a2.double() + a3.double(), a.double())
    assert float((a2.abs() / a.abs()).max()) <= 2.0 ** -8 and float((a3.abs() / a.abs()).max()) <= 2.0 ** -16
    for x, y in ((a1, b1), (a1, b2), (a2, b1), (a1, b3), (a3, b1), (a2, b2)):
        assert torch.equal((x * y).double(), x.double() * y.double())          # 8 x 8 significant bits: exact in fp32
    six = sum(x.double() * y.double() for x, y in ((a2, b2), (a3, b1), (a1, b3), (a2, b1), (a1, b2), (a1, b1)))
    exact = a.double() * b.double()
    assert float(((six - exact).abs() / exact.abs()).max()) < 2.0 ** -24

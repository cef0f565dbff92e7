"""Case table and input generators shared by ``make_golden.py`` (which writes the
fixtures) and the tests (which regenerate the inputs from the recorded seeds).

The fixtures are RESTATED-ORACLE goldens: TensorFlow / tensorflow-addons cannot be
installed offline, so they come from the float64 numpy restatement in
``oracle/np_ref.py`` -- not from TF outputs.  Nothing from the reference (source
or bytecode) is stored here; only seeds, sampled outputs and checksums.
"""
import numpy as np

# name -> (op, data_format, shape, seed, extra)
# shapes: test/test_cost_volume.py:16-21, app/test/test_cvol_equal.py:11,
#         test/test_warp.py:20-25, and the five 256x512 level shapes.
CASES = {
    "cv_test_nhwc": ("cost_volume", "channels_last", (4, 32, 64, 3), 0, {}),
    "cv_test_nchw": ("cost_volume", "channels_first", (4, 3, 32, 64), 1, {}),
    "cv_equal_nhwc": ("cost_volume", "channels_last", (1, 128, 256, 3), 2, {}),
    "cv_equal_nchw": ("cost_volume", "channels_first", (1, 3, 128, 256), 3, {}),
    "cv_L0": ("cost_volume", "channels_last", (1, 8, 16, 256), 10, {}),
    "cv_L1": ("cost_volume", "channels_last", (1, 16, 32, 256), 11, {}),
    "cv_L2": ("cost_volume", "channels_last", (1, 32, 64, 128), 12, {}),
    "cv_L3": ("cost_volume", "channels_last", (1, 64, 128, 64), 13, {}),
    "cv_L4": ("cost_volume", "channels_last", (1, 128, 256, 32), 14, {}),
    "cv_ragged": ("cost_volume", "channels_last", (2, 19, 37, 8), 15, {}),
    "cv_r2": ("cost_volume", "channels_last", (2, 9, 13, 4), 16, {"search_range": 2}),
    "warp2_test_nhwc": ("warp_v2", "channels_last", (4, 32, 64, 3), 20, {"flow_scale": 1.0}),
    "warp2_test_nchw": ("warp_v2", "channels_first", (4, 3, 32, 64), 21, {"flow_scale": 1.0}),
    "warp1_test_nhwc": ("tf_warp", "channels_last", (4, 32, 64, 3), 22, {"flow_scale": 1.0}),
    "warp1_test_nchw": ("tf_warp", "channels_first", (4, 3, 32, 64), 23, {"flow_scale": 1.0}),
    "warp2_border": ("warp_v2", "channels_last", (2, 16, 24, 8), 24, {"flow_scale": 12.0}),
    "warp1_border": ("tf_warp", "channels_last", (2, 16, 24, 8), 25, {"flow_scale": 12.0}),
    "warp2_L1": ("warp_v2", "channels_last", (1, 16, 32, 256), 31, {"flow_scale": 4.0}),
    "warp2_L2": ("warp_v2", "channels_last", (1, 32, 64, 128), 32, {"flow_scale": 4.0}),
    "warp2_L3": ("warp_v2", "channels_last", (1, 64, 128, 64), 33, {"flow_scale": 4.0}),
    "warp2_L4": ("warp_v2", "channels_last", (1, 128, 256, 32), 34, {"flow_scale": 4.0}),
    "warp1_L4": ("tf_warp", "channels_last", (1, 128, 256, 32), 35, {"flow_scale": 4.0}),
}

N_SAMPLES = 4096


def make_inputs(name, dtype=np.float32):
    op, fmt, shape, seed, extra = CASES[name]
    rng = np.random.default_rng(seed)
    if op == "cost_volume":
        # N(0,1) like test/test_cost_volume.py:20-21
        a = rng.standard_normal(shape).astype(np.float32)
        b = rng.standard_normal(shape).astype(np.float32)
        return a.astype(dtype), b.astype(dtype)
    # img ~ U[0,1), flo ~ N(0,1)*s like test/test_warp.py:24-25
    img = rng.random(shape).astype(np.float32)
    fshape = list(shape)
    fshape[3 if fmt == "channels_last" else 1] = 2
    flo = (rng.standard_normal(fshape) * extra["flow_scale"]).astype(np.float32)
    if "border" in name:
        # hit the special coordinates of tf_warp exactly: x == W-1, y == H-1,
        # (-1, 0), <= -1 -- integer flows on a few pixels
        f = flo if fmt == "channels_last" else np.moveaxis(flo, 1, 3)
        f[:, 0, :, :] = 0.0                  # zero flow on the first row
        f[:, -1, :, :] = 0.0                 # ... and on the last row (y == H-1)
        f[:, :, -1, :] = 0.0                 # last column (x == W-1)
        f[:, 3, :, 0] = -np.arange(f.shape[2], dtype=np.float32) - 0.5   # x in (-1, 0)
        f[:, 4, :, 0] = -np.arange(f.shape[2], dtype=np.float32) - 1.0   # x == -1
        f[:, 5, :, 0] = -np.arange(f.shape[2], dtype=np.float32) - 2.5   # x < -1
        f[:, 6, :, 1] = 100.0                # far below the image
        f[:, 7, :, :] = np.float32(3.0)      # pure integer shift
    return img.astype(dtype), flo.astype(dtype)


def sample_indices(name, out_size):
    seed = CASES[name][3]
    rng = np.random.default_rng(1000 + seed)
    n = min(N_SAMPLES, out_size)
    return np.sort(rng.choice(out_size, size=n, replace=False))


# ---- inverse flow / occlusion map (qpwcnet/core/occlusion.py) -------------
# name -> (data_format, flow shape, seed, noise sigma, smooth amplitude)
OCC_CASES = {
    "occ_nhwc": ("channels_last", (2, 32, 64, 2), 40, 1.0, 6.0),
    "occ_nchw": ("channels_first", (2, 2, 32, 64), 41, 1.0, 6.0),
    "occ_ragged": ("channels_last", (3, 19, 37, 2), 42, 0.5, 10.0),
    "occ_smooth_L4": ("channels_last", (1, 128, 256, 2), 43, 0.0, 8.0),
    "occ_noise": ("channels_last", (1, 24, 40, 2), 44, 4.0, 0.0),
}


def make_flow(name):
    """fp32 flow: smooth low-frequency field (amplitude `amp` px) + N(0, sigma) noise; leaves the
    image near the borders, folds over itself where the noise is large."""
    fmt, shape, seed, sigma, amp = OCC_CASES[name]
    rng = np.random.default_rng(seed)
    if fmt == "channels_last":
        n, h, w, _ = shape
    else:
        n, _, h, w = shape
    yy, xx = np.meshgrid(np.arange(h, dtype=np.float64), np.arange(w, dtype=np.float64), indexing="ij")
    f = np.empty((n, h, w, 2), np.float64)
    for b in range(n):
        ph = rng.uniform(0, 2 * np.pi, size=4)
        f[b, ..., 0] = amp * np.sin(2 * np.pi * yy / h + ph[0]) * np.cos(2 * np.pi * xx / w + ph[1])
        f[b, ..., 1] = amp * np.cos(2 * np.pi * yy / h + ph[2]) * np.sin(2 * np.pi * xx / w + ph[3])
    f += rng.standard_normal(f.shape) * sigma
    f = f.astype(np.float32)
    return f if fmt == "channels_last" else np.ascontiguousarray(np.moveaxis(f, 3, 1))

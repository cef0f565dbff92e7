#!/usr/bin/env python3
"""Headline benchmark: image-pairs/sec of full 6-level PWC-Net (qpwcnet ``build_flower``)
inference, synthetic frames and seeded random-init weights.  Default workload = BASELINE.json
configs[1] (batch 8 per GPU, 256x512 fp32; configs[2] = the same per-GPU work on 8 GPUs).

    python bench.py --gpus N --steps K --warmup W      (N > 1: starts one fresh process per GPU itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (the driver's form)

A step = one forward pass of one batch already resident in HBM: encoder / decoder / flow estimator and
the hot path (5 cost volumes + 4 warps) in the HIP kernels, the wide coarse-level pointwise GEMMs on
rocBLAS, 6 per-level EPE reductions, and (N > 1) one RCCL all-gather of the 6-float EPE vector.
Prints ONE JSON line (< 4 KB) on rank 0.  On one GPU the line also carries `extra_configs`: BASELINE
configs[3] (batch 16, 1024x2048 fp32) and configs[4] (batch 32, 256x512 fp16) measured the same way, each
with its own `value`, `ms_per_step` and `roofline` -- reported beside the headline, never as `value`.
Everything else that was measured (per-kernel times of the eager step, the other roofline blocks, method
notes) goes to `bench_detail.json` beside this file (`--detail PATH`), not into the line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from qpwcnet_amd import _hip  # noqa: E402
from qpwcnet_amd import dist as qdist  # noqa: E402
from qpwcnet_amd import metrics, non_layers, ops, synth  # noqa: E402
from qpwcnet_amd.pwcnet import GraphedForward, build_flower  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec (MI355X_MICROARCH.md); 6290 GB/s is what its float4 copy reaches
HBM_GUIDE_COPY_GBS = 6290.0
F32_MFMA_PEAK_TFS = 157.3  # dense fp32 matrix peak (same guide)
F16_MFMA_PEAK_TFS = 2500.0  # dense fp16/bf16 matrix peak
PARITY_TOL_PX = 1e-4       # north_star: "outputs matching the TF2 reference within 1e-4 fp32"; the run FAILS above it


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=50)
    p.add_argument("--warmup", type=int, default=10)
    p.add_argument("--batch", type=int, default=8, help="pairs per GPU (weak scaling)")
    p.add_argument("--global-batch", type=int, default=0,
                   help="STRONG scaling: this many pairs in total, sharded over the ranks by dist.shard_range "
                        "(BASELINE configs[2] literally: --gpus 8 --global-batch 64); 0 = --batch pairs per GPU")
    p.add_argument("--height", type=int, default=256)
    p.add_argument("--width", type=int, default=512)
    p.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    p.add_argument("--fused", dest="fused", action="store_true", default=None,
                   help="force the fused warp+cost-volume UpFlow front end at every level")
    p.add_argument("--no-fused", dest="fused", action="store_false",
                   help="never fuse warp + cost volume (default: per level, where it is faster)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-overlap", action="store_true",
                   help="decoder and flow chain on one stream (A/B of the two-stream forward)")
    p.add_argument("--inflight", type=int, default=2, help="batches in flight of the serving_throughput leg")
    p.add_argument("--no-inflight", action="store_true",
                   help="skip the extra '2 batches in flight' throughput measurement (N=1 only)")
    p.add_argument("--matmul", default="f32", choices=["f32", "bf16x3"],
                   help="arithmetic of the fp32 convolution products: f32 = the fp32 matrix instructions (the line's "
                        "default and the only form reported as the headline); bf16x3 = three-way bf16 splits "
                        "(DESIGN.md 4.11) -- the run is then labelled so in config.matmul (profiling / A-B runs)")
    p.add_argument("--no-extra", action="store_true",
                   help="skip the extra_configs legs (BASELINE configs[3] and configs[4]; N=1 default run only)")
    p.add_argument("--dtype", default="f32", choices=["f32", "f16"],
                   help="f16 = BASELINE configs[4] (fp16 storage / convs, fp32 accumulate in the hot path)")
    p.add_argument("--data-format", default="channels_last", choices=["channels_last", "channels_first"],
                   help="layout of the model's inputs and outputs (reference default for inference: "
                        "channels_first, app/optical_flow/test_infer.py:52)")
    p.add_argument("--dist-backend", default=None,
                   help="rehearsal only: 'gloo' runs N ranks on ONE GPU (EPE gathered through host memory)")
    p.add_argument("--detail", default=os.path.join(ROOT, "bench_detail.json"),
                   help="file that receives everything measured beyond the one JSON line ('' = do not write)")
    p.add_argument("--stub-forward", action="store_true",
                   help="TEST HOOK (tests/test_dist_cpu.py): no kernels, no GPU -- the launcher and the timed step "
                        "loop only, under gloo on CPU tensors; the line it prints is labelled as a rehearsal")
    p.add_argument("--cpu-pairs", type=int, default=2, help="pairs per timed CPU pass of the full net")
    p.add_argument("--cpu-reps", type=int, default=3, help="timed CPU passes of the full net (median reported)")
    return p.parse_args(argv)


# ---------------------------------------------------------------------------------------------
# labels: derived from what actually runs
class _Operand:
    """shape / dtype carrier of a device tensor for the dispatch rules (they never touch data)"""
    is_cuda = True

    def __init__(self, shape, dtype):
        self.shape, self.dtype = tuple(shape), dtype

    def dim(self):
        return len(self.shape)


def baseline_config_name(B, hw, dtype, world, global_batch=0):
    if global_batch:
        if hw == (256, 512) and dtype == "f32" and global_batch == 64 and world == 8:
            return "BASELINE configs[2] (64 pairs sharded over 8 GPUs, strong scaling)"
        return "non-BASELINE workload ({} pairs sharded over {} GPUs, strong scaling)".format(global_batch, world)
    if hw == (256, 512) and dtype == "f32" and B == 8:
        return "BASELINE configs[1]" if world == 1 else (
            "BASELINE configs[2]" if world == 8 else "BASELINE configs[1] per GPU on {} GPUs".format(world))
    if hw == (1024, 2048) and dtype == "f32" and B == 16 and world == 1:
        return "BASELINE configs[3]"
    if hw == (256, 512) and dtype == "f16" and B == 32 and world == 1:
        return "BASELINE configs[4]"
    return "non-BASELINE workload"


def metric_name(hw, dtype):
    return "image-pairs/sec at {}x{} {}".format(hw[0], hw[1], "fp32" if dtype == "f32" else "fp16")


def cost_volume_bytes(B, H, W, C, esize=4):
    """Algorithmic bytes of one cost-volume launch: read prv + nxt, write 81 channels
    (B*H*W*(2C+81)*e)."""
    return B * H * W * (2 * C + 81) * esize


def warp_bytes(B, H, W, C, esize=4):
    """B*H*W*(2C+2)*e (flow counted at the element size of the image, as there)."""
    return B * H * W * (2 * C + 2) * esize


def fused_front_bytes(B, H, W, C, esize=4):
    """SURVEY.md 8(d), fused warp + cost volume: B*H*W*(2C+2+81)*e."""
    return B * H * W * (2 * C + 2 + 81) * esize


def sepconv_flops(B, H, W, C, F):
    """depthwise 3x3 (18 flop per channel) + pointwise (2F per channel)."""
    return B * H * W * C * (2 * F + 18)


def cost_volume_symbol(B, H, W, C, dtype):
    """The kernel qpwc_cost_volume_fwd launches for an NHWC r=4 shape: asked of the library's own selection rules
    (qpwc_cost_volume_kernel, a dry run of the launchers -- host only), never a copy of them."""
    return ops.cost_volume_kernel(B, H, W, C, dtype, fused=False)


def fused_front_symbol(B, H, W, C, dtype, out_pixel_stride=0):
    """The kernel qpwc_warp_cost_volume_fwd launches for this shape (same query)."""
    return ops.cost_volume_kernel(B, H, W, C, dtype, fused=True, out_pixel_stride=out_pixel_stride)


# ---------------------------------------------------------------------------------------------
# timing helpers (HIP events on the stream the kernels are launched on = torch's current stream)
def device_copy_ceiling(dev, mib=512, reps=10):
    """GB/s (read + write bytes) of the library's own float4 copy kernel (qpwc_device_copy, 16 B per
    lane): the achievable HBM ceiling of THIS box that SURVEY 8(d) asks to report beside the 8 TB/s
    nominal peak (the guide's float4 copy: 6.29 TB/s)."""
    n = mib << 20
    src = torch.empty(n, dtype=torch.uint8, device=dev).random_(0, 255)
    dst = torch.empty_like(src)
    L = _hip.lib()
    st = torch.cuda.current_stream(dev).cuda_stream

    def run():
        _hip.check(L.qpwc_device_copy(src.data_ptr(), dst.data_ptr(), n, st))

    for _ in range(3):
        run()
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            run()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    assert torch.equal(dst[:4096], src[:4096]) and torch.equal(dst[-4096:], src[-4096:])
    return 2.0 * n * reps / (sorted(ts)[1] * 1e-3) / 1e9


def replay_launches(fn, n_rep=50, rounds=5):
    """Average duration of `fn`'s launch: n_rep back-to-back launches captured in one hipGraph and
    replayed between two HIP events (no host launch gaps, whatever the host's speed); median round."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    try:
        g, side = torch.cuda.CUDAGraph(), torch.cuda.Stream()
        with torch.cuda.stream(side):
            with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
                for _ in range(n_rep):
                    fn()
        run = g.replay
    except RuntimeError:   # capture refused (e.g. by another library's stream activity): eager replays
        def run():
            for _ in range(n_rep):
                fn()
    run()   # one untimed replay: clocks and caches in their steady state
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        run()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) / n_rep)
    return sorted(ts)[len(ts) // 2]   # median round, not the best one


def load_traffic(key):
    """HBM bytes per launch from profiles/traffic.json (separate rocprofv3 --pmc passes,
    tools/make_traffic.sh) -- only when the kernel sources still hash to what was profiled."""
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(tpath):
        return None, "profiles/traffic.json missing"
    try:
        t = json.load(open(tpath))
    except Exception as e:  # noqa: BLE001
        return None, "unreadable: {}".format(e)
    if t.get("kernel_source_sha256") != _hip.source_sha256():
        return None, "stale: kernel sources changed since the counters were read (tools/make_traffic.sh)"
    return t.get(key), "profiles/traffic.json ({})".format(t.get("tag", "untagged"))


# ---------------------------------------------------------------------------------------------
def sustained_clock(replay, dev, step_ms):
    """Shader clock the chip holds UNDER THE STEP: one probe wave (qpwc_clock_probe) stamps (s_memtime, s_memrealtime)
    pairs every ~15 us on a stream of its own while the step replays back to back; clock = delta(shader ticks) /
    delta(100 MHz ticks) x 100 MHz per interval.  A pass of its own AFTER the timed region (the probe adds a queue)."""
    n, sleeps = 3000, 4
    buf = torch.zeros(2 * n, dtype=torch.int64, device=dev)
    side = torch.cuda.Stream()
    warm = max(50, int(60.0 / step_ms))           # >= 60 ms of back-to-back steps before the first stamp
    for _ in range(warm):
        replay()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        _hip.check(_hip.lib().qpwc_clock_probe(buf.data_ptr(), n, sleeps, side.cuda_stream))
    for _ in range(max(100, int(150.0 / step_ms))):   # the replays outlast the probe (~3000 x 15 us)
        replay()
    torch.cuda.synchronize()
    v = buf.cpu().view(n, 2).double()
    dt, dr = v[1:, 0] - v[:-1, 0], v[1:, 1] - v[:-1, 1]
    ok = dr > 0
    mhz = (dt[ok] / dr[ok] * 100.0).sort().values
    if mhz.numel() < 10:
        return None
    q = lambda f: float(mhz[min(mhz.numel() - 1, int(f * mhz.numel()))])   # noqa: E731
    span_ms = float((v[-1, 1] - v[0, 1]) / 1e5)
    return {"median_mhz": q(0.5), "p10_mhz": q(0.1), "p90_mhz": q(0.9), "samples": int(mhz.numel()),
            "probe_span_ms": span_ms,
            "f32_matrix_peak_at_median_clock_TFs": 256 * 4 * 64 * q(0.5) * 1e6 / 1e12,
            "method": "one probe wave stamping s_memtime / s_memrealtime (100 MHz) every ~15 us on its own stream "
                      "while the step's hipGraph replays back to back; a pass of its own after the timed region"}


def config1_kernels(dev, hw, tdtype):
    """BASELINE configs[0]: ONE 256x512 pair, d = 4 cost volume + WarpV2 (+ the fused front end where its kernel
    applies) at the five level shapes, B = 1: average launch time (hipGraph of 50 launches, median of 5 rounds)."""
    out = {}
    chans = synth.level_channels()
    g = torch.Generator(device=dev).manual_seed(11)
    for lv in range(5):
        shp = (1, hw[0] >> (5 - lv), hw[1] >> (5 - lv), chans[lv])
        prv = torch.randn(shp, device=dev, generator=g).to(tdtype)
        nxt = torch.randn(shp, device=dev, generator=g).to(tdtype)
        flo = torch.randn(shp[:3] + (2,), device=dev, generator=g) * 2
        e = 4 if tdtype == torch.float32 else 2
        cv_ms = replay_launches(lambda: ops.cost_volume(prv, nxt))
        w_ms = replay_launches(lambda: ops.warp(nxt, flo, "clamp"))
        d = {"shape": "x".join(map(str, shp)),
             "cost_volume_us": 1e3 * cv_ms, "cost_volume_kernel": ops.cost_volume_kernel(*shp, tdtype),
             "cost_volume_GBs": cost_volume_bytes(*shp, e) / (cv_ms * 1e-3) / 1e9,
             "warp_v2_us": 1e3 * w_ms, "warp_v2_GBs": warp_bytes(*shp, e) / (w_ms * 1e-3) / 1e9}
        fk = ops.cost_volume_kernel(*shp, tdtype, fused=True)
        if fk.startswith("cost_volume_mfma_lds"):
            f_ms = replay_launches(lambda: ops.warp_cost_volume(prv, nxt, flo))
            d.update({"fused_us": 1e3 * f_ms, "fused_kernel": fk})
        out["L%d" % lv] = d
    return out


# ---------------------------------------------------------------------------------------------
def cpu_baseline(weights, pairs_np, n_pairs, reps, gpu_flows, hw):
    """The reference-algorithm CPU restatement (TF2 itself cannot run offline) on the host cores:
    the two hot-path ops op for op (oracle/torch_ref.py: 81 x slice*mul*mean + concat + lrelu; gather
    warp) at their L4 shape as MEDIAN of 10, and the full 6-level net (oracle/net_ref.py) on n_pairs
    pairs of the same batch as MEDIAN of `reps` after one warm-up."""
    from oracle import net_ref, torch_ref
    cores = torch.get_num_threads()
    n_pairs = max(1, min(n_pairs, pairs_np.shape[0]))
    net = net_ref.RefNet(weights)
    t_all = time.perf_counter()
    ref = net(pairs_np[:n_pairs])  # warm-up (thread pool, allocator) -- also the parity reference
    ts = []
    for _ in range(max(1, reps)):
        t0 = time.perf_counter()
        net(pairs_np[:n_pairs])
        ts.append(time.perf_counter() - t0)
    dt = sorted(ts)[len(ts) // 2]
    epe = [float(torch_ref.epe_error(a[:n_pairs].float().cpu(), b)) for a, b in zip(gpu_flows, ref)]
    # hot-path ops alone, one pair at the finest level shape
    g = torch.Generator().manual_seed(0)
    shp = (1, hw[0] // 2, hw[1] // 2, synth.level_channels()[-1])
    prv, nxt = torch.randn(shp, generator=g), torch.randn(shp, generator=g)
    flo = torch.randn(shp[:3] + (2,), generator=g) * 4

    def med(fn, n=10):
        fn()
        out = []
        for _ in range(n):
            t0 = time.perf_counter()
            fn()
            out.append(time.perf_counter() - t0)
        return sorted(out)[n // 2]
    t_cv = med(lambda: torch_ref.cost_volume(prv, nxt))
    t_wp = med(lambda: torch_ref.warp_v2(nxt, flo))
    total = time.perf_counter() - t_all
    return {
        "value": n_pairs / dt, "unit": "pairs/s", "cores": cores, "kind": "port",
        "sample_short": "{} pairs of the batch, full net, torch-CPU restatement, median of {} passes of {:.1f} s "
                        "(leg {:.0f} s)".format(n_pairs, len(ts), dt, total),
        "sample": "{} pairs of the same batch, full 6-level net, torch-CPU op-for-op restatement (81x "
                  "slice*mul*mean cost volume, gather warp): median of {} passes of {:.1f} s after one "
                  "warm-up; whole cpu_baseline leg {:.0f} s".format(n_pairs, len(ts), dt, total),
        "net_pass_s": [round(t, 3) for t in ts],
        "hot_path_ops": {
            "shape": "x".join(map(str, shp)), "runs": 10, "statistic": "median",
            "cost_volume_ms": 1e3 * t_cv, "cost_volume_GBs": cost_volume_bytes(*shp) / t_cv / 1e9,
            "warp_v2_ms": 1e3 * t_wp, "warp_v2_GBs": warp_bytes(*shp) / t_wp / 1e9,
        },
    }, epe


# ---------------------------------------------------------------------------------------------
def rooflines(model, model_input, B, hw, dtype, tdtype, dev, args, copy_gbs):
    """Live roofline blocks of one configuration: the L4 cost-volume launch (dominant hot-path kernel,
    HBM), the L4 WarpV2 (HBM), the fused UpFlow front end where the model uses it (HBM), and the first
    fused SeparableConv2D of L4 (the step's largest kernel; fp32 matrix pipe)."""
    esize = 4 if dtype == "f32" else 2
    chans = synth.level_channels()
    lvl4 = (B, hw[0] // 2, hw[1] // 2, chans[-1])
    model.overlap_streams = False   # one stream: events would otherwise time kernels sharing the chip
    n_prof = 5

    def forward():
        with torch.no_grad():
            return model(model_input)
    key_cv = ("cost_volume",) + lvl4
    key_fcv = ("warp_cost_volume",) + lvl4
    with ops.kernel_timing() as kt:
        for _ in range(n_prof):
            forward()
    ktimes = kt.summary()
    model.overlap_streams = not args.no_overlap
    out = {}
    up4 = model.upflows[-1]
    g = torch.Generator(device=dev).manual_seed(7)
    prv = torch.randn(lvl4, device=dev, generator=g).to(tdtype)
    nxt = torch.randn(lvl4, device=dev, generator=g).to(tdtype)
    flo = torch.randn(lvl4[:3] + (2,), device=dev, generator=g) * 4
    stride = 84 if up4.flow.wants_cost84(prv) else 81
    cbuf = torch.empty(lvl4[:3] + (stride,), dtype=tdtype, device=dev)

    def hbm_block(name, symbol, nbytes, ms, traffic_key, extra=None):
        achieved = nbytes / (ms * 1e-3) / 1e9
        # profiled shapes (tools/make_traffic.sh): the L4 launches of BASELINE configs 2, 4 and 5
        prefix = {("f32", (8, 128, 256)): "", ("f32", (16, 512, 1024)): "c4_", ("f16", (32, 128, 256)): "c5_"}.get(
            (dtype, tuple(lvl4[:3])))
        traffic, tsrc = load_traffic(prefix + traffic_key) if prefix is not None else (None, "not profiled for this shape")
        d = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
             "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": tsrc,
             "copy_ceiling_GBs": copy_gbs, "frac_of_copy_ceiling": achieved / copy_gbs,
             "frac_of_guide_copy_6290": achieved / HBM_GUIDE_COPY_GBS,
             "kernel": name, "kernel_symbol": symbol, "algorithmic_bytes_per_launch": nbytes,
             "avg_launch_ms": ms}
        d.update(extra or {})
        return d

    # -- cost volume L4: the step's own launch shape (84-float pixels where the fused first OptFlow layer
    # reads them), random inputs, 50 back-to-back launches per round
    fused4 = bool(up4.fuses(prv, flo))
    cv_ms = replay_launches(lambda: ops.cost_volume_into(prv, nxt, cbuf, 0))
    sym_by_level = {"L4": cv_ms}
    for lv in (3, 2, 1, 0):
        shp = (B, hw[0] >> (5 - lv), hw[1] >> (5 - lv), chans[lv])
        if cost_volume_symbol(*shp, dtype) != cost_volume_symbol(*lvl4, dtype):
            break
        pl = torch.randn(shp, device=dev, generator=g).to(tdtype)
        nl = torch.randn(shp, device=dev, generator=g).to(tdtype)
        blk = model.flow if lv == 0 else model.upflows[lv - 1]
        bl = torch.empty(shp[:3] + (84 if blk.flow.wants_cost84(pl) else 81,), dtype=tdtype, device=dev)
        sym_by_level["L%d" % lv] = replay_launches(lambda: ops.cost_volume_into(pl, nl, bl, 0))
    in_step = ktimes.get(key_cv)
    out["cost_volume"] = hbm_block(
        "cost_volume L4 {}".format("x".join(map(str, lvl4))), cost_volume_symbol(*lvl4, dtype),
        cost_volume_bytes(*lvl4, esize), cv_ms, "cost_volume_L4_bytes_per_launch",
        {"symbol_launch_ms_by_level": sym_by_level,
         "symbol_avg_ms": sum(sym_by_level.values()) / len(sym_by_level),
         "avg_launch_ms_inside_eager_step": in_step[1] if in_step else None,
         "launches_per_step": in_step[0] // n_prof if in_step else 0,
         "out_pixel_stride": stride,
         "method": "HIP events on the launch stream around a hipGraph of 50 back-to-back launches at the "
                   "step's own L4 shape and output pixel stride (84 = 81 channels + 3 zeroed pads; the "
                   "algorithmic bytes count 81), median of 5 rounds after a warm one"})
    # -- WarpV2 L4
    w_ms = replay_launches(lambda: ops.warp(nxt, flo, "clamp"))
    in_step = ktimes.get(("warp_clamp",) + lvl4)
    out["warp_v2"] = hbm_block(
        "WarpV2 L4 {}".format("x".join(map(str, lvl4))), "warp_nhwc_vec4_kernel", warp_bytes(*lvl4, esize),
        w_ms, "warp_clamp_L4_bytes_per_launch",
        {"avg_launch_ms_inside_eager_step": in_step[1] if in_step else None,
         "launches_per_step": in_step[0] // n_prof if in_step else 0})
    # -- fused WarpV2 + cost volume.  SURVEY 8(d): "a fused warp+cv variant is still scored
    # against these unfused bytes (so it may exceed 1.0 of 'unfused roofline'; report it separately against fused
    # bytes B*H*W*(2C+2+81)*e)": `achieved` / `frac` count the unfused pair's algorithmic bytes (cost volume +
    # warp), `achieved_vs_fused_bytes_GBs` / `frac_vs_fused_bytes` the bytes the fused launch itself has to move.
    try:
        f_ms = replay_launches(lambda: ops.cost_volume_into(prv, nxt, cbuf, 0, flo=flo))
        unf = cost_volume_bytes(*lvl4, esize) + warp_bytes(*lvl4, esize)
        fb = fused_front_bytes(*lvl4, esize)
        in_step = ktimes.get(key_fcv)
        out["warp_cost_volume_fused"] = hbm_block(
            "fused WarpV2+cost volume L4 {}".format("x".join(map(str, lvl4))),
            fused_front_symbol(*lvl4, dtype, stride),
            unf, f_ms, "warp_cost_volume_L4_bytes_per_launch",
            {"used_by_the_step_at_L4": fused4, "unfused_pair_ms": cv_ms + w_ms,
             "algorithmic_bytes_basis": "unfused pair: cost volume B*H*W*(2C+81)*e + warp B*H*W*(2C+2)*e (SURVEY 8(d))",
             "fused_algorithmic_bytes": fb, "achieved_vs_fused_bytes_GBs": fb / (f_ms * 1e-3) / 1e9,
             "frac_vs_fused_bytes": fb / (f_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
             "avg_launch_ms_inside_eager_step": in_step[1] if in_step else None,
             "launches_per_step": in_step[0] // n_prof if in_step else 0,
             "out_pixel_stride": stride})
    except (ValueError, RuntimeError) as e:
        out["warp_cost_volume_fused"] = {"error": str(e).splitlines()[0]}
    # -- first SeparableConv2D of L4's OptFlow: [cost | prv | flo] -> 128, fused depthwise + pointwise
    try:
        of = up4.flow
        of._prepare_hip()
        F_ = of.filters[0]
        cost = torch.randn(lvl4[:3] + (stride,), device=dev, generator=g).to(tdtype)
        if stride == 84:
            cost[..., 81:] = 0
            dw, pw = of._dw84, (of._pw_pad84 if dtype == "f32" else of._pw_pad84_16)
        else:
            dw, pw = of._dw[0], (of._pw_pad[0] if dtype == "f32" else of._pw_pad16[0])
        srcs = [cost, prv, flo.to(tdtype)]
        s_ms = replay_launches(lambda: ops.sepconv3x3(srcs, dw, pw, of._pw_b32[0], mish_on_store=True), n_rep=20)
        c_real = 81 + lvl4[3] + 2
        fl = sepconv_flops(lvl4[0], lvl4[1], lvl4[2], c_real, F_)
        peak = F32_MFMA_PEAK_TFS if dtype == "f32" else F16_MFMA_PEAK_TFS
        tf = fl / (s_ms * 1e-3) / 1e12
        key_s = [k for k in ktimes if k[0].startswith("sepconv3x3") and k[1:4] == lvl4[:3] and k[-1] == F_]
        out["sepconv3x3_fused_L4_first_layer"] = {
            "bound": "mfma", "achieved": tf, "peak": peak, "unit": "TFLOP/s", "frac": tf / peak, "traffic": None,
            "kernel": "SeparableConv2D {}->{} L4 {}x{}x{}".format(c_real, F_, *lvl4[:3]),
            "kernel_symbol": "sepconv3x3_fused_kernel<{}>".format(F_) if dtype == "f32"
                             else "sepconv3x3_fused_f16_kernel<{}>".format(F_),
            "algorithmic_flops_per_launch": fl, "avg_launch_ms": s_ms,
            "algorithmic_bytes_per_launch": lvl4[0] * lvl4[1] * lvl4[2] * (c_real + F_) * esize,
            "avg_launch_ms_inside_eager_step": ktimes[key_s[0]][1] if key_s else None,
            "note": "the step's largest single kernel; flops = B*H*W*C*(2F+18) with the 115 real input "
                    "channels (DESIGN.md 4.6), peak = dense {} MFMA".format("fp32" if dtype == "f32" else "fp16")}
    except (ValueError, RuntimeError, AttributeError) as e:
        out["sepconv3x3_fused_L4_first_layer"] = {"error": str(e).splitlines()[0]}
    hot_ms = sum(n * t for k, (n, t) in ktimes.items()
                 if k[0] in ("cost_volume", "warp_clamp", "warp_cost_volume")) / n_prof
    hot = {"ms_per_step_eager_events": hot_ms,
           "kernels_ms": {"{} {}".format(k[0], "x".join(map(str, k[1:]))): round(t, 5)
                          for k, (n, t) in sorted(ktimes.items())}}
    return out, hot


def measure(args, B, hw, dtype, steps, warmup, world, rank, dev, headline, copy_gbs, global_batch=0):
    """One configuration end to end -> the fields of its JSON object.  global_batch > 0: B is this rank's shard of
    that many pairs (strong scaling); otherwise every rank runs B pairs (weak scaling)."""
    tdtype = torch.float32 if dtype == "f32" else torch.float16
    weights = synth.make_weights(42, hw)
    cl = args.data_format == "channels_last"
    model = build_flower(True, hw, args.data_format, weights=weights, device=dev, fused=args.fused,
                         dtype=tdtype)
    if args.no_overlap:
        model.overlap_streams = False
    if args.matmul != "f32" and dtype == "f32":
        model.matmul = args.matmul
    pairs_np, gt_np = synth.make_frames(B, hw[0], hw[1], seed=1234 + rank)
    pairs_cl = torch.from_numpy(pairs_np).to(dev, tdtype)
    pairs = pairs_cl if cl else pairs_cl.permute(0, 3, 1, 2).contiguous()
    gt = torch.from_numpy(gt_np).to(dev)
    shapes = [(hw[0] >> s, hw[1] >> s) for s in (5, 4, 3, 2, 1, 0)]
    gt_pyr = metrics.multiscale_ground_truth(gt, shapes)
    if not cl:   # the model's flows are (B,2,h,w): so is the ground truth, the reduction reads planes
        gt_pyr = [g.permute(0, 3, 1, 2).contiguous() for g in gt_pyr]

    def epe_of(flows, out=None):
        return metrics.per_level_epe(gt_pyr, flows, data_format=args.data_format, out=out)

    def forward():
        with torch.no_grad():
            flows = model(pairs)
            return flows, epe_of(flows)

    # eager warm-up (library solver search, LDS attribute set-up) before any capture
    for _ in range(3):
        flows, epe_local = forward()
    torch.cuda.synchronize()

    # The one collective of the path: an all-gather of the 6 per-level EPE (+ shard weight) per step.
    # On RCCL it is asynchronous: the exchange of step k travels while step k+1 computes, and its result
    # is consumed (the stream waits for it) one step later; the last one is drained inside the timed region.
    # (CPU rehearsal of the N > 1 path, --dist-backend gloo: the same submit/collect pattern on host tensors)
    gloo = args.dist_backend == "gloo"
    gather = qdist.EpeGather(6, "cpu" if gloo else dev, n_local=B)
    in_place = not gloo and not args.no_graph   # the graphs' EPE reductions write the payload themselves

    graphs = None
    if not args.no_graph:
        try:
            if in_place:
                # two graphs over ONE memory pool, replayed alternately: they differ only in the payload
                # slot the captured EPE reduction writes, so no copy stands between it and the collective
                g0 = GraphedForward(model, pairs, epilogue=lambda fl: epe_of(fl, gather.payload_view(0)), warmup=0)
                g1 = GraphedForward(model, pairs, epilogue=lambda fl: epe_of(fl, gather.payload_view(1)), warmup=0,
                                    share_with=g0)
                graphs = [g0, g1]
            else:
                graphs = [GraphedForward(model, pairs, epilogue=epe_of, warmup=0)]
            flows, epe_local = graphs[0].outputs, graphs[0].extra
        except RuntimeError as e:   # capture refused: keep measuring, with eager launches
            print("bench.py: hipGraph capture failed ({}); eager launches".format(str(e).splitlines()[0]),
                  file=sys.stderr)
            graphs = None
            torch.cuda.synchronize()

    def run_step(k):
        if graphs is not None and in_place:
            s = gather.next_slot()
            graphs[s].replay()
            return s
        if graphs is not None:
            graphs[0].replay()
            e = graphs[0].extra
        else:
            _, e = forward()
        return e.cpu() if gloo else e

    elapsed, results = qdist.timed_steps(run_step, gather, steps, warmup, dev)
    per_rank, epe_mean = results[-1]
    assert len(results) == steps
    total_pairs = global_batch if global_batch else world * B
    if tuple(per_rank.shape) != (world, 6):
        raise RuntimeError("all-gather of the per-level EPE returned {} for {} ranks".format(tuple(per_rank.shape), world))

    res = {
        "metric": metric_name(hw, dtype),
        "value": total_pairs * steps / elapsed,
        "unit": "pairs/s",
        "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": 1e3 * elapsed / steps,
        "higher_is_better": True, "scaling": "strong" if global_batch else "weak", "vs_baseline": None,
        # what the one collective of the path saw in the LAST timed step: one row per rank, one column per level
        "allgather": {"ranks_in_allgather": int(per_rank.shape[0]), "levels_per_rank": int(per_rank.shape[1]),
                      "backend": "none (single process)" if world == 1 and not gather.collective else
                                 ("gloo" if gloo else "nccl (RCCL)"),
                      "finest_level_epe_per_rank": [float(x) for x in per_rank[:, -1].cpu()]},
        "dtype": dtype, "data": "synthetic",
        "config": {
            "workload": "{}: full 6-level PWC-Net (qpwcnet build_flower) inference, batch {} per GPU, "
                        "{}x{} {}, {}, d=4 cost volume + WarpV2".format(
                            baseline_config_name(B, hw, dtype, world, global_batch), B, hw[0], hw[1],
                            "fp32" if dtype == "f32" else "fp16 storage (fp32 accumulate)", args.data_format),
            "global_batch": total_pairs, "batch_per_gpu": B,
            "parallelism": "dp{} (pairs sharded, {} all-gather of the 6 per-level EPE)".format(
                world, "gloo REHEARSAL on one GPU:" if gloo else "RCCL"),
            "hipgraph": graphs is not None,
            "epe_payload": ("written by the captured EPE reduction (two graphs over one memory pool, replayed "
                            "alternately; no per-step copy)" if graphs is not None and in_place else "copied per step"),
            # levels L1..L4 whose UpFlow runs WarpV2 + cost volume as one launch
            "fused_upflow": [bool(u.fuses(_Operand((B, hw[0] >> (4 - i), hw[1] >> (4 - i),
                                                    synth.level_channels()[i + 1]), tdtype)))
                             for i, u in enumerate(model.upflows)],
            "hip_optflow": True,
            "matmul": model.matmul,
            "weights": "seeded glorot (synth.make_weights(42)), 3.09M params",
        },
        "per_level_epe_vs_ground_truth": [float(x) for x in epe_mean.cpu()],
    }

    # ---- serving-style throughput, reported BESIDE the headline (never as `value`): two batches
    # in flight, each a hipGraph replay on its own stream, so that the launch-bound coarse levels of
    # one batch run under the encoder / finest level of the other.  Same K steps, same work per step.
    if headline and graphs is not None and world == 1 and not args.no_inflight:
        n_fly = max(2, args.inflight)
        lanes = [(torch.cuda.Stream(), graphs[0])]
        for k in range(1, n_fly):
            pk, _ = synth.make_frames(B, hw[0], hw[1], seed=4321 + k)
            pk = torch.from_numpy(pk).to(dev, tdtype)
            lanes.append((torch.cuda.Stream(),
                          GraphedForward(model, pk if cl else pk.permute(0, 3, 1, 2).contiguous(),
                                         epilogue=epe_of, warmup=0)))
        torch.cuda.synchronize()

        def run(n):
            for i in range(n):
                st, g = lanes[i % n_fly]
                with torch.cuda.stream(st):
                    g.replay()

        run(max(n_fly, warmup))
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        run(steps)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        res["serving_throughput"] = {
            "batches_in_flight": n_fly, "value": B * steps / dt, "unit": "pairs/s",
            "ms_per_step": dt / steps * 1e3,
            "note": "{} hipGraph replays (batch {} each) on as many streams; not the headline value".format(n_fly, B)}
        del lanes[1:]

    # ---- the same step with the fp32 matrix products as three-way bf16 splits on the bf16 matrix instructions
    # (QpwcNet.matmul = "bf16x3", csrc/split_bf16.h): reported BESIDE the headline, which stays on the fp32 matrix
    # instructions; same inputs, same K steps, flows compared with the headline's
    if headline and graphs is not None and world == 1 and dtype == "f32" and not args.no_inflight and \
            args.matmul == "f32":
        ref_flows = [f.clone() for f in graphs[0].outputs]
        model.matmul = "bf16x3"
        try:
            with torch.no_grad():
                model(pairs)
            gx = GraphedForward(model, pairs, epilogue=epe_of, warmup=0)
            for _ in range(max(2, warmup)):
                gx.replay()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(steps):
                gx.replay()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t1
            res["matmul_bf16x3"] = {
                "value": B * steps / dt, "unit": "pairs/s", "ms_per_step": dt / steps * 1e3,
                "max_abs_flow_diff_vs_headline_px": max(float((a - b).abs().max()) for a, b in zip(gx.outputs, ref_flows)),
                "mean_abs_flow_diff_vs_headline_px": max(float((a - b).abs().mean()) for a, b in zip(gx.outputs, ref_flows)),
                "note": "opt-in arithmetic (encoder 3x3 convolutions and the 64-output SeparableConv2D layers): six bf16 "
                        "partial products per fp32 product, fp32 accumulate; not the headline value"}
            del gx
        except Exception as e:  # noqa: BLE001 -- an opt-in side leg must never lose the headline line
            res["matmul_bf16x3"] = {"error": str(e).splitlines()[0][:160] if str(e) else type(e).__name__}
            torch.cuda.synchronize()
        finally:
            model.matmul = "f32"
        del ref_flows

    # ---- the shader clock the chip sustains under this step (reported, never used to scale anything)
    if headline and graphs is not None and world == 1 and not args.no_inflight:
        try:
            res["sustained_clock"] = sustained_clock(graphs[0].replay, dev, res["ms_per_step"])
        except (RuntimeError, ValueError, AttributeError) as e:
            res["sustained_clock"] = {"error": str(e).splitlines()[0][:160]}
        try:
            res["config1_kernels_B1"] = config1_kernels(dev, hw, tdtype)
        except (RuntimeError, ValueError) as e:
            res["config1_kernels_B1"] = {"error": str(e).splitlines()[0][:160]}

    # ---- live rooflines (single stream, HIP events)
    blocks, hot = rooflines(model, pairs, B, hw, dtype, tdtype, dev, args, copy_gbs)
    # the dominant hot-path launch of THIS step: the fused WarpV2 + cost volume where UpFlow uses it at L4, the
    # plain cost volume otherwise; the other one stays in rooflines_other (the standalone CostVolume layer and
    # the coarse levels still launch the plain kernel)
    fused_dom = blocks.get("warp_cost_volume_fused", {}).get("used_by_the_step_at_L4") is True
    res["roofline"] = blocks.pop("warp_cost_volume_fused" if fused_dom else "cost_volume")
    res["rooflines_other"] = blocks
    res["hot_path"] = hot
    res["whole_step"] = whole_step_block(res, B, hw, dtype, world)
    return res, (weights, pairs_np, flows)


GFLOP_PER_PAIR_256x512 = 8.59   # encoder 3.68 + decoder 1.88 + OptFlow 2.71 + cost volume 0.32


def whole_step_block(res, B, hw, dtype, world):
    """Whole-step arithmetic rate against the dense matrix peak of the dtype: the network's algorithmic
    flops per pair (scaled with the pixel count) x pairs per step / measured step time."""
    gflop = GFLOP_PER_PAIR_256x512 * (hw[0] * hw[1]) / (256.0 * 512.0) * B * world
    tf = gflop / res["ms_per_step"]   # GFLOP / ms = TFLOP/s
    peak = (F32_MFMA_PEAK_TFS if dtype == "f32" else F16_MFMA_PEAK_TFS) * world
    return {"gflop_per_step": round(gflop, 2), "achieved": tf, "peak": peak, "unit": "TFLOP/s", "frac": tf / peak}


def _r(x, sig=5):
    """floats to `sig` significant digits (the line has a size budget); everything else unchanged"""
    if isinstance(x, float):
        return float("{:.{}g}".format(x, sig))
    if isinstance(x, (list, tuple)):
        return [_r(v, sig) for v in x]
    return x


ROOFLINE_LINE_KEYS = ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel_symbol", "kernel",
                      "algorithmic_bytes_per_launch", "avg_launch_ms", "frac_of_copy_ceiling",
                      "unfused_pair_ms", "frac_vs_fused_bytes")


def compact_roofline(r, keys=ROOFLINE_LINE_KEYS):
    return {k: _r(r[k]) for k in keys if k in r}


def compact_line(full):
    """The one JSON line: the contract's fields + roofline + cpu_baseline + whole_step + one short object per
    extra config (LAST, so that a tail of the line shows them).  Everything else stays in bench_detail.json."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data")
    line = {k: _r(full[k], 7) for k in keep}
    if "allgather" in full:
        ag = full["allgather"]
        line["allgather"] = {"ranks_in_allgather": ag["ranks_in_allgather"], "levels_per_rank": ag["levels_per_rank"],
                             "backend": ag["backend"]}
    c = full["config"]
    line["config"] = {k: c[k] for k in ("workload", "global_batch", "batch_per_gpu", "parallelism", "hipgraph",
                                        "fused_upflow") if k in c}
    if c.get("matmul", "f32") != "f32":     # an explicitly requested arithmetic is named in the line
        line["config"]["matmul"] = c["matmul"]
    line["roofline"] = compact_roofline(full["roofline"])
    cb = full.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = {k: _r(cb[k]) for k in ("value", "unit", "cores", "kind")}
        line["cpu_baseline"]["sample"] = cb["sample_short"]
    else:
        line["cpu_baseline"] = None
    if "per_level_epe_vs_oracle" in full:
        line["per_level_epe_vs_oracle"] = _r(full["per_level_epe_vs_oracle"], 3)
    if "parity_gate" in full:
        line["parity_gate"] = {"tol": full["parity_gate"]["tolerance_px"], "pass": full["parity_gate"]["pass"]}
        if "skipped" in full["parity_gate"]:
            line["parity_gate"]["skipped"] = full["parity_gate"]["skipped"]
    if "serving_throughput" in full:
        st = full["serving_throughput"]
        line["serving_throughput"] = {"value": _r(st["value"]), "batches_in_flight": st["batches_in_flight"]}
    if "matmul_bf16x3" in full and "error" in full["matmul_bf16x3"]:
        line["matmul_bf16x3"] = {"error": full["matmul_bf16x3"]["error"]}
    elif "matmul_bf16x3" in full:
        mx = full["matmul_bf16x3"]
        line["matmul_bf16x3"] = {"value": _r(mx["value"]), "ms_per_step": _r(mx["ms_per_step"]),
                                 "max_abs_flow_diff_vs_headline_px": _r(mx["max_abs_flow_diff_vs_headline_px"], 3),
                                 "mean_abs_flow_diff_vs_headline_px": _r(mx.get("mean_abs_flow_diff_vs_headline_px"), 3)}
    lib = full.get("library") or {}
    line["library"] = "{} v{}{}".format(lib.get("build"), lib.get("version"), "" if lib.get("product") else " NOT-PRODUCT")
    line["detail"] = full.get("detail_file")
    line["whole_step"] = {k: _r(v) for k, v in full["whole_step"].items()}
    sc = full.get("sustained_clock")
    if sc and "median_mhz" in sc:
        line["whole_step"]["clock_mhz"] = _r(sc["median_mhz"], 4)
        line["whole_step"]["frac_of_peak_at_that_clock"] = _r(
            full["whole_step"]["achieved"] / sc["f32_matrix_peak_at_median_clock_TFs"], 3) \
            if full.get("dtype") == "f32" else None
    if "extra_configs" in full:
        ex = []
        for r in full["extra_configs"]:
            if "error" in r:
                ex.append({"config": r["config"]["workload"], "metric": r["metric"], "error": r["error"][:160]})
                continue
            ex.append({"config": r["config"]["workload"].split(":")[0], "metric": r["metric"],
                       "value": _r(r["value"], 7), "unit": r["unit"], "ms_per_step": _r(r["ms_per_step"], 7),
                       "steps": r["steps"], "dtype": r["dtype"], "batch_per_gpu": r["config"]["batch_per_gpu"],
                       "fused_upflow": r["config"].get("fused_upflow"),
                       "roofline": compact_roofline(r["roofline"], (
                           "bound", "achieved", "peak", "unit", "frac", "traffic", "kernel_symbol",
                           "algorithmic_bytes_per_launch", "avg_launch_ms", "unfused_pair_ms", "frac_vs_fused_bytes")),
                       "whole_step_frac": _r(r["whole_step"]["frac"])})
        line["extra_configs"] = ex
    return line


def self_launch(n):
    """`python bench.py --gpus N` without a torchrun environment: start N fresh processes (one per GPU) through
    torch.distributed.run and pass rank 0's line through.  This parent has made NO GPU call (it runs before
    anything touches torch.cuda) and does not re-exec itself: it waits for the children and returns their
    exit code, non-zero if any rank failed."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("bench.py: --gpus {} without WORLD_SIZE: launching {}".format(n, " ".join(cmd[1:9])), file=sys.stderr)
    return subprocess.call(cmd, env=env)


def rehearse_stub(args):
    """--stub-forward: the launcher path and THE timed step loop (dist.timed_steps) with a stub forward on CPU
    tensors under gloo -- what tests/test_dist_cpu.py runs where there is no GPU.  Not a measurement."""
    world, rank, _ = qdist.init("gloo")
    batch = args.batch
    if args.global_batch:
        lo, hi = qdist.shard_range(args.global_batch, rank, world)
        batch = hi - lo
    gather = qdist.EpeGather(6, "cpu", n_local=batch)
    base = torch.arange(6, dtype=torch.float32)

    def run_step(k):
        time.sleep(0.002)
        return base + 10.0 * k + 1000.0 * rank

    elapsed, results = qdist.timed_steps(run_step, gather, args.steps, args.warmup, "cpu")
    assert len(results) == args.steps
    per_rank, mean = results[-1]
    if rank == 0:
        print(json.dumps({"metric": "REHEARSAL (stub forward, no kernels): launcher + step loop only",
                          "value": None, "unit": "pairs/s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "rehearsal": True,
                          "scaling": "strong" if args.global_batch else "weak",
                          "global_batch": args.global_batch or world * args.batch, "batch_rank0": batch,
                          "allgather": {"ranks_in_allgather": int(per_rank.shape[0]),
                                        "levels_per_rank": int(per_rank.shape[1]), "backend": "gloo"},
                          "last_step_per_rank": per_rank.tolist(), "last_step_mean": mean.tolist()}))
    if world > 1:
        qdist.barrier()
        torch.distributed.destroy_process_group()
    return 0


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args.gpus)      # before any GPU call; the children are fresh processes
    if args.gpus != qdist.env_world()[0]:   # before the rendezvous: a wrong environment must not wait for peers
        raise SystemExit("--gpus {} but WORLD_SIZE={}".format(args.gpus, qdist.env_world()[0]))
    if args.stub_forward:
        return rehearse_stub(args)
    world, rank, local_rank = qdist.init(args.dist_backend)
    if args.dist_backend == "gloo":
        local_rank = 0  # rehearsal: every rank shares cuda:0
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU fallback)")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    torch.backends.cudnn.benchmark = True  # MIOpen find mode (fp16 library convolutions): fastest solver once

    info = _hip.build_info()
    copy_gbs = device_copy_ceiling(dev)
    hw = (args.height, args.width)
    batch = args.batch
    if args.global_batch:
        if args.global_batch < world:
            raise SystemExit("--global-batch {} < {} ranks: every rank needs at least one pair".format(
                args.global_batch, world))
        lo, hi = qdist.shard_range(args.global_batch, rank, world)
        batch = hi - lo
    result, (weights, pairs_np, flows) = measure(args, batch, hw, args.dtype, args.steps, args.warmup,
                                                 world, rank, dev, True, copy_gbs, args.global_batch)
    result["library"] = info
    default_run = (world == 1 and batch == 8 and not args.global_batch and hw == (256, 512) and args.dtype == "f32" and
                   args.data_format == "channels_last" and not args.no_graph and args.matmul == "f32")
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            base, epe_oracle = cpu_baseline(weights, pairs_np, args.cpu_pairs, args.cpu_reps, flows, hw)
            result["cpu_baseline"] = base
            result["per_level_epe_vs_oracle"] = epe_oracle
        else:
            result["cpu_baseline"] = None
    del flows
    if default_run and not args.no_extra:
        # BASELINE configs[3] and configs[4] on the driver's clock too: same procedure, fewer steps
        # (config 3 is ~25 ms per step), their own roofline blocks; never the headline `value`
        extra = []
        for (b, h, w, dt, st, wu) in ((16, 1024, 2048, "f32", 10, 3), (32, 256, 512, "f16", 30, 5)):
            torch.cuda.empty_cache()
            try:
                r, keep = measure(args, b, (h, w), dt, st, wu, world, rank, dev, False, copy_gbs)
                del keep
                extra.append(r)
            except (RuntimeError, ValueError) as e:  # report, do not lose the headline line
                extra.append({"metric": metric_name((h, w), dt), "error": str(e).splitlines()[0],
                              "config": {"workload": baseline_config_name(b, (h, w), dt, world)}})
        result["extra_configs"] = extra
    rc = 0
    if rank == 0 and "per_level_epe_vs_oracle" in result and args.dtype == "f32" and args.matmul == "f32":
        # the number is gated where it is produced: the flows compared are the ones the TIMED hipGraph wrote
        # (B-pair batch, two-stream forward), against the CPU oracle on the first --cpu-pairs pairs
        worst = max(result["per_level_epe_vs_oracle"])
        result["parity_gate"] = {"tolerance_px": PARITY_TOL_PX, "worst_level_epe_px": worst,
                                 "pass": bool(worst < PARITY_TOL_PX)}
        if not worst < PARITY_TOL_PX:   # also catches NaN
            print("bench.py: PARITY FAILURE: per-level EPE vs the oracle {} >= {} px".format(
                result["per_level_epe_vs_oracle"], PARITY_TOL_PX), file=sys.stderr)
            rc = 3
    elif rank == 0:
        # said, not silent: this run did not compare its flows with the oracle (--no-cpu-baseline, N > 1 ranks, fp16 or bf16x3)
        result["parity_gate"] = {"tolerance_px": PARITY_TOL_PX, "pass": None,
                                 "skipped": "no oracle comparison in this run (the default N=1 fp32 run makes it)"}
    if rank == 0:
        result["detail_file"] = None
        if args.detail:
            try:
                with open(args.detail, "w") as f:
                    json.dump(result, f, indent=1)
                result["detail_file"] = os.path.relpath(args.detail, ROOT)
            except OSError as e:
                print("bench.py: could not write {} ({})".format(args.detail, e), file=sys.stderr)
        print(json.dumps(compact_line(result), separators=(",", ":")))
    if world > 1:
        qdist.barrier()
        torch.distributed.destroy_process_group()
    return rc


if __name__ == "__main__":
    sys.exit(main())

#!/usr/bin/env python3
"""Headline benchmark: image-pairs/sec of full 6-level PWC-Net (qpwcnet ``build_flower``)
inference at 256x512 fp32, batch 8 per GPU (BASELINE.json configs[1]; configs[2] = the same
per-GPU work on 8 GPUs), synthetic frames and seeded random-init weights.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one forward pass of one batch already resident in HBM: encoder/decoder/flow
estimator convolutions on PyTorch-ROCm, 5 cost volumes + 4 warps in the HIP kernels,
6 per-level EPE reductions, and (N > 1) one RCCL all-gather of the 6-float EPE vector.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from qpwcnet_amd import dist as qdist  # noqa: E402
from qpwcnet_amd import metrics, ops, synth  # noqa: E402
from qpwcnet_amd.pwcnet import GraphedForward, build_flower  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s copy-measured


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=50)
    p.add_argument("--warmup", type=int, default=10)
    p.add_argument("--batch", type=int, default=8, help="pairs per GPU (weak scaling)")
    p.add_argument("--height", type=int, default=256)
    p.add_argument("--width", type=int, default=512)
    p.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    p.add_argument("--fused", dest="fused", action="store_true", default=False,
                   help="fused warp+cost-volume UpFlow front end")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-overlap", action="store_true",
                   help="decoder and flow chain on one stream (A/B of the two-stream forward)")
    p.add_argument("--inflight", type=int, default=2, help="batches in flight of the serving_throughput leg")
    p.add_argument("--no-inflight", action="store_true",
                   help="skip the extra '2 batches in flight' throughput measurement (N=1 only)")
    p.add_argument("--dtype", default="f32", choices=["f32", "f16"],
                   help="f16 = BASELINE configs[4] (fp16 storage / convs, fp32 accumulate in the hot path)")
    p.add_argument("--dist-backend", default=None,
                   help="rehearsal only: 'gloo' runs N ranks on ONE GPU (EPE gathered through host memory)")
    p.add_argument("--cpu-pairs", type=int, default=8, help="pairs timed on the host for cpu_baseline")
    return p.parse_args()


def cost_volume_bytes(B, H, W, C, esize=4):
    """Algorithmic bytes of one cost-volume launch: read prv + nxt, write 81 channels
    (B*H*W*(2C+81)*e)."""
    return B * H * W * (2 * C + 81) * esize


def device_copy_ceiling(dev, mib=512, reps=10):
    """GB/s (read + write bytes) of a dense device-to-device copy: the achievable HBM ceiling of this
    box that SURVEY 8(d) asks to report beside the 8 TB/s nominal peak."""
    src = torch.empty(mib << 20, dtype=torch.uint8, device=dev).random_(0, 255)
    dst = torch.empty_like(src)
    for _ in range(2):
        dst.copy_(src)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        dst.copy_(src)
    e1.record()
    e1.synchronize()
    return 2.0 * src.numel() * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9


def cpu_baseline(weights, pairs_np, n_pairs, gpu_flows):
    """The reference-algorithm CPU restatement (oracle/net_ref.py; TF2 itself cannot run
    offline) on the host cores, on the first n_pairs of the same workload."""
    from oracle import net_ref, torch_ref
    n_pairs = max(1, min(n_pairs, pairs_np.shape[0]))
    net = net_ref.RefNet(weights)
    cores = torch.get_num_threads()
    net(pairs_np[:1])  # warm-up (thread pool, allocator)
    t0 = time.perf_counter()
    ref = net(pairs_np[:n_pairs])
    dt = time.perf_counter() - t0
    epe = [float(torch_ref.epe_error(a[:n_pairs].float().cpu(), b)) for a, b in zip(gpu_flows, ref)]
    return {
        "value": n_pairs / dt, "unit": "pairs/s", "cores": cores, "kind": "port",
        "sample": "{} pairs of the same batch, full 6-level net, torch-CPU op-for-op restatement "
                  "(81x slice*mul*mean cost volume, gather warp), {:.1f} s".format(n_pairs, dt),
    }, epe


def main():
    args = parse_args()
    world, rank, local_rank = qdist.init(args.dist_backend)
    if args.dist_backend == "gloo":
        local_rank = 0  # rehearsal: every rank shares cuda:0
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus {} needs one process per GPU: launch with "
                             "python -m torch.distributed.run --nproc-per-node {} bench.py ...".format(
                                 args.gpus, args.gpus))
        raise SystemExit("--gpus {} but WORLD_SIZE={}".format(args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU fallback)")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    torch.backends.cudnn.benchmark = True  # MIOpen find mode: pick the fastest conv solver once

    hw = (args.height, args.width)
    B = args.batch
    weights = synth.make_weights(42, hw)
    tdtype = torch.float32 if args.dtype == "f32" else torch.float16
    esize = 4 if args.dtype == "f32" else 2
    model = build_flower(True, hw, "channels_last", weights=weights, device=dev, fused=args.fused,
                         dtype=tdtype)
    if args.no_overlap:
        model.overlap_streams = False
    pairs_np, gt_np = synth.make_frames(B, hw[0], hw[1], seed=1234 + rank)
    pairs = torch.from_numpy(pairs_np).to(dev, tdtype)
    gt = torch.from_numpy(gt_np).to(dev)
    shapes = [(hw[0] >> s, hw[1] >> s) for s in (5, 4, 3, 2, 1, 0)]
    gt_pyr = metrics.multiscale_ground_truth(gt, shapes)

    def forward():
        with torch.no_grad():
            flows = model(pairs)
            return flows, metrics.per_level_epe(gt_pyr, flows)

    # eager warm-up (MIOpen solver search, LDS attribute set-up) before any capture
    for _ in range(3):
        flows, epe_local = forward()
    torch.cuda.synchronize()

    graph = None
    if not args.no_graph:
        try:
            graph = GraphedForward(model, pairs, epilogue=lambda fl: metrics.per_level_epe(gt_pyr, fl), warmup=0)
            flows, epe_local = graph.outputs, graph.extra
        except RuntimeError as e:   # capture refused: keep measuring, with eager launches
            print("bench.py: hipGraph capture failed ({}); eager launches".format(str(e).splitlines()[0]),
                  file=sys.stderr)
            graph = None
            torch.cuda.synchronize()

    # The one collective of the path: an all-gather of the 6 per-level EPE (+ shard weight) per step.
    # On RCCL it is asynchronous: the exchange of step k travels while step k+1 computes, and its result
    # is consumed (the stream waits for it) one step later; the last one is drained inside the timed region.
    # (CPU rehearsal of the N > 1 path, --dist-backend gloo: the same submit/collect pattern on host tensors)
    gloo = args.dist_backend == "gloo"
    gather = qdist.EpeGather(6, "cpu" if gloo else dev, n_local=B)

    def step():
        if graph is not None:
            graph.replay()
            e = epe_local
        else:
            _, e = forward()
        gather.submit(e.cpu() if gloo else e)
        return gather.collect() if gather.outstanding() > 1 else None

    def drain():
        out = None
        while gather.outstanding():
            out = gather.collect()
        return out

    for _ in range(args.warmup):
        step()
    drain()
    qdist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = None
    for _ in range(args.steps):
        res = step() or res
    res = drain() or res
    qdist.barrier()
    torch.cuda.synchronize()
    elapsed = qdist.max_over_ranks(time.perf_counter() - t0, dev)
    per_rank, epe_mean = res

    # ---- serving-style throughput, reported BESIDE the headline (never as `value`): two batches of 8
    # in flight, each a hipGraph replay on its own stream, so that the launch-bound coarse levels of
    # one batch run under the encoder / finest level of the other.  Same K steps, same work per step.
    serving = None
    if graph is not None and world == 1 and not args.no_inflight:
        n_fly = max(2, args.inflight)
        lanes = [(torch.cuda.Stream(), graph)]
        for k in range(1, n_fly):
            pk, _ = synth.make_frames(B, hw[0], hw[1], seed=4321 + k)
            lanes.append((torch.cuda.Stream(), GraphedForward(model, torch.from_numpy(pk).to(dev, tdtype),
                                                              epilogue=lambda fl: metrics.per_level_epe(gt_pyr, fl),
                                                              warmup=0)))
        torch.cuda.synchronize()

        def run(n):
            for i in range(n):
                st, g = lanes[i % n_fly]
                with torch.cuda.stream(st):
                    g.replay()

        run(max(n_fly, args.warmup))
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        run(args.steps)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        serving = {"batches_in_flight": n_fly, "value": B * args.steps / dt, "unit": "pairs/s",
                   "ms_per_step": dt / args.steps * 1e3,
                   "note": "{} hipGraph replays (batch {} each) on as many streams; not the headline value".format(
                       n_fly, B)}
        del lanes[1:]

    # ---- live roofline of the dominant hot-path kernel: HIP events on the launch stream
    # (single stream for this pass: with the decoder running beside it on the side stream the
    # events would time the kernel while it shares the chip)
    copy_gbs = device_copy_ceiling(dev)
    n_prof = max(5, min(args.steps, 20))
    lvl4 = (B, hw[0] // 2, hw[1] // 2, synth.level_channels()[-1])
    dom_name = "warp_cost_volume" if args.fused else "cost_volume"
    dom_key = (dom_name,) + lvl4
    model.overlap_streams = False
    lvl3 = (B, hw[0] // 4, hw[1] // 4, synth.level_channels()[-2])
    with ops.kernel_timing(capture=dom_key) as kt:
        for _ in range(n_prof):
            forward()
    ktimes = kt.summary()
    model.overlap_streams = not args.no_overlap
    _, dom_ms_eager = ktimes[dom_key]
    dom_ms = dom_ms_eager
    def replay_launches(fn, n_rep=50):
        """Average duration of `fn`'s launch: n_rep back-to-back launches captured in one hipGraph and
        replayed between two HIP events (no host launch gaps, whatever the host's speed)."""
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
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            run()
            e1.record()
            e1.synchronize()
            ts.append(e0.elapsed_time(e1) / n_rep)
        return sorted(ts)[len(ts) // 2]   # median round, not the best one

    if kt.captured is not None:
        # The eager pass times event -> (host launch latency) -> kernel -> event.  For the launch
        # DURATION rocprofv3 reports, replay the step's own launch (same inputs, same output pixel stride)
        # back to back: the queue never drains, so (t1 - t0) / n is the kernel time.
        cp, cn, cstride = kt.captured
        cbuf = torch.empty(cp.shape[:3] + (cstride,), dtype=cp.dtype, device=cp.device)
        dom_ms = replay_launches(lambda: ops.cost_volume_into(cp, cn, cbuf, 0))
    dom_bytes = cost_volume_bytes(*lvl4, esize)
    # The same kernel symbol also serves the coarser levels that give it >= 256 regions of 8x8 pixels (L2
    # and L3 at B=8, 256x512; one launch per step each): time those launches the same way, so that the
    # per-level figures compare with rocprofv3's per-grid durations and their mean with its per-symbol
    # AverageNs.
    sym_avg_ms, sym_by_level = None, None
    if kt.captured is not None and not args.fused:
        sym_by_level = {"L4": dom_ms}
        chans = synth.level_channels()
        for lv in (3, 2, 1, 0):
            shp = (B, hw[0] >> (5 - lv), hw[1] >> (5 - lv), chans[lv])
            regions = B * ((shp[1] + 7) // 8) * ((shp[2] + 7) // 8)
            if shp[3] % 32 or regions < 256:
                break
            pl = torch.randn(shp, device=dev, dtype=tdtype)
            nl = torch.randn(shp, device=dev, dtype=tdtype)
            # the step writes 84-channel pixels where the first OptFlow layer is fused, dense 81 elsewhere
            blk = model.flow if lv == 0 else model.upflows[lv - 1]
            stride = kt.captured[2] if blk.flow.wants_cost84(pl) else 81
            bl = torch.empty(shp[:3] + (stride,), dtype=tdtype, device=dev)
            sym_by_level["L%d" % lv] = replay_launches(lambda: ops.cost_volume_into(pl, nl, bl, 0))
        sym_avg_ms = sum(sym_by_level.values()) / len(sym_by_level)
    achieved = dom_bytes / (dom_ms * 1e-3) / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")  # from a separate rocprofv3 --pmc run
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get(dom_name + "_L4_bytes_per_launch")
        except Exception:
            traffic = None
    hot_ms = sum(n * t for (n, t) in ktimes.values()) / n_prof

    result = {
        "metric": "image-pairs/sec at 256x512 fp32",
        "value": world * B * args.steps / elapsed,
        "unit": "pairs/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {
            "workload": "BASELINE configs[1]: full 6-level PWC-Net (qpwcnet build_flower) inference, "
                        "batch {} per GPU, {}x{} {}, d=4 cost volume + WarpV2".format(
                            B, hw[0], hw[1], "fp32" if args.dtype == "f32" else "fp16"),
            "global_batch": world * B, "batch_per_gpu": B,
            "parallelism": "dp{} (pairs sharded, RCCL all-gather of the 6 per-level EPE)".format(world),
            "hipgraph": graph is not None, "fused_upflow": bool(args.fused),
            "hip_optflow": True,
            "weights": "seeded glorot (synth.make_weights(42)), 3.09M params",
        },
        "roofline": {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
            # the ceiling a plain device copy reaches on this box, measured now
            "copy_ceiling_GBs": copy_gbs, "frac_of_copy_ceiling": achieved / copy_gbs,
            "kernel": "{} L4 {}".format(dom_name, "x".join(map(str, lvl4))),
            "algorithmic_bytes_per_launch": dom_bytes, "avg_launch_ms": dom_ms,
            "avg_launch_ms_inside_eager_step": dom_ms_eager,
            "kernel_symbol": "cost_volume_mfma_lds_kernel" if args.dtype == "f32" else "cost_volume_mfma_lds_f16_kernel",
            # compare with rocprofv3: per-grid durations of the symbol / its --stats AverageNs
            "symbol_launch_ms_by_level": sym_by_level, "symbol_avg_ms": sym_avg_ms,
            "launches_timed": ktimes[dom_key][0],
            "out_pixel_stride": kt.captured[2] if kt.captured is not None else None,
            "method": "HIP events on the launch stream around a hipGraph of 50 back-to-back replays of the "
                      "step's own L4 launch (inputs and output pixel stride captured from the forward; "
                      "stride 84 = 81 channels + 3 zeroed pads, the algorithmic bytes count 81); the "
                      "eager-step figure also contains the host launch gap",
        },
        "hot_path": {
            "ms_per_step_eager_events": hot_ms,
            "kernels_ms": {"{} {}".format(k[0], "x".join(map(str, k[1:]))): round(t, 5)
                           for k, (n, t) in sorted(ktimes.items())},
        },
        "per_level_epe_vs_ground_truth": [float(x) for x in epe_mean.cpu()],
        "serving_throughput": serving,
    }
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            base, epe_oracle = cpu_baseline(weights, pairs_np, args.cpu_pairs, flows)
            result["cpu_baseline"] = base
            result["per_level_epe_vs_oracle"] = epe_oracle
        else:
            result["cpu_baseline"] = None
        print(json.dumps(result))
    if world > 1:
        qdist.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

"""Model assembly mirroring qpwcnet/core/pwcnet.py: ``encoder`` (134-168),
``decoder`` (171-207), ``flower`` (28-67), ``build_flower`` (210-244).

Every layer of the forward runs on the hand-written gfx950 kernels of
``csrc/`` (cost volume, warps, SeparableConv2D, encoder / decoder convolutions,
flow heads, EPE) through the C ABI; PyTorch supplies device memory, streams and
hipGraph capture.  The only library launches left are the two wide pointwise
GEMMs (``torch.addmm`` -> hipBLASLt) of the first OptFlow layer at the two
coarsest levels.  Inference only.
"""
import numpy as np
import torch

from . import ops
from .backend import CHANNELS_FIRST, CHANNELS_LAST, get_axis, image_data_format
from .non_layers import (DownConv, Downsample, Flow, Flower, FrameInterpolate, Split, UpConv, UpFlow,
                         Upsample)
from .synth import DEC_FILTERS, ENC_FILTERS, make_interpolator_weights, make_weights


def encoder(layers, img_prv, img_nxt, output_features=False):
    """pwcnet.py:134-168: the same five DownConv blocks applied to both frames."""
    f = img_prv
    feats_prv = [f]
    for l in layers:
        f = l(f)
        feats_prv.append(f)
    f = img_nxt
    feats_nxt = [f]
    for l in layers:
        f = l(f)
        feats_nxt.append(f)
    if output_features:
        return feats_prv, feats_nxt
    return feats_prv[-1], feats_nxt[-1]


def decoder(layers, encs_prv, encs_nxt, axis, use_skip=True):
    """pwcnet.py:171-207: four UpConv blocks, each concatenated with the encoder
    feature of the same resolution."""
    def run(encs):
        f = encs[-1]
        i = -2
        decs = []
        for l in layers:
            f = l(f)
            if use_skip:
                f = torch.cat([f, encs[i]], dim=axis)
                i -= 1
            decs.append(f)
        return decs
    return run(encs_prv), run(encs_nxt)


def flower(flow, upflows, enc_prv, enc_nxt, decs_prv, decs_nxt, data_format,
           output_multiscale=True):
    """pwcnet.py:28-67: coarsest Flow, then per level Upsample(x2, *2) + UpFlow,
    and a final upsample-only full-resolution flow."""
    flo_01 = flow((enc_prv, enc_nxt))
    flos = [flo_01]
    for i in range(len(decs_prv)):
        flo_01_u = Upsample(scale=2.0, data_format=data_format)(flo_01)
        flo_01 = upflows[i]((decs_prv[i], decs_nxt[i], flo_01_u))
        flos.append(flo_01)
    flo_01 = Upsample(scale=2.0, data_format=data_format)(flo_01)
    flos.append(flo_01)
    return flos if output_multiscale else [flo_01]


def interpolator(blocks, img_prv, img_nxt, decs_prv, decs_nxt, flos_01, flos_10, data_format,
                 output_multiscale=True):
    """pwcnet.py:70-131: frame-interpolation stack.  `blocks`: the n+1 FrameInterpolate functors
    (img_0 ... img_n).  The coarsest image comes from the twice-per-level average-pooled input
    frames, every finer one from the decoder features of that level plus the upsampled previous
    image; the full-resolution output is upsample-only (:123-124)."""
    n = len(decs_prv)
    pool = Downsample(data_format=data_format)
    up = Upsample(scale=1.0, data_format=data_format)
    imgs_prv, imgs_nxt = [img_prv], [img_nxt]
    for _ in range(n + 1):                                   # :87-90
        imgs_prv.append(pool(imgs_prv[-1]))
        imgs_nxt.append(pool(imgs_nxt[-1]))
    img = blocks[0]((imgs_prv[-1], imgs_nxt[-1], flos_01[0], flos_10[0]))   # :101-102
    imgs = [img]
    for i in range(n):                                       # :107-121
        img_u = up(img)
        img = blocks[i + 1]((decs_prv[i], decs_nxt[i], flos_01[i + 1], flos_10[i + 1], img_u))
        imgs.append(img)
    imgs.append(up(img))                                     # :124
    return imgs if output_multiscale else imgs[-1]


class QpwcNet:
    """What ``build_flower`` returns: ``model(inputs)`` / ``model.predict(inputs)``
    with inputs (B,H,W,6) ('channels_last') or (B,6,H,W); returns the list of 6
    multi-scale flows when ``train`` (pwcnet.py:237-239) else the final flow only."""

    def __init__(self, weights, train=True, input_shape=(256, 512), data_format=None,
                 use_tfa=True, device="cuda", dtype=torch.float32, fused=None, hip_optflow=True,
                 batch_frames=True, overlap_streams=True, internal_channels_last=True):
        self.data_format = image_data_format() if data_format is None else data_format
        self.axis = get_axis(self.data_format)
        self.train = train
        self.batch_frames = bool(batch_frames)
        self.overlap_streams = bool(overlap_streams)
        self._side = None
        self._sides = []
        # side stream of each decoder level in the two-stream forward (see _forward_two_streams)
        self.dec_stream_of = (0, 0, 0, 0)
        # round 4 (tools/capture_crosswait.py): the two suspects of the capture SIGSEGV as switches
        self.allow_returning_dec_streams = False    # a mapping like (0,1,0,1): two side streams waiting on each other
        self.record_stream_under_capture = True     # Tensor.record_stream on private-pool tensors while capturing
        # the skip halves of the decoder's concat buffers copied under the encoder (_prefill): measured SLOWER -- 1.230 vs 1.133
        # ms/step (config 5: 1.781 vs 1.645; config 4: 30.0 vs 29.8): a fork that early turns the encoder's chain into one branch
        # of a two-branch graph for its whole length (tools/step_time.py prefill_skips=True).  Off.
        self.prefill_skips = False
        self.skip_copy_first = ()    # see _forward_two_streams: (3, 2) 1.140-1.145, (3,) 1.138-1.140, (3, 2, 1, 0) 1.142-1.145 vs () 1.130-1.133 ms/step
        self._prefilled = {}
        # launches per decoder level on the side stream (slices of the 2B stacked frames), see _forward_two_streams
        # round 3 (tools/step_time.py "dec_chunks=...", three interleaved runs each in one call, ms/step): (2,4,4,4)
        # 1.2147, (2,4,4,2) 1.1996, (2,4,4,1) 1.1991, (2,4,2,2) 1.2005, (2,2,4,1) 1.2031, (2,4,8,8) 1.296: the finest
        # decoder level runs beside flow level 3, whose kernels fill the chip themselves -- one efficient launch that is
        # over sooner disturbs them less than four that trickle
        self.dec_chunks = (2, 4, 4, 1)
        # stacked frames (2B) up to which dec_chunks applies; beyond: one launch per level.  Config 5 (B=32: 64 frames),
        # tools/step_time.py --batch=32 --dtype=f16, two runs each in one call: one launch per level 1.7199 / 1.7222 ms,
        # (2,4,4,1) 1.7092 / 1.7005, (2,4,4,2) 1.7066 / 1.7072, (1,1,2,1) 1.7205 / 1.7191
        self.dec_chunk_max_frames = 64
        # ... and input pixels (frames x H x W) up to which it applies: at config 4's size (32 frames of 1024x2048) every
        # launch is many rounds of workgroups anyway and one launch per level is 0.4 % faster (29.89 / 30.00 vs 30.00 / 30.16 ms)
        self.dec_chunk_max_pixels = 64 * 256 * 512
        # launch order of flow levels (F) and decoder levels (D) in the two-stream forward, see _forward_two_streams
        self.capture_order = ("F0", "D0", "D1", "D2", "D3", "F1", "F2", "F3", "F4")
        self.dec_after_flow = {}     # {decoder level: flow level it waits for}; see _forward_two_streams
        self._matmul = "f32"
        self.skip_redundant_join = True   # see the end of _forward_two_streams
        self.input_shape = tuple(input_shape)
        self.device = torch.device(device)
        self.dtype = dtype
        # 'channels_first' (the reference's inference default, app/optical_flow/test_infer.py:52) on a HIP
        # device: inputs and outputs are (B,C,H,W), everything in between runs on the channels-last kernels.
        # The first encoder kernel reads the six input planes itself, every flow_head writes its (B,2,h,w)
        # output itself and the Upsample kernel reads / writes planes, so no transposition launch is added.
        self._df = CHANNELS_LAST if (self.data_format == CHANNELS_FIRST and self.batch_frames and hip_optflow and
                                     internal_channels_last and self.device.type == "cuda") else self.data_format
        self.params = {}
        for k, v in weights.items():
            t = torch.as_tensor(np.asarray(v)).to(self.device, dtype)
            if t.dim() == 4 and self._df == CHANNELS_LAST:
                t = t.contiguous(memory_format=torch.channels_last)
            self.params[k] = t
        df = self._df
        self.split = Split(2, axis=self.axis, data_format=df)
        self.enc = [DownConv(self.params, "enc.{}.".format(i), data_format=df)
                    for i in range(len(ENC_FILTERS))]
        self.dec = [UpConv(self.params, "dec.{}.".format(i), data_format=df)
                    for i in range(len(DEC_FILTERS))]
        self.flow = Flow(self.params, "flow.", use_tfa=use_tfa, hip_optflow=hip_optflow, data_format=df)
        self.upflows = [UpFlow(self.params, "upflow.{}.".format(i), use_tfa=use_tfa, fused=fused,
                               hip_optflow=hip_optflow, data_format=df)
                        for i in range(len(DEC_FILTERS))]
        if self._df != self.data_format:
            for blk in [self.flow] + self.upflows:
                blk.flow.out_format = self.data_format   # flow_head writes the (B,2,h,w) output itself
        self.fuse_flow_upsample = True

    @property
    def fuse_flow_upsample(self):
        """Flow head + the x2 upsampling of its flow in one launch (qpwc_flow_head_up_fwd) on the levels that run the
        separate flow-head kernel (round 4)."""
        return self._fuse_flow_upsample

    @fuse_flow_upsample.setter
    def fuse_flow_upsample(self, v):
        self._fuse_flow_upsample = bool(v)
        for blk in [self.flow] + self.upflows:
            blk.flow.fuse_upsample = self._fuse_flow_upsample and self._df == self.data_format == CHANNELS_LAST

    @property
    def matmul(self):
        """Arithmetic of the fp32 matrix products in the convolution kernels: "f32" (fp32 matrix instructions) or
        "bf16x3" (three-way bf16 splits on the bf16 matrix instructions, csrc/split_bf16.h)."""
        return self._matmul

    @matmul.setter
    def matmul(self, v):
        if v not in ("f32", "bf16x3"):
            raise ValueError("matmul must be 'f32' or 'bf16x3', got {!r}".format(v))
        self._matmul = v
        for layer in self.enc + self.dec:
            layer.matmul = v
        for blk in [self.flow] + self.upflows:
            blk.flow.matmul = v

    def _up(self, flo, last=False):
        """Upsample(scale=2.0) between levels (pwcnet.py:55,60).  A channels_first model on the
        channels-last kernels: the flow comes in the declared layout (flow_head wrote it so), the
        next level wants it channels-last, the last one is an output again."""
        up = getattr(flo, "_qpwc_up2", None)
        if up is not None and self._df == self.data_format == CHANNELS_LAST:
            return up       # written by the flow head's own launch (OptFlow.fuse_upsample, qpwc_flow_head_up_fwd)
        if self._df != self.data_format and flo.is_cuda and flo.shape[1] == 2:
            return ops.upsample2x_flow(flo, 2.0, in_format=self.data_format,
                                       out_format=self.data_format if last else self._df)
        return Upsample(scale=2.0, data_format=self.data_format)(flo)

    def __call__(self, inputs):
        exp = (self.input_shape + (6,)) if self.data_format == CHANNELS_LAST \
            else ((6,) + self.input_shape)
        if tuple(inputs.shape[1:]) != exp:
            raise ValueError("expected input shape (B,)+{}, got {}".format(exp, tuple(inputs.shape)))
        if self.batch_frames:
            nb = inputs.shape[0]
            overlap = self.overlap_streams and inputs.is_cuda
            self._prefilled = {}
            encs = self._encode_stacked(inputs, prefill=overlap and self.prefill_skips)
            try:
                return self._forward_stacked(encs, nb, overlap)
            finally:
                self._prefilled = {}
        img_prv, img_nxt = self.split(inputs)
        encs_prv, encs_nxt = encoder(self.enc, img_prv, img_nxt, True)
        decs_prv, decs_nxt = decoder(self.dec, encs_prv, encs_nxt, self.axis, True)
        outs = flower(self.flow, self.upflows, encs_prv[-1], encs_nxt[-1], decs_prv, decs_nxt,
                      self.data_format, output_multiscale=self.train)
        return outs if self.train else outs[0]

    def _prefill(self, li, f):
        """Encoder level li (li < 4) is the skip of decoder level 3 - li: allocate that level's concat buffer and copy the
        skip half NOW, on the decoder's side stream, under the encoder's remaining levels (whose kernels leave the memory
        system idle) instead of beside the coarse flow levels, where both queues are full (round 4: the four copies are
        41 us of the second queue's ~310 us in that phase).  The side stream only ever waits for the caller's stream here."""
        i = len(self.dec) - 1 - li
        if i < 0 or self.dec_stream_of[i] != 0:
            return
        main = torch.cuda.current_stream()
        if self._side is None:
            self._side = torch.cuda.Stream(device=f.device)
        if not self._sides:
            self._sides.append(self._side)
        side = self._sides[0]
        c_in = ENC_FILTERS[-1] if i == 0 else DEC_FILTERS[i - 1] + ENC_FILTERS[-1 - i]   # channels of decoder level i's input
        ev = torch.cuda.Event()
        ev.record(main)
        side.wait_event(ev)
        with torch.cuda.stream(side):
            buf = self.dec[i].prefill_skip(f, c_in)
        if buf is not None:
            if self.record_stream_under_capture or not torch.cuda.is_current_stream_capturing():
                f.record_stream(side)
            self._prefilled[i] = buf

    def _encode_stacked(self, inputs, prefill=False):
        """The encoder/decoder weights are shared by both frames (pwcnet.py:145-162, 179-206): run
        the encoder once on the 2B stacked frames [prv; nxt] -> [frames, enc_0 .. enc_4]."""
        h, w = self.input_shape
        first = self.enc[0].first_layer(inputs, self.data_format)
        split_axis = self.axis
        if first is None and self._df != self.data_format:
            inputs = ops.layout_transpose(inputs, self._df)      # (no first-layer kernel for this input: odd sizes)
            split_axis = get_axis(self._df)                      # the six channels moved with the transposition
        if first is not None:
            # the frames themselves (entry 0 of the reference's feature lists) are not used downstream
            f, padded = inputs, None
        elif (self._df == CHANNELS_LAST and inputs.is_cuda and h % 2 == 0 and w % 2 == 0 and
                inputs.dtype in (torch.float32, torch.float16)):
            # split + stack + 'SAME' padding of the first stride-2 conv in one pass
            padded = ops.split_frames_pad(inputs, 1, 1)
            f = padded[:, :h, :w, :]
        else:
            f, padded = torch.cat(torch.chunk(inputs, 2, dim=split_axis), dim=0), None
        encs = [f]
        for li, l in enumerate(self.enc):
            # the activation epilogue of level li lays down the 'SAME' padding level li+1 needs
            f, padded = l.forward_padded(f, padded, want_padded=li + 1 < len(self.enc),
                                         after_a=first if li == 0 else None)
            encs.append(f)
            if prefill and li + 1 < len(self.enc):
                self._prefill(li, f)
        return encs

    def _forward_stacked(self, encs, nb, overlap):
        """Decoder + flow chain on the stacked encoder outputs.  overlap: the two-stream form."""
        if overlap:
            return self._forward_two_streams(encs, nb)
        f, decs, i = encs[-1], [], -2
        for l in self.dec:
            f = l.cat_skip(f, encs[i])
            i -= 1
            decs.append(f)
        flo = self.flow((encs[-1][:nb], encs[-1][nb:]))
        flos = [flo]
        for i, upflow in enumerate(self.upflows):
            flo = upflow((decs[i][:nb], decs[i][nb:], self._up(flo)))
            flos.append(flo)
        flos.append(self._up(flo, last=True))
        return flos if self.train else flos[-1]

    def _forward_two_streams(self, encs, nb):
        """Decoder chain on a side stream, flow chain on the caller's stream: the coarse
        levels' launches are far too small to fill 256 CUs one at a time, and the decoder
        of level i+1 does not depend on the flow of level i (pwcnet.py:179-206 vs 39-57).
        Level i's UpFlow waits on an event recorded after decoder i.  Captured by hipGraph
        as two parallel branches."""
        main = torch.cuda.current_stream()
        if self._side is None:
            self._side = torch.cuda.Stream(device=encs[-1].device)
        # decoder level i runs on side stream dec_stream_of[i] (default: one side stream for the whole decoder).
        # Round 3, tools/dec_streams_ab.py (one mapping per process): (0,1,1,1) 1.216 vs 1.221 ms/step, (0,1,2,2) 1.223,
        # (0,1,2,3) 1.223 -- level 1's late release (see below) does not come from what is queued behind decoder level
        # 0 on ITS stream; a mapping that RETURNS to an earlier stream, e.g. (0,1,0,1), makes two side streams wait on
        # each other alternately and the runtime dies with SIGSEGV while capturing / instantiating the graph -- the
        # round-1 "four staggered streams" crash.  Unsupported by this build: refused here.
        if (any(b < a for a, b in zip(self.dec_stream_of, self.dec_stream_of[1:])) and
                not self.allow_returning_dec_streams) or min(self.dec_stream_of) < 0:
            raise ValueError("dec_stream_of must be non-decreasing side-stream indices, got {}".format(self.dec_stream_of))
        # tensors allocated under capture come from the graph's private pool and stay referenced (decs) until every
        # stream has been joined: the graph's own edges order their reuse, the allocator needs no cross-stream note
        note_streams = self.record_stream_under_capture or not torch.cuda.is_current_stream_capturing()
        n_side = max(self.dec_stream_of) + 1
        while len(self._sides) < n_side:
            self._sides.append(self._side if not self._sides else torch.cuda.Stream(device=encs[-1].device))
        sides = self._sides[:n_side]
        for sd in sides:
            sd.wait_stream(main)            # encoder outputs are ready
        # One library launch over the 2B stacked frames fills the chip with long-lived workgroups and the small
        # critical-path kernels beside it wait for CUs; two launches of B frames each leave room for them
        # (B=8: 1.343 -> 1.309 ms/step; 4 chunks: no further gain; B=32: the launches are multi-round anyway, -1 %)
        # (own transposed-convolution kernel: four launches over quarters of the batch, 256 short-lived workgroups
        # each = one per CU: 1.308 -> 1.257 ms/step; 8 / 16 launches: 1.43 / 1.95; two launches for the coarsest
        # level, whose quarter launches are only 128 workgroups: another -0.7 %)
        small = (encs[-1].shape[0] <= self.dec_chunk_max_frames and
                 encs[-1].shape[0] * self.input_shape[0] * self.input_shape[1] <= self.dec_chunk_max_pixels)
        chunks = 2 if small else 1
        # The coarsest flow block first, THEN the decoder launches: in the captured graph the first node
        # created after the fork stays on the encoder's hardware queue and the other branch starts on a second
        # queue some 50-100 us later (kernel trace of the replay: with the decoder captured first, the flow
        # chain -- the critical path -- was the branch that waited).  The decoder has that much slack.
        # Launch (= capture) order of the two branches: "F<i>" = flow level i on the caller's stream, "D<i>" =
        # decoder level i on the side stream.  F0 first (see above); then the whole decoder, then F1..F4.
        # A node that waits on the other queue is released late when much of that queue's work was enqueued
        # before it: in this order level 1's first launch starts 35-45 us after decoder level 0 has finished
        # (kernel traces: it follows the START of the last launch of decoder level 1 wherever that is moved).
        # Interleaving (F0 D0 F1 D1 F2 D2 F3 D3 F4) removes that wait and delays every later decoder level by
        # more: 1.253 vs 1.197-1.207 ms/step; five orders in between all land within 1 % of this one
        # (tools/capture_order_ab.py, one call; dropping the wait altogether -- a race -- would be worth 12-18 us).
        # Round 3, "M<i>" = decoder level i on the caller's stream, one launch (tools/capture_order_ab.py main, one
        # call, ms/step): default 1.2305; M0 F0 D1 D2 D3 F1.. 1.2290; M0 F0 D1 F1 D2 D3 F2.. 1.2237; F0 M0 D1.. 1.2608;
        # M0 F0 M1 F1 D2.. 1.2527.  Kernel trace of the best one: level 1 then starts without a gap, but level 0 starts
        # 31 us later (decoder level 0 in front of it) and level 2 takes 115 instead of 98 us beside decoder levels 2
        # and 3 -- level 3 begins at 720 instead of 705 us of a forward either way.  The two chains of this phase
        # (decoder levels 0-2: ~200 us alone; flow levels 0-2: ~200 us alone) share the chip for ~300 us whatever the
        # order; the default order stays.
        order = self.capture_order
        # Round 4: the skip halves of the finest decoder levels' concat buffers are copied FIRST on the side stream -- beside
        # the coarsest flow level, whose launches leave the chip almost empty -- instead of after their level's transposed
        # convolution, beside flow levels 2 and 3 where both queues are full (skip_copy_first: decoder levels, in this order).
        # Measured slower in every order tried (the decoder's first level is then late for flow level 1): off.
        for i in self.skip_copy_first:
            if i in self._prefilled or not 0 <= i < len(self.dec) or ("D%d" % i) not in order:
                continue
            side = sides[self.dec_stream_of[i]]
            if side is not sides[0]:
                continue
            c_in = ENC_FILTERS[-1] if i == 0 else DEC_FILTERS[i - 1] + ENC_FILTERS[-1 - i]
            with torch.cuda.stream(side):
                buf = self.dec[i].prefill_skip(encs[-2 - i], c_in)
            if buf is not None:
                self._prefilled[i] = buf
        ready, decs, flos, ran_on, flow_done = {}, {}, [], {}, {}
        waited_on_main = set()      # decoder levels whose `ready` event the caller's stream has waited for
        f, k, flo = encs[-1], -2, None
        for tok in order:
            i = int(tok[1:])
            if tok[0] in "DM":
                # "D<i>": decoder level i on its side stream; "M<i>": on the caller's stream (one launch over the whole
                # batch: a level the next flow level needs at once then costs no cross-queue wait)
                side = main if tok[0] == "M" else sides[self.dec_stream_of[i]]
                if i > 0 and ran_on[i - 1] is not side:
                    side.wait_event(ready[i - 1])       # the previous decoder level ran on another stream
                # dec_after_flow (A/B, round 4): decoder level i is held back until flow level j has finished (j must come
                # before D<i> in capture_order) -- to choose WHICH flow-chain kernels the level's launches share the chip with
                j_hold = self.dec_after_flow.get(i)
                if j_hold is not None and side is not main:
                    if j_hold not in flow_done:
                        raise ValueError("dec_after_flow: F{} must precede D{} in capture_order".format(j_hold, i))
                    side.wait_event(flow_done[j_hold])
                with torch.cuda.stream(side):
                    hip_chunks = (self.dec_chunks[i] if small else 1) if tok[0] == "D" else 1
                    f = self.dec[i].cat_skip(f, encs[k], batch_chunks=chunks if tok[0] == "D" else 1, hip_chunks=hip_chunks,
                                             buf=self._prefilled.get(i) if side is sides[0] else None)
                    k -= 1
                    # allocated on `side`, read by UpFlow on `main`: tell the caching allocator, so that the block
                    # is not handed to a later side-stream allocation while main may still be reading it (the
                    # join at the end orders main after side, not side's NEXT use after main's reads)
                    for sd in ([main] + sides) if note_streams else ():
                        if sd is not side:
                            f.record_stream(sd)
                    decs[i] = f
                    ready[i] = torch.cuda.Event()
                    ready[i].record(side)
                    ran_on[i] = side
            elif i == 0:
                flo = self.flow((encs[-1][:nb], encs[-1][nb:]))
                flos.append(flo)
            else:
                flo_u = self._up(flo)
                if ran_on[i - 1] is not main:
                    main.wait_event(ready[i - 1])
                    waited_on_main.add(i - 1)
                flo = self.upflows[i - 1]((decs[i - 1][:nb], decs[i - 1][nb:], flo_u))
                flos.append(flo)
            if tok[0] == "F" and i in self.dec_after_flow.values():
                flow_done[i] = torch.cuda.Event()
                flow_done[i].record(main)
        flos.append(self._up(flo, last=True))
        # Join before anything is freed or returned.  A side stream whose LAST operation main has already waited for
        # (ready[i] of the last decoder level it ran, consumed by flow level i + 1) is joined already: another wait on
        # it would be one more cross-queue barrier in front of whatever the caller launches next (5-6 us in front of
        # the EPE reduction in the captured step) for no ordering it adds.
        for sd in sides:
            last = max((i for i, st in ran_on.items() if st is sd), default=None)
            if not (self.skip_redundant_join and last is not None and last in waited_on_main):
                main.wait_stream(sd)
        return flos if self.train else flos[-1]

    @torch.no_grad()
    def predict(self, inputs):
        if isinstance(inputs, np.ndarray):
            inputs = torch.from_numpy(inputs)
        return self(inputs.to(self.device, self.dtype))


class GraphedForward:
    """hipGraph capture of one model forward (plus an optional epilogue on its outputs) at a fixed
    input shape: `replay(new_inputs)` copies into the static input and relaunches the captured
    kernels -- the launch-bound inner loop of inference (160 -> 1 host launches per batch)."""

    def __init__(self, model, example_inputs, epilogue=None, warmup=3, share_with=None):
        """share_with: another GraphedForward whose static input and memory pool this one reuses (same
        model, same shapes, another epilogue) -- the two must then be replayed alternately in capture
        order, never concurrently (bench.py, N > 1: the two graphs differ only in the all-gather payload
        slot their EPE reduction writes)."""
        self.model = model
        self.static_in = example_inputs.clone() if share_with is None else share_with.static_in

        def run():
            with torch.no_grad():
                out = model(self.static_in)
                return out, (epilogue(out) if epilogue is not None else None)

        for _ in range(warmup):  # library solver search and lazy initialisation happen eagerly
            run()
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            run()
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        pool = None if share_with is None else share_with.graph.pool()
        with torch.cuda.graph(self.graph, pool=pool, capture_error_mode="thread_local"):
            self.outputs, self.extra = run()

    def replay(self, inputs=None):
        if inputs is not None:
            self.static_in.copy_(inputs)
        self.graph.replay()
        return self.outputs, self.extra


class QpwcInterpolator(QpwcNet):
    """What ``build_interpolator`` returns (pwcnet.py:247-281): encoder/decoder as in
    ``build_flower``, the Flower block run in both directions with shared weights --
    ``flows_01 = flower(enc_nxt, enc_prv, decs_nxt, decs_prv)``, ``flows_10`` with the roles
    swapped (:271-278, argument order as written there) -- then the interpolation stack.
    Returns the list of n+2 images (coarse to fine, last = full resolution) or only the last."""

    def __init__(self, weights, input_shape=(256, 512), data_format=None, use_tfa=True,
                 output_multiscale=True, device="cuda", dtype=torch.float32, hip_optflow=True,
                 batch_directions=True):
        super().__init__(weights, train=True, input_shape=input_shape, data_format=data_format,
                         use_tfa=use_tfa, device=device, dtype=dtype, hip_optflow=hip_optflow,
                         batch_frames=True, overlap_streams=False, internal_channels_last=False)
        df = self.data_format
        self.output_multiscale = bool(output_multiscale)
        self.batch_directions = bool(batch_directions)
        self.flower_block = Flower(self.params, len(DEC_FILTERS), output_multiscale=True, use_tfa=use_tfa,
                                   hip_optflow=hip_optflow, data_format=df)
        self.img_blocks = [FrameInterpolate(self.params, "img.{}.".format(k), up=k > 0, data_format=df)
                           for k in range(len(DEC_FILTERS) + 1)]

    def __call__(self, inputs):
        exp = (self.input_shape + (6,)) if self.data_format == CHANNELS_LAST \
            else ((6,) + self.input_shape)
        if tuple(inputs.shape[1:]) != exp:
            raise ValueError("expected input shape (B,)+{}, got {}".format(exp, tuple(inputs.shape)))
        img_prv, img_nxt = self.split(inputs)
        nb = inputs.shape[0]
        # shared encoder/decoder weights: both frames as one batch of 2B (as in QpwcNet)
        encs = self._encode_stacked(inputs)
        f = encs[-1]
        decs, i = [], -2
        for l in self.dec:
            f = l.cat_skip(f, encs[i])
            i -= 1
            decs.append(f)
        if self.batch_directions and self.data_format == CHANNELS_LAST:
            return self._both_directions_batched(img_prv, img_nxt, encs[-1], decs, nb)
        enc_prv, enc_nxt = encs[-1][:nb], encs[-1][nb:]
        decs_prv, decs_nxt = [d[:nb] for d in decs], [d[nb:] for d in decs]
        flows_01 = self.flower_block((enc_nxt, enc_prv, decs_nxt, decs_prv))
        flows_10 = self.flower_block((enc_prv, enc_nxt, decs_prv, decs_nxt))
        return interpolator(self.img_blocks, img_prv, img_nxt, decs_prv, decs_nxt, flows_01, flows_10,
                            self.data_format, self.output_multiscale)


    def _both_directions_batched(self, img_prv, img_nxt, enc, decs, nb):
        """The two Flower passes (shared weights, roles swapped: pwcnet.py:271-278) as ONE pass over
        2*nb pairs -- pair b is (prv-role = nxt_b, nxt-role = prv_b), pair nb+b the opposite -- and
        the two half-flow warps of every FrameInterpolate block as one launch.  Same arithmetic per
        pair; half the launches on the launch-bound coarse levels."""
        def swap(x):
            return torch.cat([x[nb:], x[:nb]], dim=0)
        enc_s, decs_s = swap(enc), [swap(d) for d in decs]
        flows = self.flower_block((enc_s, enc, decs_s, decs))     # [flows_01; flows_10] per level
        n = len(decs)
        pool = Downsample(data_format=self.data_format)
        up = Upsample(scale=1.0, data_format=self.data_format)
        small = torch.cat([img_nxt, img_prv], dim=0)              # [nxt; prv], pwcnet.py:87-90
        for _ in range(n + 1):
            small = pool(small)
        img = self.img_blocks[0].call_stacked(small, flows[0], nb)
        imgs = [img]
        for i in range(n):
            img = self.img_blocks[i + 1].call_stacked(decs_s[i], flows[i + 1], nb, up(img))
            imgs.append(img)
        imgs.append(up(img))
        return imgs if self.output_multiscale else imgs[-1]


def build_interpolator(input_shape=(256, 512), data_format=None, use_tfa=True, weights=None,
                       output_multiscale=True, device="cuda", dtype=torch.float32, hip_optflow=True,
                       batch_directions=True):
    """pwcnet.py:247-281.  ``weights``: flat dict from ``synth.make_interpolator_weights``."""
    if weights is None:
        weights = make_interpolator_weights(42, input_shape)
    return QpwcInterpolator(weights, input_shape=input_shape, data_format=data_format, use_tfa=use_tfa,
                            output_multiscale=output_multiscale, device=device, dtype=dtype,
                            hip_optflow=hip_optflow, batch_directions=batch_directions)


def build_flower(train=True, input_shape=(256, 512), data_format=None, use_tfa=True,
                 weights=None, device="cuda", dtype=torch.float32, fused=None, hip_optflow=True):
    """pwcnet.py:210-244.  ``weights``: flat dict from ``synth.make_weights`` (default:
    seed 42), since no checkpoint ships with the reference."""
    if weights is None:
        weights = make_weights(42, input_shape)
    return QpwcNet(weights, train=train, input_shape=input_shape, data_format=data_format,
                   use_tfa=use_tfa, device=device, dtype=dtype, fused=fused, hip_optflow=hip_optflow)

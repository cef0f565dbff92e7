#!/usr/bin/env python3
"""Which layout / solver is fastest for the encoder-decoder convolutions?  Dev tool."""
import torch
import torch.nn.functional as F

dev = "cuda:0"
torch.backends.cudnn.benchmark = True
shapes = [  # (name, B, Cin, H, W, Cout, k, stride, transposed)
    ("enc0.a", 16, 3, 256, 512, 16, 3, 2, False),
    ("enc0.b", 16, 16, 128, 256, 16, 3, 1, False),
    ("enc1.a", 16, 16, 128, 256, 32, 3, 2, False),
    ("enc1.b", 16, 32, 64, 128, 32, 3, 1, False),
    ("enc2.b", 16, 64, 32, 64, 64, 3, 1, False),
    ("enc3.b", 16, 128, 16, 32, 128, 3, 1, False),
    ("enc4.b", 16, 256, 8, 16, 256, 3, 1, False),
    ("dec0", 16, 256, 8, 16, 128, 4, 2, True),
    ("dec3", 16, 64, 64, 128, 16, 4, 2, True),
]


def timeit(fn, n=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, B, ci, H, W, co, k, s, tr in shapes:
    res = []
    for fmt in (torch.channels_last, torch.contiguous_format):
        for dt in (torch.float32, torch.float16):
            x = torch.randn(B, ci, H, W, device=dev, dtype=dt).contiguous(memory_format=fmt)
            if tr:
                w = torch.randn(ci, co, k, k, device=dev, dtype=dt).contiguous(memory_format=fmt)
                fn = lambda: F.conv_transpose2d(x, w, None, stride=2, padding=1)
            else:
                w = torch.randn(co, ci, k, k, device=dev, dtype=dt).contiguous(memory_format=fmt)
                xp = F.pad(x, (0, 1, 0, 1)) if s == 2 else x
                fn = (lambda: F.conv2d(xp, w, None, stride=2)) if s == 2 else (lambda: F.conv2d(x, w, None, padding=1))
            try:
                res.append("%s/%s %7.1f us" % ("NHWC" if fmt == torch.channels_last else "NCHW", str(dt)[6:], timeit(fn)))
            except Exception as e:
                res.append("ERR %s" % type(e).__name__)
    print("%-7s" % name, " | ".join(res))

import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from qpwcnet_amd import ops
DEV = "cuda:0"
g = torch.Generator(device=DEV).manual_seed(0)
B, H, W = 7, 100, 200
chans, F = (84, 32, 2), int(sys.argv[1]) if len(sys.argv) > 1 else 128
C = sum(chans)
srcs = [torch.randn(B, H, W, c, device=DEV, generator=g) for c in chans]
dw = torch.randn(C, 9, device=DEV, generator=g)
pw = torch.randn(F, C, device=DEV, generator=g) / C ** 0.5
bias = torch.randn(F, device=DEV, generator=g)
pwp = ops.pad_pointwise(pw)
out = ops.sepconv3x3(srcs, dw, pwp, bias)
ref = torch.cat([ops.sepconv3x3([s[b:b + 1].contiguous() for s in srcs], dw, pwp, bias) for b in range(B)])
bad = (out != ref)
print("mismatched elements", int(bad.sum()), "of", bad.numel())
idx = bad.nonzero().cpu().numpy()
if len(idx):
    b, y, x, c = idx.T
    print("by ft (c//16):", np.bincount(c // 16, minlength=F // 16))
    print("by g (c%16//4):", np.bincount(c % 16 // 4, minlength=4))
    print("by in-tile row (y%8):", np.bincount(y % 8, minlength=8))
    print("by in-tile col (x%16):", np.bincount(x % 16, minlength=16))
    tiles_x, tiles_y = (W + 15) // 16, (H + 7) // 8
    tile = (b * tiles_y + y // 8) * tiles_x + x // 16
    ut = np.unique(tile)
    print("distinct bad tiles:", len(ut), "of", B * tiles_x * tiles_y, "first:", ut[:20])
    print("bad pixels per bad tile (hist):", np.bincount(np.bincount(tile)[ut] // 1)[:5] if False else np.unique(np.bincount(tile)[ut], return_counts=True))
    print("by image:", np.bincount(b, minlength=B))
    d = (out - ref).abs()
    print("max abs diff", float(d.max()))

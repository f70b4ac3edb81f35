#!/usr/bin/env python3
"""The six matcher calls of a cfg-2 pass (top / mid x level 1 / level 2 / global), each `--iters` times on frame-correlated
fp16 tokens: a fixed mix to run under `rocprofv3 --kernel-trace --stats` when tuning the matcher's side kernels."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from vidtome_amd import _lib  # noqa: E402

SHAPES = {"top_l1": (2, 49152, 16384, 320), "top_l2": (2, 12288, 28672, 320), "top_g": (2, 34816, 34816, 320),
          "mid_l1": (2, 12288, 4096, 640), "mid_l2": (2, 3072, 7168, 640), "mid_g": (2, 8704, 8704, 640)}
ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=5)
a = ap.parse_args()
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
for name, (B, Ns, Nd, C) in SHAPES.items():
    base = torch.randn(B, 1024, C, generator=g, device=dev)
    x = (base[:, torch.arange(Ns + Nd, device=dev) % 1024] + 0.5 * torch.randn(B, Ns + Nd, C, generator=g, device=dev)).half()
    ra = torch.arange(Ns, dtype=torch.int32, device=dev).expand(B, Ns).contiguous()
    rb = torch.arange(Ns, Ns + Nd, dtype=torch.int32, device=dev).expand(B, Nd).contiguous()
    for _ in range(a.iters + 1):
        best = _lib.match_filtered(x, None, ra, rb, False)
        _lib.sort_desc(best)
torch.cuda.synchronize()

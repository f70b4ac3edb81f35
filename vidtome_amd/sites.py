"""SD transformer-block *sites* for the hot-path harness (bench.py, tests).

Diffusers and model weights are not available offline, so the measured unit of work is one
pass over the UNet's transformer-block sites executing the patched self-attention segment
``norm1 -> compute_merge -> attn1 -> unmerge -> + residual`` (vidtome/patch.py:139-169) with synthetic
hidden states and random-init weights of the real shapes.  A site is a module with exactly the attributes
the patched block reads (norm1, attn1.{to_q,to_k,to_v,to_out,heads,scale}); it is patched by
``vidtome_amd.apply_patch`` like a Diffusers ``BasicTransformerBlock`` (the class is *named* so).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Tuple

import torch


@dataclass(frozen=True)
class Site:
    name: str
    downsample: int      # tokens per frame = (H/8/downsample) * (W/8/downsample)
    channels: int
    heads: int


def sd15_sites() -> List[Site]:
    """SD-1.5 UNet: 16 transformer blocks (down0 x2, down1 x2, down2 x2, mid, up1 x3, up2 x3, up3 x3);
    heads = 8 everywhere (head dims 40 / 80 / 160)."""
    s = []
    s += [Site(f"down0.{i}", 1, 320, 8) for i in range(2)]
    s += [Site(f"down1.{i}", 2, 640, 8) for i in range(2)]
    s += [Site(f"down2.{i}", 4, 1280, 8) for i in range(2)]
    s += [Site("mid", 8, 1280, 8)]
    s += [Site(f"up1.{i}", 4, 1280, 8) for i in range(3)]
    s += [Site(f"up2.{i}", 2, 640, 8) for i in range(3)]
    s += [Site(f"up3.{i}", 1, 320, 8) for i in range(3)]
    return s


def sd21_sites() -> List[Site]:
    """SD-2.1: same topology, head dim 64 (heads 5 / 10 / 20 / 20)."""
    s = []
    s += [Site(f"down0.{i}", 1, 320, 5) for i in range(2)]
    s += [Site(f"down1.{i}", 2, 640, 10) for i in range(2)]
    s += [Site(f"down2.{i}", 4, 1280, 20) for i in range(2)]
    s += [Site("mid", 8, 1280, 20)]
    s += [Site(f"up1.{i}", 4, 1280, 20) for i in range(3)]
    s += [Site(f"up2.{i}", 2, 640, 10) for i in range(3)]
    s += [Site(f"up3.{i}", 1, 320, 5) for i in range(3)]
    return s


class Attention(torch.nn.Module):
    def __init__(self, C: int, heads: int):
        super().__init__()
        self.heads = heads
        self.scale = (C // heads) ** -0.5
        self.to_q = torch.nn.Linear(C, C, bias=False)
        self.to_k = torch.nn.Linear(C, C, bias=False)
        self.to_v = torch.nn.Linear(C, C, bias=False)
        self.to_out = torch.nn.ModuleList([torch.nn.Linear(C, C), torch.nn.Dropout(0.0)])


class BasicTransformerBlock(torch.nn.Module):
    """Only the self-attention segment's members; the rest of a real block is outside the hot path."""

    def __init__(self, site: Site):
        super().__init__()
        self.site = site
        self.norm1 = torch.nn.LayerNorm(site.channels)
        self.attn1 = Attention(site.channels, site.heads)
        self.attn2 = None
        self.only_cross_attention = False


class ModelMixin(torch.nn.Module):
    pass


class SiteUNet(ModelMixin):
    def __init__(self, sites: List[Site], seed: int = 0):
        super().__init__()
        self.blocks = torch.nn.ModuleList([BasicTransformerBlock(s) for s in sites])
        g = torch.Generator().manual_seed(seed)
        with torch.no_grad():
            for p in self.parameters():
                if p.ndim == 2:
                    p.copy_(torch.randn(p.shape, generator=g) * p.shape[-1] ** -0.5)   # N(0, 1/sqrt(C))
            for b in self.blocks:
                b.norm1.weight.fill_(1.0)
                b.norm1.bias.zero_()
                b.attn1.to_out[0].bias.zero_()

    def set_size(self, latent_hw: Tuple[int, int]) -> None:
        """What hook_tome_model records from the latent (patch.py:208-210)."""
        self._tome_info["size"] = latent_hw


def synthetic_hidden(site: Site, batch: int, frames: int, latent_hw: Tuple[int, int], dtype, device,
                     seed: int, frame_noise: float = 0.5, clip_seed: int = None) -> torch.Tensor:
    """(B*F, N, C) hidden states: per-sample base + frame_noise * N(0,1) per frame (frames of a clip are
    correlated).  Batch layout [uncond frames | cond frames] like generate.py:245.  With ``clip_seed`` the base comes
    from that seed and only the frame noise from ``seed``: different ``seed``s are then different CHUNKS OF ONE CLIP
    (same content, independent frames) -- what consecutive chunks of a video look like to the global level."""
    h, w = latent_hw[0] // site.downsample, latent_hw[1] // site.downsample
    N = h * w
    g = torch.Generator().manual_seed(seed)
    gb = g if clip_seed is None else torch.Generator().manual_seed(clip_seed)
    base = torch.randn(batch, 1, N, site.channels, generator=gb)
    x = base + frame_noise * torch.randn(batch, frames, N, site.channels, generator=g)
    return x.reshape(batch * frames, N, site.channels).to(device=device, dtype=dtype)


def run_segment_pass(unet: SiteUNet, hiddens: List[torch.Tensor]) -> List[torch.Tensor]:
    """One hot-path pass: the patched self-attention segment of every site (patch.py:139-169)."""
    from . import patch
    outs = []
    for blk, h in zip(unet.blocks, hiddens):
        if not hasattr(blk, "generator"):
            blk.generator = patch.init_generator(h.device)        # what hook_tome_module does
        outs.append(patch.patched_self_attention_segment(blk, h, patch.layer_norm(blk.norm1, h)))
    return outs

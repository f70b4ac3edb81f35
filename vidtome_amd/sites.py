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


class CrossAttention(Attention):
    """attn2 of an SD block: queries from the tokens, keys / values from the text conditioning (77 x 768 in SD-1.5)."""

    def __init__(self, C: int, heads: int, cond_dim: int):
        super().__init__(C, heads)
        self.to_k = torch.nn.Linear(cond_dim, C, bias=False)
        self.to_v = torch.nn.Linear(cond_dim, C, bias=False)


class GEGLU(torch.nn.Module):
    def __init__(self, C: int, D: int):
        super().__init__()
        self.proj = torch.nn.Linear(C, 2 * D)

    def forward(self, x):
        h, gate = self.proj(x).chunk(2, dim=-1)
        return h * torch.nn.functional.gelu(gate)


class FeedForward(torch.nn.Module):
    """Diffusers' FeedForward of SD blocks: [GEGLU(C -> 4C, from a C -> 8C projection), Dropout, Linear 4C -> C]."""

    def __init__(self, C: int):
        super().__init__()
        self.net = torch.nn.ModuleList([GEGLU(C, 4 * C), torch.nn.Dropout(0.0), torch.nn.Linear(4 * C, C)])

    def forward(self, x):
        for m in self.net:
            x = m(x)
        return x


class BasicTransformerBlock(torch.nn.Module):
    """The self-attention segment's members; with ``full`` also the rest of an SD block (norm2 / attn2 over the text
    conditioning, norm3 / GEGLU feed-forward: vidtome/patch.py:171-199) for the secondary full-block measurement."""

    def __init__(self, site: Site, full: bool = False, cond_dim: int = 768):
        super().__init__()
        self.site = site
        self.norm1 = torch.nn.LayerNorm(site.channels)
        self.attn1 = Attention(site.channels, site.heads)
        self.attn2 = None
        self.only_cross_attention = False
        if full:
            self.norm2 = torch.nn.LayerNorm(site.channels)
            self.attn2 = CrossAttention(site.channels, site.heads, cond_dim)
            self.norm3 = torch.nn.LayerNorm(site.channels)
            self.ff = FeedForward(site.channels)


class ModelMixin(torch.nn.Module):
    pass


class SiteUNet(ModelMixin):
    def __init__(self, sites: List[Site], seed: int = 0, full: bool = False):
        super().__init__()
        self.blocks = torch.nn.ModuleList([BasicTransformerBlock(s, full) for s in sites])
        g = torch.Generator().manual_seed(seed)
        with torch.no_grad():
            for p in self.parameters():
                if p.ndim == 2:
                    p.copy_(torch.randn(p.shape, generator=g) * p.shape[-1] ** -0.5)   # N(0, 1/sqrt(C))
            for b in self.blocks:
                b.norm1.weight.fill_(1.0)
                b.norm1.bias.zero_()
                b.attn1.to_out[0].bias.zero_()
                if full:
                    for n in (b.norm2, b.norm3):
                        n.weight.fill_(1.0)
                        n.bias.zero_()
                    b.attn2.to_out[0].bias.zero_()
                    b.ff.net[0].proj.bias.zero_()
                    b.ff.net[2].bias.zero_()

    def set_size(self, latent_hw: Tuple[int, int]) -> None:
        """What hook_tome_model records from the latent (patch.py:208-210)."""
        self._tome_info["size"] = latent_hw


DATA_REGIMES = {
    # name: (frame_noise, flat fraction, duplicate fraction) -- SURVEY.md 8d names the first two; `corr05` is what rounds
    # 1-3 measured; `flat25` / `dup` load the matcher's candidate logic; `corr002` / `smooth` are what SURVEY.md
    # section 7 says LayerNorm'd video tokens look like (cross-frame cosine near 1, spatially smooth content)
    "n01": (None, 0.0, 0.0),        # h ~ N(0, 1), no cross-frame correlation: one candidate per row, low cosines
    "corr01": (0.1, 0.0, 0.0),      # h[f] = base + 0.1 N(0, 1): realistic high cross-frame cosine (~0.99)
    "corr002": (0.02, 0.0, 0.0),    # h[f] = base + 0.02 N(0, 1): a static shot (cosine ~0.9996 between the frames of a position:
                                    # every same-position dst token sits inside the filter's window)
    "corr05": (0.5, 0.0, 0.0),      # h[f] = base + 0.5 N(0, 1) (cosine ~0.8 between the frames of a position)
    "flat25": (0.5, 0.25, 0.0),     # corr05 + a flat region: a quarter of the positions (every frame) hold ONE content vector
                                    # + 2 % noise (a sky, a wall): ~N/4 dst rows per frame inside every such row's window
    "dup": (0.5, 0.0, 0.2),         # corr05 + exact copies: a fifth of the positions repeat another position's tokens bit for
                                    # bit in every frame (what anchor updates do to the global level, patch.py:80)
    "smooth": ("smooth", 0.0, 0.0),  # a spatially low-passed field (Gaussian, sigma = SMOOTH_SIGMA tokens, unit variance per
                                    # channel) that drifts by SMOOTH_SHIFT tokens per frame (sub-pixel, bilinear) + 0.05 N(0, 1):
                                    # neighbouring positions AND neighbouring frames are near-maximal matches of a token
}
SMOOTH_SIGMA, SMOOTH_SHIFT, SMOOTH_NOISE = 2.0, 0.1, 0.05


def _smooth_tokens(batch: int, frames: int, N: int, C: int, g: torch.Generator, gb: torch.Generator,
                   frame0: int = 0) -> torch.Tensor:
    """(batch, frames, N, C): one low-passed random field per sample, frame f = the field shifted by (frame0 + f) *
    SMOOTH_SHIFT tokens along both axes (periodic, bilinear) + SMOOTH_NOISE * N(0, 1).  N must be a square grid (the
    bench's latents are)."""
    h = int(round(N ** 0.5))
    if h * h != N:
        raise ValueError(f"the smooth regime needs a square token grid, got N = {N}")
    dev = g.device
    base = torch.randn(batch, C, h, h, generator=gb, device=dev)
    # separable periodic Gaussian blur via the FFT (exact circular convolution: the field tiles seamlessly)
    fy = torch.fft.fftfreq(h, device=dev).view(h, 1)
    fx = torch.fft.rfftfreq(h, device=dev).view(1, h // 2 + 1)
    kern = torch.exp(-2.0 * (torch.pi * SMOOTH_SIGMA) ** 2 * (fy * fy + fx * fx))
    field = torch.fft.irfft2(torch.fft.rfft2(base) * kern, s=(h, h))
    field = field / field.std(dim=(2, 3), keepdim=True)
    out = torch.empty(batch, frames, N, C, device=dev)
    for f in range(frames):
        s = (frame0 + f) * SMOOTH_SHIFT
        k, a = int(s // 1), float(s - s // 1)
        sh = lambda t, dy, dx: torch.roll(t, shifts=(dy, dx), dims=(2, 3))
        fr = ((1 - a) * (1 - a) * sh(field, k, k) + a * (1 - a) * sh(field, k + 1, k)
              + (1 - a) * a * sh(field, k, k + 1) + a * a * sh(field, k + 1, k + 1))
        out[:, f] = fr.reshape(batch, C, N).transpose(1, 2)
    return out + SMOOTH_NOISE * torch.randn(batch, frames, N, C, generator=g, device=dev)


def regime_tokens(regime: str, batch: int, frames: int, N: int, C: int, g: torch.Generator,
                  gb: torch.Generator = None, frame0: int = 0) -> torch.Tensor:
    """(batch, frames, N, C) fp32 tokens of one data regime (DATA_REGIMES); `gb` draws the clip content (base, flat
    vector, duplicate pattern), `g` the per-frame noise.  `frame0` = index of the chunk's first frame in the clip (only
    the drifting `smooth` regime looks at it).  The tensors are drawn on the generators' device (CPU generators: the
    streams the tests and rounds 1-4 used; a CUDA generator draws on the GPU -- bench.py, seconds faster per regime)."""
    noise, flat, dup = DATA_REGIMES[regime]
    gb = g if gb is None else gb
    dev = g.device
    if noise is None:
        return torch.randn(batch, frames, N, C, generator=g, device=dev)
    if noise == "smooth":
        return _smooth_tokens(batch, frames, N, C, g, gb, frame0)
    base = torch.randn(batch, 1, N, C, generator=gb, device=dev)
    x = base + noise * torch.randn(batch, frames, N, C, generator=g, device=dev)
    if flat > 0:
        nf = int(N * flat)
        content = torch.randn(batch, 1, 1, C, generator=gb, device=dev)
        x[:, :, :nf] = content + 0.02 * torch.randn(batch, frames, nf, C, generator=g, device=dev)
    if dup > 0:
        nd = int(N * dup)
        src = torch.randint(nd, N, (nd,), generator=gb, device=dev)
        x[:, :, :nd] = x[:, :, src]
    return x


def synthetic_hidden(site: Site, batch: int, frames: int, latent_hw: Tuple[int, int], dtype, device,
                     seed: int, frame_noise: float = 0.5, clip_seed: int = None, regime: str = None,
                     frame0: int = 0, gen_device=None) -> torch.Tensor:
    """(B*F, N, C) hidden states: per-sample base + frame_noise * N(0,1) per frame (frames of a clip are
    correlated).  Batch layout [uncond frames | cond frames] like generate.py:245.  With ``clip_seed`` the base comes
    from that seed and only the frame noise from ``seed``: different ``seed``s are then different CHUNKS OF ONE CLIP
    (same content, independent frames) -- what consecutive chunks of a video look like to the global level.
    ``regime`` selects one of DATA_REGIMES instead (``corr05`` = the default arithmetic, same random stream)."""
    h, w = latent_hw[0] // site.downsample, latent_hw[1] // site.downsample
    N = h * w
    gdev = torch.device("cpu") if gen_device is None else torch.device(gen_device)
    g = torch.Generator(device=gdev).manual_seed(seed)
    gb = g if clip_seed is None else torch.Generator(device=gdev).manual_seed(clip_seed)
    if regime is not None and regime != "corr05":
        x = regime_tokens(regime, batch, frames, N, site.channels, g, gb, frame0)
    else:
        base = torch.randn(batch, 1, N, site.channels, generator=gb, device=gdev)
        x = base + frame_noise * torch.randn(batch, frames, N, site.channels, generator=g, device=gdev)
    return x.reshape(batch * frames, N, site.channels).to(device=device, dtype=dtype)


def run_segment_pass(unet: SiteUNet, hiddens: List[torch.Tensor]) -> List[torch.Tensor]:
    """One hot-path pass: the patched self-attention segment of every site (patch.py:139-169)."""
    from . import patch
    outs = []
    for blk, h in zip(unet.blocks, hiddens):
        if not hasattr(blk, "generator"):
            blk.generator = patch.init_generator(h.device, mode=blk._tome_info["args"].get("generator_device"))   # what hook_tome_module does
        outs.append(patch.self_attention_segment(blk, h))
    return outs


def run_block_pass(unet: SiteUNet, hiddens: List[torch.Tensor], cond: torch.Tensor) -> List[torch.Tensor]:
    """One FULL-block pass (secondary measurement): the patched block's whole forward (patch.py:128-201) at every site --
    the hot-path segment, then the cross-attention over ``cond`` (B*F, 77, 768) and the GEGLU feed-forward."""
    return [blk(h, encoder_hidden_states=cond) for blk, h in zip(unet.blocks, hiddens)]


class ClipStream:
    """The chunk stream bench.py (and tools / tests) feed the patched sites with: chunk c of the run holds frame set
    c % n_sets of ONE synthetic clip (per-sample base shared by all sets, independent frame noise), so the anchor tokens a
    chunk merges against always come from a different chunk (generate.py:215-219).

    The reference resets the anchors after every denoising step (generate.py:233-236) and the first chunk of a step only
    stores its local tokens (patch.py:82), so the anchor CHAIN a chunk sees is 1 .. chunks_per_step - 1 updates long.  A
    measurement of steady-state passes (every pass has a global level) must not let the chain grow without bound -- each
    local-is-src update copies matched rows into the anchors (patch.py:80) and an endless chain accumulates duplicates no
    real run has -- so every `chunks_per_step - 1` passes the anchors are re-seeded with what the first chunk of a step
    would have stored (its local merged tokens, computed once per frame set before the timed region; a pointer swap, no
    kernel).  `same_chunk` is rounds 1-2's regime: one frame set, fed to every pass, never re-seeded.
    """

    def __init__(self, unet: "SiteUNet", site_list: List[Site], batch: int, frames: int, latent_hw: Tuple[int, int], dtype,
                 device, n_sets: int = 3, chunks_per_step: int = 8, same_chunk: bool = False, rank: int = 0,
                 reseed: bool = True, sets=None, cond: torch.Tensor = None, regime: str = None, gen_device=None,
                 inflight: int = 1):
        self.unet, self.site_list = unet, site_list
        # `inflight` chunks at a time: chunk c is issued on HIP stream c % inflight (the host issues whole chunks in order; on the
        # device chunk c + 1 waits block by block for chunk c's anchors -- patch.mark_anchors_ready / await_anchors), so the
        # dispatch gaps and the small launches of one chunk run beside the big kernels of the other.  Same results as one stream.
        self.streams = [torch.cuda.Stream(device=device) for _ in range(inflight)] if inflight > 1 else None
        self.cond = cond                      # not None: full-block passes (run_block_pass)
        self.K = 1 if same_chunk else max(2, n_sets)
        self.same_chunk = same_chunk
        self.reseed_every = 0 if (same_chunk or not reseed) else max(1, chunks_per_step - 1)
        self.steady = 0                       # steady-state passes run so far
        self.seeds = {}                       # frame set -> per-block first-chunk anchors
        want = range(self.K) if sets is None else sorted(set(sets))

        def make(j):
            if same_chunk:
                return [synthetic_hidden(s, batch, frames, latent_hw, dtype, device, seed=1234 + 97 * rank + i, regime=regime,
                                         gen_device=gen_device) for i, s in enumerate(site_list)]
            return [synthetic_hidden(s, batch, frames, latent_hw, dtype, device, seed=1234 + 97 * j + i,
                                     clip_seed=4321 + i, regime=regime, frame0=j * frames, gen_device=gen_device)
                    for i, s in enumerate(site_list)]
        self.sets = {j: make(j) for j in want}

    def _first_chunk(self, j: int) -> None:
        """What the first chunk of a step leaves behind when it is frame set j: every merging block's local tokens."""
        for b in self.unet.blocks:
            b.global_tokens = None
        with torch.no_grad():
            self._run(self.sets[j])
        self.seeds[j] = [getattr(b, "global_tokens", None) for b in self.unet.blocks]

    def populate(self) -> None:
        """Untimed: the first-chunk pass of every frame set (the seeds), leaving the anchors of set K - 1 in place -- the
        state in front of steady pass 0, which processes set 0."""
        order = sorted(self.sets)
        if self.reseed_every:
            for j in order:
                self._first_chunk(j)
        else:
            self._first_chunk(order[-1])

    def step(self, chunk: int):
        """One steady-state pass: chunk index `chunk` of the run (frame set chunk % K)."""
        j = chunk % self.K
        if self.reseed_every and self.steady % self.reseed_every == 0:
            prev = self.seeds[(chunk - 1) % self.K]
            for b, a in zip(self.unet.blocks, prev):
                b.global_tokens = a
        self.steady += 1
        with torch.no_grad():
            if self.streams is None:
                return self._run(self.sets[j])
            with torch.cuda.stream(self.streams[chunk % len(self.streams)]):
                return self._run(self.sets[j])

    def _run(self, hiddens):
        return run_segment_pass(self.unet, hiddens) if self.cond is None else run_block_pass(self.unet, hiddens, self.cond)

#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING the reference.

Run in the build container only (the reference lives at /root/reference there and nowhere else):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Nothing of the reference (source, bytecode) is copied: the script calls the reference's own
functions -- vidtome/merge.py `bipartite_soft_matching_randframe` / `bipartite_soft_matching_2s`,
vidtome/patch.py `apply_patch` + the patched block forward, utils/pnp_utils.py
`register_attention_control` -- on seeded inputs and stores inputs + outputs as .npz data.

Every case is *screened*: the reference is run in float32 and in float64 and the case is kept only
if both give identical indices, i.e. the result does not depend on the (unspecified) summation
order of torch's CPU kernels.  That is what makes "bit-exact indices vs the reference" a
well-defined statement (hard parts).
"""
import hashlib
import importlib.util
import json
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
import vidtome  # noqa: E402  (the reference package)
from vidtome import merge as ref_merge  # noqa: E402
from vidtome import patch as ref_patch  # noqa: E402

_spec = importlib.util.spec_from_file_location("ref_pnp_utils", os.path.join(REF, "utils", "pnp_utils.py"))
ref_pnp = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(ref_pnp)

sys.path.insert(0, HERE)
from inputs import planted_inputs, portable_weight  # noqa: E402  (shared, reference-free input builders)

torch.set_grad_enabled(False)


def cells(fn):
    """Read the index tensors a reference closure captured (merge.py:119,135)."""
    out = {n: c.cell_contents for n, c in zip(fn.__code__.co_freevars, fn.__closure__)}
    if "split" in out:      # a_idx / b_idx are captured by the inner `split` (merge.py:76-81)
        out.update(cells(out["split"]))
    return out


def np64(t):
    return t.detach().cpu().numpy().astype(np.int64)


def fork_generator(seed):
    """What hook_tome_module/init_generator do on CPU (patch.py:219, vidtome/utils.py:22-23)."""
    torch.manual_seed(seed)
    return torch.Generator(device="cpu").set_state(torch.get_rng_state())


MARGIN = 1e-6   # >= ~16 ulp of a cosine near 1; summation-order noise is ~1-3 ulp


def margins(metric, a_idx, b_idx, align):
    """fp64 recomputation of merge.py:84-113 to measure how well separated a case is: returns
    (min gap between adjacent sorted node_max values, min top-1/top-2 score gap)."""
    x = metric.double()
    x = x / x.norm(dim=-1, keepdim=True)
    a = x[:, a_idx.reshape(-1)]
    b = x[:, b_idx.reshape(-1)]
    sc = a @ b.transpose(-1, -2)
    if align:
        sc = torch.cat([*sc], dim=-1)[None]
    if torch.isnan(sc).any():
        return 1.0, 1.0

    def min_real_gap(g):
        # exact ties (duplicate rows: anchors hold copies of matched rows, patch.py:80) are resolved
        # by index order, not by rounding -> fine; only 0 < gap < MARGIN is ambiguous
        g = g[g > 1e-12]
        return float(g.min()) if g.numel() else 1.0

    top2 = sc.topk(min(2, sc.shape[-1]), dim=-1).values
    g2 = min_real_gap(top2[..., 0] - top2[..., -1]) if sc.shape[-1] > 1 else 1.0
    nm = top2[..., 0].sort(dim=-1).values
    g1 = min_real_gap(nm[..., 1:] - nm[..., :-1]) if nm.shape[-1] > 1 else 1.0
    return g1, g2


def well_separated(metric, c, align):
    g1, g2 = margins(metric, c["a_idx"], c["b_idx"], align)
    return g1 >= MARGIN and g2 >= MARGIN


# --------------------------------------------------------------------------------------------
# 1. direct calls of the two matchers
# --------------------------------------------------------------------------------------------
def run_randframe(x, F, ratio, unm_pre, seed, stride, align, dtype):
    gen = fork_generator(seed)
    m, u, ret = ref_merge.bipartite_soft_matching_randframe(
        x.to(dtype), F, ratio, unm_pre, gen, stride, align)
    if m is ref_merge.do_nothing:
        return {"noop": True, "unm_num": ret["unm_num"]}
    c = cells(m)
    merged = m(x.to(dtype))
    return {"noop": False, "unm_num": ret["unm_num"], "a_idx": np64(c["a_idx"])[0, :, 0],
            "b_idx": np64(c["b_idx"])[0, :, 0], "unm_idx": np64(c["unm_idx"])[..., 0],
            "src_idx": np64(c["src_idx"])[..., 0], "dst_idx": np64(c["dst_idx"])[..., 0],
            "merged": merged, "m": m, "u": u}


def same_idx(a, b):
    return all(np.array_equal(a[k], b[k]) for k in ("a_idx", "b_idx", "unm_idx", "src_idx", "dst_idx"))


def gen_randframe_cases():
    cases = []
    spec = [
        # B, F, tnum, C, unm_pre, ratio, stride, align
        (2, 4, 4, 8, 0, 0.5, 4, False),      # the SURVEY tiny KAT shape
        (2, 4, 4, 8, 0, 0.5, 4, True),
        (2, 2, 24, 16, 0, 0.5, 4, False),
        (2, 3, 20, 16, 0, 0.6, 4, False),    # F=3 -> ts=3
        (2, 4, 32, 16, 0, 0.9, 4, False),
        (2, 4, 32, 16, 0, 1.0, 4, False),    # r == Ns, no unmerged rows
        (2, 4, 32, 16, 0, 0.0, 4, False),    # ratio<=0 early-out (merge.py:45-46)
        (2, 8, 16, 16, 0, 0.5, 4, False),    # F=8: two dst frames
        (2, 16, 12, 24, 0, 0.5, 4, False),   # F=16
        (3, 16, 12, 24, 0, 0.5, 4, True),    # PnP batch, aligned
        (2, 6, 10, 16, 0, 0.5, 4, False),    # F % ts != 0
        (2, 4, 16, 16, 37, 0.5, 4, False),   # unm_pre > 0 (second level)
        (3, 4, 16, 16, 37, 0.6, 4, True),
        (2, 2, 16, 16, 21, 0.9, 4, False),
        (2, 5, 13, 16, 7, 0.5, 2, False),    # stride 2, odd sizes
        (1, 4, 16, 16, 0, 0.5, 4, False),    # B = 1
        (2, 4, 18, 40, 0, 0.3, 4, False),
    ]
    for ci, (B, F, tnum, C, unm_pre, ratio, stride, align) in enumerate(spec):
        N = unm_pre + tnum * F
        for attempt in range(50):
            seed = 1000 + 17 * ci + attempt
            g = torch.Generator().manual_seed(seed)
            x = torch.randn(B, N, C, generator=g)
            r32 = run_randframe(x, F, ratio, unm_pre, seed, stride, align, torch.float32)
            r64 = run_randframe(x, F, ratio, unm_pre, seed, stride, align, torch.float64)
            if r32["noop"] or (same_idx(r32, r64) and well_separated(
                    x, {"a_idx": torch.from_numpy(r32["a_idx"]), "b_idx": torch.from_numpy(r32["b_idx"])}, align)):
                break
        else:
            raise RuntimeError(f"could not screen randframe case {ci}")
        case = {"kind": "randframe", "B": B, "F": F, "N": N, "C": C, "unm_pre": unm_pre, "ratio": ratio,
                "stride": stride, "align": align, "seed": seed, "x": x.numpy(), "noop": r32["noop"],
                "unm_num": r32["unm_num"]}
        if not r32["noop"]:
            ts = min(stride, F)
            # recover the draw from the partition: first dst position // tnum
            randf = int((r32["b_idx"][0] - unm_pre) // tnum) if len(r32["b_idx"]) > unm_pre else -1
            gen = fork_generator(seed)
            drawn = int(torch.randint(0, ts, torch.Size([1]), generator=gen))
            assert randf in (drawn, -1), (randf, drawn)
            y = torch.randn(B, r32["merged"].shape[1], C, generator=g)
            case.update({k: r32[k] for k in ("a_idx", "b_idx", "unm_idx", "src_idx", "dst_idx")})
            case.update({"randf": drawn, "merged": r32["merged"].numpy(), "y": y.numpy(),
                         "unmerged": r32["u"](y).numpy()})
        cases.append(case)
    return cases


def run_2s(x, src_len, ratio, align, chunk, dtype):
    m, u, ret = ref_merge.bipartite_soft_matching_2s(x.to(dtype), src_len, ratio, align, unmerge_chunk=chunk)
    c = cells(m)
    return {"a_idx": np64(c["a_idx"])[0, :, 0], "b_idx": np64(c["b_idx"])[0, :, 0],
            "unm_idx": np64(c["unm_idx"])[..., 0], "src_idx": np64(c["src_idx"])[..., 0],
            "dst_idx": np64(c["dst_idx"])[..., 0], "merged": m(x.to(dtype)), "m": m, "u": u,
            "unm_num": ret["unm_num"]}


def gen_2s_cases():
    cases = []
    spec = [
        # B, src_len, dst_len, C, ratio, align, unmerge_chunk
        (2, 40, 40, 16, 0.5, False, 0),
        (2, 40, 40, 16, 0.5, False, 1),
        (2, 40, 24, 16, 0.5, False, 0),     # rectangular (F=1 chunk vs longer anchors)
        (2, 24, 40, 16, 0.8, False, 1),
        (3, 33, 47, 24, 0.8, True, 0),
        (3, 47, 33, 24, 0.6, True, 1),
        (2, 32, 32, 16, 1.0, False, 0),
        (1, 20, 30, 8, 0.5, False, 1),
    ]
    for ci, (B, sl, dl, C, ratio, align, chunk) in enumerate(spec):
        for attempt in range(50):
            seed = 5000 + 13 * ci + attempt
            g = torch.Generator().manual_seed(seed)
            x = torch.randn(B, sl + dl, C, generator=g)
            r32 = run_2s(x, sl, ratio, align, chunk, torch.float32)
            r64 = run_2s(x, sl, ratio, align, chunk, torch.float64)
            if same_idx(r32, r64) and well_separated(
                    x, {"a_idx": torch.from_numpy(r32["a_idx"]), "b_idx": torch.from_numpy(r32["b_idx"])}, align):
                break
        else:
            raise RuntimeError(f"could not screen 2s case {ci}")
        y = torch.randn(B, r32["merged"].shape[1], C, generator=g)
        case = {"kind": "2s", "B": B, "src_len": sl, "N": sl + dl, "C": C, "ratio": ratio, "align": align,
                "unmerge_chunk": chunk, "seed": seed, "x": x.numpy(), "unm_num": r32["unm_num"],
                "merged": r32["merged"].numpy(), "y": y.numpy(), "unmerged": r32["u"](y).numpy()}
        case.update({k: r32[k] for k in ("a_idx", "b_idx", "unm_idx", "src_idx", "dst_idx")})
        cases.append(case)
    return cases


def gen_nan_case():
    """Zero token -> 0/0 = NaN row (merge.py:84 has no eps): NaN max at index 0 and NaN sorts first."""
    g = torch.Generator().manual_seed(77)
    B, F, tnum, C = 2, 4, 8, 8
    x = torch.randn(B, F * tnum, C, generator=g)
    x[0, 3] = 0.0      # a src row of sample 0 (frame 0)
    gen = fork_generator(123)   # randf = 2 -> frame 2 is dst
    m, u, ret = ref_merge.bipartite_soft_matching_randframe(x, F, 0.5, 0, gen, 4, False)
    c = cells(m)
    return {"kind": "nan", "x": x.numpy(), "F": F, "ratio": 0.5, "seed": 123, "randf": 2,
            "a_idx": np64(c["a_idx"])[0, :, 0], "b_idx": np64(c["b_idx"])[0, :, 0],
            "unm_idx": np64(c["unm_idx"])[..., 0], "src_idx": np64(c["src_idx"])[..., 0],
            "dst_idx": np64(c["dst_idx"])[..., 0]}


def gen_survey_kat():
    """The tiny known-answer case quoted in SURVEY.md 8c, regenerated from the reference."""
    torch.manual_seed(7)
    x = torch.randn(2, 16, 8)
    out = {"kind": "kat", "x": x.numpy()}
    for align in (False, True):
        gen = torch.Generator().manual_seed(123)
        m, u, ret = ref_merge.bipartite_soft_matching_randframe(x, 4, 0.5, 0, gen, 4, align)
        c = cells(m)
        sfx = "_al" if align else ""
        out["a_idx"] = np64(c["a_idx"])[0, :, 0]
        out["b_idx"] = np64(c["b_idx"])[0, :, 0]
        out["unm_idx" + sfx] = np64(c["unm_idx"])[..., 0]
        out["src_idx" + sfx] = np64(c["src_idx"])[..., 0]
        out["dst_idx" + sfx] = np64(c["dst_idx"])[..., 0]
    g = torch.Generator().manual_seed(123)
    out["randint_0_4_x8"] = np.array([int(torch.randint(0, 4, torch.Size([1]), generator=g)) for _ in range(8)])
    g = torch.Generator().manual_seed(123)
    seq = []
    for _ in range(2):
        seq += [float(torch.randint(0, 4, torch.Size([1]), generator=g)),
                float(torch.randint(0, 4, torch.Size([1]), generator=g)),
                float(torch.rand(1, generator=g))]
    out["interleaved"] = np.array(seq, dtype=np.float64)
    out["int_trunc"] = np.array([int(49152 * 0.5), int(12288 * 0.9), int(110592 * 0.6), int(27648 * 0.6),
                                 int(34816 * 0.8)])
    return out


# --------------------------------------------------------------------------------------------
# 2. apply_patch on a stand-in UNet: compute_merge chain, global tokens, RNG lock-step, PnP attention
# --------------------------------------------------------------------------------------------
class Attention(torch.nn.Module):
    """Stand-in for diffusers' Attention with exactly the attributes sa_forward reads
    (pnp_utils.py:41-95): heads, scale, to_q/to_k/to_v, to_out[0], head_to_batch_dim, batch_to_head_dim."""

    def __init__(self, C, heads):
        super().__init__()
        self.heads = heads
        self.scale = (C // heads) ** -0.5
        self.to_q = torch.nn.Linear(C, C, bias=False)
        self.to_k = torch.nn.Linear(C, C, bias=False)
        self.to_v = torch.nn.Linear(C, C, bias=False)
        self.to_out = torch.nn.ModuleList([torch.nn.Linear(C, C), torch.nn.Dropout(0.0)])
        self.seen = []

    def head_to_batch_dim(self, t):
        b, n, c = t.shape
        h = self.heads
        return t.reshape(b, n, h, c // h).permute(0, 2, 1, 3).reshape(b * h, n, c // h)

    def batch_to_head_dim(self, t):
        bh, n, d = t.shape
        h = self.heads
        return t.reshape(bh // h, h, n, d).permute(0, 2, 1, 3).reshape(bh // h, n, h * d)

    def forward(self, x, encoder_hidden_states=None, attention_mask=None, **kw):
        raise RuntimeError("replaced by register_attention_control's sa_forward")


class Zero(torch.nn.Module):
    def forward(self, x):
        return torch.zeros_like(x)


class CrossAttention(Attention):
    """attn2 of an SD block (make_golden_fullblock.py): queries from the tokens, keys / values from the conditioning."""

    def __init__(self, C, heads, cond_dim):
        super().__init__(C, heads)
        self.to_k = torch.nn.Linear(cond_dim, C, bias=False)
        self.to_v = torch.nn.Linear(cond_dim, C, bias=False)


class GEGLU(torch.nn.Module):
    """Diffusers' GEGLU (third-party, SURVEY.md 8c: restated from its published form): one Linear to 2 D, value * gelu(gate),
    exact (erf) gelu."""

    def __init__(self, C, D):
        super().__init__()
        self.proj = torch.nn.Linear(C, 2 * D)

    def forward(self, x):
        h, gate = self.proj(x).chunk(2, dim=-1)
        return h * torch.nn.functional.gelu(gate)


class FeedForward(torch.nn.Module):
    """Diffusers' FeedForward of an SD block: [GEGLU(C -> 4C), Dropout(0), Linear(4C -> C)]."""

    def __init__(self, C):
        super().__init__()
        self.net = torch.nn.ModuleList([GEGLU(C, 4 * C), torch.nn.Dropout(0.0), torch.nn.Linear(4 * C, C)])

    def forward(self, x):
        for m in self.net:
            x = m(x)
        return x


class BasicTransformerBlock(torch.nn.Module):   # class NAME is what apply_patch looks for (patch.py:319)
    def __init__(self, C, heads, full=False, cond_dim=0):
        super().__init__()
        self.norm1 = torch.nn.LayerNorm(C)
        self.attn1 = Attention(C, heads)
        self.only_cross_attention = False
        if full:        # make_golden_fullblock.py: the whole SD block (patch.py:171-199)
            self.norm2 = torch.nn.LayerNorm(C)
            self.attn2 = CrossAttention(C, heads, cond_dim)
            self.norm3 = torch.nn.LayerNorm(C)
            self.ff = FeedForward(C)
        else:
            self.attn2 = None
            self.norm2 = None
            self.norm3 = torch.nn.Identity()
            self.ff = Zero()


class _Attn2D(torch.nn.Module):
    """Stand-in for Diffusers' Transformer2DModel: it calls its block with the FULL keyword set the patched forward
    declares (patch.py:128-137; caller side per SURVEY.md 8b)."""

    def __init__(self, C, heads, full=False, cond_dim=0):
        super().__init__()
        self.transformer_blocks = torch.nn.ModuleList([BasicTransformerBlock(C, heads, full, cond_dim)])

    def forward(self, hidden_states, encoder_hidden_states=None, timestep=None, class_labels=None,
                cross_attention_kwargs=None, attention_mask=None, encoder_attention_mask=None):
        for block in self.transformer_blocks:
            hidden_states = block(hidden_states, attention_mask=attention_mask, encoder_hidden_states=encoder_hidden_states,
                                  encoder_attention_mask=encoder_attention_mask, timestep=timestep,
                                  cross_attention_kwargs=cross_attention_kwargs, class_labels=class_labels)
        return hidden_states


class _UpBlock(torch.nn.Module):
    def __init__(self, C, heads, n, ds, full=False, cond_dim=0):
        super().__init__()
        self.ds = ds
        if n:
            self.attentions = torch.nn.ModuleList([_Attn2D(C, heads, full, cond_dim) for _ in range(n)])


class ModelMixin(torch.nn.Module):              # class NAME checked by isinstance_str (patch.py:279-280)
    pass


class StandInUNet(ModelMixin):
    """up_blocks[1..3] x 3 transformer blocks at downsample 4 / 2 / 1, like SD's decoder."""

    def __init__(self, C, heads, full=False, cond_dim=0):
        super().__init__()
        self.C = C
        self.full = full
        self.up_blocks = torch.nn.ModuleList([_UpBlock(C, heads, 0, 8), _UpBlock(C, heads, 3, 4, full, cond_dim),
                                              _UpBlock(C, heads, 3, 2, full, cond_dim), _UpBlock(C, heads, 3, 1, full, cond_dim)])
        self.embed = torch.nn.Linear(4, C)
        self.mix = torch.nn.Linear(C, C)
        self.gen = torch.Generator().manual_seed(0)
        self.records = []

    def blocks(self):
        for ub in self.up_blocks:
            if hasattr(ub, "attentions"):
                for a in ub.attentions:
                    yield ub.ds, a.transformer_blocks[0]

    def transformers(self):
        for ub in self.up_blocks:
            if hasattr(ub, "attentions"):
                for a in ub.attentions:
                    yield a

    def forward(self, latent, t=None, encoder_hidden_states=None):
        outs = []
        t2d = list(self.transformers())
        for bi, (ds, blk) in enumerate(self.blocks()):
            z = torch.nn.functional.avg_pool2d(latent, ds) if ds > 1 else latent
            tok = z.flatten(2).transpose(1, 2)                    # (B*F, N, 4)
            # i.i.d. Gaussian token directions mixed with a little latent content: cosines are well spread,
            # which keeps the case separable from rounding noise (see MARGIN)
            hidden = self.mix(torch.tanh(2.0 * self.embed(tok) + 0.3 * bi)) * 0.25 + \
                getattr(self, "hidden_noise", 1.0) * torch.randn(tok.shape[0], tok.shape[1], self.C, generator=self.gen).to(tok.dtype)
            if getattr(self, "fp16_grid", False):     # make_golden_chain16.py: hidden states an fp16 model can hold exactly
                hidden = hidden.half().to(tok.dtype)
            if self.full:   # through the stand-in Transformer2DModel: the block is called with the full keyword set
                out = t2d[bi](hidden, encoder_hidden_states=encoder_hidden_states, timestep=t)
            else:
                out = blk(hidden, encoder_hidden_states=encoder_hidden_states)
            self.records.append({"block": bi, "ds": ds, "hidden": hidden.clone(), "out": out.clone()})
            outs.append(out)
        return outs


class Pipe:
    """`model.unet` holder for register_attention_control (pnp_utils.py:98-105)."""

    def __init__(self, unet):
        self.unet = unet


def run_chain(dtype, cfg, weights_seed):
    C, heads, H, W = cfg["C"], cfg["heads"], cfg["H"], cfg["W"]
    torch.manual_seed(weights_seed)
    full = bool(cfg.get("full"))
    unet = StandInUNet(C, heads, full, cfg.get("cond_dim", 0)) if full else StandInUNet(C, heads)
    for p in unet.parameters():     # drawn in fp32 so the fp64 screening run sees the same weights
        p.copy_(torch.randn_like(p) * (0.5 if p.ndim == 1 else p.shape[-1] ** -0.5))
    if full:                        # LayerNorm gains near 1 (a gain drawn like a bias would be a different model)
        for _, blk in unet.blocks():
            for nrm in (blk.norm2, blk.norm3):
                nrm.weight.copy_(1.0 + 0.1 * torch.randn_like(nrm.weight))
        # the blocks' matrices come from a portable integer generator keyed by the parameter name (inputs.portable_weight):
        # the fixture stores the 1-D parameters only, the test rebuilds the matrices
        for n, p in unet.named_parameters():
            if p.ndim == 2 and "transformer_blocks" in n:
                p.copy_(torch.from_numpy(portable_weight(n, tuple(p.shape))))
    if cfg.get("fp16_grid"):
        # make_golden_chain16.py: weights and hidden states on the fp16 grid (an fp16 model holds them exactly; the
        # reference still computes in `dtype`), positions of one clip correlated across frames
        for p in unet.parameters():
            p.copy_(p.half().float())
        unet.fp16_grid = True
        unet.hidden_noise = cfg.get("hidden_noise", 1.0)
    unet = unet.to(dtype)
    if cfg.get("round_norm1"):
        # make_golden_chain16.py: the model an fp16 run holds -- norm1's output on the fp16 grid (what the matcher and the
        # attention of an fp16 model see).  An integer > 1 additionally moves a random 2e-4 of the elements by one fp16
        # ulp (screening: another correctly-implemented fp16 LayerNorm may round that many elements the other way)
        class RoundedNorm(torch.nn.Module):
            def __init__(self, inner, seed):
                super().__init__()
                self.inner = inner
                self.gen = torch.Generator().manual_seed(seed) if seed > 1 else None

            def forward(self, x):
                y = self.inner(x).half()
                if self.gen is not None:
                    hit = torch.rand(y.shape, generator=self.gen) < 2e-4
                    step = torch.where(torch.rand(y.shape, generator=self.gen) < 0.5, 1, -1).to(torch.int16)
                    y = torch.where(hit, (y.view(torch.int16) + step).view(torch.float16), y)
                return y.to(x.dtype)
        for bi, (_, blk) in enumerate(unet.blocks()):
            blk.norm1 = RoundedNorm(blk.norm1, int(cfg["round_norm1"]) * 100 + bi if int(cfg["round_norm1"]) > 1 else 1)
    pipe = Pipe(unet)
    # PnP attention override: the reference's own attention arithmetic (pnp_utils.py:39-106)
    ref_pnp.register_attention_control(pipe, cfg["injection"], cfg["B"])
    for _, blk in unet.blocks():
        if not hasattr(blk.attn1, "injection_schedule"):
            # up_blocks[1].attentions[0] is not overridden by the reference (pnp_utils.py:100);
            # give it the same arithmetic without injection so every block has a forward.
            blk.attn1.injection_schedule = None
            blk.attn1.forward = _plain_sa(blk.attn1)
        blk.attn1.t = cfg["t"]
    if full:
        # attn2 = the reference's own attention arithmetic too: register_attention_control is pointed at a holder whose
        # "attn1" slots are the blocks' attn2 modules, so sa_forward's closure (its is_cross branch, pnp_utils.py:51-53,
        # 71-75) becomes attn2.forward; the one block it skips (pnp_utils.py:100) gets the same arithmetic restated
        class _Holder(torch.nn.Module):
            pass
        fake = _Holder()
        fake.unet = _Holder()
        fake.unet.up_blocks = []
        for ub in unet.up_blocks:
            h_ub = _Holder()
            h_ub.attentions = []
            for a in (ub.attentions if hasattr(ub, "attentions") else []):
                h_a, h_b = _Holder(), _Holder()
                h_b.attn1 = a.transformer_blocks[0].attn2
                h_a.transformer_blocks = [h_b]
                h_ub.attentions.append(h_a)
            fake.unet.up_blocks.append(h_ub)
        ref_pnp.register_attention_control(fake, None, cfg["B"])
        for _, blk in unet.blocks():
            if not hasattr(blk.attn2, "injection_schedule"):
                blk.attn2.injection_schedule = None
                blk.attn2.forward = _plain_sa(blk.attn2)
            blk.attn2.t = cfg["t"]
    # record what attn1 receives = compute_merge's merged tokens
    for bi, (_, blk) in enumerate(unet.blocks()):
        inner = blk.attn1.forward

        def wrapped(x, encoder_hidden_states=None, attention_mask=None, _inner=inner, _blk=blk, **kw):
            _blk.attn1.seen.append(x.clone())
            return _inner(x, encoder_hidden_states=encoder_hidden_states, attention_mask=attention_mask, **kw)

        blk.attn1.forward = wrapped

    # record the matcher closures (indices) without touching the reference code
    trace = []
    orig_rf, orig_2s = ref_merge.bipartite_soft_matching_randframe, ref_merge.bipartite_soft_matching_2s

    def rec_rf(*a, **k):
        res = orig_rf(*a, **k)
        if res[0] is not ref_merge.do_nothing:
            c = cells(res[0])
            trace.append({"kind": "local", "margins": margins(a[0], c["a_idx"], c["b_idx"], a[6]),
                          **{n: np64(c[n]) for n in ("unm_idx", "src_idx", "dst_idx")}})
        return res

    def rec_2s(*a, **k):
        res = orig_2s(*a, **k)
        c = cells(res[0])
        trace.append({"kind": "global", "src_len": int(a[1]), "margins": margins(a[0], c["a_idx"], c["b_idx"], a[3]),
                      **{n: np64(c[n]) for n in ("unm_idx", "src_idx", "dst_idx")}})
        return res

    ref_merge.bipartite_soft_matching_randframe = rec_rf
    ref_merge.bipartite_soft_matching_2s = rec_2s
    try:
        vidtome.apply_patch(unet, local_merge_ratio=cfg["local_ratio"], merge_global=cfg["merge_global"],
                            global_merge_ratio=cfg["global_ratio"], batch_size=cfg["B"],
                            align_batch=cfg["align"], target_stride=4, global_rand=0.5)
        torch.manual_seed(cfg["rng_seed"])       # the block generators fork THIS state (patch.py:219)
        rng_state = torch.get_rng_state().clone()
        g = torch.Generator().manual_seed(cfg["data_seed"])
        unet.gen.manual_seed(cfg["data_seed"] + 1)
        chunks = []
        for ck, F in enumerate(cfg["chunk_frames"]):
            if ck in cfg.get("reset_before", []):
                # generate.py:233-236: anchors live for one denoising step only
                vidtome.update_patch(unet, global_tokens=None)
            base = torch.randn(1, 4, H, W, generator=g)
            lat = (base + cfg["frame_noise"] * torch.randn(cfg["B"] * F, 4, H, W, generator=g)).to(dtype)
            unet.records.clear()
            for _, blk in unet.blocks():
                blk.attn1.seen.clear()
            n0 = len(trace)
            cond = None
            if full:    # the text conditioning: one embedding per batch group, repeated over the chunk's frames (generate.py:245)
                cond = torch.randn(cfg["B"], 1, cfg["cond_tokens"], cfg["cond_dim"], generator=g).half().float()
                cond = cond.expand(-1, F, -1, -1).reshape(cfg["B"] * F, cfg["cond_tokens"], cfg["cond_dim"]).to(dtype)
            unet(lat, cfg["t"] if full else 0, cond)
            gts = vidtome.collect_from_patch(unet, attr="global_tokens")
            chunks.append({
                "latent": lat.clone(), "cond": None if cond is None else cond.clone(), "records": [dict(r) for r in unet.records],
                "merged": [blk.attn1.seen[0].clone() for _, blk in unet.blocks()],
                "global_tokens": {k: (None if v is None else v.clone()) for k, v in gts.items()},
                "trace": trace[n0:],
            })
        names = [n for n, m in unet.named_modules() if m.__class__.__name__ == "ToMeBlock"]
        vidtome.remove_patch(unet)
    finally:
        ref_merge.bipartite_soft_matching_randframe = orig_rf
        ref_merge.bipartite_soft_matching_2s = orig_2s
    weights = {n.replace("norm1.inner.", "norm1."): p.detach().clone() for n, p in unet.state_dict().items()}
    return chunks, weights, names, rng_state


def _plain_sa(attn):
    def forward(x, encoder_hidden_states=None, attention_mask=None, **kw):
        ctx = x if encoder_hidden_states is None else encoder_hidden_states      # pnp_utils.py:51-53
        q = attn.head_to_batch_dim(attn.to_q(x))
        k = attn.head_to_batch_dim(attn.to_k(ctx))
        v = attn.head_to_batch_dim(attn.to_v(ctx))
        sim = torch.einsum("b i d, b j d -> b i j", q, k) * attn.scale
        out = torch.einsum("b i j, b j d -> b i d", sim.softmax(dim=-1), v)
        return attn.to_out[0](attn.batch_to_head_dim(out))
    return forward


def gen_chain_cases():
    cfgs = [
        # CFG batch (B=2), F=4 chunks, local+global, no injection
        dict(name="chain_cfg_f4", B=2, C=16, heads=2, H=6, W=6, chunk_frames=[4, 4, 4, 1, 4], reset_before=[3],
             local_ratio=0.5, merge_global=True, global_ratio=0.5, align=False, injection=None, t=500,
             rng_seed=2, data_seed=1234, frame_noise=1.0),
        # two local levels (F=8), default-like ratios
        dict(name="chain_cfg_f8", B=2, C=16, heads=2, H=6, W=6, chunk_frames=[8, 8, 3],
             local_ratio=0.9, merge_global=True, global_ratio=0.8, align=False, injection=None, t=500,
             rng_seed=2, data_seed=99, frame_noise=1.0),
        # PnP batch (B=3), aligned matching, shared-probability attention in the injected blocks
        dict(name="chain_pnp_f4", B=3, C=16, heads=2, H=6, W=6, chunk_frames=[4, 4, 4],
             local_ratio=0.6, merge_global=True, global_ratio=0.6, align=True, injection=[500], t=500,
             rng_seed=6, data_seed=4321, frame_noise=1.0),
        # local only (BASELINE cfg-1 shape in miniature)
        dict(name="chain_local_f4", B=2, C=24, heads=3, H=6, W=6, chunk_frames=[4, 2],
             local_ratio=0.5, merge_global=False, global_ratio=0.5, align=False, injection=None, t=500,
             rng_seed=11, data_seed=5, frame_noise=1.0),
    ]
    out = []
    for cfg in cfgs:
        for attempt in range(200):
            cfg = dict(cfg, data_seed=cfg["data_seed"] + 1000 * attempt)
            c32, w32, names, rng_state = run_chain(torch.float32, cfg, weights_seed=2024)
            c64, _, _, _ = run_chain(torch.float64, cfg, weights_seed=2024)
            ok = True
            for a, b in zip(c32, c64):
                if len(a["trace"]) != len(b["trace"]):
                    ok = False
                    break
                for ta, tb in zip(a["trace"], b["trace"]):
                    for n in ("unm_idx", "src_idx", "dst_idx"):
                        ok &= np.array_equal(ta[n], tb[n])
                    ok &= min(tb["margins"]) >= MARGIN
            if ok:
                print(cfg["name"], "screened after", attempt + 1, "attempts")
                break
        else:
            raise RuntimeError(f"could not screen chain case {cfg['name']}")
        data = {"cfg_json": json.dumps({k: v for k, v in cfg.items()}), "rng_state": rng_state.numpy(),
                "block_names": np.array(names)}
        for k, v in w32.items():
            data["w/" + k] = v.numpy()
        for ck, ch in enumerate(c32):
            data[f"c{ck}/latent"] = ch["latent"].numpy()
            for r, mg in zip(ch["records"], ch["merged"]):
                bi = r["block"]
                data[f"c{ck}/b{bi}/hidden"] = r["hidden"].numpy()
                data[f"c{ck}/b{bi}/out"] = r["out"].numpy()
                data[f"c{ck}/b{bi}/merged"] = mg.numpy()
            for k, v in ch["global_tokens"].items():
                if v is not None and k != "":
                    data[f"c{ck}/gt/{k}"] = v.numpy()
            for ti, tr in enumerate(ch["trace"]):
                data[f"c{ck}/t{ti}/kind"] = np.array(tr["kind"])
                for n in ("unm_idx", "src_idx", "dst_idx"):
                    data[f"c{ck}/t{ti}/{n}"] = tr[n][..., 0].astype(np.int32)
        out.append((cfg["name"], data))
    return out


# --------------------------------------------------------------------------------------------
# 3. attention fixtures through the reference's sa_forward
# --------------------------------------------------------------------------------------------
def gen_attention_cases():
    shapes = [  # (B, heads, M, d, inject)
        (2, 2, 333, 80, False), (2, 4, 1332, 40, False), (2, 3, 1000, 64, False), (3, 1, 640, 160, True),
        (3, 4, 257, 40, True)]
    cases = []
    for si, (B, h, M, d, inject) in enumerate(shapes):
        C = h * d
        torch.manual_seed(300 + si)
        unet = StandInUNet(C, h)
        pipe = Pipe(unet)
        ref_pnp.register_attention_control(pipe, [500] if inject else None, B)
        attn = unet.up_blocks[3].attentions[0].transformer_blocks[0].attn1
        for p in attn.parameters():
            # fp16-representable weights: the fixture stores them as fp16
            p.copy_((torch.randn_like(p) * (0.1 if p.ndim == 1 else 1.5 * p.shape[-1] ** -0.5)).half().float())
        attn.t = 500
        g = torch.Generator().manual_seed(900 + si)
        x = torch.randn(B, M, C, generator=g)
        x = x.half().float()       # fp16-representable input
        y = attn.forward(x)
        # store only a strided subset of output rows for the big ones
        sel = np.unique(np.concatenate([np.arange(0, M, max(1, M // 48)), [M - 1]]))
        cases.append({"kind": "attn", "B": B, "heads": h, "M": M, "d": d, "inject": inject, "x": x.numpy().astype(np.float16),
                      "wq": attn.to_q.weight.numpy().astype(np.float16), "wk": attn.to_k.weight.numpy().astype(np.float16),
                      "wv": attn.to_v.weight.numpy().astype(np.float16), "wo": attn.to_out[0].weight.numpy().astype(np.float16),
                      "bo": attn.to_out[0].bias.numpy().astype(np.float16),
                      "rows": sel, "y_rows": y[:, sel, :].numpy()})
    return cases


# --------------------------------------------------------------------------------------------
# 4. full-size planted case (cfg-2 top block level 1): stored as sha256 of the index arrays
# --------------------------------------------------------------------------------------------
def gen_planted(full):
    cases = []
    for name, Ns, Nd, C, F in ([("planted_small", 3072, 1024, 64, 4)] +
                               ([("planted_cfg2_top_l1", 49152, 16384, 320, 16)] if full else [])):
        a, b = planted_inputs(Ns, Nd, C, seed=42)
        # arrange as a joined chunk: frames 0..F-1, dst frames are those with f % 4 == randf
        tnum = (Ns + Nd) // F
        gen = fork_generator(123)
        randf = int(torch.randint(0, 4, torch.Size([1]), generator=fork_generator(123)))
        x = np.empty((1, Ns + Nd, C), np.float32)
        frames = np.arange(Ns + Nd) // tnum
        is_dst = frames % 4 == randf
        x[0, is_dst] = b[0]
        x[0, ~is_dst] = a[0]
        m, u, ret = ref_merge.bipartite_soft_matching_randframe(torch.from_numpy(x), F, 0.5, 0, gen, 4, False)
        c = cells(m)
        idx = {n: np64(c[n])[0, :, 0].astype(np.int32) for n in ("unm_idx", "src_idx", "dst_idx")}
        cases.append({"kind": "planted", "name": name, "Ns": Ns, "Nd": Nd, "C": C, "F": F, "seed": 42,
                      "randf": randf, "ratio": 0.5,
                      **{n + "_sha256": hashlib.sha256(v.tobytes()).hexdigest() for n, v in idx.items()},
                      **{n + "_head": v[:16] for n, v in idx.items()}})
        print("planted", name, {k: v for k, v in cases[-1].items() if k.endswith("sha256")})
    return cases


def save_cases(fname, cases):
    flat = {}
    for i, c in enumerate(cases):
        for k, v in c.items():
            if k in ("m", "u"):
                continue
            if isinstance(v, torch.Tensor):
                v = v.numpy()
            flat[f"{i}/{k}"] = np.asarray(v)
    flat["n_cases"] = np.array(len(cases))
    np.savez_compressed(os.path.join(HERE, fname), **flat)
    print("wrote", fname, len(cases), "cases", os.path.getsize(os.path.join(HERE, fname)) // 1024, "KiB")


def main():
    full = "--no-full" not in sys.argv
    if "--only-attention" in sys.argv:
        save_cases("attention.npz", gen_attention_cases())
        return
    if "--only-chain" in sys.argv:
        for name, data in gen_chain_cases():
            np.savez_compressed(os.path.join(HERE, name + ".npz"), **data)
            print("wrote", name, os.path.getsize(os.path.join(HERE, name + ".npz")) // 1024, "KiB")
        return
    if "--only-planted" in sys.argv:
        save_cases("planted.npz", gen_planted(True))
        return
    save_cases("randframe.npz", gen_randframe_cases())
    save_cases("twos.npz", gen_2s_cases())
    save_cases("misc.npz", [gen_nan_case(), gen_survey_kat()])
    for name, data in gen_chain_cases():
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **data)
        print("wrote", name, os.path.getsize(os.path.join(HERE, name + ".npz")) // 1024, "KiB")
    save_cases("attention.npz", gen_attention_cases())
    save_cases("planted.npz", gen_planted(full))


if __name__ == "__main__":
    main()

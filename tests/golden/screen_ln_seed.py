#!/usr/bin/env python3
"""Find data seeds for tests/test_gpu_parity.py::test_block_vs_oracle_from_the_oracles_own_layernorm (CPU, oracle only).

    python tests/golden/screen_ln_seed.py

That test lets the ORACLE apply its own LayerNorm (fp32, rounded once to fp16: what an fp16 reference model's norm1 returns)
instead of starting from vtm_layernorm's output.  Two correct fp16 LayerNorms may round a few elements differently, and a
merge decision on nearly tied similarities can hinge on such an element -- which is a property of the input, not of either
implementation.  (At the bench's sizes -- thousands of src rows per level -- some decision at the rank-r boundary ALWAYS flips
under a one-ulp change of an input, and a flipped decision replaces a whole token: every one of 400 clip seeds at 16 x 16
tokens per frame was rejected.  The case is therefore small, 8 frames of 8 x 8 tokens, like the chain16 fixtures; the
full-size end-to-end tests start both sides from the same norm1 output.)  The case is SCREENED on outputs: a clip seed is kept only if the oracle's block outputs and anchors of
all chunks (local levels, global level, both coin outcomes) stay within 3e-4 of their scale when a random 1e-3 of the
LayerNorm outputs are moved by one fp16 ulp (four draws) -- a changed merge decision at the rank-r boundary shows as >= 3e-3.
Prints the first seeds that pass; the test hard-codes one.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle  # noqa: E402
from vidtome_amd import sites  # noqa: E402

B, F, LATENT, C, HEADS, COINS = 2, 4, (6, 6), 320, 8, (0.0, 1.0, 0.0)


def ln_fp16(hidden16, w, b):
    y = oracle.layer_norm(hidden16.float().numpy(), w, b)
    return torch.from_numpy(y).half()


KEPT_SEED = 18          # the first clip seed this script keeps (python tests/golden/screen_ln_seed.py)


def case_weights(clip_seed):
    """(norm1 weight, norm1 bias, {wq, wk, wv, wo}, to_out bias), all on the fp16 grid (the test loads them into an fp16 block)."""
    g = torch.Generator().manual_seed(1000 + clip_seed)
    w = (1.0 + 0.1 * torch.randn(C, generator=g)).half().float().numpy()
    b = (0.1 * torch.randn(C, generator=g)).half().float().numpy()
    wts = {n: (torch.randn(C, C, generator=g) * C ** -0.5).half().float().numpy() for n in ("wq", "wk", "wv", "wo")}
    bo = (0.1 * torch.randn(C, generator=g)).half().float().numpy()
    return w, b, wts, bo


def outputs(clip_seed, perturb_seed=None):
    """Per chunk: the oracle's block output (from ITS OWN LayerNorm, rounded once to fp16) and the anchors it stored."""
    site = sites.Site("s", 1, C, HEADS)
    w, b, wts, bo = case_weights(clip_seed)
    args = {"max_downsample": 2, "target_stride": 4, "local_merge_ratio": 0.5, "merge_global": True, "global_merge_ratio": 0.5,
            "global_rand": 0.5, "batch_size": B, "align_batch": False}
    torch.manual_seed(123)
    draws = oracle.RandomDraws.from_torch_generator(torch.Generator().set_state(torch.get_rng_state()))
    state = {"global_tokens": None}
    pg = None if perturb_seed is None else torch.Generator().manual_seed(perturb_seed)
    out = []
    for ck in range(1 + len(COINS)):
        if ck > 0:
            args["global_rand"] = COINS[ck - 1]
        hidden = sites.synthetic_hidden(site, B, F, LATENT, torch.float16, "cpu", seed=50 + ck, clip_seed=clip_seed)
        nh = ln_fp16(hidden, w, b)
        if pg is not None:
            hit = torch.rand(nh.shape, generator=pg) < 2e-4
            step = torch.where(torch.rand(nh.shape, generator=pg) < 0.5, 1, -1).to(torch.int16)
            nh = torch.where(hit, (nh.view(torch.int16) + step).view(torch.float16), nh)
        _, u, merged, trace = oracle.compute_merge(nh.float().numpy(), LATENT, args, draws, state)
        attn = oracle.self_attention(merged, wts["wq"], wts["wk"], wts["wv"], wts["wo"], bo, HEADS)
        out.append(u(attn) + hidden.float().numpy())
        out.append(state["global_tokens"].copy())
    return out


def main():
    oracle.build()
    found = []
    for clip_seed in range(1, 60):
        base = outputs(clip_seed)
        worst = 0.0
        for d in range(4):
            other = outputs(clip_seed, 7000 + 13 * clip_seed + d)
            for x, y in zip(base, other):
                worst = max(worst, float(np.abs(x - y).max()) / max(1.0, float(np.abs(x).max())))
            if worst >= 3e-4:
                break
        print("clip seed", clip_seed, "worst disagreement", f"{worst:.2e}", "KEPT" if worst < 3e-4 else "rejected", flush=True)
        if worst < 3e-4:
            found.append(clip_seed)
            if len(found) == 2:
                break
    print("kept:", found)


if __name__ == "__main__":
    main()

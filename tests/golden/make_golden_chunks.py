#!/usr/bin/env python3
"""Golden chunk lists from the reference's own Generator.get_chunks (generate.py:172-203).

generate.py imports diffusers-dependent helpers that are not installed here; only `get_chunks` is needed, so the
unavailable imports are stubbed in sys.modules and the method is called on a plain namespace object (it reads
self.chunk_size, self.merge_global, self.chunk_ord, self.perm_div only).  Run in the build container:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_chunks.py
"""
import os
import random
import sys
import types
from unittest import mock

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
for name in ("utils",):
    sys.modules[name] = mock.MagicMock()
import generate as ref_generate  # noqa: E402


def seed_everything(seed):     # utils/pnp_utils.py:6-10
    torch.manual_seed(seed)
    random.seed(seed)
    np.random.seed(seed)


def main():
    out = {}
    cases = []
    for ci, (flen, chunk_size, merge_global, chunk_ord) in enumerate([
            (32, 4, True, "mix-4"), (64, 16, True, "mix-4"), (17, 4, True, "mix"), (40, 8, True, "rand"),
            (40, 8, True, "seq"), (33, 4, False, "mix-4"), (5, 8, True, "mix-4"), (128, 8, True, "mix-2")]):
        me = types.SimpleNamespace(chunk_size=chunk_size, merge_global=merge_global, chunk_ord=chunk_ord)
        if "mix" in me.chunk_ord:                      # generate.py:87-89
            me.perm_div = float(me.chunk_ord.split("-")[-1]) if "-" in me.chunk_ord else 3.
            me.chunk_ord = "mix"
        seed_everything(123 + ci)
        for step in range(6):                          # six consecutive denoising steps from one seed
            chunks = ref_generate.Generator.get_chunks(me, flen)
            flat = np.concatenate([c.numpy() for c in chunks])
            lens = np.array([len(c) for c in chunks])
            out[f"{ci}/{step}/flat"] = flat.astype(np.int32)
            out[f"{ci}/{step}/lens"] = lens.astype(np.int32)
        cases.append((flen, chunk_size, int(merge_global), chunk_ord, 123 + ci))
    out["cases"] = np.array([f"{a}|{b}|{c}|{d}|{e}" for a, b, c, d, e in cases])
    np.savez_compressed(os.path.join(HERE, "chunks.npz"), **out)
    print("wrote chunks.npz", len(cases), "cases")


if __name__ == "__main__":
    main()


def ddim_cases():
    """CFG combine (generate.py:276-278) + DDIM update (generate.py:281-311) evaluated by the reference's own
    `pred_next_x` on a stand-in scheduler object (it reads timesteps, alphas_cumprod, final_alpha_cumprod)."""
    betas = torch.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=torch.float32) ** 2
    alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
    timesteps = torch.arange(981, -1, -20)          # 50 DDIM steps
    sched = types.SimpleNamespace(timesteps=timesteps, alphas_cumprod=alphas_cumprod,
                                  final_alpha_cumprod=torch.tensor(1.0))
    me = types.SimpleNamespace(scheduler=sched, guidance_scale=7.5)
    g = torch.Generator().manual_seed(99)
    out = {}
    k = 0
    for dtype in (torch.float32, torch.float16):
        for (i, inversion) in [(0, False), (17, False), (49, False), (5, True), (0, True)]:
            x = torch.randn(4, 4, 8, 8, generator=g).to(dtype)
            eu = torch.randn(4, 4, 8, 8, generator=g).to(dtype)
            ec = torch.randn(4, 4, 8, 8, generator=g).to(dtype)
            t = (reversed(timesteps) if inversion else timesteps)[i]
            eps = eu + me.guidance_scale * (ec - eu)                              # generate.py:277
            xn = ref_generate.Generator.pred_next_x.__wrapped__(me, x, eps, t, i, inversion=inversion) \
                if hasattr(ref_generate.Generator.pred_next_x, "__wrapped__") else \
                ref_generate.Generator.pred_next_x(me, x, eps, t, i, inversion=inversion)
            ts = reversed(timesteps) if inversion else timesteps
            apt = float(alphas_cumprod[t])
            if inversion:
                app = float(alphas_cumprod[ts[i - 1]]) if i > 0 else 1.0
            else:
                app = float(alphas_cumprod[ts[i + 1]]) if i < len(ts) - 1 else 1.0
            out[f"{k}/x"], out[f"{k}/eu"], out[f"{k}/ec"] = x.numpy(), eu.numpy(), ec.numpy()
            out[f"{k}/eps"], out[f"{k}/xn"] = eps.numpy(), xn.numpy()
            # the scalar coefficients exactly as pred_next_x derives them (0-dim fp32 tensor arithmetic,
            # generate.py:299-302); stored so that the test does not depend on the host's scalar pow
            a_t, a_p = torch.tensor(apt, dtype=torch.float32), torch.tensor(app, dtype=torch.float32)
            coef = [float(a_t ** 0.5), float((1 - a_t) ** 0.5), float(a_p ** 0.5), float((1 - a_p) ** 0.5)]
            out[f"{k}/meta"] = np.array([apt, app, float(inversion), 7.5] + coef, dtype=np.float64)
            k += 1
    out["n"] = np.array(k)
    np.savez_compressed(os.path.join(HERE, "ddim.npz"), **out)
    print("wrote ddim.npz", k, "cases")


if __name__ == "__main__":
    ddim_cases()

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

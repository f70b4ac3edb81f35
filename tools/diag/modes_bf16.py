import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from helpers import load_cases
from vidtome_amd import merge
for c in load_cases("modes.npz")[:1]:
    x = torch.from_numpy(c["x"]).cuda()
    torch.manual_seed(123)
    gen = torch.Generator(device="cpu").set_state(torch.get_rng_state())
    m, u, info = merge.bipartite_soft_matching_randframe(x, int(c["F"]), float(c["ratio"]), int(c["unm_pre"]), gen, 4, bool(c["align"]))
    for name, dt in (("f16", torch.float16), ("bf16", torch.bfloat16)):
        for mode in ("sum", "mean", "amax"):
            got = m(x.to(dt), mode=mode).cpu()
            want = torch.from_numpy(c[f"{name}/{mode}"]).view(dt)
            bad = (got.view(torch.int16) != want.view(torch.int16))
            print(name, mode, int(bad.sum()), "of", bad.numel())
            if bad.any():
                idx = bad.nonzero()[:5]
                for b, r, ch in idx.tolist():
                    print("   ", (b, r, ch), float(got[b, r, ch]), float(want[b, r, ch]), "x16:", float(x.to(dt)[b, 0, ch]))

#!/usr/bin/env python3
"""The WHOLE patched block against the reference -> tests/golden/fullblock16_*.npz.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_fullblock.py      (build container only: imports /root/reference)

The chain / chain16 fixtures drive the reference's `ToMeBlock.forward` on a stand-in block whose `attn2` is None and whose
feed-forward returns zeros, called positionally -- the self-attention segment (patch.py:139-169) is pinned to the reference,
the rest of the block (patch.py:171-199) only to fp32 PyTorch (missing 3).  Here the stand-in is a full SD block:

* `norm2` / `attn2` over a text conditioning (77 tokens, repeated over the frames like generate.py:245) -- attn2's forward is
  the reference's OWN attention arithmetic (`sa_forward`'s cross branch, utils/pnp_utils.py:47-95, registered on attn2 through
  a holder object; make_golden.run_chain), `norm3` / GEGLU `FeedForward` (Diffusers' published form: Linear C -> 8C,
  value * gelu(gate), Linear 4C -> C);
* the block is called by a stand-in `Transformer2DModel` with the FULL keyword set of patch.py:128-137
  (attention_mask, encoder_hidden_states, encoder_attention_mask, timestep, cross_attention_kwargs, class_labels);
* the reference runs in fp32 on the CPU; weights, hidden states, the conditioning and norm1's output lie on the fp16 grid
  (the model an fp16 run holds), and the cases are screened like chain16 (no merge decision may change in fp64 or when a
  random 2e-4 of norm1's fp16 outputs move by one ulp).

tests/test_gpu_parity.py::test_full_block_vs_reference_chain runs the fp16 model's patched forward (panel-GEMM and library
modes) on the recorded hidden states and holds block outputs and anchors to the recorded run.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (imports the reference)
from make_golden_chain16 import disagreement  # noqa: E402

CFGS = [
    # SD-1.5 top-block geometry in miniature: C = 320, 8 heads of 40; CFG batch, both coin outcomes, a single-frame chunk.
    # 8 x 8 latent -> 64 / 16 / 4 tokens per frame: the 4-token (un-merged, downsample 4) sites take the padded cross-attention
    dict(name="fullblock16_cfg_f4_d40", B=2, C=320, heads=8, H=8, W=8, chunk_frames=[4, 4, 1, 4],
         local_ratio=0.5, merge_global=True, global_ratio=0.5, align=False, injection=None, t=500,
         rng_seed=2, data_seed=9001, frame_noise=0.6, full=True, cond_tokens=77, cond_dim=128, keep_blocks=[0, 4, 8]),
    # SD-2.1's head dim 64, PnP batch 3 (aligned matching, shared probabilities in attn1; attn2 is never injected)
    dict(name="fullblock16_pnp_f4_d64", B=3, C=128, heads=2, H=8, W=8, chunk_frames=[4, 4, 4],
         local_ratio=0.6, merge_global=True, global_ratio=0.6, align=True, injection=[500], t=500,
         rng_seed=6, data_seed=9002, frame_noise=0.6, full=True, cond_tokens=77, cond_dim=64, keep_blocks=[0, 1, 4, 8]),
]


def main():
    only = sys.argv[1:]
    for base in CFGS:
        if only and base["name"] not in only:
            continue
        for attempt in range(400):
            cfg = dict(base, data_seed=base["data_seed"] + 1000 * attempt, fp16_grid=True, round_norm1=1)
            plain, w, names, rng_state = mg.run_chain(torch.float32, cfg, weights_seed=2025)
            worst = 0.0
            for kind in ("fp64", 2, 3, 4):
                other, _, _, _ = mg.run_chain(torch.float64 if kind == "fp64" else torch.float32,
                                              cfg if kind == "fp64" else dict(cfg, round_norm1=kind), weights_seed=2025)
                worst = max(worst, disagreement(plain, other))
                if worst >= 2e-3:
                    break
            if worst < 2e-3:
                print(cfg["name"], "kept after", attempt + 1, "attempts; worst disagreement of the screening runs:",
                      f"{worst:.2e}", flush=True)
                break
        else:
            raise RuntimeError("could not screen " + base["name"])
        data = {"cfg_json": json.dumps(cfg), "rng_state": rng_state.numpy(), "block_names": np.array(names),
                "screen_disagreement": np.array(worst)}
        # blocks are independent of each other (own generator fork, own anchors): the fixture keeps one block per token
        # resolution (+ the one decoder block PnP does not inject into); their 1-D parameters are stored, their matrices
        # are inputs.portable_weight(name, shape) -- rebuilt by the test, checked here
        keep = cfg["keep_blocks"]
        kept_names = [names[bi] for bi in keep]
        for k, v in w.items():          # on the fp16 grid: stored as fp16
            assert torch.equal(v, v.half().float()), k
            if not any(k.startswith(nm + ".") for nm in kept_names):
                continue
            if v.ndim == 2:
                from inputs import portable_weight
                assert np.array_equal(v.numpy(), portable_weight(k, tuple(v.shape))), k
            else:
                data["w/" + k] = v.half().numpy()
        for ck, ch in enumerate(plain):
            data[f"c{ck}/latent_shape"] = np.array(ch["latent"].shape)
            cond = ch["cond"].numpy()
            assert np.array_equal(cond, cond.astype(np.float16).astype(np.float32))
            data[f"c{ck}/cond"] = cond[::cond.shape[0] // cfg["B"]].astype(np.float16)   # one row per batch group (repeated over frames)
            for r in ch["records"]:
                bi = r["block"]
                if bi not in keep:
                    continue
                hid = r["hidden"].numpy()
                assert np.array_equal(hid, hid.astype(np.float16).astype(np.float32))
                data[f"c{ck}/b{bi}/hidden"] = hid.astype(np.float16)
                data[f"c{ck}/b{bi}/out"] = r["out"].numpy()
            for k, v in ch["global_tokens"].items():
                if v is not None and k in kept_names:
                    data[f"c{ck}/gt/{k}"] = v.numpy().astype(np.float16)   # row copies of norm1's (fp16-grid) output: exact
                    assert np.array_equal(data[f"c{ck}/gt/{k}"].astype(np.float32), v.numpy())
        path = os.path.join(HERE, cfg["name"] + ".npz")
        np.savez_compressed(path, **data)
        print("wrote", path, os.path.getsize(path) // 1024, "KiB", flush=True)


if __name__ == "__main__":
    main()

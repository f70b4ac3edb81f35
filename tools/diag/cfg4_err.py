"""Diagnostic (GPU): error distribution of the cfg-4 top / mid block outputs against the oracle on sampled rows."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import vidtome_amd
from vidtome_amd import patch as vpatch, sites
from vidtome_amd.utils import join_frame
from oracle import oracle
oracle.build()
dev = torch.device("cuda:0")
B, F, latent = 2, 8, (64, 64)
sl = [sites.Site("top", 1, 320, 8), sites.Site("mid", 2, 640, 8)]
for dtype in (torch.float16, torch.float32):
    unet = sites.SiteUNet(sl, seed=0).to(device=dev, dtype=dtype)
    vidtome_amd.apply_patch(unet, local_merge_ratio=0.5, merge_global=True, global_merge_ratio=0.5, batch_size=B)
    unet.set_size(latent)
    torch.manual_seed(123)
    seen = {}
    orig = vpatch.compute_merge
    def rec(module, x, info, **kw):
        res = orig(module, x, info, **kw)
        seen[id(module)] = res[0].plan
        return res
    vpatch.compute_merge = rec
    for ck in range(2):
        hid = [sites.synthetic_hidden(s, B, F, latent, dtype, dev, seed=101 * ck + i) for i, s in enumerate(sl)]
        with torch.no_grad():
            outs = sites.run_segment_pass(unet, hid)
        for bi, blk in enumerate(unet.blocks):
            plan = seen[id(blk)]
            a = blk.attn1
            f32 = lambda t: t.detach().float().cpu().numpy()
            wq, wk, wv, wo, bo = f32(a.to_q.weight), f32(a.to_k.weight), f32(a.to_v.weight), f32(a.to_out[0].weight), f32(a.to_out[0].bias)
            merged = f32(plan.merged[:, :plan.M])
            L = plan.inv.shape[1]
            g = np.random.default_rng(ck)
            idx = np.unique(g.integers(0, L, 256))
            inv = plan.inv.cpu().numpy()
            m = inv[:, idx]
            k, v = merged @ wk.T, merged @ wv.T
            q = np.stack([merged[b, m[b]] for b in range(B)]) @ wq.T
            o = oracle.attention_qkv(np.ascontiguousarray(q), np.ascontiguousarray(k), v, a.heads)
            hj, oj = f32(join_frame(hid[bi], F)), f32(join_frame(outs[bi], F))
            attn_ref = o @ wo.T + bo
            ref = attn_ref + hj[:, idx]
            d = np.abs(oj[:, idx] - ref)
            sc = max(1.0, np.abs(ref).max())
            # logits scale
            qh = q.reshape(B, -1, a.heads, q.shape[-1] // a.heads)
            print(f"{dtype} chunk {ck} block {bi} M={plan.M}: scale {sc:.2f} |attn| max {np.abs(attn_ref).max():.2f} err max {d.max():.5f} "
                  f"({d.max() / sc:.2e} of scale) p99.9 {np.quantile(d, 0.999):.5f} mean {d.mean():.6f} frac>1e-3*scale {(d > 1e-3 * sc).mean():.2e}", flush=True)
    vpatch.compute_merge = orig
    vidtome_amd.remove_patch(unet)

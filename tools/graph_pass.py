#!/usr/bin/env python3
"""MEASURE what a HIP graph buys for one hot-path pass ("measure one site pass captured as a hipGraph
per coin outcome instead of reasoning about it").

    python tools/graph_pass.py [--sites all|top|mid] [--iters 20] [--data corr01]

A pass over the chosen sites is run (a) eagerly -- ctypes launches on torch's current stream, the product path -- and (b)
captured once into a torch.cuda.CUDAGraph (stream capture: every vtm_* launch, memset and device copy of the pass becomes
a graph node; the host-side draws of the block generators are baked in, i.e. ONE graph per (randf, randf, coin) outcome)
and replayed.  Both are timed between synchronisations over `--iters` repetitions.  The replay executes the same kernels
with the same arguments on the same buffers, so the difference is launch overhead + dispatch gaps, nothing else.
Shapes do not depend on the draws at F = 16 (the frame partition is 12 / 4 and 3 / 1 whatever randf is), so a product
implementation would need 4 x 4 x 2 graphs per site and chunk length; this tool only answers what ONE of them saves.
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import vidtome_amd  # noqa: E402
from vidtome_amd import sites  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sites", default="all", choices=["all", "top", "mid"])
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--data", default="corr01")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    B, F, LAT = 2, 16, (64, 64)
    sl = sites.sd15_sites()
    if a.sites == "top":
        sl = [s for s in sl if s.downsample == 1][:1]
    elif a.sites == "mid":
        sl = [s for s in sl if s.downsample == 2][:1]
    unet = sites.SiteUNet(sl, seed=0).to(device=dev, dtype=torch.float16)
    vidtome_amd.apply_patch(unet, local_merge_ratio=0.5, merge_global=True, global_merge_ratio=0.5, batch_size=B)
    unet.set_size(LAT)
    torch.manual_seed(123)
    stream = sites.ClipStream(unet, sl, B, F, LAT, torch.float16, dev, regime=a.data, gen_device=dev, reseed=False)
    stream.populate()
    for c in range(3):
        stream.step(c)
    torch.cuda.synchronize()

    def timed(fn, n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            fn(i)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    # the pass that will be captured: chunk index 3 (frame set 0), anchors = whatever pass 2 left
    anchors0 = [getattr(b, "global_tokens", None) for b in unet.blocks]
    gens0 = [b.generator.get_state() if hasattr(b, "generator") else None for b in unet.blocks]

    def restore():
        for b, t, g in zip(unet.blocks, anchors0, gens0):
            b.global_tokens = t
            if g is not None:
                b.generator.set_state(g)

    def eager(_):
        restore()                      # the same draws and the same anchors every time: the SAME pass as the graph's
        stream.step(3)
    eager(0)
    ms_eager = timed(eager, a.iters)

    restore()
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g):
            stream.step(3)
    except Exception as e:  # noqa: BLE001
        print(f"graph capture failed: {type(e).__name__}: {e}")
        print(f"eager pass over {len(sl)} site(s): {ms_eager:.3f} ms")
        return
    g.replay()
    ms_graph = timed(lambda _: g.replay(), a.iters)
    print(f"sites={a.sites} ({len(sl)}) data={a.data}: eager {ms_eager:.3f} ms per pass, graph replay {ms_graph:.3f} ms "
          f"(-{ms_eager - ms_graph:.3f} ms, {100 * (1 - ms_graph / ms_eager):.1f} %)")


if __name__ == "__main__":
    main()

"""Chunk-parallel runs of the HIP path (vidtome_amd/chunk_parallel.py wired into patch.compute_merge):

* cfg-4 (BASELINE.json configs[3]: 64 frames of 512x512 in 8 chunks of 8 frames, one chunk per GPU): consecutive
  F = 8 chunks of a top (N = 4096, C = 320) and a mid (N = 1024, C = 640) SD-1.5 block through the exact ring mode on an
  in-process transport (three fake ranks) must equal the sequential run BIT FOR BIT;
* the same glue over a real process group: two processes (gloo, both on cuda:0), five chunks of unequal length.
"""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu
B = 2


def _make(sites_list, dev, rng_state, **patch_kw):
    """A patched SiteUNet whose block generators fork `rng_state` (what hook_tome_module does, patch.py:215-231)."""
    import vidtome_amd
    from vidtome_amd import sites
    unet = sites.SiteUNet(sites_list, seed=0).to(device=dev, dtype=torch.float16)
    kw = dict(local_merge_ratio=0.5, merge_global=True, global_merge_ratio=0.5, batch_size=B, target_stride=4,
              global_rand=0.5)
    kw.update(patch_kw)
    vidtome_amd.apply_patch(unet, **kw)
    for blk in unet.blocks:
        blk.generator = torch.Generator(device="cpu").set_state(rng_state)
    return unet


def _hiddens(sites_list, frames, latent, dev, step, chunk):
    from vidtome_amd import sites
    return [sites.synthetic_hidden(s, B, frames, latent, torch.float16, dev, seed=7919 * step + 101 * chunk + i)
            for i, s in enumerate(sites_list)]


def _sequential(sites_list, steps, latent, dev, rng_state, oracle=None):
    """The reference order on one device: chunks one after the other, anchors reset after every step.  With `oracle` every
    chunk's block outputs are also compared with the CPU oracle on sampled token positions (test_gpu_parity's
    _block_rows_vs_oracle: 1e-3 of the output scale), so the bit-equality of the chunk-parallel runs with this run is an
    equality with something pinned, not HIP against HIP."""
    import vidtome_amd
    from vidtome_amd import patch as vpatch
    from vidtome_amd import sites
    unet = _make(sites_list, dev, rng_state)
    unet.set_size(latent)
    out = {}
    seen = {}
    orig = vpatch.compute_merge

    def rec(module, x, info, **kw):
        res = orig(module, x, info, **kw)
        seen[id(module)] = getattr(res[0], "plan", None)
        return res

    with torch.no_grad():
        for s, frames in enumerate(steps):
            for ck, F in enumerate(frames):
                hid = _hiddens(sites_list, F, latent, dev, s, ck)
                vpatch.compute_merge = rec
                try:
                    outs = sites.run_segment_pass(unet, hid)
                finally:
                    vpatch.compute_merge = orig
                if oracle is not None:
                    from test_gpu_parity import _block_rows_vs_oracle
                    for bi, blk in enumerate(unet.blocks):
                        _block_rows_vs_oracle(oracle, blk, seen[id(blk)], hid[bi], outs[bi], F, n_rows=96, seed=ck, out_ulp=True)
                gts = [blk.global_tokens.clone() for blk in unet.blocks]
                out[(s, ck)] = ([o.clone() for o in outs], gts)
            vidtome_amd.update_patch(unet, global_tokens=None)                 # generate.py:233-236
    end = [blk.generator.get_state() for blk in unet.blocks]
    return out, end


def _cfg4_sites():
    from vidtome_amd import sites
    return [sites.Site("top", 1, 320, 8), sites.Site("mid", 2, 640, 8)]


def test_cfg4_ring_fake_ranks_equal_sequential(oracle):
    """cfg-4 chunk size (F = 8, 512x512) at a top and a mid block: 4 chunks on 3 fake ranks (the fourth wraps around
    to rank 0) through RingExchange == sequential compute_merge, bit for bit (block outputs, anchors, generators); the
    sequential run's block outputs are themselves held to the oracle on sampled rows (every chunk: first-chunk pass, both
    kinds of global level)."""
    from vidtome_amd import chunk_parallel as cp
    from vidtome_amd import sites
    dev = torch.device("cuda:0")
    sl, latent, steps = _cfg4_sites(), (64, 64), [[8, 8, 8, 8]]
    torch.manual_seed(123)
    rng_state = torch.get_rng_state()
    ref, ref_end = _sequential(sl, steps, latent, dev, rng_state, oracle)
    # sizes SURVEY.md 8d lists for cfg-4: top M = 18 432 (-> 27 648 with the global level), mid 4 608
    assert ref[(0, 0)][1][0].shape == (B, 18432, 320) and ref[(0, 1)][1][1].shape == (B, 4608, 640)

    W = 3
    fabric = cp.LocalTransport.fabric(W)
    ranks = []
    for r in range(W):
        unet = _make(sl, dev, rng_state)
        unet.set_size(latent)
        ex = cp.RingExchange(transport=fabric[r])
        cp.enable(unet, ex)
        ranks.append((unet, ex))
    with torch.no_grad():
        for s, frames in enumerate(steps):
            for _, ex in ranks:
                ex.begin_step(frames)
            for ck, F in enumerate(frames):
                unet, ex = ranks[ck % W]
                ex.begin_chunk(ck)
                outs = sites.run_segment_pass(unet, _hiddens(sl, F, latent, dev, s, ck))
                for bi, blk in enumerate(unet.blocks):
                    assert torch.equal(outs[bi], ref[(s, ck)][0][bi]), ("block output", ck, bi)
                    assert torch.equal(blk.global_tokens, ref[(s, ck)][1][bi]), ("anchors", ck, bi)
            for _, ex in ranks:
                ex.end_step()
    for unet, ex in ranks:
        for bi, blk in enumerate(unet.blocks):
            assert torch.equal(blk.generator.get_state(), ref_end[bi])
    # rank 0 ran chunk 0 (sent to rank 1) and chunk 3 (the last of the step: nothing to forward)
    assert ranks[0][1].bytes_sent == B * (18432 * 320 + 4608 * 640) * 2 + B * (18432 + 4608) * 4      # tokens + content ids


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


STEPS = [[3, 6, 4, 1, 4], [2, 4, 4]]
SMALL_LATENT = (16, 16)


def _worker(rank, world, port, mode, q, backend="gloo"):
    try:
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dev = torch.device("cuda", rank if backend == "nccl" else 0)     # RCCL: one GPU per rank
        torch.cuda.set_device(dev)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        import vidtome_amd
        from vidtome_amd import chunk_parallel as cp
        from vidtome_amd import sites
        sl = _cfg4_sites()
        torch.manual_seed(123)
        rng_state = torch.get_rng_state()
        why = []
        # the all-gather is a collective per round: whole rounds only
        steps = STEPS if mode != "allgather" else [f[:len(f) // world * world] for f in STEPS]
        if mode == "ring":
            ref, ref_end = _sequential(sl, steps, SMALL_LATENT, dev, rng_state)
        else:
            # parallel anchors: the reference is the one-rank run of the same semantics (in-process transport)
            ref = {}
            u1 = _make(sl, dev, rng_state)
            u1.set_size(SMALL_LATENT)
            e1 = cp.NeighbourExchange(transport=cp.LocalTransport.fabric(1)[0])
            cp.enable(u1, e1)
            with torch.no_grad():
                for s, frames in enumerate(steps):
                    def run1(i, s=s, frames=frames):
                        outs = sites.run_segment_pass(u1, _hiddens(sl, frames[i], SMALL_LATENT, dev, s, i))
                        ref[(s, i)] = ([o.clone() for o in outs], [b.global_tokens.clone() for b in u1.blocks])
                    cp.run_step(u1, e1, frames, run1)
            ref_end = [b.generator.get_state() for b in u1.blocks]
        unet = _make(sl, dev, rng_state)
        unet.set_size(SMALL_LATENT)
        ex = cp.AnchorExchange(mode)
        cp.enable(unet, ex)
        with torch.no_grad():
            for s, frames in enumerate(steps):
                def run(i, s=s, frames=frames):
                    outs = sites.run_segment_pass(unet, _hiddens(sl, frames[i], SMALL_LATENT, dev, s, i))
                    for bi, blk in enumerate(unet.blocks):
                        if not torch.equal(outs[bi], ref[(s, i)][0][bi]):
                            why.append(f"block output s{s} c{i} b{bi}")
                        if not torch.equal(blk.global_tokens, ref[(s, i)][1][bi]):
                            why.append(f"anchors s{s} c{i} b{bi}")
                mine = cp.run_step(unet, ex, frames, run)
                if mine != list(range(rank, len(frames), world)):
                    why.append("assignment")
                if any(b.global_tokens is not None for b in unet.blocks):
                    why.append("anchors not reset")
        for bi, blk in enumerate(unet.blocks):
            if not torch.equal(blk.generator.get_state(), ref_end[bi]):
                why.append(f"generator b{bi}")
        if ex.bytes_received == 0:
            why.append("nothing received")
        torch.cuda.synchronize()
        q.put((rank, not why, "; ".join(why)))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, False, traceback.format_exc()))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["ring", "neighbour", "allgather"])
def test_two_processes_one_gpu_through_compute_merge(mode):
    """patch.compute_merge's exchange glue with the HIP kernels over a real process group (gloo; both ranks on
    cuda:0): ring == the sequential run bit for bit; neighbour / all-gather == the one-rank run of the same mode."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, mode, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, ok, err in results:
        assert ok, f"rank {rank} failed: {err}"


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL over xGMI)")
@pytest.mark.parametrize("mode", ["ring", "neighbour", "allgather"])
def test_two_processes_two_gpus_rccl(mode):
    """SURVEY.md 8e, second clause: the real N-rank RCCL run equals the replay -- ring == the sequential run bit for bit
    (block outputs, anchors, generator states), neighbour / all-gather == the one-rank run of the same semantics."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, mode, q, "nccl")) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, ok, err in results:
        assert ok, f"rank {rank} failed: {err}"

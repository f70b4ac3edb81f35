"""Chunk scheduler against chunk lists produced by the reference's own
Generator.get_chunks (tests/golden/make_golden_chunks.py)."""
import os
import random

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "chunks.npz")


def test_chunk_lists_match_reference():
    from vidtome_amd.scheduler import ChunkScheduler
    z = np.load(GOLDEN)
    for ci, desc in enumerate(z["cases"]):
        flen, chunk_size, mg, chunk_ord, seed = str(desc).split("|")
        sch = ChunkScheduler(int(chunk_size), bool(int(mg)), chunk_ord)
        torch.manual_seed(int(seed))
        random.seed(int(seed))
        np.random.seed(int(seed))
        for step in range(6):
            chunks = sch.get_chunks(int(flen))
            assert np.array_equal(np.concatenate([c.numpy() for c in chunks]), z[f"{ci}/{step}/flat"]), (ci, step)
            assert [len(c) for c in chunks] == z[f"{ci}/{step}/lens"].tolist(), (ci, step)
            assert sorted(np.concatenate([c.numpy() for c in chunks]).tolist()) == list(range(int(flen)))


def test_run_step_resets_anchors_and_rank_assignment():
    import vidtome_amd
    from standin import StandInUNet
    from vidtome_amd.scheduler import ChunkScheduler, assign_chunks_to_ranks, run_step
    unet = StandInUNet(16, 2)
    vidtome_amd.apply_patch(unet, merge_global=True)
    blocks = list(unet.blocks())
    seen = []

    def process(chunk):
        seen.append(chunk.tolist())
        for b in blocks:
            b.global_tokens = torch.zeros(1)        # what compute_merge leaves behind after a chunk

    np.random.seed(0)
    torch.manual_seed(0)
    chunks = run_step(unet, ChunkScheduler(4, True, "mix-4"), 18, process)
    assert [c.tolist() for c in chunks] == seen
    assert all(b.global_tokens is None for b in blocks)              # generate.py:233-236
    ranks = assign_chunks_to_ranks(chunks, 2)
    assert sorted(ranks[0] + ranks[1]) == list(range(len(chunks))) and ranks[0][0] == 0 and ranks[1][0] == 1
    vidtome_amd.remove_patch(unet)

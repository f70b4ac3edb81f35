"""Chunk scheduling of a denoising step and the anchor-token lifetime -- the caller side of the hot path
(reference behaviour: generate.py:172-203 `get_chunks`, :205-224 the chunk loop of `ddim_sample`, :233-236
`post_iter`).

A step's schedule is fully described by three random draws, so this module separates them from the index
arithmetic:

    StepDraws      what is random:  first chunk length, walk direction, a permutation of the chunk ids
    cut_frames()   what is not:     [0, flen) cut into the (start, stop) frame ranges the draws imply
    visit_order()                   the order in which those ranges are processed ("seq" / "rand" / "mix-k")

`ChunkScheduler.draw()` takes the draws from the SAME global generators in the SAME sequence as the reference
(`np.random.randint`, `np.random.rand`, then `torch.randperm` on the number of chunks the first draw produced), so a
process seeded like the reference's `seed_everything` walks the same schedule; tests/golden/chunks.npz (recorded from
the reference's own `Generator.get_chunks`) pins that.  With global merging the order matters: the per-block anchor
tokens flow from one chunk to the next (patch.py:59-82) and die at the end of the step.  Because a schedule is plain
data it can be drawn once (rank 0) and broadcast, or replayed, by the chunk-parallel runner (chunk_parallel.py).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np
import torch


@dataclass(frozen=True)
class StepDraws:
    first_len: int                       # length of the first chunk, 1..chunk_size      (generate.py:176)
    backwards: bool                      # walk the video from its end                   (generate.py:179-180)
    perm: Optional[Tuple[int, ...]]      # permutation of the chunk ids, None for "seq"  (generate.py:187,190)


def cut_frames(flen: int, first_len: int, chunk_size: int) -> List[Tuple[int, int]]:
    """Frame ranges of one step: a first chunk of `first_len` frames, then runs of `chunk_size`, the last one as
    short as what is left.  A first chunk that already covers the video leaves a single range."""
    head = min(first_len, flen)
    return [(0, head)] + [(s, min(s + chunk_size, flen)) for s in range(head, flen, chunk_size)]


def visit_order(n: int, mode: str, perm: Optional[Sequence[int]], perm_div: float) -> List[int]:
    """Processing order of n chunks.  "seq": as cut; "rand": the permutation; "mix": the first n / perm_div entries of
    the permutation are visited first, the remaining chunks follow in ascending or descending order -- whichever end
    lies closer to the last randomly visited chunk (generate.py:189-199)."""
    if mode == "rand":
        return list(perm)
    if mode != "mix":
        return list(range(n))
    n_rand = int(n / perm_div)
    lead, rest = list(perm[:n_rand]), sorted(perm[n_rand:])
    if lead and abs(rest[-1] - lead[-1]) < abs(rest[0] - lead[-1]):
        rest.reverse()
    return lead + rest


class ChunkScheduler:
    """Configuration as in generate.py:70-71,86-89: `chunk_ord` is "seq", "rand", "mix" or "mix-<divisor>"."""

    def __init__(self, chunk_size: int, merge_global: bool = True, chunk_ord: str = "mix-4"):
        self.chunk_size = int(chunk_size)
        self.merge_global = bool(merge_global)
        name, _, div = chunk_ord.partition("-")
        self.perm_div = float(div) if ("mix" in chunk_ord and div) else 3.0
        self.chunk_ord = "mix" if "mix" in chunk_ord else name

    def n_chunks(self, flen: int, first_len: int) -> int:
        return len(cut_frames(flen, first_len, self.chunk_size))

    def draw(self, flen: int) -> StepDraws:
        """The step's draws, consumed from the global numpy / torch generators in the reference's sequence.  The
        permutation is only drawn when the order matters (global merging) and the mode uses one."""
        first_len = int(np.random.randint(0, self.chunk_size)) + 1
        backwards = bool(np.random.rand() > 0.5)
        perm = None
        if self.merge_global and self.chunk_ord in ("rand", "mix"):
            perm = tuple(torch.randperm(self.n_chunks(flen, first_len)).tolist())
        return StepDraws(first_len, backwards, perm)

    def schedule(self, flen: int, draws: StepDraws) -> List[Tuple[int, int]]:
        """Pure function of the draws: the (start, stop) ranges in processing order."""
        spans = cut_frames(flen, draws.first_len, self.chunk_size)
        if draws.backwards:
            spans.reverse()
        if not self.merge_global:                    # the order only matters for the anchor chain
            return spans
        return [spans[i] for i in visit_order(len(spans), self.chunk_ord, draws.perm, self.perm_div)]

    def get_chunks(self, flen: int) -> List[torch.Tensor]:
        """Same result as the reference's `Generator.get_chunks`: frame-index tensors in processing order."""
        return [torch.arange(a, b) for a, b in self.schedule(flen, self.draw(flen))]


def run_step(model, scheduler: ChunkScheduler, n_frames: int,
             process_chunk: Callable[[torch.Tensor], None], streams: Optional[Sequence] = None) -> List[torch.Tensor]:
    """One denoising step's chunk loop (generate.py:215-219) followed by the anchor reset of `post_iter`
    (generate.py:233-236): anchors live for exactly one step.

    ``streams`` (round 6, optional): a few `torch.cuda.Stream`s -- chunk i is then ISSUED on streams[i % len(streams)], in the
    reference's order, and chunk i + 1 waits on the device, block by block, for the anchors chunk i leaves behind
    (patch.mark_anchors_ready / await_anchors): same results as the sequential loop, the dispatch gaps and small launches of one
    chunk hidden behind the other's big kernels (two streams: +15 % chunk-steps/s at cfg-2 on one MI355X).  `process_chunk`
    must not synchronise the device and whatever it keeps of a chunk (the noise prediction) is the caller's to order: the
    streams are joined before this returns."""
    from . import patch
    chunks = scheduler.get_chunks(n_frames)
    if streams:
        first = torch.cuda.current_stream()
        for st in streams:
            st.wait_stream(first)                  # the chunks see everything the caller enqueued so far
        for i, chunk in enumerate(chunks):
            with torch.cuda.stream(streams[i % len(streams)]):
                process_chunk(chunk)
        for st in streams:
            first.wait_stream(st)                  # ... and the caller's stream sees every chunk's results
    else:
        for chunk in chunks:
            process_chunk(chunk)
    if scheduler.merge_global:
        patch.update_patch(model, global_tokens=None)
    return chunks


def assign_chunks_to_ranks(chunks: Sequence, world: int) -> List[List[int]]:
    """Chunk-parallel mapping used by chunk_parallel.py: rank r processes chunks r, r + world, ... of the schedule,
    so the predecessor of a chunk in the anchor chain always lives on rank (r - 1) mod world (rank world-1 hands
    over to rank 0 for the next round)."""
    return [list(range(r, len(chunks), world)) for r in range(world)]

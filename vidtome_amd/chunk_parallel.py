"""Chunk-parallel execution of the global-merge anchor chain: one process per GPU, the chunks of a denoising step
dealt round-robin to the ranks (chunk i runs on rank i mod W).

In the reference the chunks of a step run sequentially on one device and the per-block anchor tokens flow
chunk -> chunk through ``module.global_tokens`` (generate.py:215-219, patch.py:59-82).  Local merging is independent
per chunk; only this hand-off couples ranks.  Three exchange modes, one class (`AnchorExchange`):

``ring``  (exact)      chunk i receives, point-to-point over ONE xGMI link, the anchors chunk i-1 produced at the
                       SAME block, right before its own global level, and forwards its updated anchors to the rank of
                       chunk i+1 -- rank W-1 hands over to rank 0 for the next round, so a step may have any number
                       of chunks.  A wavefront pipeline whose skew is one compute_merge per hop.  Reproduces the
                       sequential run bit for bit.
``neighbour``          every chunk merges against the LOCAL merged tokens of chunk i-1.  No serial chain: all ranks of a
                       round run concurrently.  A documented semantic deviation (anchors are "parallel", not chained).
                       The hand-over is split so that the bytes move WHILE the local levels run: when a block starts, every
                       rank ships its joined chunk (B, F N, C) to its successor and posts the receive of its predecessor's;
                       when its local levels are done it ships the composed local merge MAP
                       (B, M_local int32, a few hundred KB) the same way, and the receiver gathers the predecessor's local
                       tokens out of the pool it already holds.  Twice the bytes of sending the merged tokens -- on a link
                       that is otherwise idle -- but nothing is waited for where the anchors are consumed (sending the
                       merged tokens after the local levels leaves the whole transfer, 0.45 ms per cfg-2 top block, exposed).
                       VIDTOME_NEIGHBOUR_EARLY=0 selects that single-message form.
``allgather``          the same semantics with one RCCL all-gather per merging block (north_star's wording).  What is
                       gathered is the composed local merge MAP of every rank (B, M_local int32, padded to the round's
                       longest: a few hundred KB per rank), not the tokens: the joined chunk itself travels point-to-point to
                       the one rank that needs it when the block starts, exactly as in the neighbour mode, so the collective on
                       the critical path in front of the global level moves KBs (rounds 1-3 gathered the tokens: W - 1 times
                       44.6 MB per cfg-2 top block, of which a rank used one shard).

What makes this work without host round trips:

* The random draws of a block are host-side and depend only on the frame counts of the chunks seen so far, so every
  rank REPLAYS the draws of the chunks it does not process (`simulate_block_draws`): its own draws then equal the
  sequential run's, and it knows the merged length of every other chunk -- i.e. the SHAPE of what it is about to
  receive (the first chunk of a step has a random length, generate.py:176-178, so shapes differ between ranks).
  No shape header, no `.tolist()`.
* Receives are posted as soon as a block starts (before its local levels), sends as soon as the tensor exists; both
  are asynchronous (`isend` / `irecv`: RCCL runs them on its own stream, the compute stream only waits -- on the
  device -- right where the anchors are consumed).

The classes move tensors through a `Transport` (``torch.distributed``: backend "nccl" == RCCL on ROCm, "gloo" in the
CPU tests; or an in-process mailbox for the single-process N-rank replay of SURVEY.md 8e) and are independent of how
the merge itself is computed.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from . import _lib


# ----------------------------------------------------------------------------------------------------
# RNG replay
# ----------------------------------------------------------------------------------------------------
def simulate_block_draws(generator: torch.Generator, frames: int, tokens_per_frame: int, args: Dict,
                         has_anchors: bool, downsample_ok: bool = True) -> Dict[str, object]:
    """Advance ``generator`` exactly as ``compute_merge`` would for one chunk of ``frames`` frames at a block
    with ``tokens_per_frame`` tokens (patch.py:44-54,59-62; merge.py:57-58), without touching any tensor.
    Returns the draws and the sizes they imply (also used by tests)."""
    out = {"randf": [], "coin": None, "M_local": frames * tokens_per_frame}
    if not downsample_ok:
        return out
    n_cur, unm, curF, tsize = frames * tokens_per_frame, 0, frames, tokens_per_frame
    while curF > 1:
        ratio = args["local_merge_ratio"]
        tnum = (n_cur - unm) // curF
        if ratio <= 0:
            unm += tnum
        else:
            ts = min(args["target_stride"], curF)
            randf = int(torch.randint(0, ts, torch.Size([1]), generator=generator, device=generator.device))
            out["randf"].append(randf)
            Ns, Nd = _lib.partition_counts(n_cur, unm, tnum, ts, randf)
            r = min(Ns, int(Ns * ratio))
            unm += Ns - r
            n_cur = (Ns - r) + Nd
        curF = (n_cur - unm) // tsize
    out["M_local"] = n_cur
    if args["merge_global"] and has_anchors:
        out["coin"] = float(torch.rand(1, generator=generator, device=generator.device))
    return out


def replay_draws(generator: torch.Generator, chunk_frames: Sequence[int], tokens_per_frame: int, args: Dict,
                 first_has_anchors: bool = False) -> List[int]:
    """Consume the draws of a run of consecutive chunks (the ones other ranks process); returns their merged
    local lengths."""
    has, lens = first_has_anchors, []
    for f in chunk_frames:
        lens.append(simulate_block_draws(generator, f, tokens_per_frame, args, has)["M_local"])
        has = True
    return lens


# ----------------------------------------------------------------------------------------------------
# transports
# ----------------------------------------------------------------------------------------------------
class _Done:
    def __init__(self, tensor=None):
        self.tensor = tensor

    def wait(self):
        return self.tensor

    def done(self) -> bool:
        return True


class DistTransport:
    """``torch.distributed`` point-to-point + all-gather.  With RCCL the operations run on communicator streams;
    `wait()` makes the CURRENT stream wait for them (no host block).

    Every DIRECTED edge of the ring (rank r -> rank r + 1, and W-1 -> 0) gets a process group -- i.e. a communicator and
    a stream -- of its own.  The protocol needs that independence: rank W-1's successor is rank 0's chunk of the NEXT
    round, so its sends wait for a receiver that is a whole round behind; if they shared a stream with rank W-1's own
    receives (one communicator per rank, or with two ranks one per PAIR: 0 -> 1 and 1 -> 0), those receives would queue
    behind the waiting sends, rank W-2's sends behind them, and the ring would lock up within W-1 blocks.  (torch
    happens to keep one communicator per pair of ranks for un-batched point-to-point operations, which covers W > 2;
    the explicit edge groups do not depend on that.)

    gloo (the CPU test backend) moves host memory only: device tensors are staged through the host there -- a
    test-only path, it synchronises."""

    def __init__(self, group=None):
        if group is not None and dist.get_world_size(group) != dist.get_world_size():
            # the edge groups below are made with dist.new_group, which EVERY rank of the default group has to enter --
            # members of a sub-group constructing the transport on their own would wait for the others forever
            raise NotImplementedError("DistTransport needs the default (WORLD) group: its per-edge process groups are "
                                      "created collectively by all ranks (dist.new_group)")
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self._gloo = dist.get_backend(group) == "gloo"
        self._edge = {}
        if self.world > 1:
            ranks = list(range(dist.get_world_size())) if group is None else dist.get_process_group_ranks(group)
            for r in range(self.world):                       # collective: every rank constructs every edge group
                pair = sorted({ranks[r], ranks[(r + 1) % self.world]})
                self._edge[r] = dist.new_group(pair, backend=dist.get_backend(group))
            if not self._gloo:
                # RCCL creates a communicator at its first use, and that creation blocks the HOST until both ends have
                # arrived: do it here, edge by edge (an order in which nobody waits for a rank that waits for it),
                # instead of in the middle of the first blocks
                tok = torch.zeros(1, device=torch.device("cuda", torch.cuda.current_device()))
                for r in range(self.world):
                    nxt = (r + 1) % self.world
                    if self.rank == r:
                        dist.send(tok, dst=ranks[nxt], group=self._edge[r])
                    elif self.rank == nxt:
                        dist.recv(tok, src=ranks[r], group=self._edge[r])
                torch.cuda.synchronize()

    def close(self) -> None:
        """Destroy the per-edge process groups (communicators + streams).  Collective like their creation: every rank
        calls it, after its last transfer has completed."""
        for g in self._edge.values():
            try:
                dist.destroy_process_group(g)
            except Exception:
                pass
        self._edge = {}

    def _group_for(self, src: int, dst: int):
        if (src + 1) % self.world != dst:
            raise RuntimeError(f"DistTransport: {src} -> {dst} is not an edge of the ring")
        return self._edge[src]

    def _global(self, r: int) -> int:
        return r if self.group is None else dist.get_global_rank(self.group, r)

    def isend(self, tensor: torch.Tensor, dst: int):
        if self._gloo and tensor.is_cuda:
            tensor = tensor.cpu()
        work = dist.isend(tensor, dst=self._global(dst), group=self._group_for(self.rank, dst))

        class _Send:
            keep = tensor                                  # the buffer must outlive the transfer

            def wait(_self):
                work.wait()

            def done(_self) -> bool:
                return work.is_completed()
        return _Send()

    def irecv(self, shape, dtype, device, src: int):
        stage = self._gloo and torch.device(device).type == "cuda"
        buf = torch.empty(tuple(shape), dtype=dtype, device="cpu" if stage else device)
        work = dist.irecv(buf, src=self._global(src), group=self._group_for(src, self.rank))

        class _Recv:
            def wait(_self):
                work.wait()
                return buf.to(device) if stage else buf
        return _Recv()

    def all_gather(self, tensor: torch.Tensor) -> torch.Tensor:
        if self._gloo:
            src = tensor.cpu() if tensor.is_cuda else tensor
            parts = [torch.empty_like(src) for _ in range(self.world)]
            dist.all_gather(parts, src, group=self.group)
            return torch.stack(parts).to(tensor.device)
        out = torch.empty((self.world,) + tuple(tensor.shape), dtype=tensor.dtype, device=tensor.device)
        dist.all_gather_into_tensor(out, tensor, group=self.group)
        return out


class LocalTransport:
    """In-process mailbox: N "ranks" of one process executed one after the other in chunk order (the
    single-process N-fake-rank replay).  `LocalTransport.fabric(n)` returns the n endpoints of one fabric."""

    def __init__(self, rank: int, world: int, boxes: Dict):
        self.rank, self.world, self._boxes = rank, world, boxes

    @staticmethod
    def fabric(world: int) -> List["LocalTransport"]:
        boxes: Dict[Tuple[int, int], List[torch.Tensor]] = {}
        return [LocalTransport(r, world, boxes) for r in range(world)]

    def isend(self, tensor: torch.Tensor, dst: int):
        self._boxes.setdefault((self.rank, dst), []).append(tensor)
        return _Done()

    def irecv(self, shape, dtype, device, src: int):
        boxes, key = self._boxes, (src, self.rank)

        class _Recv:
            def wait(_self):
                q = boxes.get(key)
                if not q:
                    raise RuntimeError(f"LocalTransport: nothing was sent {key[0]} -> {key[1]} (run the fake ranks in "
                                       "chunk order)")
                t = q.pop(0)
                if tuple(t.shape) != tuple(shape) or t.dtype != dtype:
                    raise RuntimeError(f"LocalTransport: expected {tuple(shape)} {dtype}, got {tuple(t.shape)} {t.dtype}")
                return t
        return _Recv()

    def all_gather(self, tensor):
        raise RuntimeError("LocalTransport has no collectives: use the 'ring' or 'neighbour' mode")

    def close(self) -> None:
        pass


# ----------------------------------------------------------------------------------------------------
# the exchange
# ----------------------------------------------------------------------------------------------------
class _BlockState:
    __slots__ = ("module", "tsize", "args", "drawn", "lens", "recv", "recv_ids", "carry", "steps_done", "pool")

    def __init__(self, module, tsize, args, steps_done=0):
        self.module, self.tsize, self.args = module, tsize, args
        self.steps_done = steps_done   # finished steps whose draws this block's generator has consumed in full
        self.drawn = 0          # chunks of the current step whose draws this block's generator has consumed
        self.lens: Dict[int, int] = {}    # chunk index -> merged local length (simulated or own)
        self.recv = None        # pending receive of the predecessor's tokens
        self.recv_ids = None    # ring mode: ... and of their content ids (patch.compute_merge's key folding)
        self.carry = None       # tokens of this rank's previous chunk (world == 1 / all-gather wrap-around)
        self.pool = None        # neighbour mode, early hand-over: the predecessor's joined chunk on its way here


def _with_positions(tokens: torch.Tensor, row_map: Optional[torch.Tensor], tsize: int) -> torch.Tensor:
    """Attach the token positions of an anchor set to its tensor (`_vtm_pos`, (B, M) int32: what compute_merge hands the
    matcher's seed kernel; never changes a result).  The anchors are rows `row_map` (B, M) of a joined chunk whose frames hold
    `tsize` tokens each (None: the chunk's rows themselves)."""
    if not tokens.is_cuda:
        return tokens
    B, M = tokens.shape[0], tokens.shape[1]
    if row_map is None:
        pos = (torch.arange(M, dtype=torch.int32, device=tokens.device) % tsize).expand(B, M).contiguous()
    else:
        pos = (row_map.to(torch.int32) % tsize).contiguous()
    tokens._vtm_pos = pos
    return tokens


class AnchorExchange:
    """See the module docstring.  Protocol, per denoising step::

        ex.begin_step(frames_per_chunk)            # the step's schedule (same on every rank)
        for i in ex.my_chunks():                   # i = rank, rank + W, ...
            ex.begin_chunk(i)
            model(chunk i)                         # patched blocks call begin_block / anchors_for / publish
        ex.end_step()                              # flush draws + outstanding sends; the caller resets the anchors
    """

    def __init__(self, mode: str = "ring", transport=None, group=None):
        if mode not in ("ring", "neighbour", "allgather"):
            raise ValueError(f"unknown exchange mode {mode!r}")
        self.mode = mode
        self.t = transport if transport is not None else DistTransport(group)
        self.rank, self.world = self.t.rank, self.t.world
        self.bytes_sent = 0
        self.bytes_received = 0
        self._frames: List[int] = []
        self._cur: Optional[int] = None
        self._blocks: Dict[str, _BlockState] = {}
        self._inflight: List[Tuple[object, torch.Tensor]] = []     # (work, tensor kept alive until the send is done)
        # A rank may sit out whole steps (fewer chunks than ranks; the first chunk of a step has a random length, so
        # the chunk count varies, generate.py:176-178).  A block it has not run yet has no _BlockState -- its token
        # count per frame is only known once the model reaches it -- so the schedules of the finished steps are kept
        # and a block replays the ones it missed when it first appears (begin_block).
        import os
        self.early = os.environ.get("VIDTOME_NEIGHBOUR_EARLY", "1") != "0"   # neighbour mode: joined chunk first, map later
        self._history: List[List[int]] = []
        self._modules: Dict[str, object] = {}                      # patched blocks registered by enable()
        self.where = "idle"                                        # last bookkeeping entry (bench.py's watchdog prints it)

    def close(self) -> None:
        """Release the transport's communicators (collective; after end_step)."""
        if self._cur is not None:
            raise RuntimeError("close inside a step (call end_step first)")
        self.t.close()

    # ---- step / chunk bookkeeping (host side only)
    @property
    def exact(self) -> bool:
        return self.mode == "ring"

    def begin_step(self, frames_per_chunk: Sequence[int]) -> None:
        if self._cur is not None:
            raise RuntimeError("begin_step inside a step (call end_step first)")
        self._frames = [int(f) for f in frames_per_chunk]
        if self.mode == "allgather" and len(self._frames) % self.world:
            raise ValueError("the all-gather mode is a collective per round: the step needs a multiple of "
                             f"{self.world} chunks (got {len(self._frames)}); use 'neighbour' or 'ring'")
        for st in self._blocks.values():
            st.drawn, st.lens, st.recv, st.recv_ids, st.carry, st.pool = 0, {}, None, None, None, None
        # The sequential run forks every block generator at the block's first forward of the FIRST step
        # (patch.py:215-231), i.e. right after that step's schedule was drawn.  A rank whose first chunk comes later
        # would fork later -- after further draws from the global generator (the next steps' schedules) -- and its
        # randf / coin stream would differ from the other ranks'.  Every registered block forks here instead: the same
        # point of the global stream on every rank.
        for module in self._modules.values():
            if not hasattr(module, "generator"):
                from .utils import init_generator
                module.generator = init_generator(None)

    def register(self, key: str, module) -> None:
        """A patched block this exchange serves (called by `enable`)."""
        self._modules[key] = module

    def reset(self) -> None:
        """Forget every block and every finished step (a new model / a new run on the same process group)."""
        if self._cur is not None:
            raise RuntimeError("reset inside a step (call end_step first)")
        self._blocks.clear()
        self._modules.clear()
        self._history.clear()

    def my_chunks(self) -> List[int]:
        return list(range(self.rank, len(self._frames), self.world))

    def begin_chunk(self, index: int) -> None:
        if index % self.world != self.rank:
            raise RuntimeError(f"chunk {index} belongs to rank {index % self.world}, not {self.rank}")
        self._cur = index

    def end_step(self) -> None:
        """Consume the draws of the chunks after this rank's last one (the next step continues the same generator
        streams, like the sequential run) and retire the sends."""
        n = len(self._frames)
        self.where = f"end_step: draining {len(self._inflight)} sends"
        self._history.append(self._frames)
        for st in self._blocks.values():
            self._replay(st, n)
            st.steps_done = len(self._history)
            if st.recv is not None:                       # a posted receive nobody consumed (cannot happen in a
                st.recv.wait()                            # well-formed step; drain it rather than leak it)
                st.recv = None
            if st.recv_ids is not None:
                st.recv_ids.wait()
                st.recv_ids = None
        for work, _ in self._inflight:
            if work is not None:
                work.wait()
        self._inflight.clear()
        self._cur = None
        self.where = "idle (step finished)"

    def _replay(self, st: _BlockState, upto: int) -> None:
        """Advance the block's generator over chunks [st.drawn, upto) that other ranks process."""
        while st.drawn < upto:
            c = st.drawn
            sim = simulate_block_draws(st.module.generator, self._frames[c], st.tsize, st.args, has_anchors=c > 0)
            st.lens[c] = sim["M_local"]
            st.drawn += 1

    def _catch_up(self, st: _BlockState) -> None:
        """Consume the draws of the finished steps this block took no part in on this rank (it sat them out)."""
        while st.steps_done < len(self._history):
            replay_draws(st.module.generator, self._history[st.steps_done], st.tsize, st.args)
            st.steps_done += 1

    def _owner(self, chunk: int) -> int:
        return chunk % self.world

    # ---- called by patch.compute_merge
    def begin_block(self, module, key: str, fsize: int, tsize: int, args: Dict, like: torch.Tensor) -> None:
        """Before the block's first draw: bring its generator to where the sequential run would be, and post the
        receive of the predecessor chunk's tokens (their shape follows from the replayed draws)."""
        i = self._cur
        if i is None:
            raise RuntimeError("chunk-parallel exchange: begin_chunk() was not called")
        if fsize != self._frames[i]:
            raise RuntimeError(f"chunk {i} has {fsize} frames, the step schedule says {self._frames[i]}")
        st = self._blocks.get(key)
        if st is None:
            st = self._blocks[key] = _BlockState(module, tsize, args)
        elif st.module is not module:                      # the block object was replaced: its generator is its own
            st = self._blocks[key] = _BlockState(module, tsize, args, len(self._history))
        st.tsize, st.args = tsize, args
        self._catch_up(st)
        self._replay(st, i)
        st.drawn = i + 1                                   # compute_merge itself makes chunk i's draws
        self.where = f"chunk {i} block {key}: begin_block ({self.mode}; predecessor rank {self._owner(i - 1)}, " \
                     f"successor rank {self._owner(i + 1)})"
        if not args["merge_global"]:
            return
        if self.mode in ("neighbour", "allgather"):
            if self.world > 1 and (self.early or self.mode == "allgather"):
                # ship the joined chunk now, receive the predecessor's: both overlap this block's local levels
                n = len(self._frames)
                send = like if i + 1 < n else None
                spec = ((like.shape[0], self._frames[i - 1] * tsize, like.shape[2]), like.dtype, like.device) if i > 0 else None
                st.pool = self._exchange(send, self._owner(i + 1), spec, self._owner(i - 1))
            return                                         # (single-message form: everything happens in anchors_for)
        if i == 0:
            return
        src = self._owner(i - 1)
        if src == self.rank:                               # world == 1: the predecessor ran here
            return
        B = like.shape[0]
        st.recv = self.t.irecv((B, st.lens[i - 1], like.shape[2]), like.dtype, like.device, src)
        # ... and their content ids (equal id = identical rows; -1 = none): the sequential run folds the anchors' exact copies
        # into one attention key each (patch.MergePlan.key_fold), so the exact mode has to know them too.  Always sent --
        # a second message whose presence depended on the predecessor's coin would have to be predicted here
        st.recv_ids = self.t.irecv((B, st.lens[i - 1]), torch.int32, like.device, src)

    def anchors_for(self, key: str, local_tokens_fn: Callable[[], torch.Tensor], like: torch.Tensor,
                    local_map: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
        """The tokens chunk i merges against at this block (None for the first chunk of a step).  The parallel modes
        first publish this chunk's own local merged tokens (``local_tokens_fn()``; ``local_map`` (B, M_local) int32 = the
        rows of ``like`` they are, None when the chunk has no local level -- what the early hand-over ships instead)."""
        i, st = self._cur, self._blocks[key]
        n = len(self._frames)
        self.where = f"chunk {i} block {key}: anchors_for ({self.mode}): waiting for rank {self._owner(i - 1)}'s tokens / " \
                     f"handing over to rank {self._owner(i + 1)}"
        if self.mode == "ring":
            if i == 0:
                return None
            if st.recv is None:
                return st.carry                            # world == 1
            got, st.recv = st.recv.wait(), None
            ids, st.recv_ids = st.recv_ids.wait(), None
            self.bytes_received += got.numel() * got.element_size() + ids.numel() * 4
            if got.is_cuda:
                from .patch import tag_content_ids
                tag_content_ids(got, ids)
            return got
        if self.mode == "neighbour" and self.world > 1:
            B, C = like.shape[0], like.shape[2]
            nxt, prv = self._owner(i + 1), self._owner(i - 1)
            if self.early:
                # second half of the hand-over: the composed local merge map (None: the chunk has no local level, its
                # local tokens ARE its joined chunk -- the receiver knows that from the schedule)
                prev_has_map = i > 0 and self._has_local_levels(self._frames[i - 1], st.args)
                send = local_map.contiguous() if (i + 1 < n and local_map is not None) else None
                spec = ((B, st.lens[i - 1]), torch.int32, like.device) if prev_has_map else None
                maps = self._exchange(send, nxt, spec, prv)
                st.lens[i] = like.shape[1] if local_map is None else local_map.shape[1]
                pool, st.pool = (st.pool.wait() if st.pool is not None else None), None
                got_map = maps.wait() if maps is not None else None
                if i == 0:
                    return None
                self.bytes_received += pool.numel() * pool.element_size() + (0 if got_map is None else got_map.numel() * 4)
                got = pool if got_map is None else _lib.gather_rows(pool, None, got_map) if pool.is_cuda else \
                    torch.gather(pool, 1, got_map.long()[:, :, None].expand(-1, -1, C))
                return _with_positions(got, got_map, st.tsize)
            local = local_tokens_fn().contiguous()
            st.lens[i] = local.shape[1]
            spec = ((B, st.lens[i - 1], C), like.dtype, like.device) if i > 0 else None
            h = self._exchange(local if i + 1 < n else None, nxt, spec, prv)
            got = h.wait() if h is not None else None
            if got is not None:
                self.bytes_received += got.numel() * got.element_size()
            return got
        if self.world == 1:                                # the predecessor ran here: hand over in place (both parallel modes)
            local = _with_positions(local_tokens_fn().contiguous(), local_map, st.tsize)
            st.lens[i] = local.shape[1]
            got, st.carry = st.carry, local
            return got if i > 0 else None
        # all-gather: the joined chunks went point-to-point when the block started (begin_block); the collective moves the
        # composed local merge MAPS.  Chunk lengths differ between ranks -> pad to the round's maximum (known from the replay)
        B, C = like.shape[0], like.shape[2]
        my_map = local_map if local_map is not None else \
            torch.arange(like.shape[1], dtype=torch.int32, device=like.device).expand(B, -1)   # no local level: every row
        Ml = my_map.shape[1]
        st.lens[i] = Ml
        first = i - self.rank
        st_lens = dict(st.lens)
        # lengths of the later chunks of this round: simulate on a COPY of the generator (their draws are consumed
        # for real when this rank replays them before its next chunk)
        g0 = st.module.generator
        gen = torch.Generator(device=g0.device).set_state(g0.get_state())
        self._skip_own_draws(gen, st, i)
        for c in range(i + 1, min(first + self.world, n)):
            st_lens[c] = simulate_block_draws(gen, self._frames[c], st.tsize, st.args, has_anchors=c > 0)["M_local"]
        round_chunks = range(first, min(first + self.world, n))
        m_max = max(st_lens[c] for c in round_chunks)
        padded = my_map if Ml == m_max else torch.cat([my_map, my_map.new_zeros(B, m_max - Ml)], dim=1)
        gathered = self.t.all_gather(padded.contiguous())  # (W, B, m_max) int32; every rank calls it once per round
        prev_round_last, st.carry = st.carry, None
        last = first + self.world - 1
        if last < n:
            st.carry = gathered[self.world - 1][:, :st_lens[last]]
        pool, st.pool = (st.pool.wait() if st.pool is not None else None), None
        self.bytes_received += gathered.numel() * gathered.element_size() + (0 if pool is None else pool.numel() * pool.element_size())
        if i == 0:
            return None
        # chunk i-1 ran on rank W-1 in the previous round (rank 0), else on rank - 1 in this one
        pmap = (prev_round_last if self.rank == 0 else gathered[self.rank - 1][:, :st_lens[i - 1]]).contiguous()
        got = _lib.gather_rows(pool, None, pmap) if pool.is_cuda else \
            torch.gather(pool, 1, pmap.long()[:, :, None].expand(-1, -1, C))
        return _with_positions(got, pmap, st.tsize)

    def _skip_own_draws(self, gen: torch.Generator, st: _BlockState, i: int) -> None:
        """all-gather only: `gen` is a copy taken after this chunk's LOCAL draws; the coin of its global level (made by
        compute_merge right after anchors_for returns) still lies ahead of the later chunks' draws."""
        if st.args["merge_global"] and i > 0:
            torch.rand(1, generator=gen, device=gen.device)

    def publish(self, key: str, anchors: torch.Tensor) -> None:
        """After the block stored its new anchors (patch.py:80,82).  Exact mode forwards them to the next chunk."""
        i, st = self._cur, self._blocks[key]
        st.lens.setdefault(i, anchors.shape[1])
        self.where = f"chunk {i} block {key}: published"
        if self.mode != "ring":
            return
        if i + 1 >= len(self._frames):
            return
        dst = self._owner(i + 1)
        if dst == self.rank:
            st.carry = anchors
            return
        self._send(anchors.contiguous(), dst, st)
        from .patch import content_ids
        cid = content_ids(anchors, anchors.device, anchors.dtype)
        ids = cid.contiguous() if cid is not None else \
            torch.full(tuple(anchors.shape[:2]), -1, dtype=torch.int32, device=anchors.device)
        self._send(ids, dst, st)

    @staticmethod
    def _has_local_levels(frames: int, args: Dict) -> bool:
        """patch.py:44-54: a chunk builds a local merge map iff it has more than one frame and a positive ratio."""
        return frames > 1 and args["local_merge_ratio"] > 0

    def _exchange(self, send: Optional[torch.Tensor], dst: int, recv_spec, src: int):
        """An asynchronous send and an asynchronous receive (either may be absent), INDEPENDENT of each other: the send
        to the successor and the receive from the predecessor travel on different communicators (torch keeps one per
        pair of ranks, each with its own stream), so a send that has to wait for its receiver -- rank W-1's successor is
        rank 0's chunk of the NEXT round -- never holds up this rank's receive.  (Grouping the two into one
        ncclGroup would couple them and deadlock on exactly that wrap-around.)  Returns the receive handle or None."""
        # (receive first: were a communicator still to be created at this point, the creation blocks the host until the
        # peer arrives -- with every rank posting its receive first, the ring unblocks from rank 0 onwards)
        h = None if recv_spec is None else self.t.irecv(recv_spec[0], recv_spec[1], recv_spec[2], src)
        if send is not None:
            self._send(send, dst, None)
        return h

    def _send(self, tensor: torch.Tensor, dst: int, st: _BlockState) -> None:
        if dst == self.rank:
            return
        work = self.t.isend(tensor, dst)
        self._inflight.append((work, tensor))
        self.bytes_sent += tensor.numel() * tensor.element_size()
        if len(self._inflight) > 64:                       # long steps: let finished transfers (and their buffers) go
            self._inflight = [(w, t) for w, t in self._inflight if not getattr(w, "done", lambda: True)()]


class RingExchange(AnchorExchange):
    """Exact mode (bit-identical to the sequential run)."""

    def __init__(self, group=None, transport=None):
        super().__init__("ring", transport, group)


class NeighbourExchange(AnchorExchange):
    """Parallel anchors from the predecessor chunk's local tokens, point-to-point."""

    def __init__(self, group=None, transport=None):
        super().__init__("neighbour", transport, group)


class AllGatherExchange(AnchorExchange):
    """Parallel anchors through one RCCL all-gather per merging block (north_star's wording)."""

    def __init__(self, group=None, transport=None):
        super().__init__("allgather", transport, group)

    @property
    def bytes_gathered(self) -> int:
        return self.bytes_received


# ----------------------------------------------------------------------------------------------------
# wiring into the patched model
# ----------------------------------------------------------------------------------------------------
def enable(model: torch.nn.Module, exchange: AnchorExchange) -> None:
    """Attach ``exchange`` to every patched block; compute_merge consults ``module._vtm_exchange``.  The blocks are
    registered with the exchange, which forks their generators at the first ``begin_step`` (every rank at the same
    point of the global RNG stream, also the ranks that sit the first step out)."""
    root = model.unet if hasattr(model, "unet") else model
    exchange.reset()
    for name, m in root.named_modules():
        if m.__class__.__name__ == "ToMeBlock":
            # The exchange forks / replays the block generators on the HOST (begin_step, simulate_block_draws): with the
            # opt-in device stream (apply_patch(generator_device="device") / VIDTOME_GENERATOR=device) the patched pre-hook
            # would re-fork from torch.cuda.get_rng_state() at each rank's own first forward -- a rank-dependent fork point --
            # and every replayed draw would cost a device sync.  Chunk-parallel runs therefore pin the CPU stream (the
            # sequential run they are compared with uses the same one).
            args = getattr(m, "_tome_info", {}).get("args", {})
            from . import utils as _utils
            if (args.get("generator_device") or _utils.GENERATOR_MODE) == "device":
                raise RuntimeError("chunk_parallel.enable: the device generator stream (generator_device='device' / "
                                   "VIDTOME_GENERATOR=device) cannot be combined with an AnchorExchange; patch the model with "
                                   "generator_device='cpu'")
            m._vtm_exchange = exchange
            m._vtm_key = name
            exchange.register(name, m)


def disable(model: torch.nn.Module) -> None:
    root = model.unet if hasattr(model, "unet") else model
    for _, m in root.named_modules():
        if hasattr(m, "_vtm_exchange"):
            del m._vtm_exchange


def run_step(model: torch.nn.Module, exchange: AnchorExchange, frames_per_chunk: Sequence[int],
             process_chunk: Callable[[int], None], reset_anchors: bool = True) -> List[int]:
    """This rank's share of one denoising step (generate.py:215-219 + :233-236): its chunks in schedule order, then
    the anchor reset.  ``process_chunk(i)`` runs the patched model on chunk i.  Returns the chunk indices processed."""
    from . import patch
    exchange.begin_step(frames_per_chunk)
    mine = exchange.my_chunks()
    for i in mine:
        exchange.begin_chunk(i)
        process_chunk(i)
    exchange.end_step()
    if reset_anchors:
        patch.update_patch(model, global_tokens=None)
    return mine

"""Cross-frame bipartite soft matching on MI355X -- the host side of vidtome/merge.py.

Public surface mirrors the reference (same names, argument meaning, return protocol and error behaviour):

* ``do_nothing``                           <- vidtome/merge.py:5-6
* ``bipartite_soft_matching_randframe``    <- vidtome/merge.py:20-159   (local merging)
* ``bipartite_soft_matching_2s``           <- vidtome/merge.py:343-463  (global merging)

Each matcher returns ``(merge, unmerge, ret_dict)`` closures like the reference, but the closures hold
*composed row maps* on the device instead of index tensors + gather/scatter code: in ``replace`` mode (the
only mode the reference's ``compute_merge`` ever uses, patch.py:45-46,73-75) merged tokens are a pure row
selection, so ``merge`` is one gather and ``unmerge`` is one gather with the inverse map.

The arithmetic lives in libvidtome_hip.so (``_lib``): fused normalise+split, fused score+row-max on the
fp32 MFMA, radix argsort, index planning.  Nothing here touches the CPU except the two RNG draws the
reference also makes on its generator (merge.py:57-58, patch.py:62).
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Callable, Optional, Tuple

import torch

from . import _lib

# "filtered" (default) or "exact": both produce bit-identical packed results (tests/test_gpu_parity.py)
MATCH_MODE = os.environ.get("VIDTOME_MATCH", "filtered")


def do_nothing(x: torch.Tensor, mode: str = None, **kwargs):
    """vidtome/merge.py:5-6."""
    return x


@dataclass
class Level:
    """Result of one matching level, everything device-resident (int32)."""
    N_in: int
    Ns: int
    Nd: int
    r: int
    new_cur: torch.Tensor              # (B, U + Nd) pool row id of every merged token  (merge closure)
    inv: torch.Tensor                  # (B, N_in)  merged position each input position is restored from
    a_pos: torch.Tensor                # (Ns,)  = a_idx of the reference
    b_pos: torch.Tensor                # (Nd,)  = b_idx
    best: torch.Tensor                 # packed row maxima (see include/vidtome_hip.h)
    unm_idx: Optional[torch.Tensor] = None
    src_idx: Optional[torch.Tensor] = None
    dst_idx: Optional[torch.Tensor] = None

    @property
    def unm_num(self) -> int:
        return self.Ns - self.r


# Launch plan of the filtered matcher (same bits either way): "auto" = merge.MatchPlanner decides per
# block from the previous call's counters, "one" = always the one-launch filter (rounds 1-4), "range" = always scout + range
MATCH_PLAN = os.environ.get("VIDTOME_MATCH_PLAN", "auto")
# ... and whether the planner may let the scout test after one channel step ("0": always at the filter's own depth)
SHALLOW_SCOUT = os.environ.get("VIDTOME_SHALLOW_SCOUT", "1") != "0"


class MatchPlanner:
    """How the next call of ONE matching level of one block is launched (vtm_match_filtered_plan / _ordered).

    The scout + range plan is ~30 % faster when nearly all 256 x 128 tile pairs die at the scout's test -- frames of one clip
    at low noise -- and ~40 % slower when most stay alive (uncorrelated tokens; a noisy clip, whose matches do not clear the
    rest bound at that depth).  How many stay alive is a property of the data at that block, and consecutive calls (the next
    chunk, the next denoising step) see similar data, so the planner steers by the PREVIOUS call's counters: every
    scout + range call copies its 8 counters asynchronously into this planner's pinned host buffer; the next call reads
    whatever has arrived -- never waiting, never synchronising; a stale or missing reading only delays a switch -- and falls
    back to the one-launch plan for `COOL` calls when the spans the second launch had to stream covered more than `HIGH` of
    the level, then tries again.  Measured per call (profiles/r05_n_scout_range_plan.txt, r05_q_position_order.txt): the plan
    wins up to 0.08 of the level inside the spans (top level 1 on low-noise clips 0.06: -24 %; position-ordered top global level
    0.014: -43 % on top of the ordering; a smooth field's level 2 0.077: -8 %) and loses from 0.11 on (0.25: -11 % ... +50 %;
    ~1.0 on noisy clips and uncorrelated tokens: +40 %).

    Levels behind the first are handed to the matcher in position order (vtm_position_order) -- which the plan needs, and which
    by itself costs a 40 us sort and saves 12-26 % of a top global call (more blocks die) but nothing at level 2.  So
    `order_alone` says what the level does while the plan is off: keep the ordering (global level) or go back to the reference's
    row order (level 2); and a level in which NOTHING died at the scout's test (uncorrelated tokens) drops the ordering too.
    Results never depend on any of this.

    Two pinned buffers, one per KIND of reading (round 6): the scout + range calls copy into `buf`, the one
    one-launch call that is asked for the filter's own counters (the order probe) into `probe_buf` -- the 32-byte copy of
    an earlier scout + range call that lands late can then never be read as the probe's answer.  `__del__` waits for the
    copies still in flight before the pinned memory goes back to torch's host allocator."""

    # COOL (round 6: 32, was 256): how long a level stays on the one-launch plan before the scout is tried again.  It bounds
    # BOTH costs of being wrong (tests/test_gpu_parity.py::test_planner_cost_is_bounded_on_alternating_regimes): on data the plan
    # never pays for, one exploring call (+40 % of ONE call) per COOL + 1 calls = +1.2 % of the level; on a clip that turns
    # plan-friendly again, at most COOL calls at the one-launch price (a shot of 300 calls: < 3 % of the level against a
    # per-call oracle; 256 lost 11 % there).  Fast alternation (every 2 - 8 calls) cannot be followed from the previous call's
    # counters at all; the planner then costs at most the exploring calls against the better FIXED plan.
    HIGH, LOW, COOL = 0.09, 0.07, 32

    def __init__(self, order_alone: bool = False):
        self.mode = _lib.MATCH_SCOUT_RANGE
        self.buf = torch.zeros(8, dtype=torch.int32).pin_memory()
        self.view = self.buf.numpy()
        self.probe_buf = torch.zeros(8, dtype=torch.int32).pin_memory()
        self.probe_view = self.probe_buf.numpy()
        self.cool = 0
        self.switches = 0
        self.order_alone = order_alone
        self.order_off = False
        self.order_probe = False
        # the scout's depth: the filter's own test depth, or ONE 64-channel step (VTM_MATCH_SCOUT_STEPS(1): -20 ... -28 % per
        # call where a tile that is dead at 40 % of the channels is dead at 20 % already -- low-noise clips: same spans;
        # elsewhere more tiles stay marked).  Tried once the deep scout's spans are below LOW, kept while its own are below HIGH
        self.shallow = False
        self.shallow_ban = 0
        self.issued_shallow = False

    def next(self):
        """-> (mode, pinned stats buffer or None, position-order the level?, scout steps) for the call about to be issued."""
        if self.mode == _lib.MATCH_SCOUT_RANGE:
            tested, in_spans = int(self.view[4]), int(self.view[7])        # the last call whose counters have arrived
            if tested > 0:
                self.view[4] = 0                                           # judged once
                wide = in_spans > self.HIGH * tested
                if self.issued_shallow:                                    # (that call scouted after one step)
                    if wide:
                        self.shallow, self.shallow_ban = False, self.COOL
                elif wide:
                    self.mode, self.cool = _lib.MATCH_ONE_LAUNCH, self.COOL
                    self.switches += 1
                    # does the ordering pay without the plan?  Not when nothing dies in the ONE-LAUNCH filter either (its
                    # running maxima grow along the dst axis, the scout only has the seeds: a drifting smooth field leaves
                    # every block alive in the scout and 29 % in the filter) -- asked of the first one-launch call's counters
                    self.order_off, self.order_probe = False, self.order_alone
                    self.probe_view[4] = 0
                elif self.shallow_ban <= 0 and in_spans < self.LOW * tested:      # (some margin: a level at the edge of the
                    self.shallow = True                                           # plan is not the place for a cheaper scout)
            if self.shallow_ban > 0:
                self.shallow_ban -= 1
        else:
            if self.order_probe and int(self.probe_view[4]) > 0:
                self.order_off = int(self.probe_view[5]) > 0.9 * int(self.probe_view[4])
                self.order_probe = False
            self.cool -= 1
            if self.cool <= 0:
                self.mode, self.shallow = _lib.MATCH_SCOUT_RANGE, False
                self.view[4] = 0
        if self.mode == _lib.MATCH_SCOUT_RANGE:
            self.issued_shallow = self.shallow
            return self.mode, self.buf, True, (1 if self.shallow else 0)
        return self.mode, (self.probe_buf if self.order_probe else None), self.order_alone and not self.order_off, 0

    def __del__(self):
        # an asynchronous counter copy may still be in flight towards the pinned buffers: wait for the device
        # before they are released (garbage collection of a block, bench.py dropping a block's planners)
        try:
            if torch.cuda.is_available() and torch.cuda.is_initialized():
                torch.cuda.synchronize()
        except Exception:          # interpreter shutdown: the process' memory goes with it
            pass


# Levels 2 / global meet their rows in position order (vtm_position_order + vtm_match_filtered_ordered: same bits, fewer live
# blocks; "0" = the reference's sequence order as in rounds 1-4) ...
POSITION_ORDER = os.environ.get("VIDTOME_POSITION_ORDER", "1") != "0"
# ... when the level is large enough for the sort (3 launches, ~40 us) to pay: src rows x dst rows per sample
# ... also with align_batch (ONE order for all samples, sample 0's: entry i is the same original index in every sample)
ALIGNED_ORDER = os.environ.get("VIDTOME_ALIGNED_ORDER", "1") != "0"
POSITION_ORDER_MIN_PAIRS = int(os.environ.get("VIDTOME_POSITION_ORDER_MIN_PAIRS", str(1 << 26)))


def order_level(Ns: int, Nd: int, tokens: int, align_batch: bool) -> bool:
    """Does a level of this geometry (not the first local one) meet its rows in position order?"""
    return (POSITION_ORDER and (ALIGNED_ORDER or not align_batch) and Ns * Nd >= POSITION_ORDER_MIN_PAIRS and
            0 < tokens <= _lib.POSITION_ORDER_MAX_N)


def _run_level(x0: torch.Tensor, x1: Optional[torch.Tensor], parts, ratio: float, align_batch: bool,
               want_indices: bool, seed=None, planner: Optional[MatchPlanner] = None, reorder: bool = False) -> Level:
    """normalise+split -> fused score/top-1 -> argsort -> index split  (merge.py:84-117 / 389-421).  ``reorder``: the level's
    rows are NOT in (frame, position) order (every level but the first local one) -- the matcher is then handed both lists
    sorted by token position and reports in the original indexing; everything behind it sees the reference's order."""
    a_pos, b_pos, a_rows, b_rows = parts
    Ns, Nd = a_rows.shape[1], b_rows.shape[1]
    r = min(Ns, int(Ns * ratio))                       # merge.py:90 (Python float -> int truncation)
    if MATCH_MODE == "exact" or x0.shape[2] > 1280:    # plain fp32-MFMA kernel (filter window holds for C <= 1280)
        a_op, _ = _lib.normalize_gather(x0, x1, a_rows)
        b_op, _ = _lib.normalize_gather(x0, x1, b_rows)
        best = _lib.match(a_op, b_op, Ns, Nd, align_batch)
    else:                                              # fp16 filter + fp32 refine: same bits, ~4x faster
        mode, stats, order, scout = _lib.MATCH_ONE_LAUNCH, None, None, 0
        m_a, m_b = a_rows, b_rows
        seeded = seed is not None and _lib.SEED_MATCHER
        can_order = reorder and seeded and order_level(Ns, Nd, seed[0], align_batch)
        use_order = can_order
        if seeded and (can_order or not reorder) and MATCH_PLAN != "one":     # (rows in similarity-rank order: no plan)
            if MATCH_PLAN == "range":
                mode = _lib.MATCH_SCOUT_RANGE
            elif planner is not None:
                mode, stats, keep, scout = planner.next()
                use_order = can_order and keep
                if not SHALLOW_SCOUT:
                    scout = 0
        if use_order:
            tokens, L, pos1, _ = seed
            m_a, a_order, m_b, b_order, table = _lib.position_order(a_rows, b_rows, L, tokens, pos1, x0.shape[1],
                                                                    shared=align_batch)
            order, seed = (a_order, b_order), (tokens, L, pos1, table)     # the sort's offsets ARE the position -> dst table
        elif can_order and seed[2] is not None and seed[3] is None:
            seed = None        # a global level left in the reference's order without a position -> dst table: unseeded
        best = _lib.match_filtered(x0, x1, m_a, m_b, align_batch, seed=seed, mode=mode, stats_host=stats, order=order,
                                   scout_steps=scout)
    perm = _lib.sort_desc(best)
    new_cur, inv, unm_idx, src_idx, dst_idx = _lib.plan_apply(best, perm, a_pos, b_pos, a_rows, b_rows, r,
                                                              align_batch, want_indices)
    return Level(Ns + Nd, Ns, Nd, r, new_cur, inv, a_pos, b_pos, best, unm_idx, src_idx, dst_idx)


def local_level(x0: torch.Tensor, cur: Optional[torch.Tensor], N_in: int, F: int, ratio: float, unm_pre: int,
                randf: int, target_stride: int, align_batch: bool, want_indices: bool = False,
                tokens: Optional[int] = None, planner: Optional[MatchPlanner] = None) -> Level:
    """One level of local merging on the joined chunk x0 (B, L, C); ``cur`` maps the current sequence to
    rows of x0 (None = identity).  ``tokens`` = tokens per frame of the joined chunk when its rows are (frame, position)
    ordered: the matcher is then seeded with, for every src token, the token at the same position of the first dst frame
    (dst index = position; never changes the result)."""
    B = x0.shape[0]
    tnum = (N_in - unm_pre) // F                       # merge.py:43
    ts = min(target_stride, F)                         # merge.py:56
    parts = _lib.partition_local(cur, B, N_in, unm_pre, tnum, ts, randf, x0.device)
    seed = (tokens, x0.shape[1], None, None) if (tokens and tokens == tnum) else None
    # the scout + range plan is for position-ordered rows on both sides: the first level (cur None) as it comes, the later
    # ones through vtm_position_order.  (Measured with a planner on every level while levels 2 / global still met their rows in
    # similarity-rank order: they switched themselves off on every regime and only paid the exploring calls.)
    return _run_level(x0, None, parts, ratio, align_batch, want_indices, seed, planner, reorder=cur is not None)


def global_level(x0: torch.Tensor, anchors: torch.Tensor, cur_local: Optional[torch.Tensor], Ml: int,
                 local_is_src: bool, ratio: float, align_batch: bool, want_indices: bool = False,
                 tokens: Optional[int] = None, anchor_positions: Optional[torch.Tensor] = None,
                 planner: Optional[MatchPlanner] = None) -> Level:
    """Global merging of the chunk's local tokens against the block's anchor tokens (patch.py:59-82).  ``tokens`` (tokens
    per frame) / ``anchor_positions`` (B, Mg) int32 seed the matcher with the dst token at every src token's position."""
    B, L, _ = x0.shape
    if cur_local is None:
        cur_local = torch.arange(Ml, dtype=torch.int32, device=x0.device).expand(B, Ml).contiguous()
    seed = table = None
    # a seed pairs a src row with the dst row at the same token position: one side of the level is the anchors, so their
    # positions must be known whichever side it is (local-is-src: the position -> dst row table is built from them;
    # local-is-dst: they are the src rows).  Anchors that arrived without positions (an exchange's single-message form, a
    # user-supplied tensor): no table, no seed launch
    if tokens and _lib.SEED_MATCHER and anchor_positions is not None:
        src_len = Ml if local_is_src else anchors.shape[1]
        if not order_level(src_len, Ml + anchors.shape[1] - src_len, tokens, align_batch):   # (position-ordered calls get their table from the sort)
            table = torch.empty((B, tokens), dtype=torch.int32, device=x0.device)
        seed = (tokens, L, anchor_positions, table)
    parts = _lib.partition_global(cur_local, L, anchors.shape[1], local_is_src, table, tokens or 0, anchor_positions)
    return _run_level(x0, anchors, parts, ratio, align_batch, want_indices, seed, planner, reorder=True)


def draw_randf(generator: torch.Generator, ts: int) -> int:
    """merge.py:57-58: ``torch.randint(0, target_stride, [1], generator=generator)`` on the CPU generator."""
    return int(torch.randint(0, ts, torch.Size([1]), generator=generator, device=generator.device))


def _check_metric(metric: torch.Tensor) -> torch.Tensor:
    if metric.dim() != 3:
        raise ValueError("metric must be [B, N, C]")
    if not metric.is_cuda:
        raise RuntimeError("vidtome_amd runs on the GPU only (no CPU path); got a CPU tensor")
    return metric.contiguous()


def _merge_with_mode(level: Level, x: torch.Tensor, mode: str) -> torch.Tensor:
    """merge.py:119-133 / 423-437: ``cat([unm, dst])``; in the modes other than "replace" the matched src rows are first
    folded into their dst rows (``scatter_reduce(..., reduce=mode, include_self=True)``, merge.py:127-131) -- never reached
    from compute_merge, provided for code written against the closure protocol."""
    x = x.contiguous()
    out = _lib.gather_rows(x, None, level.new_cur)                  # [unm | dst]: the replace-mode result
    if mode == "replace":
        return out
    B = x.shape[0]
    src_rows = level.a_pos.long()[level.src_idx.long()].to(torch.int32).contiguous()       # (B, r) rows of x
    dst_rows = level.b_pos.to(torch.int32).expand(B, -1).contiguous()                       # (B, Nd)
    return _lib.merge_reduce(x, src_rows, dst_rows, level.dst_idx.contiguous(), mode, out, level.Ns - level.r)


def _make_closures(level: Level, N: int, out_slice: Optional[Tuple[int, int]] = None, merge_mode="replace"):
    def merge(x: torch.Tensor, mode=None) -> torch.Tensor:
        return _merge_with_mode(level, x, mode if mode is not None else merge_mode)

    def unmerge(x: torch.Tensor, **kwarg) -> torch.Tensor:
        inv = level.inv
        if out_slice is not None:                       # merge.py:459
            inv = inv[:, out_slice[0]:out_slice[1]].contiguous()
        return _lib.unmerge_add(x.contiguous(), inv, None)

    return merge, unmerge


def bipartite_soft_matching_randframe(metric: torch.Tensor, F: int, ratio: float, unm_pre: int,
                                      generator: torch.Generator, target_stride: int = 4,
                                      align_batch: bool = False, merge_mode: str = "replace"
                                      ) -> Tuple[Callable, Callable, dict]:
    """vidtome/merge.py:20-159, same signature.  ``ret_dict`` additionally carries the index tensors the
    reference keeps in closure cells (a_idx, b_idx, unm_idx, src_idx, dst_idx as int32) for parity tests."""
    metric = _check_metric(metric)
    B, N, _ = metric.shape
    tnum = (N - unm_pre) // F
    if ratio <= 0:
        return do_nothing, do_nothing, {"unm_num": tnum}          # merge.py:45-46
    with torch.no_grad():
        randf = draw_randf(generator, min(target_stride, F))
        level = local_level(metric, None, N, F, ratio, unm_pre, randf, target_stride, align_batch, True,
                            tokens=tnum if unm_pre == 0 else None)
    merge, unmerge = _make_closures(level, N, None, merge_mode)
    ret_dict = {"unm_num": level.unm_num, "a_idx": level.a_pos, "b_idx": level.b_pos,
                "unm_idx": level.unm_idx, "src_idx": level.src_idx, "dst_idx": level.dst_idx,
                "level": level}
    return merge, unmerge, ret_dict


def bipartite_soft_matching_2s(metric: torch.Tensor, src_len: int, ratio: float, align_batch: bool,
                               merge_mode: str = "replace", unmerge_chunk: int = 0):
    """vidtome/merge.py:343-463, same signature (including the 2-tuple returned for ratio <= 0,
    merge.py:364-365)."""
    metric = _check_metric(metric)
    B, N, _ = metric.shape
    if ratio <= 0:
        return do_nothing, do_nothing
    with torch.no_grad():
        # [src | dst] = [src_len | N - src_len]: expressed as "local = src part, anchors = dst part"
        x_src = metric[:, :src_len].contiguous()
        x_dst = metric[:, src_len:].contiguous()
        level = global_level(x_src, x_dst, None, src_len, True, ratio, align_batch, True)
    sl = (0, src_len) if unmerge_chunk == 0 else (src_len, N)

    def merge(x: torch.Tensor, mode=None) -> torch.Tensor:
        return _merge_with_mode(level, x, mode if mode is not None else merge_mode)

    def unmerge(x: torch.Tensor, **kwarg) -> torch.Tensor:
        return _lib.unmerge_add(x.contiguous(), level.inv[:, sl[0]:sl[1]].contiguous(), None)

    ret_dict = {"unm_num": level.unm_num, "a_idx": level.a_pos, "b_idx": level.b_pos,
                "unm_idx": level.unm_idx, "src_idx": level.src_idx, "dst_idx": level.dst_idx,
                "level": level}
    return merge, unmerge, ret_dict

"""apply_patch / remove_patch / update_patch / collect_from_patch -- the hook API of vidtome/patch.py,
with the patched self-attention segment executed by libvidtome_hip.so.

Mirrors (file:line under the reference):
* ``compute_merge``               <- vidtome/patch.py:14-91
* ``make_diffusers_tome_block``   <- vidtome/patch.py:119-203   (class *named* ToMeBlock, ``_parent``)
* ``hook_tome_model`` / ``hook_tome_module`` <- vidtome/patch.py:206-231
* ``apply_patch``                 <- vidtome/patch.py:234-334   (same kwargs and defaults)
* ``remove_patch``                <- vidtome/patch.py:337-355
* ``update_patch``                <- vidtome/patch.py:358-370
* ``collect_from_patch``          <- vidtome/patch.py:373-387

Differences that are design, not omissions:
* the merge chain of a block is ONE composed gather map and the unmerge chain ONE inverse map (all levels
  use merge mode "replace", so merged tokens are a row selection of [chunk tokens | anchor tokens]);
* ``module.global_tokens`` stays on the device (the reference parks it on the CPU and syncs per block,
  patch.py:65,70,80,82);
* the block generator forks the CPU RNG state by default (utils.init_generator; `generator_device="device"` opts into the
  reference's device rule);
* ``attn1`` is evaluated from the module's own weights by the fused path (projection GEMMs fed through the composed
  merge map or as panel GEMMs, attention core = vtm_attention: no library GEMM on the default path), including the
  reference's PnP injection branch when ``utils/pnp_utils.py``-style control is registered on the module;
* with a global level the attention runs only for the DISTINCT merged rows the chunk's local tokens read
  (MergePlan.q_rows / q_count), and the rest of the block (norm2 / attn2, norm3 / GEGLU feed-forward) runs as panel GEMMs
  (csrc/ff.hip) when the modules are the plain arithmetic.
"""
from __future__ import annotations

import math
import os
from typing import Any, Callable, Dict, Optional, Tuple, Type

import torch
import torch.nn.functional as F

from . import _lib, merge
from .utils import init_generator, isinstance_str, join_frame, split_frame

# Attention over the merged sequence computes outputs only for the rows unmerge() reads (see MergePlan.q_rows);
# VIDTOME_LIVE_QUERIES=0 computes every row like the reference does (same block output, more work).
LIVE_QUERIES = os.environ.get("VIDTOME_LIVE_QUERIES", "1") != "0"
# With the local chunk on the src side of the global level several local tokens may have merged into the SAME anchor
# token; its attention output is then computed once and shared (device-side compaction of the live queries, no host
# sync: the attention launch is sized for the upper bound and query blocks past the per-sample count exit).
# VIDTOME_COMPACT_QUERIES=0 computes one query per local token (round 2's behaviour; same block output).
COMPACT_QUERIES = os.environ.get("VIDTOME_COMPACT_QUERIES", "1") != "0"


# ----------------------------------------------------------------------------------------------------
# compute_merge
# ----------------------------------------------------------------------------------------------------
class MergePlan:
    """What ``compute_merge`` produces for one block call: composed maps + the merged tokens."""

    __slots__ = ("fsize", "L", "M", "gather_map", "_inv", "_inv_parts", "_merged", "levels", "global_level", "local_chunk",
                 "x_joined", "anchors_in", "q_rows", "inv_q", "q_count", "pad_to", "fold_args", "_key_fold", "aligned")

    def __init__(self):
        self.levels = []
        self.global_level = None
        self.local_chunk = None
        self.gather_map = None
        self._inv = None
        self._inv_parts = None          # (local-levels inverse map, local position -> merged position): composed on demand
        self.anchors_in = None
        # Global level only: the rows of the merged sequence whose attention output unmerge() ever reads
        # (q_rows[b, t] = merged row of local token t) and the local-levels-only inverse map.  The merged
        # sequence also contains the other chunk's tokens; they are keys / values, but nobody reads their
        # attention OUTPUT (merge.py:459 returns the local part only), so the block computes attention for
        # the q_rows queries only -- a third fewer at global_merge_ratio 0.5, same block output.
        # q_count (B,) int32 on the device, or None: with the local chunk on the src side q_rows lists every DISTINCT
        # merged position once (several local tokens may share an anchor row) and only its first q_count[b] entries
        # are queries; inv_q then maps through that compact list.
        self.q_rows = None
        self.inv_q = None
        self.q_count = None
        self.aligned = False            # align_batch: every sample shares the levels' indices (so also q_rows / inv_q)
        # The anchors' exact duplicates (patch.py:80 copies an anchor row to every local token that merged into it) are
        # duplicate KEYS of the next block: (merged-position -> pool row map, L, content id per anchor row, number of ids)
        # when the anchors carry ids, folded on first use by the attention path that can take a per-key multiplicity
        self.fold_args = None
        self._key_fold = None
        self._merged = None
        self.pad_to = 8

    @property
    def inv(self) -> Optional[torch.Tensor]:
        """The composed unmerge map of ALL levels (B, L): merged row every position of the joined chunk is restored from.
        With a global level the patched block unmerges through `inv_q` (the rows its live queries produce) and never needs
        this map, so it is composed on first use (API users, tests, VIDTOME_LIVE_QUERIES=0)."""
        if self._inv is None and self._inv_parts is not None:
            inv_local, loc = self._inv_parts
            self._inv = _lib.compose(inv_local, loc, self.L) if inv_local is not None else loc
        return self._inv

    def key_fold(self, dtype) -> Optional[Tuple[torch.Tensor, torch.Tensor, torch.Tensor]]:
        """(key_sel, k_bias, k_count) of _lib.fold_keys, or None when the anchors carried no content ids."""
        if self._key_fold is None and self.fold_args is not None:
            cur, L, cid, n_ids = self.fold_args
            self._key_fold = _lib.fold_keys(cur, L, cid, n_ids, dtype)
        return self._key_fold

    @property
    def merged(self) -> torch.Tensor:
        """The merged tokens (B, Mp, C) -- what the reference's composed merge closure returns (patch.py:84).  The
        patched block itself does not need them (its projections read the pool through `gather_map`), so they are
        materialised on first use."""
        if self._merged is None:
            self._merged = self.x_joined if self.gather_map is None else \
                _lib.gather_rows(self.x_joined, self.anchors_in, self.gather_map, pad_to=self.pad_to)
        return self._merged


CHECK_CONTENT_IDS = os.environ.get("VIDTOME_CHECK_CID", "0") == "1"


def tag_content_ids(tokens: torch.Tensor, ids: torch.Tensor) -> None:
    """Attach content ids (B, M) int32 -- equal id = byte-identical rows; -1 = none -- to an anchor tensor.  RESULTS depend
    on them (the attention folds rows with equal ids into one key), so the tag also records what the tensor looked like
    when the ids were computed: torch's version counter, the storage address and the shape.  `content_ids` drops the ids
    when any of them changed.  What no tag can see is a write that bypasses torch (`.data`, a raw pointer, another library's
    kernel): `module.global_tokens` must not be edited that way; VIDTOME_CHECK_CID=1 verifies the claim on every use (one
    device sync per block: a debugging aid)."""
    tokens._vtm_cid = (ids, tokens._version, tokens.data_ptr(), tuple(tokens.shape))


def mark_anchors_ready(tokens: Optional[torch.Tensor]) -> None:
    """(round 6) Consecutive chunks may run on DIFFERENT HIP streams (two chunks in flight hide each other's dispatch gaps and
    small launches: bench.py `two_in_flight`, +15 % chunk-steps/s on one GPU).  The anchors are the one thing chunk k + 1 takes
    from chunk k (patch.py:60-82), so the producer leaves an event behind every anchor tensor it stores -- recorded on its
    stream after the last kernel that writes the anchors, their token positions or their content ids."""
    if tokens is not None and tokens.is_cuda:
        s = torch.cuda.current_stream(tokens.device)
        ev = torch.cuda.Event()
        ev.record(s)
        tokens._vtm_ready = (ev, s.cuda_stream)


def await_anchors(tokens: Optional[torch.Tensor]) -> None:
    """... and a consumer on another stream waits for that event ON THE DEVICE (no host synchronisation) and tells torch's
    caching allocator that the tensors are in use on its stream too (`record_stream`: the producer's reference may be dropped
    -- the next anchor update replaces `module.global_tokens` -- while this stream's kernels still read the old rows)."""
    ready = getattr(tokens, "_vtm_ready", None) if tokens is not None else None
    if ready is None or not tokens.is_cuda:
        return
    ev, sid = ready
    cur = torch.cuda.current_stream(tokens.device)
    if cur.cuda_stream == sid:
        return
    cur.wait_event(ev)
    tokens.record_stream(cur)
    pos = getattr(tokens, "_vtm_pos", None)
    if pos is not None and pos.is_cuda:
        pos.record_stream(cur)
    tag = getattr(tokens, "_vtm_cid", None)
    if tag is not None and len(tag) == 4 and tag[0].is_cuda:
        tag[0].record_stream(cur)


def content_ids(tokens: torch.Tensor, device, dtype) -> Optional[torch.Tensor]:
    """The ids `tag_content_ids` attached, or None when the tag is missing or stale."""
    tag = getattr(tokens, "_vtm_cid", None)
    if tag is None or len(tag) != 4:
        return None
    ids, version, ptr, shape = tag
    if (version != tokens._version or ptr != tokens.data_ptr() or shape != tuple(tokens.shape)
            or tuple(ids.shape) != tuple(tokens.shape[:2]) or ids.device != torch.device(device) or tokens.dtype != dtype):
        return None
    if CHECK_CONTENT_IDS:
        # rows with equal ids must be equal: compare every row with the first row of its id
        B, M = ids.shape
        for b in range(B):
            idb = ids[b].long()
            valid = idb >= 0
            if not bool(valid.any()):
                continue
            first = torch.full((int(idb.max()) + 1,), M, dtype=torch.long, device=ids.device)
            first.scatter_reduce_(0, idb[valid], torch.arange(M, device=ids.device)[valid], reduce="amin")
            ref = tokens[b][first[idb.clamp(min=0)].clamp(max=M - 1)]
            same = (tokens[b].view(torch.int16 if tokens.element_size() == 2 else torch.int32)
                    == ref.view(torch.int16 if tokens.element_size() == 2 else torch.int32)).all(dim=1)
            if not bool((same | ~valid).all()):
                raise RuntimeError("vidtome_amd: anchor rows with equal content ids differ -- module.global_tokens was "
                                   "modified behind torch's back (VIDTOME_CHECK_CID=1)")
    return ids


def _planner(module: torch.nn.Module, key, order_alone: bool = False) -> "merge.MatchPlanner":
    """The launch planner of one matching level of the block (merge.MatchPlanner), kept on the module next to its generator
    and its anchors, one per level geometry; `remove_patch` drops them."""
    plans = module.__dict__.get("_vtm_match_plans")
    if plans is None:
        plans = module.__dict__["_vtm_match_plans"] = {}
    p = plans.get(key)
    if p is None:
        p = plans[key] = merge.MatchPlanner(order_alone)
    return p


def _draw_coin(generator: torch.Generator) -> float:
    """patch.py:62: ``torch.rand(1, generator=generator, device=generator.device)``."""
    return float(torch.rand(1, generator=generator, device=generator.device))


def compute_merge(module: torch.nn.Module, x: torch.Tensor, tome_info: Dict[str, Any],
                  pad_to: int = 8, want_indices: bool = False, materialize: bool = True
                  ) -> Tuple[Callable, Callable, torch.Tensor]:
    """vidtome/patch.py:14-91.  Returns ``(m, u, merged_tokens)``; ``u`` accepts ``resid=`` to fuse the
    residual add of patch.py:169 into the unmerge gather.  ``merged_tokens`` is (B, Mp, C) with
    Mp = M rounded up to ``pad_to`` rows (zero rows); ``m.plan`` exposes the MergePlan.  With ``materialize=False``
    (the patched block's own call: its projections gather through the map) the third value is None and
    ``m.plan.merged`` materialises the tokens on demand."""
    original_h, original_w = tome_info["size"]
    original_tokens = original_h * original_w
    downsample = int(math.ceil(math.sqrt(original_tokens // x.shape[1])))        # patch.py:17
    args = tome_info["args"]
    generator = module.generator
    fsize = x.shape[0] // args["batch_size"]                                       # patch.py:23
    tsize = x.shape[1]                                                             # patch.py:24

    if downsample > args["max_downsample"]:                                        # patch.py:27,86-88
        return merge.do_nothing, merge.do_nothing, x

    gmode = args.get("generator_device", None)
    if args["generator"] is None:                                                  # patch.py:29-33
        args["generator"] = init_generator(x.device, mode=gmode)
    elif args["generator"].device != x.device and (gmode or GENERATOR_MODE_DEFAULT()) == "device":
        args["generator"] = init_generator(x.device, fallback=args["generator"], mode=gmode)

    plan = MergePlan()
    plan.fsize = fsize
    plan.aligned = bool(args["align_batch"])
    # chunk-parallel runs (chunk_parallel.py): the exchange replays the draws of the chunks other ranks process
    # and posts the receive of the predecessor's anchor tokens before this block's own first draw
    exchange = getattr(module, "_vtm_exchange", None)
    xkey = getattr(module, "_vtm_key", "")
    with torch.no_grad():
        xj = join_frame(x.contiguous(), fsize)                                     # patch.py:37 (a view)
        B, L, C = xj.shape
        if exchange is not None:
            exchange.begin_block(module, xkey, fsize, tsize, args, xj)
        plan.x_joined, plan.L = xj, L
        cur: Optional[torch.Tensor] = None      # pool row id of every token of the current sequence
        inv: Optional[torch.Tensor] = None      # composed unmerge map so far
        n_cur, unm, curF = L, 0, fsize
        while curF > 1:                                                            # patch.py:44-54
            ratio = args["local_merge_ratio"]
            if ratio <= 0:                                                         # merge.py:45-46
                unm += (n_cur - unm) // curF
            else:
                randf = merge.draw_randf(generator, min(args["target_stride"], curF))
                lv = merge.local_level(xj, cur, n_cur, curF, ratio, unm, randf, args["target_stride"],
                                       args["align_batch"], want_indices, tokens=tsize,
                                       planner=_planner(module, (xj.shape[1], curF, n_cur)))
                plan.levels.append(lv)
                unm += lv.unm_num
                cur = lv.new_cur
                inv = _lib.compose(inv, lv.inv, L) if inv is not None else lv.inv
                n_cur = cur.shape[1]
            curF = (n_cur - unm) // tsize                                          # patch.py:54
        Ml = n_cur

        anchors_out = anchors_pos = None
        if args["merge_global"]:                                                   # patch.py:59-82
            if exchange is not None:
                gt = exchange.anchors_for(xkey, lambda: xj if cur is None else _lib.gather_rows(xj, None, cur), xj, cur)
            else:
                gt = getattr(module, "global_tokens", None)
                await_anchors(gt)              # (a predecessor chunk on another stream: device-side wait, see mark_anchors_ready)
            # token positions of the anchors (the matcher's seeds) ride on the tensor THIS function stored (an attribute of the
            # tensor object: they live and die with it); anchors that came from anywhere else (the user, an exchange) have none
            gt_pos = getattr(gt, "_vtm_pos", None) if gt is not None else None
            if gt_pos is not None and (tuple(gt_pos.shape) != tuple(gt.shape[:2]) or gt_pos.device != xj.device):
                gt_pos = None
            # content ids (equal id = identical rows), same lifetime rule -- and, because RESULTS depend on them, only while
            # nobody has written to the tensor since (in-place edits bump torch's version counter)
            gt_cid = content_ids(gt, xj.device, xj.dtype) if gt is not None else None
            if gt is not None:
                gt = gt.to(xj).contiguous()                                        # patch.py:65,70
                coin = _draw_coin(generator)
                local_is_src = coin > args["global_rand"]                          # patch.py:62
                res_ratio = args["global_merge_ratio"]
                if res_ratio <= 0:
                    # merge.py:364-365 returns a 2-tuple which patch.py:73 unpacks into 3 names
                    raise ValueError("not enough values to unpack (expected 3, got 2)")
                gl = merge.global_level(xj, gt, cur, Ml, local_is_src, res_ratio, args["align_batch"],
                                        want_indices, tokens=tsize, anchor_positions=gt_pos,
                                        planner=_planner(module, ("global", xj.shape[1]), order_alone=True))
                plan.global_level, plan.local_chunk = gl, (0 if local_is_src else 1)
                plan.anchors_in = gt
                off = 0 if local_is_src else gt.shape[1]                           # merge.py:459
                # loc: local position -> merged position.  patch.py:80: new anchors = u(merged) = the local tokens with every
                # merged local src row replaced by its matched global row -> one gather (amap) from [chunk | old anchors];
                # their token positions ride along for the next chunk's seeds
                loc, amap, anchors_pos = _lib.anchor_maps(gl.inv, off, gl.new_cur, Ml, L, tsize, gt_pos, _lib.SEED_MATCHER)
                anchors_out = _lib.gather_rows(xj, gt, amap)
                anchors_cid = None
                if gt_cid is not None and _lib.FOLD_KEYS and gl.new_cur.shape[1] <= 131072:
                    plan.fold_args = (gl.new_cur, L, gt_cid, gt.shape[1])
                if local_is_src and COMPACT_QUERIES and gl.Nd <= 131072:      # (vtm_compact_queries' bitmap lives in LDS)
                    qc, tmap, plan.q_count = _lib.compact_queries(loc, gl.Ns - gl.r, gl.Nd)
                    plan.q_rows = qc
                    plan.inv_q = _lib.compose(inv, tmap, L) if inv is not None else tmap
                    # local tokens that share a merged position become identical rows of the new anchors: tmap is an id
                    # per distinct position (old anchor rows that were copies of each other keep distinct ids -- a
                    # missed fold, never a wrong one)
                    anchors_cid = tmap
                else:
                    plan.q_rows, plan.inv_q = loc, inv
                plan._inv_parts, inv = (inv, loc), None      # composed lazily (MergePlan.inv)
                cur = gl.new_cur
                n_cur = cur.shape[1]

        plan.M = n_cur
        plan.gather_map, plan._inv, plan.pad_to = cur, inv, pad_to
        merged = plan.merged if (materialize or cur is None) else None             # cur None: F == 1, a view
        if args["merge_global"]:
            if anchors_out is not None:
                module.global_tokens = anchors_out
                if anchors_pos is not None:
                    anchors_out._vtm_pos = anchors_pos
                if anchors_cid is not None:
                    tag_content_ids(anchors_out, anchors_cid)
            elif plan.global_level is None:
                # patch.py:82: first chunk of a step stores its local tokens (device-resident, shared
                # with `merged`, which nothing mutates)
                if merged is None:
                    merged = plan.merged
                module.global_tokens = merged[:, :Ml] if merged.shape[1] != Ml else merged
                if _lib.SEED_MATCHER:      # (cur None: a single-frame chunk, the local tokens are the chunk's rows themselves)
                    module.global_tokens._vtm_pos = _lib.anchor_pos(cur, B, Ml, L, tsize, None, xj.device)
            mark_anchors_ready(getattr(module, "global_tokens", None))
            if exchange is not None:
                exchange.publish(xkey, module.global_tokens)

    def m(t: torch.Tensor, **kwarg) -> torch.Tensor:                               # patch.py:84
        tj = join_frame(t.contiguous(), fsize)
        if plan.gather_map is None:
            return tj
        if plan.anchors_in is not None:
            raise RuntimeError("the composed merge map of a global level needs the anchor tokens; "
                               "use the merged_tokens compute_merge returned")
        return _lib.gather_rows(tj, None, plan.gather_map)

    def u(t: torch.Tensor, resid: Optional[torch.Tensor] = None, **kwarg) -> torch.Tensor:   # patch.py:85
        if plan.inv is None:
            t = t[:, :plan.L]                                  # drop the 8-row padding of the attention path
            out = t if resid is None else t + join_frame(resid, fsize)
            return split_frame(out, fsize)
        r = None if resid is None else join_frame(resid.contiguous(), fsize)
        return split_frame(_lib.unmerge_add(t.contiguous(), plan.inv, r), fsize)

    m.plan = plan
    u.plan = plan
    return m, u, merged


# ----------------------------------------------------------------------------------------------------
# attn1 on merged tokens
# ----------------------------------------------------------------------------------------------------
def _pnp_num_inputs(attn: torch.nn.Module) -> Optional[int]:
    """If the reference's ``register_attention_control`` (utils/pnp_utils.py:39-106) replaced
    ``attn.forward`` with its ``sa_forward`` closure, recover ``num_inputs`` from that closure; our own
    ``vidtome_amd.pnp.register_attention_control`` stores it as ``attn.vtm_num_inputs``."""
    n = getattr(attn, "vtm_num_inputs", None)
    if n is not None:
        return int(n)
    fwd = attn.__dict__.get("forward")
    clo = getattr(fwd, "__closure__", None)
    if fwd is not None and clo:
        cells = dict(zip(fwd.__code__.co_freevars, clo))
        if "num_inputs" in cells:
            return int(cells["num_inputs"].cell_contents)
    return None


def _pnp_share_groups(attn: torch.nn.Module) -> int:
    """1, or the number of batch groups that share the source group's attention probabilities at this timestep
    (pnp_utils.py:57-67)."""
    sched = getattr(attn, "injection_schedule", None)
    if sched is not None:
        t = getattr(attn, "t", None)
        if t is not None and (t in sched or t == 1000):                 # pnp_utils.py:57-58
            n = _pnp_num_inputs(attn)
            if n is None:
                raise RuntimeError("PnP injection is registered on attn1 but num_inputs is unknown")
            return n
    return 1


def _built_here(device):
    """(event, stream id) behind a cached device tensor that was just built on the current stream: a chunk on ANOTHER stream may
    be the next to use it (scheduler.run_step(streams=...): the first two chunks of a run meet the caches empty) ..."""
    if not torch.cuda.is_available() or torch.device(device).type != "cuda":
        return None
    st = torch.cuda.current_stream(device)
    ev = torch.cuda.Event()
    ev.record(st)
    return ev, st.cuda_stream


def _built_before(mark, device) -> None:
    """... and waits for the build on the device before its own kernels read the tensor (same stream: nothing to do)."""
    if mark is not None:
        cur = torch.cuda.current_stream(device)
        if cur.cuda_stream != mark[1]:
            cur.wait_event(mark[0])


def _fused_weights(attn: torch.nn.Module, dtype, device):
    """[Wq; Wk] stacked once per module (one projection GEMM for q and k), cached on the module."""
    cache = attn.__dict__.get("_vtm_wcache")
    wq = attn.to_q.weight
    key = (wq.data_ptr(), attn.to_k.weight.data_ptr(), wq._version, attn.to_k.weight._version, dtype, device)
    if cache is None or cache[0] != key:
        wqk = torch.cat([attn.to_q.weight, attn.to_k.weight], dim=0).to(device=device, dtype=dtype).contiguous()
        bqk = None
        if getattr(attn.to_q, "bias", None) is not None or getattr(attn.to_k, "bias", None) is not None:
            zq = attn.to_q.bias if attn.to_q.bias is not None else torch.zeros_like(attn.to_q.weight[:, 0])
            zk = attn.to_k.bias if attn.to_k.bias is not None else torch.zeros_like(attn.to_k.weight[:, 0])
            bqk = torch.cat([zq, zk]).to(device=device, dtype=dtype)
        cache = (key, wqk, bqk, _built_here(device))
        attn.__dict__["_vtm_wcache"] = cache
    _built_before(cache[3], device)
    return cache[1], cache[2]


_HEAD_DIMS = (8, 16, 32, 40, 64, 80, 96, 128, 160)          # instantiations of attention_kernel (attention.hip)
_PLAIN_PROCESSORS = ("AttnProcessor", "AttnProcessor2_0", "XFormersAttnProcessor")
_warned = set()


def _warn_once(key: str, msg: str) -> None:
    if key not in _warned:
        _warned.add(key)
        import warnings
        warnings.warn(msg, stacklevel=3)


def _plain_linear(m) -> bool:
    """A projection the fused path may read `.weight` / `.bias` from: a plain Linear (or Diffusers' LoRA-compatible
    subclass with no LoRA attached).  PEFT / LoRA wrappers compute more than `x W^T + b`; reading the base weight
    would silently drop the adapter."""
    return (isinstance(m, torch.nn.Linear) and type(m).__name__ in ("Linear", "LoRACompatibleLinear")
            and getattr(m, "lora_layer", None) is None)


def _out_linear(attn: torch.nn.Module):
    to_out = attn.to_out
    return to_out[0] if isinstance(to_out, (torch.nn.ModuleList, torch.nn.Sequential, list, tuple)) else to_out  # pnp_utils.py:41-45


def fused_attention_ok(attn: torch.nn.Module, x: torch.Tensor, self_attn: bool = True) -> bool:
    """True when `attn(x)` is exactly `to_out[0](softmax(to_q(x) to_k(.)^T * scale) to_v(.))` -- the arithmetic of
    utils/pnp_utils.py:47-95 -- so that the fused path (projection GEMMs + vtm_attention) computes what the module
    would.  Anything else (LoRA / PEFT projections, custom processors or a replaced forward, group / cross norms,
    output rescaling or an inner residual, head dims without a kernel instantiation, dropout in training) makes the
    patched block call the module itself on the merged tokens, like the reference does (patch.py:157-162)."""
    if not (x.is_cuda and x.dim() == 3 and x.dtype in (torch.float16, torch.bfloat16, torch.float32)):
        return False
    if not all(hasattr(attn, a) for a in ("to_q", "to_k", "to_v", "to_out", "heads")):
        return False
    if not all(_plain_linear(m) for m in (attn.to_q, attn.to_k, attn.to_v, _out_linear(attn))):
        return False
    C = x.shape[-1]
    if attn.to_q.out_features != C or C % attn.heads or (C // attn.heads) not in _HEAD_DIMS:
        return False
    if any(getattr(attn, a, None) is not None for a in ("group_norm", "spatial_norm", "norm_cross", "norm_q", "norm_k")):
        return False
    if getattr(attn, "rescale_output_factor", 1.0) != 1.0 or getattr(attn, "residual_connection", False):
        return False
    proc = getattr(attn, "processor", None)
    if proc is not None and type(proc).__name__ not in _PLAIN_PROCESSORS:
        return False
    if "forward" in attn.__dict__ and not (self_attn and _pnp_num_inputs(attn) is not None):
        return False                              # replaced forward: only the PnP closure is understood
    to_out = attn.to_out
    if attn.training and isinstance(to_out, (torch.nn.ModuleList, torch.nn.Sequential)) and len(to_out) > 1 \
            and getattr(to_out[1], "p", 0.0) != 0.0:
        return False
    return True


# How attn1's projections are computed.  "rows": vtm_linear_rows, GEMMs whose A rows are fetched through the composed
# merge map (no merged tensor, V produced channel-major; the C ABI then covers attn1 end to end).  "blas": library GEMMs
# (torch -> hipBLASLt) over materialised merged tokens.  "auto" (default) picks per site what measures faster on MI355X
# (profiles/r02_*): the gather-fused kernel at C <= 320 (cfg-2 top blocks: k 73 vs 99 us, v^T 65 vs 81 us), the library
# at C >= 640, where its 256 x 256 macro-tiles move half the operand bytes of the 128 x 160 tiles of linear.hip.
PROJ_MODE = os.environ.get("VIDTOME_PROJ", "auto")
FUSED_PROJ = PROJ_MODE != "blas"                      # (tests toggle this to compare the two paths)
# Round 3: at C >= 640 "auto" no longer leaves the C ABI either -- the projections are PANEL GEMMs (csrc/ff.hip: the merged
# rows are gathered straight into the k-panel layout, vtm_gather_panels, and k / v^T / q / out are vtm_linear_panels
# launches; the un-merged C = 1280 sites take their panels from the LayerNorm itself).  VIDTOME_PROJ=panels forces that
# path everywhere, "blas" keeps the library GEMMs of rounds 1-2.


def _proj_dtypes_ok(attn: torch.nn.Module, dtype) -> bool:
    """All four projection weights (and the biases that exist) are of the tokens' dtype: the in-house projection kernels
    read raw 16-bit words and take the dtype from the TOKENS (a mixed-dtype module -- an fp32 to_out next to fp16
    to_q / to_k / to_v, say -- would otherwise be packed and multiplied as garbage instead of leaving the fused path)."""
    for lin in (attn.to_q, attn.to_k, attn.to_v, _out_linear(attn)):
        if lin.weight.dtype != dtype or (getattr(lin, "bias", None) is not None and lin.bias.dtype != dtype):
            return False
    return True


def fused_projections_ok(attn: torch.nn.Module, x: torch.Tensor) -> bool:
    """The gather-fused projection GEMM (vtm_linear_rows) takes fp16 / bf16 tokens with C % 32 == 0."""
    if not FUSED_PROJ or x.dtype not in (torch.float16, torch.bfloat16) or x.shape[-1] % 32 \
            or not _proj_dtypes_ok(attn, x.dtype):
        return False
    return PROJ_MODE == "rows" or (PROJ_MODE == "auto" and x.shape[-1] <= 320)


def panel_projections_ok(attn: torch.nn.Module, x: torch.Tensor) -> bool:
    """The panel-GEMM projections (vtm_gather_panels / vtm_layernorm_panels + vtm_linear_panels): fp16 / bf16 tokens,
    C % 64 == 0, no bias on to_v (V^T = W_v X^T is computed with the roles of the operands swapped)."""
    return (PROJ_MODE in ("auto", "panels") and x.dtype in (torch.float16, torch.bfloat16) and x.shape[-1] % 64 == 0
            and _proj_dtypes_ok(attn, x.dtype) and getattr(attn.to_v, "bias", None) is None)


def _panel_weight(lin: torch.nn.Module):
    """(weight as k-panels, fp32 bias or None) of a Linear, cached on the module."""
    return _packed(lin, "rows", lambda: (_lib.to_panels(lin.weight.detach().contiguous()),
                                        None if lin.bias is None else lin.bias.detach().float().contiguous()))


def self_attention_panels(attn: torch.nn.Module, x0: torch.Tensor, x1: Optional[torch.Tensor], rows: Optional[torch.Tensor],
                          q_rows: Optional[torch.Tensor] = None, q_count: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``attn1(merged)`` like self_attention_rows, with the projections as panel GEMMs: the merged rows (and the live-query
    rows) are gathered ONCE into the k-panel layout, sample b at rows b * Mpad (Mpad = M rounded up to 256), and
        k    = X W_k^T                     one launch over all samples  -> (B, Mpad, C)
        v^T  = W_v X_b^T                   one launch per sample, the weight as "token" operand -> (B, C, Mpad): no transpose
        q    = X_q W_q^T, out = O W_o^T + b
    Returns (B, Mq rounded up to 256, C); rows >= Mq (or the per-sample q_count) are not meaningful."""
    B, _, C = x0.shape
    M = x0.shape[1] if rows is None else rows.shape[1]
    heads = attn.heads
    scale = getattr(attn, "scale", None) or (C // heads) ** -0.5
    share = _pnp_share_groups(attn)
    # (q_rows under PnP sharing: the caller vouches that every sample of a group has the same rows -- align_batch)
    Mpad = _lib.panel_rows(M)
    xp = _lib.gather_panels(x0, x1, rows, None, M)                       # (C / 8, B * Mpad, 8)
    wq, bq = _panel_weight(attn.to_q)
    wk, bk = _panel_weight(attn.to_k)
    wv, _ = _panel_weight(attn.to_v)
    k_op = _lib.linear_panels(xp, B * Mpad, wk, C, bk).view(B, Mpad, C)
    vt = torch.empty((B, C, Mpad), dtype=x0.dtype, device=x0.device)
    M8 = (M + 7) // 8 * 8
    for b in range(B):
        _lib.linear_panels(wv, C, xp[:, b * Mpad:(b + 1) * Mpad], M8, None, out=vt[b])
    if q_rows is None:
        q_op = _lib.linear_panels(xp, B * Mpad, wq, C, bq).view(B, Mpad, C)
        o = _lib.attention(q_op, k_op, vt, heads, M, scale, share)
        Mq, Mqpad = M, Mpad
    else:
        Mq = q_rows.shape[1]
        Mqpad = _lib.panel_rows(Mq)
        qp = _lib.gather_panels(x0, x1, rows, q_rows, Mq)
        q_op = _lib.linear_panels(qp, B * Mqpad, wq, C, bq).view(B, Mqpad, C)
        o = _lib.attention_kv(q_op, k_op, vt, heads, Mq, M, scale, q_count=q_count, share_groups=share)
    wo, bo = _panel_weight(_out_linear(attn))
    op = _lib.to_panels(o.view(B * Mqpad, C))
    return _lib.linear_panels(op, B * Mqpad, wo, C, bo).view(B, Mqpad, C)


def unmerged_site(block: torch.nn.Module, x: torch.Tensor) -> bool:
    """patch.py:15-17,27: the block is too deep in the UNet to merge (downsample > max_downsample)."""
    info = block._tome_info
    if info["size"] is None:
        return False
    downsample = int(math.ceil(math.sqrt((info["size"][0] * info["size"][1]) // x.shape[1])))
    return downsample > info["args"]["max_downsample"]


def unmerged_self_attention_ok(block: torch.nn.Module, x: torch.Tensor) -> bool:
    norm, attn = block.norm1, block.attn1
    return (x.is_cuda and x.dim() == 3 and x.shape[1] % 8 == 0 and type(norm) is torch.nn.LayerNorm and len(norm.normalized_shape) == 1
            and norm.normalized_shape[0] == x.shape[-1] and (norm.weight is None or norm.weight.dtype == x.dtype)
            and (norm.bias is None or norm.bias.dtype == x.dtype) and fused_attention_ok(attn, x)
            and panel_projections_ok(attn, x) and not fused_projections_ok(attn, x) and _pnp_share_groups(attn) == 1
            and unmerged_site(block, x))


def unmerged_self_attention_residual(block: torch.nn.Module, hidden_states: torch.Tensor) -> torch.Tensor:
    """patch.py:139-169 at a site that does not merge (per-frame attention): ``attn1(norm1(h)) + h`` with norm1 writing
    k-panels (its output is only read by the three projections), ONE GEMM for q | k | v over all frames, the attention core
    per frame, and the output projection with bias and residual in its epilogue.  The caller checked
    ``unmerged_self_attention_ok``."""
    attn, norm = block.attn1, block.norm1
    BF, N, C = hidden_states.shape
    heads = attn.heads
    scale = getattr(attn, "scale", None) or (C // heads) ** -0.5
    hs = hidden_states.contiguous()
    n = BF * N

    def pack_qkv():
        w = torch.cat([attn.to_q.weight, attn.to_k.weight, attn.to_v.weight], dim=0).detach().contiguous()
        bs = [getattr(m, "bias", None) for m in (attn.to_q, attn.to_k, attn.to_v)]
        b = None if all(x is None for x in bs) else torch.cat(
            [torch.zeros(C, device=w.device) if x is None else x.detach().float() for x in bs]).contiguous()
        return _lib.to_panels(w), b
    wqkv, bqkv = _packed(attn, "qkv", pack_qkv)
    xp = _lib.layernorm_panels(hs, norm.weight, norm.bias, norm.eps)
    qkv = _lib.linear_panels(xp, n, wqkv, 3 * C, bqkv).view(BF, N, 3 * C)
    vt = _lib.transpose_cols(qkv, 2 * C, C)                              # (BF, C, N): V channel-major for the PV contraction
    o = _lib.attention(qkv[:, :, :C], qkv[:, :, C:2 * C], vt, heads, N, scale, 1)
    wo, bo = _panel_weight(_out_linear(attn))
    op = _lib.to_panels(o.view(n, C))
    return _lib.linear_panels(op, n, wo, C, bo, resid=hs.view(n, C)).view(BF, N, C)


def self_attention_segment(block: torch.nn.Module, hidden_states: torch.Tensor, encoder_hidden_states=None,
                           attention_mask=None, cross_attention_kwargs=None) -> torch.Tensor:
    """patch.py:146-169 for the plain-LayerNorm block: norm1 -> compute_merge -> attn1 -> unmerge -> + residual."""
    if (encoder_hidden_states is None or not block.only_cross_attention) and attention_mask is None \
            and not cross_attention_kwargs and unmerged_self_attention_ok(block, hidden_states):
        return unmerged_self_attention_residual(block, hidden_states)
    return patched_self_attention_segment(block, hidden_states, layer_norm(block.norm1, hidden_states),
                                          encoder_hidden_states, attention_mask, cross_attention_kwargs, None)


def _weight(m: torch.nn.Module, dtype) -> torch.Tensor:
    w = m.weight
    return w if (w.dtype == dtype and w.is_contiguous()) else w.to(dtype).contiguous()


def self_attention_rows(attn: torch.nn.Module, x0: torch.Tensor, x1: Optional[torch.Tensor],
                        rows: Optional[torch.Tensor], q_rows: Optional[torch.Tensor] = None,
                        q_count: Optional[torch.Tensor] = None, plan: Optional["MergePlan"] = None) -> torch.Tensor:
    """``attn1(merged)`` (patch.py:157-162; arithmetic of pnp_utils.py:47-95) with ``merged[b, i] = pool[b, rows[b, i]]``
    never materialised: every projection is a vtm_linear_rows GEMM that fetches its A rows through the composed merge
    map (pool = x0 | x1, ``rows`` None = the rows of x0 as they are), k and v for all M rows (v channel-major, what the
    PV contraction reads), q -- and the output projection -- only for the ``q_rows`` positions when given (the rows
    unmerge() reads).  Returns (B, Mq rounded up to 8, C); rows >= Mq are not meaningful.
    With a ``plan`` whose anchors carried content ids and a head dim that has spare contraction slots (40), k and v are
    projected for the duplicate-free key list only (MergePlan.key_fold) and every key carries log2 of its multiplicity."""
    B, _, C = x0.shape
    M = x0.shape[1] if rows is None else rows.shape[1]
    Mp = (M + 7) // 8 * 8
    heads = attn.heads
    scale = getattr(attn, "scale", None) or (C // heads) ** -0.5
    share = _pnp_share_groups(attn)
    # (q_rows under PnP sharing: the caller vouches that every sample of a group has the same rows -- align_batch)
    dt = x0.dtype
    wqk, bqk = _fused_weights(attn, dt, x0.device)
    bv = getattr(attn.to_v, "bias", None)
    fold = None
    if plan is not None and rows is not None and share == 1 and (C // heads) in (8, 40):
        fold = plan.key_fold(dt)
    if fold is not None:
        key_sel, k_bias, k_count = fold
        vt = _lib.linear_rows(x0, x1, rows, key_sel, M, _weight(attn.to_v, dt), None if bv is None else bv.to(dt),
                              transposed=True)
        k_op = _lib.linear_rows(x0, x1, rows, key_sel, M, wqk[C:], None if bqk is None else bqk[C:])
        Mq = M if q_rows is None else q_rows.shape[1]
        q_op = _lib.linear_rows(x0, x1, rows, q_rows, Mq, wqk[:C], None if bqk is None else bqk[:C])
        o = _lib.attention_kv(q_op, k_op, vt, heads, Mq, M, scale, q_count=q_count, k_fold=(k_count, k_bias))
        to_out = _out_linear(attn)
        return _lib.linear_rows(o, None, None, None, Mq, _weight(to_out, dt), None if to_out.bias is None else to_out.bias.to(dt))
    vt = _lib.linear_rows(x0, x1, rows, None, M, _weight(attn.to_v, dt), None if bv is None else bv.to(dt),
                          transposed=True)                                                       # (B, C, Mp)
    if q_rows is None:
        qk = _lib.linear_rows(x0, x1, rows, None, M, wqk, bqk)                                   # (B, Mp, 2C): q | k
        o = _lib.attention(qk[:, :, :C], qk[:, :, C:], vt, heads, M, scale, share)
        Mq = M
    else:
        Mq = q_rows.shape[1]
        k_op = _lib.linear_rows(x0, x1, rows, None, M, wqk[C:], None if bqk is None else bqk[C:])
        q_op = _lib.linear_rows(x0, x1, rows, q_rows, Mq, wqk[:C], None if bqk is None else bqk[:C])
        o = _lib.attention_kv(q_op, k_op, vt, heads, Mq, M, scale, q_count=q_count, share_groups=share)
    to_out = _out_linear(attn)
    return _lib.linear_rows(o, None, None, None, Mq, _weight(to_out, dt), None if to_out.bias is None else to_out.bias.to(dt))


def self_attention(attn: torch.nn.Module, x: torch.Tensor, M: Optional[int] = None,
                   q_rows: Optional[torch.Tensor] = None, q_count: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``attn1(x)`` for self-attention without mask (patch.py:157-162), arithmetic of pnp_utils.py:47-95:
    q,k,v projections -> softmax(q k^T * scale) v per head -> to_out[0] (+ dropout(0)), on MATERIALISED tokens with
    library GEMMs (torch -> hipBLASLt) for the projections: the path of fp32 models, of channel counts the
    gather-fused GEMM does not take, and of VIDTOME_PROJ=blas (see self_attention_rows for the default).
    x is (B, Mp, C) whose first M rows per sample are the sequence.  With ``q_rows`` (B, Mq) only those rows act
    as queries (every row stays a key / value) and the result is (B, Mq rounded up to 8, C) in q_rows order.
    The caller has checked ``fused_attention_ok(attn, x)``.

    fp16 / bf16 models run everything in their dtype.  fp32 models keep the four projections in fp32 and only the
    attention core's operands (q, k, v^T) are rounded to fp16 for the MFMA (fp32 accumulation, fp32 softmax): the
    result is within the 1e-3 class of the fp32 reference, and a warning says so once."""
    B, Mp, C = x.shape
    M = Mp if M is None else M
    heads = attn.heads
    scale = getattr(attn, "scale", None) or (C // heads) ** -0.5
    share = _pnp_share_groups(attn)
    # (q_rows under PnP sharing: the caller vouches that every sample of a group has the same rows -- align_batch)
    if x.shape[1] % 8:
        # keep the transposed V (B, C, Mp) 16-byte aligned per row
        pad = 8 - x.shape[1] % 8
        x = F.pad(x, (0, 0, 0, pad))
        Mp = x.shape[1]
    core = x.dtype
    if x.dtype != torch.float32 and PROJ_MODE != "blas":
        # an fp16 / bf16 model normally never gets here: say once that (and why) this module's projections are library GEMMs
        _warn_once("lib-proj", f"vidtome_amd: attn1 projections of a {x.dtype} model run as library GEMMs (C = {C}: the "
                               "hand-written projection kernels take C % 32 == 0 with all four projection weights in the "
                               "tokens' dtype and no bias on to_v at C > 320)")
    if x.dtype == torch.float32:
        core = torch.float16
        _warn_once("fp32-core", "vidtome_amd: fp32 model -- the self-attention core (QK^T, softmax, PV) runs on the "
                                "fp16 MFMA with fp32 accumulation; projections, matching and merging stay fp32")
    wqk, bqk = _fused_weights(attn, x.dtype, x.device)
    if q_rows is None:
        qk = F.linear(x, wqk, bqk).to(core)                              # (B, Mp, 2C): one GEMM for q and k
        q_op, k_op = qk[:, :, :C], qk[:, :, C:]
    else:
        xq = _lib.gather_rows(x, None, q_rows, pad_to=8)                 # (B, Mqp, C) query tokens
        q_op = F.linear(xq, wqk[:C], None if bqk is None else bqk[:C]).to(core)
        k_op = F.linear(x, wqk[C:], None if bqk is None else bqk[C:]).to(core)
    wv = attn.to_v.weight.to(x.dtype)
    if B <= 4:                                                           # merged sites: few long sequences
        vt = torch.empty((B, C, Mp), dtype=x.dtype, device=x.device)     # V^T straight from the GEMM:
        for bi in range(B):                                              # W_v @ x_b^T, transposed operand
            torch.mm(wv, x[bi].t(), out=vt[bi])                          # handled by the BLAS (no copy)
    else:                                                                # un-merged sites: many short ones
        vt = torch.matmul(wv, x.transpose(1, 2))
    if getattr(attn.to_v, "bias", None) is not None:
        vt = vt + attn.to_v.bias.to(x.dtype)[None, :, None]
    vt = vt.to(core)
    if q_rows is None:
        o = _lib.attention(q_op, k_op, vt, heads, M, scale, share)
    else:
        o = _lib.attention_kv(q_op, k_op, vt, heads, q_rows.shape[1], M, scale, q_count=q_count, share_groups=share)
        # (q_count: rows past the count are undefined; the output projection is row-wise, so they stay confined to
        # rows unmerge() never reads)
    to_out = _out_linear(attn)
    o = o.to(x.dtype)
    return F.linear(o, to_out.weight.to(o.dtype), None if to_out.bias is None else to_out.bias.to(o.dtype))


def norm_cross_attention_residual(norm: torch.nn.Module, attn: torch.nn.Module, hidden_states: torch.Tensor,
                                  encoder_hidden_states: torch.Tensor) -> torch.Tensor:
    """patch.py:171-185 for the plain case: ``attn2(norm2(hidden_states), encoder_hidden_states) + hidden_states`` with the
    query projection as a panel GEMM fed by the LayerNorm (vtm_layernorm_panels -> vtm_linear_panels: the normalised
    tokens are only ever read by to_q, so they are written once, as panels), k / v^T of the (few) conditioning tokens by the
    library, the attention core on vtm_attention_kv and the output projection + bias + residual as a panel GEMM too.
    The caller has checked ``fused_cross_ok``."""
    B, N, C = hidden_states.shape
    heads = attn.heads
    scale = getattr(attn, "scale", None) or (C // heads) ** -0.5
    dt = hidden_states.dtype
    enc = encoder_hidden_states.to(dt)
    Mk = enc.shape[1]
    Mkp = (Mk + 7) // 8 * 8
    if Mkp != Mk:
        enc = F.pad(enc, (0, 0, 0, Mkp - Mk))
    hs = hidden_states.contiguous()
    n = B * N
    wq, bq = _packed(attn.to_q, "rows", lambda: (_lib.to_panels(attn.to_q.weight.detach().contiguous()),
                                                None if attn.to_q.bias is None else attn.to_q.bias.detach().float().contiguous()))
    to_out = _out_linear(attn)
    wo, bo = _packed(to_out, "rows", lambda: (_lib.to_panels(to_out.weight.detach().contiguous()),
                                              None if to_out.bias is None else to_out.bias.detach().float().contiguous()))
    xp = _lib.layernorm_panels(hs, norm.weight, norm.bias, norm.eps)
    q = _lib.linear_panels(xp, n, wq, C, bq).view(B, N, C)
    if N % 8:
        raise RuntimeError("norm_cross_attention_residual: token count must be a multiple of 8")
    if enc.shape[2] % 64 == 0 and attn.to_k.weight.dtype == dt and attn.to_v.weight.dtype == dt:
        ep = _lib.to_panels(enc.reshape(B * Mkp, enc.shape[2]))          # the (few) conditioning tokens: 77 per frame in SD
        wk, bk = _panel_weight(attn.to_k)
        wv, bv = _panel_weight(attn.to_v)
        k = _lib.linear_panels(ep, B * Mkp, wk, C, bk).view(B, Mkp, C)
        vt = _lib.linear_panels(ep, B * Mkp, wv, C, bv).view(B, Mkp, C).transpose(1, 2).contiguous()   # (B, C, Mkp)
    else:
        _warn_once("lib-cross-kv", f"vidtome_amd: attn2's k / v projections run as library GEMMs (conditioning width "
                                   f"{enc.shape[2]} is not a multiple of 64, or to_k / to_v are not of the tokens' dtype)")
        lin = lambda m, t: F.linear(t, m.weight.to(t.dtype), None if m.bias is None else m.bias.to(t.dtype))
        k = lin(attn.to_k, enc)
        vt = lin(attn.to_v, enc).transpose(1, 2).contiguous()
    o = _lib.attention_kv(q, k, vt, heads, N, Mk, scale)
    op = _lib.to_panels(o.view(n, C))
    return _lib.linear_panels(op, n, wo, C, bo, resid=hs.view(n, C)).view(B, N, C)


def fused_cross_ok(norm: torch.nn.Module, attn: torch.nn.Module, x: torch.Tensor, encoder_hidden_states,
                   attention_mask, kwargs) -> bool:
    return (FF_MODE == "panels" and encoder_hidden_states is not None and attention_mask is None and not kwargs
            and encoder_hidden_states.dim() == 3 and x.dim() == 3 and x.is_cuda and x.dtype in (torch.float16, torch.bfloat16)
            and type(norm) is torch.nn.LayerNorm and len(norm.normalized_shape) == 1
            and norm.normalized_shape[0] == x.shape[-1] and x.shape[-1] % 64 == 0 and x.shape[1] % 8 == 0
            and (norm.weight is None or norm.weight.dtype == x.dtype) and (norm.bias is None or norm.bias.dtype == x.dtype)
            and _proj_dtypes_ok(attn, x.dtype) and fused_attention_ok(attn, x, self_attn=False))


def cross_attention(attn: torch.nn.Module, x: torch.Tensor, encoder_hidden_states: Optional[torch.Tensor],
                    attention_mask=None, **kwargs) -> torch.Tensor:
    """`self.attn2(norm_hidden_states, encoder_hidden_states=..., attention_mask=...)` (patch.py:178-183) -- the
    un-merged tokens attending to the conditioning (77 text tokens in SD).  The plain case (projection Linears, no
    mask, no processor kwargs) runs on vtm_attention_kv; everything else is the module's own forward."""
    plain = (encoder_hidden_states is not None and attention_mask is None and not kwargs
             and encoder_hidden_states.dim() == 3 and x.dtype in (torch.float16, torch.bfloat16)
             and fused_attention_ok(attn, x, self_attn=False))
    if not plain:
        return attn(x, encoder_hidden_states=encoder_hidden_states, attention_mask=attention_mask, **kwargs)
    B, N, C = x.shape
    enc = encoder_hidden_states.to(x.dtype)
    Mk = enc.shape[1]
    heads = attn.heads
    scale = getattr(attn, "scale", None) or (C // heads) ** -0.5
    Np, Mkp = (N + 7) // 8 * 8, (Mk + 7) // 8 * 8
    if Np != N:
        x = F.pad(x, (0, 0, 0, Np - N))
    if Mkp != Mk:
        enc = F.pad(enc, (0, 0, 0, Mkp - Mk))
    lin = lambda m, t: F.linear(t, m.weight.to(t.dtype), None if m.bias is None else m.bias.to(t.dtype))
    q = lin(attn.to_q, x)
    k = lin(attn.to_k, enc)
    vt = lin(attn.to_v, enc).transpose(1, 2).contiguous()               # (B, C, Mkp): 77 keys, negligible
    o = _lib.attention_kv(q, k, vt, heads, N, Mk, scale)
    return lin(_out_linear(attn), o)[:, :N]


# ----------------------------------------------------------------------------------------------------
# the rest of the block as panel GEMMs (csrc/ff.hip)
# ----------------------------------------------------------------------------------------------------
# VIDTOME_FF: "panels" (default) = norm3 -> GEGLU projection with the gated activation in the GEMM's epilogue -> output
# Linear with bias and residual, all hand-written (vtm_layernorm_panels, vtm_ff_geglu, vtm_linear_panels: the 8C-wide
# projection is never written); "blas" = library GEMMs around vtm_geglu (rounds 1-2).  The same switch covers the
# cross-attention's query projection (norm2 -> panels -> vtm_linear_panels).
FF_MODE = os.environ.get("VIDTOME_FF", "panels")


def _packed(module: torch.nn.Module, key: str, build):
    """Weights repacked for the panel GEMMs, cached on the module and rebuilt when the parameter changes."""
    cache = module.__dict__.setdefault("_vtm_packed", {})
    params = [p for p in module.parameters()]
    tag = tuple((p.data_ptr(), p._version, p.dtype, p.device) for p in params)
    hit = cache.get(key)
    dev = params[0].device if params else None
    if hit is None or hit[0] != tag:
        hit = cache[key] = (tag, build(), _built_here(dev) if dev is not None else None)
    if dev is not None:
        _built_before(hit[2], dev)
    return hit[1]


def _geglu_ff(ff: torch.nn.Module):
    """(GEGLU projection Linear, output Linear) of a Diffusers FeedForward [GEGLU, Dropout, Linear], else None."""
    net = getattr(ff, "net", None)
    if (net is not None and len(net) == 3 and net[0].__class__.__name__ == "GEGLU" and hasattr(net[0], "proj")
            and _plain_linear(net[0].proj) and _plain_linear(net[2])
            and (not ff.training or getattr(net[1], "p", 0.0) == 0.0)):      # Dropout(0.0) is the identity in any mode
        return net[0].proj, net[2]
    return None


def fused_ff_ok(norm: torch.nn.Module, ff: torch.nn.Module, x: torch.Tensor) -> bool:
    if FF_MODE != "panels" or not x.is_cuda or x.dtype not in (torch.float16, torch.bfloat16):
        return False
    lin = _geglu_ff(ff)
    if lin is None or type(norm) is not torch.nn.LayerNorm or len(norm.normalized_shape) != 1:
        return False
    proj, out = lin
    C = x.shape[-1]
    D = proj.out_features // 2
    return (norm.normalized_shape[0] == C and C % 64 == 0 and C <= 2048 and proj.in_features == C and D % 64 == 0
            and out.in_features == D and out.out_features == C and proj.weight.dtype == x.dtype
            and out.weight.dtype == x.dtype and (norm.weight is None or norm.weight.dtype == x.dtype)
            and (norm.bias is None or norm.bias.dtype == x.dtype))


def norm_feed_forward_residual(norm: torch.nn.Module, ff: torch.nn.Module, hidden_states: torch.Tensor) -> torch.Tensor:
    """patch.py:187-199 for the plain-LayerNorm block: ``ff(norm3(hidden_states)) + hidden_states`` as three launches:
    LayerNorm -> k-panels, GEGLU projection with the gated activation in its epilogue -> k-panels, output Linear + bias +
    residual -> token rows.  The caller has checked ``fused_ff_ok``."""
    proj, out = _geglu_ff(ff)
    C = hidden_states.shape[-1]
    D = proj.out_features // 2

    def pack_w1():
        t = torch.arange(D // 64, device=proj.weight.device)[:, None] * 64 + torch.arange(64, device=proj.weight.device)[None, :]
        order = torch.cat([t, t + D], dim=1).reshape(-1).to(torch.int32)       # tile t: 64 value rows, then their gate rows
        b = None if proj.bias is None else proj.bias.detach().float()[order.long()].contiguous()
        return _lib.to_panels(proj.weight.detach().contiguous(), order), b

    def pack_w2():
        b = None if out.bias is None else out.bias.detach().float().contiguous()
        return _lib.to_panels(out.weight.detach().contiguous()), b

    w1, b1 = _packed(proj, "geglu", pack_w1)
    w2, b2 = _packed(out, "rows", pack_w2)
    hs = hidden_states.contiguous()
    n = hs.numel() // C
    xp = _lib.layernorm_panels(hs, norm.weight, norm.bias, norm.eps)
    hp = _lib.ff_geglu(xp, n, w1, D, b1)
    return _lib.linear_panels(hp, n, w2, C, b2, resid=hs.view(n, C)).view(hidden_states.shape)


def feed_forward(ff: torch.nn.Module, x: torch.Tensor) -> torch.Tensor:
    """`self.ff(norm_hidden_states)` (patch.py:192).  The Diffusers feed-forward of SD blocks is
    [GEGLU(proj: Linear C -> 8C), Dropout, Linear 4C -> C]; when that shape is recognised the gated activation
    runs as vtm_geglu (the two Linears stay library GEMMs), otherwise the module runs unchanged."""
    net = getattr(ff, "net", None)
    if (net is not None and len(net) == 3 and net[0].__class__.__name__ == "GEGLU" and hasattr(net[0], "proj")
            and _plain_linear(net[0].proj) and _plain_linear(net[2]) and x.is_cuda and not ff.training
            and x.dtype in (torch.float16, torch.bfloat16, torch.float32) and net[0].proj.out_features % 16 == 0):
        return net[2](_lib.geglu(net[0].proj(x)))
    return ff(x)


# ----------------------------------------------------------------------------------------------------
# patched block
# ----------------------------------------------------------------------------------------------------
def patched_self_attention_segment(block: torch.nn.Module, hidden_states: torch.Tensor,
                                   norm_hidden_states: torch.Tensor, encoder_hidden_states=None,
                                   attention_mask=None, cross_attention_kwargs=None, gate_msa=None
                                   ) -> torch.Tensor:
    """patch.py:148-169: compute_merge -> attn1(merged) -> unmerge -> + residual."""
    cross_attention_kwargs = cross_attention_kwargs if cross_attention_kwargs is not None else {}
    custom = encoder_hidden_states is not None and block.only_cross_attention
    fused = not (custom or attention_mask is not None or cross_attention_kwargs) \
        and fused_attention_ok(block.attn1, norm_hidden_states)
    by_rows = fused and fused_projections_ok(block.attn1, norm_hidden_states)
    by_panels = fused and not by_rows and panel_projections_ok(block.attn1, norm_hidden_states)
    m_a, u_a, merged = compute_merge(block, norm_hidden_states, block._tome_info, materialize=not (by_rows or by_panels))
    plan = getattr(m_a, "plan", None)
    if not fused:
        # not the hot path (SD never masks self-attention nor makes attn1 a cross-attention; LoRA'd / custom
        # attention modules are not the plain arithmetic): run the module's own attention on the merged tokens
        # exactly as the reference does
        M = plan.M if plan is not None else merged.shape[1]
        attn_output = block.attn1(merged[:, :M],
                                  encoder_hidden_states=encoder_hidden_states if block.only_cross_attention else None,
                                  attention_mask=attention_mask, **cross_attention_kwargs)
    else:
        # (PnP injection reads the source sample's q for every group: the live rows must be the same rows in every sample,
        # which align_batch guarantees -- the levels' indices are computed once on the batch-folded tokens, merge.py:73-76)
        live = (plan is not None and plan.q_rows is not None and gate_msa is None and LIVE_QUERIES
                and (_pnp_share_groups(block.attn1) == 1 or plan.aligned))
        q_rows = plan.q_rows if live else None
        q_count = plan.q_count if live else None
        if by_rows or by_panels:
            sa = self_attention_rows if by_rows else self_attention_panels
            if plan is None:                                              # block does not merge: per-frame attention
                attn_output = sa(block.attn1, norm_hidden_states.contiguous(), None, None)
            else:
                attn_output = sa(block.attn1, plan.x_joined, plan.anchors_in, plan.gather_map, q_rows, q_count,
                                 **({"plan": plan} if by_rows else {}))
        else:
            attn_output = self_attention(block.attn1, merged, plan.M if plan is not None else None, q_rows, q_count)
        if live:
            # rows are the chunk's local merged tokens: unmerge with the local levels' map alone
            # (= the global level's unmerge, merge.py:439-460, folded into the choice of queries)
            fs = plan.fsize
            r = join_frame(hidden_states.contiguous(), fs)
            if plan.inv_q is None:                                        # single-frame chunk: no local level
                return split_frame(attn_output[:, :plan.L] + r, fs)
            return split_frame(_lib.unmerge_add(attn_output.contiguous(), plan.inv_q, r), fs)
    if gate_msa is not None:
        attn_output = gate_msa.unsqueeze(1) * attn_output
    if plan is None:
        return attn_output[:, :hidden_states.shape[1]] + hidden_states
    return u_a(attn_output, resid=hidden_states)                          # patch.py:168-169 fused


def layer_norm(norm: torch.nn.Module, x: torch.Tensor) -> torch.Tensor:
    """`self.norm1(hidden_states)` (patch.py:146; also norm2 / norm3, patch.py:173-176, 187): a plain
    torch.nn.LayerNorm over the channel axis runs as vtm_layernorm; anything else (AdaLayerNorm variants,
    unusual shapes) is the module's own business."""
    if (type(norm) is torch.nn.LayerNorm and x.is_cuda and len(norm.normalized_shape) == 1
            and x.shape[-1] == norm.normalized_shape[0] and x.shape[-1] % 8 == 0 and x.shape[-1] <= 2048
            and x.dtype in (torch.float16, torch.bfloat16, torch.float32)
            and (norm.weight is None or norm.weight.dtype == x.dtype)
            and (norm.bias is None or norm.bias.dtype == x.dtype)):
        return _lib.layernorm(x, norm.weight, norm.bias, norm.eps)
    return norm(x)


def make_diffusers_tome_block(block_class: Type[torch.nn.Module]) -> Type[torch.nn.Module]:
    """vidtome/patch.py:119-203: patched class made on the fly, named ToMeBlock, keeps ``_parent``."""

    class ToMeBlock(block_class):
        _parent = block_class

        def forward(self, hidden_states, attention_mask=None, encoder_hidden_states=None,
                    encoder_attention_mask=None, timestep=None, cross_attention_kwargs=None,
                    class_labels=None) -> torch.Tensor:
            gate_msa = None
            if self.use_ada_layer_norm:                                            # patch.py:139-146
                norm_hidden_states = self.norm1(hidden_states, timestep)
            elif self.use_ada_layer_norm_zero:
                norm_hidden_states, gate_msa, shift_mlp, scale_mlp, gate_mlp = self.norm1(
                    hidden_states, timestep, class_labels, hidden_dtype=hidden_states.dtype)
            else:
                norm_hidden_states = None

            # 1. self-attention on merged tokens (the hot path)                    # patch.py:148-169
            if norm_hidden_states is None:
                hidden_states = self_attention_segment(self, hidden_states, encoder_hidden_states, attention_mask,
                                                       cross_attention_kwargs)
            else:
                hidden_states = patched_self_attention_segment(
                    self, hidden_states, norm_hidden_states, encoder_hidden_states, attention_mask,
                    cross_attention_kwargs, gate_msa)

            cross_attention_kwargs = cross_attention_kwargs if cross_attention_kwargs is not None else {}
            if self.attn2 is not None:                                             # patch.py:171-185
                if not self.use_ada_layer_norm and fused_cross_ok(self.norm2, self.attn2, hidden_states,
                                                                  encoder_hidden_states, encoder_attention_mask,
                                                                  cross_attention_kwargs):
                    hidden_states = norm_cross_attention_residual(self.norm2, self.attn2, hidden_states,
                                                                  encoder_hidden_states)
                else:
                    norm_hidden_states = (self.norm2(hidden_states, timestep) if self.use_ada_layer_norm
                                          else layer_norm(self.norm2, hidden_states))
                    attn_output = cross_attention(self.attn2, norm_hidden_states, encoder_hidden_states,
                                                  encoder_attention_mask, **cross_attention_kwargs)
                    hidden_states = attn_output + hidden_states

            if not self.use_ada_layer_norm_zero and fused_ff_ok(self.norm3, self.ff, hidden_states):   # patch.py:187-199
                return norm_feed_forward_residual(self.norm3, self.ff, hidden_states)
            norm_hidden_states = layer_norm(self.norm3, hidden_states)
            if self.use_ada_layer_norm_zero:
                norm_hidden_states = norm_hidden_states * (1 + scale_mlp[:, None]) + shift_mlp[:, None]
            ff_output = feed_forward(self.ff, norm_hidden_states)
            if self.use_ada_layer_norm_zero:
                ff_output = gate_mlp.unsqueeze(1) * ff_output
            return ff_output + hidden_states

    return ToMeBlock


def hook_tome_model(model: torch.nn.Module):
    """vidtome/patch.py:206-212: forward pre-hook recording the latent (H, W)."""
    def hook(module, args):
        module._tome_info["size"] = (args[0].shape[2], args[0].shape[3])
        return None

    model._tome_info["hooks"].append(model.register_forward_pre_hook(hook))


def hook_tome_module(module: torch.nn.Module):
    """vidtome/patch.py:215-231: lazily create the block generator; all blocks fork the same state so their
    draws stay in lock-step within one pass."""
    def hook(module, args):
        gmode = module._tome_info["args"].get("generator_device", None)
        if not hasattr(module, "generator"):
            module.generator = init_generator(args[0].device, mode=gmode)
        elif module.generator.device != args[0].device and (gmode or GENERATOR_MODE_DEFAULT()) == "device":
            # patch.py:221-224: the model moved to another device -> fork that device's state (opt-in stream only: the
            # default CPU stream does not depend on where the tensors live)
            module.generator = init_generator(args[0].device, fallback=module.generator, mode=gmode)
        return None

    module._tome_info["hooks"].append(module.register_forward_pre_hook(hook))


def GENERATOR_MODE_DEFAULT() -> str:
    from . import utils
    return utils.GENERATOR_MODE


def apply_patch(model: torch.nn.Module, local_merge_ratio: float = 0.9, merge_global: bool = False,
                global_merge_ratio=0.8, max_downsample: int = 2, seed: int = 123, batch_size: int = 2,
                include_control: bool = False, align_batch: bool = False, target_stride: int = 4,
                global_rand=0.5, *, generator_device: Optional[str] = None):
    """vidtome/patch.py:234-334 -- same arguments, defaults, return value and errors.

    One keyword-only addition: ``generator_device`` = None (the process default, VIDTOME_GENERATOR, "cpu" unless set) |
    "cpu" | "device".  "cpu": the block generators fork the CPU RNG state wherever the model lives -- the draw stream of the
    reference's CPU path (the parity oracle).  "device": the reference's own rule (vidtome/utils.py:18-30, patch.py:31-32,
    221-224): on a GPU they fork ``torch.cuda.get_rng_state()`` and draw on the device generator (re-forked when the model
    changes device), which reproduces a GPU run of the reference draw for draw at the price of a host read-back per draw."""
    if generator_device not in (None, "cpu", "device"):
        raise ValueError(f"generator_device must be None, 'cpu' or 'device', got {generator_device!r}")
    _lib.lib()   # fail loudly right here if the HIP library is missing: there is no fallback
    remove_patch(model)                                                            # patch.py:277
    is_diffusers = isinstance_str(model, "DiffusionPipeline") or isinstance_str(model, "ModelMixin")
    if not is_diffusers:
        if not hasattr(model, "model") or not hasattr(model.model, "diffusion_model"):
            raise RuntimeError(
                "Provided model was not a Stable Diffusion / Latent Diffusion model, as expected.")
        # the reference's LDM branch (make_tome_block, patch.py:94-116) unpacks 6 values from the 3-tuple
        # compute_merge returns and cannot run; it is out of scope here as well
        raise RuntimeError("LDM (non-diffusers) models are not supported: the reference's own LDM patch "
                           "path is broken (patch.py:105-106 vs :91)")
    diffusion_model = model.unet if hasattr(model, "unet") else model              # patch.py:290
    if isinstance_str(model, "StableDiffusionControlNetPipeline") and include_control:
        diffusion_models = [diffusion_model, model.controlnet]                     # patch.py:292-295
    else:
        diffusion_models = [diffusion_model]

    for diffusion_model in diffusion_models:
        diffusion_model._tome_info = {                                             # patch.py:298-313
            "size": None,
            "hooks": [],
            "args": {
                "max_downsample": max_downsample,
                "generator": None,
                "seed": seed,
                "batch_size": batch_size,
                "align_batch": align_batch,
                "merge_global": merge_global,
                "global_merge_ratio": global_merge_ratio,
                "local_merge_ratio": local_merge_ratio,
                "global_rand": global_rand,
                "target_stride": target_stride,
                "generator_device": generator_device,          # (not a reference key; see the docstring)
            },
        }
        hook_tome_model(diffusion_model)
        for _, module in diffusion_model.named_modules():
            if isinstance_str(module, "BasicTransformerBlock"):                    # patch.py:319
                module.__class__ = make_diffusers_tome_block(module.__class__)
                module._tome_info = diffusion_model._tome_info
                hook_tome_module(module)
                if not hasattr(module, "use_ada_layer_norm_zero"):                 # patch.py:330-332
                    module.use_ada_layer_norm = False
                    module.use_ada_layer_norm_zero = False
    return model


def _patched_roots(model: torch.nn.Module, controlnet_on_unet: bool):
    """The module trees the reference walks: the UNet (`model.unet` for a pipeline) and, if present, a ControlNet.
    `remove_patch` looks for `.controlnet` on the UNet it just unwrapped (patch.py:338-341, a quirk: a pipeline's
    ControlNet hangs off the pipeline), `update_patch` / `collect_from_patch` on the object they were given
    (patch.py:359-362, 374-377)."""
    unet = model.unet if hasattr(model, "unet") else model
    owner = unet if controlnet_on_unet else model
    return unet, ([unet, owner.controlnet] if hasattr(owner, "controlnet") else [unet])


def remove_patch(model: torch.nn.Module):
    """vidtome/patch.py:337-355: drop the hooks, give every ToMeBlock its parent class back; returns the UNet."""
    unet, roots = _patched_roots(model, controlnet_on_unet=True)
    # the launch planners' pinned counter buffers are written by asynchronous copies the LIBRARY issued (torch's host
    # allocator does not know about them): let the last ones land before the buffers go back to its pool
    if torch.cuda.is_available() and any("_vtm_match_plans" in m.__dict__ for root in roots for m in root.modules()):
        torch.cuda.synchronize()
    for root in roots:
        for module in root.modules():
            info = getattr(module, "_tome_info", None)
            if info is not None:
                for hook in info["hooks"]:
                    hook.remove()
                info["hooks"].clear()
            if module.__class__.__name__ == "ToMeBlock":
                module.__class__ = module._parent
            # the fused path's derived copies of the weights (panel-layout packs, stacked q|k weights): ~0.4-0.5 GB of fp16
            # for SD-1.5; they are rebuilt on demand if the model is patched again
            module.__dict__.pop("_vtm_packed", None)
            module.__dict__.pop("_vtm_wcache", None)
            module.__dict__.pop("_vtm_match_plans", None)      # the matcher's launch planners (pinned 32-byte buffers)
    _lib.release_workspaces()            # the cached scratch buffers of the patched path (re-created on demand)
    return roots[-1]                     # the reference returns its loop variable: the last tree walked


def update_patch(model: torch.nn.Module, **kwargs):
    """vidtome/patch.py:358-370: setattr on every module that carries ``_tome_info`` (the root included);
    returns the last tree walked, like the reference."""
    _, roots = _patched_roots(model, controlnet_on_unet=False)
    for root in roots:
        for module in root.modules():
            if hasattr(module, "_tome_info"):
                for k, v in kwargs.items():
                    setattr(module, k, v)
    return roots[-1]


def collect_from_patch(model: torch.nn.Module, attr="tome"):
    """vidtome/patch.py:373-387: {module name: getattr(module, attr)} over the patched trees."""
    _, roots = _patched_roots(model, controlnet_on_unet=False)
    found = {}
    for root in roots:
        for name, module in root.named_modules():
            if hasattr(module, attr):
                found[name] = getattr(module, attr)
    return found

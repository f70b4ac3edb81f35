"""Chunk-parallel execution of the global-merge anchor chain: one process per GPU, one video chunk per rank.

In the reference the chunks of a denoising step run sequentially on one device and the per-block anchor
tokens flow chunk -> chunk through ``module.global_tokens`` (generate.py:215-219, patch.py:59-82).  Local
merging is independent per chunk; only this anchor hand-off couples ranks.  Two exchanges are provided:

``RingExchange`` (exact)   rank k receives the anchors rank k-1 produced at the SAME block (point-to-point,
                           one xGMI link) right before its global level and sends its own updated anchors to
                           rank k+1 right after it -- a wavefront pipeline whose skew is one compute_merge per
                           hop.  Together with ``replay_draws`` (every rank replays the generator draws of the
                           chunks before it) this reproduces the sequential run's indices bit-exactly.
``AllGatherExchange``      the north-star mode: every rank all-gathers its *local* merged tokens per block
                           (RCCL all-gather over xGMI) and merges against the tokens of rank (k-1) mod n.  No
                           serial dependency; a documented semantic deviation (anchors are "parallel", not
                           chained), reported separately.

The classes only move tensors through ``torch.distributed`` (backend "nccl" == RCCL on ROCm; "gloo" in the CPU
tests) and are independent of how the merge itself is computed.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence

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
                 first_has_anchors: bool = False) -> None:
    """Consume the draws of a run of consecutive chunks (the ones another rank processes)."""
    has = first_has_anchors
    for f in chunk_frames:
        simulate_block_draws(generator, f, tokens_per_frame, args, has)
        has = True


# ----------------------------------------------------------------------------------------------------
# exchanges
# ----------------------------------------------------------------------------------------------------
class RingExchange:
    """Exact mode.  ``anchors_for`` blocks (on the stream for NCCL) until the previous rank's anchors of this
    block arrive; ``publish`` forwards the updated anchors to the next rank."""

    def __init__(self, group=None):
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.bytes_sent = 0

    def anchors_for(self, key: str, local_tokens_fn: Callable[[], torch.Tensor], like: torch.Tensor
                    ) -> Optional[torch.Tensor]:
        if self.rank == 0:
            return None                                    # first chunk of the step (generate.py:233-236)
        # chunk lengths differ (first chunk is random-length, generate.py:176-178): shape header first
        hdr = torch.empty(3, dtype=torch.int64, device=like.device)
        dist.recv(hdr, src=self.rank - 1, group=self.group)
        buf = torch.empty(tuple(int(v) for v in hdr.tolist()), dtype=like.dtype, device=like.device)
        dist.recv(buf, src=self.rank - 1, group=self.group)
        return buf

    def publish(self, key: str, anchors: torch.Tensor) -> None:
        if self.rank + 1 < self.world:
            a = anchors.contiguous()
            hdr = torch.tensor(list(a.shape), dtype=torch.int64, device=a.device)
            dist.send(hdr, dst=self.rank + 1, group=self.group)
            dist.send(a, dst=self.rank + 1, group=self.group)
            self.bytes_sent += a.numel() * a.element_size()


class AllGatherExchange:
    """North-star mode: all-gather of every rank's local merged tokens; rank k merges against rank k-1's."""

    def __init__(self, group=None):
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.bytes_gathered = 0

    def anchors_for(self, key: str, local_tokens_fn: Callable[[], torch.Tensor], like: torch.Tensor
                    ) -> Optional[torch.Tensor]:
        if self.world == 1:
            return None
        local = local_tokens_fn().contiguous()
        out = torch.empty((self.world,) + tuple(local.shape), dtype=local.dtype, device=local.device)
        if dist.get_backend(self.group) == "gloo":
            parts = [torch.empty_like(local) for _ in range(self.world)]
            dist.all_gather(parts, local, group=self.group)
            out = torch.stack(parts)
        else:
            dist.all_gather_into_tensor(out, local, group=self.group)
        self.bytes_gathered += out.numel() * out.element_size()
        return out[(self.rank - 1) % self.world]

    def publish(self, key: str, anchors: torch.Tensor) -> None:
        return None


# ----------------------------------------------------------------------------------------------------
# wiring into the patched model
# ----------------------------------------------------------------------------------------------------
def enable(model: torch.nn.Module, exchange) -> None:
    """Attach ``exchange`` to every patched block; compute_merge consults ``module._vtm_exchange``."""
    root = model.unet if hasattr(model, "unet") else model
    for name, m in root.named_modules():
        if m.__class__.__name__ == "ToMeBlock":
            m._vtm_exchange = exchange
            m._vtm_key = name


def disable(model: torch.nn.Module) -> None:
    root = model.unet if hasattr(model, "unet") else model
    for _, m in root.named_modules():
        if hasattr(m, "_vtm_exchange"):
            del m._vtm_exchange

"""ctypes binding of libvidtome_hip.so (the C ABI declared in include/vidtome_hip.h).

PyTorch is plumbing here: it owns device memory (caching allocator) and the current HIP stream; every
function below passes raw device pointers + sizes to the library.  There is NO fallback: if the library
is missing or a call fails, a RuntimeError is raised.

torch must be imported before the library is loaded so that both bind the same HIP runtime
(libamdhip64.so.7 is resolved by SONAME against the copy torch already mapped).
"""
from __future__ import annotations

import ctypes
import functools
import os
import threading
from typing import Optional, Tuple

import torch  # noqa: F401  (must precede the CDLL load, see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
ABI_VERSION = 2   # include/vidtome_hip.h VTM_ABI_VERSION (2: flags_out of the filtered matchers is 8 int32)
LIB_PATH = os.environ.get("VIDTOME_HIP_LIB") or os.path.join(_HERE, "lib", "libvidtome_hip.so")

VTM_F32, VTM_F16, VTM_BF16 = 0, 1, 2
ROW_PAD, K_PAD = 256, 32

_DT = {torch.float32: VTM_F32, torch.float16: VTM_F16, torch.bfloat16: VTM_BF16}

_vp, _i64, _int, _f32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_float
_lib: Optional[ctypes.CDLL] = None

_SIGNATURES = {
    "vtm_version": ([], _int),
    "vtm_last_error": ([], ctypes.c_char_p),
    "vtm_build_ablations": ([], _int),
    "vtm_pad_rows": ([_i64], _i64),
    "vtm_pad_k": ([_i64], _i64),
    "vtm_normalize_gather": ([_vp, _i64, _vp, _i64, _int, _i64, _i64, _vp, _i64, _vp, _vp, _i64, _i64, _vp], _int),
    "vtm_match": ([_vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _int, _vp, _vp], _int),
    "vtm_match_filtered_ws_bytes": ([_i64, _i64, _i64, _i64, _int], ctypes.c_size_t),
    "vtm_match_filtered": ([_vp, _i64, _vp, _i64, _int, _i64, _i64, _vp, _i64, _vp, _i64, _int, _vp, ctypes.c_size_t,
                            _vp, _vp, _vp], _int),
    "vtm_match_filtered_plan": ([_vp, _i64, _vp, _i64, _int, _i64, _i64, _vp, _i64, _vp, _i64, _int, _vp, ctypes.c_size_t,
                                 _vp, _vp, _i64, _i64, _vp, _vp, _int, _vp], _int),
    "vtm_match_filtered_ordered": ([_vp, _i64, _vp, _i64, _int, _i64, _i64, _vp, _i64, _vp, _i64, _int, _vp, ctypes.c_size_t,
                                    _vp, _vp, _i64, _i64, _vp, _vp, _int, _vp, _vp, _vp], _int),
    "vtm_position_order_counter_ints": ([_i64, _i64], ctypes.c_size_t),
    "vtm_position_order_ws_bytes": ([_i64, _i64, _i64, _i64], ctypes.c_size_t),
    "vtm_position_order": ([_vp, _i64, _vp, _i64, _i64, _i64, _i64, _vp, _i64, _i64, _vp, _vp, ctypes.c_size_t, _vp, _vp, _vp,
                            _vp, _vp, _int, _vp], _int),
    "vtm_match_filtered_seeded": ([_vp, _i64, _vp, _i64, _int, _i64, _i64, _vp, _i64, _vp, _i64, _int, _vp, ctypes.c_size_t,
                                   _vp, _vp, _i64, _i64, _vp, _vp, _vp], _int),
    "vtm_anchor_pos": ([_vp, _i64, _i64, _i64, _i64, _vp, _i64, _vp, _vp], _int),
    "vtm_decode_best": ([_vp, _i64, _vp, _vp, _vp], _int),
    "vtm_sort_ws_bytes": ([_i64, _i64], ctypes.c_size_t),
    "vtm_sort_desc": ([_vp, _i64, _i64, _vp, _vp, ctypes.c_size_t, _vp], _int),
    "vtm_partition_counts": ([_i64, _i64, _i64, _i64, _i64, ctypes.POINTER(_i64), ctypes.POINTER(_i64)], _int),
    "vtm_partition_local": ([_vp, _i64, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _i64, _i64, _vp], _int),
    "vtm_partition_global": ([_vp, _i64, _i64, _i64, _i64, _int, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp], _int),
    "vtm_plan_apply": ([_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _int, _vp, _vp, _vp, _vp,
                        _vp, _vp], _int),
    "vtm_compose": ([_vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp], _int),
    "vtm_gather_rows": ([_vp, _i64, _vp, _i64, _int, _i64, _i64, _vp, _i64, _vp, _i64, _vp], _int),
    "vtm_unmerge_add": ([_vp, _i64, _vp, _vp, _int, _i64, _i64, _i64, _vp, _vp], _int),
    "vtm_merge_reduce": ([_vp, _int, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _i64, _i64, _int, _vp, _i64, _i64, _vp], _int),
    "vtm_attention": ([_vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _int, _i64, _i64, _i64, _i64, _i64, _f32,
                       _int, _vp, ctypes.c_size_t, _vp], _int),
    "vtm_attention_ws_bytes": ([_i64, _i64, _i64, _i64, _i64], ctypes.c_size_t),
    "vtm_attention_kv": ([_vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _int, _i64, _i64, _i64, _i64, _i64, _i64, _i64,
                          _f32, _int, _vp, ctypes.c_size_t, _vp], _int),
    "vtm_attention_kv_bounded": ([_vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _int, _i64, _i64, _i64, _i64, _i64, _i64, _i64,
                                  _f32, _vp, _vp, ctypes.c_size_t, _vp], _int),
    "vtm_attention_kv_shared_bounded": ([_vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _int, _i64, _i64, _i64, _i64, _i64, _i64, _i64,
                                         _f32, _int, _vp, _vp, ctypes.c_size_t, _vp], _int),
    "vtm_attention_kv_bounded_ws_bytes": ([_i64, _i64, _i64, _i64, _i64], ctypes.c_size_t),
    "vtm_anchor_maps": ([_vp, _i64, _i64, _vp, _i64, _i64, _i64, _i64, _i64, _vp, _i64, _vp, _vp, _vp, _vp], _int),
    "vtm_transpose_cols": ([_vp, _i64, _int, _i64, _i64, _i64, _vp, _i64, _vp], _int),
    "vtm_fold_keys_ws_bytes": ([_i64, _i64, _i64], ctypes.c_size_t),
    "vtm_fold_keys": ([_vp, _i64, _i64, _i64, _vp, _i64, _i64, _int, _vp, ctypes.c_size_t, _vp, _vp, _i64, _vp, _vp], _int),
    "vtm_attention_kv_folded": ([_vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _int, _i64, _i64, _i64, _i64, _i64, _i64, _i64,
                                 _f32, _vp, _vp, _vp, _i64, _vp, ctypes.c_size_t, _vp], _int),
    "vtm_compact_queries_ws_bytes": ([_i64, _i64], ctypes.c_size_t),
    "vtm_compact_queries": ([_vp, _i64, _i64, _i64, _i64, _vp, ctypes.c_size_t, _vp, _vp, _vp, _vp], _int),
    "vtm_panel_rows": ([_i64], _i64),
    "vtm_to_panels": ([_vp, _int, _i64, _i64, _vp, _vp, _i64, _vp], _int),
    "vtm_layernorm_panels": ([_vp, _vp, _vp, _int, _i64, _i64, _f32, _vp, _i64, _vp], _int),
    "vtm_gather_panels": ([_vp, _i64, _vp, _i64, _int, _i64, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp], _int),
    "vtm_ff_geglu": ([_vp, _i64, _i64, _vp, _i64, _i64, _i64, _vp, _int, _vp, _vp], _int),
    "vtm_linear_panels": ([_vp, _i64, _i64, _vp, _i64, _i64, _i64, _vp, _vp, _int, _vp, _i64, _vp], _int),
    "vtm_cfg_ddim": ([_vp, _vp, _vp, _int, _i64, _f32, _f32, _f32, _f32, _f32, _vp, _vp, _vp], _int),
    "vtm_layernorm": ([_vp, _vp, _vp, _int, _i64, _i64, _f32, _vp, _vp], _int),
    "vtm_geglu": ([_vp, _int, _i64, _i64, _vp, _vp], _int),
    "vtm_linear_rows": ([_vp, _i64, _vp, _i64, _int, _i64, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _i64, _vp, _i64, _i64,
                         _int, _vp], _int),
}


def exported_symbols():
    """Names include/vidtome_hip.h declares (used by the CPU test that checks the .so exports them)."""
    return sorted(_SIGNATURES)


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -m vidtome_amd.build` (hipcc, gfx950). "
                "vidtome_amd has no CPU / eager fallback.")
        L = ctypes.CDLL(LIB_PATH)
        for name, (argtypes, restype) in _SIGNATURES.items():
            fn = getattr(L, name)   # AttributeError here = the .so is stale w.r.t. the header
            fn.argtypes = argtypes
            fn.restype = restype
        if L.vtm_version() != ABI_VERSION:
            raise RuntimeError("libvidtome_hip.so ABI version mismatch")
        if L.vtm_build_ablations() != 0 and os.environ.get("VIDTOME_ALLOW_ABLATED") != "1":
            # an experiment build (csrc/ablate.h) computes wrong results by construction: only tools/ may load one, knowingly
            raise RuntimeError(f"{LIB_PATH} was built with ablation switches (mask {L.vtm_build_ablations():#x}); "
                               "set VIDTOME_ALLOW_ABLATED=1 to load it for a timing experiment")
        _lib = L
    return _lib


def _check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().vtm_last_error()
        raise RuntimeError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")


def _stream() -> int:
    """Raw HIP stream the launch goes to: torch's current stream of the CURRENT device -- every wrapper below runs
    under `_on_device`, which makes the operands' device the current one first."""
    return torch.cuda.current_stream().cuda_stream


def _on_device(fn):
    """Run the wrapper with its first tensor argument's device as the current HIP device (a model moved to cuda:1
    in a process whose current device is cuda:0 must launch on cuda:1's stream, not enqueue foreign pointers on
    cuda:0).  The common case -- already current -- costs one integer comparison."""
    @functools.wraps(fn)
    def guarded(*args, **kwargs):
        dev = next((a.device if isinstance(a, torch.Tensor) else a for a in args
                    if isinstance(a, (torch.Tensor, torch.device))), None)
        if dev is None or dev.type != "cuda" or dev.index is None or dev.index == torch.cuda.current_device():
            return fn(*args, **kwargs)
        with torch.cuda.device(dev.index):
            return fn(*args, **kwargs)
    return guarded


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


_WS: dict = {}
_WS_LOCK = threading.Lock()


def _workspace(tag: str, nbytes: int, device: torch.device) -> torch.Tensor:
    """Scratch memory of a call, cached per (device, stream, purpose) and grown on demand: calls on one stream are
    ordered, so the next user of the buffer starts after the previous one finished -- no allocator round trip per
    call (a block makes ~40 of them).  The library itself stays stateless: it is handed the pointer and the size.
    The host thread is part of the key: two threads launching on the same stream never share a buffer (their launches
    are not ordered against each other by anything the caller can rely on).  `remove_patch` drops the cache."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream, tag, threading.get_ident())
    buf = _WS.get(key)                     # (a thread only ever reads / replaces its OWN keys outside the lock)
    if buf is None or buf.numel() < nbytes:
        new = torch.empty((max(int(nbytes), 16),), dtype=torch.uint8, device=device)
        with _WS_LOCK:                     # insertion + clean-up mutate the dict other threads iterate
            if buf is None:
                # a new (thread, stream, purpose): the moment to let go of the buffers of threads that no longer exist
                alive = {t.ident for t in threading.enumerate()}
                for k in [k for k in list(_WS) if k[3] not in alive]:
                    _WS.pop(k, None)
            _WS[key] = new
        buf = new
    return buf


def release_workspaces() -> None:
    """Drop the cached scratch buffers (they are re-created on demand)."""
    with _WS_LOCK:
        _WS.clear()
        _ZEROED.clear()


def _req(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError(f"{name} must live on the GPU (vidtome_amd has no CPU path)")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")
    return t


def dtype_code(t: torch.Tensor) -> int:
    try:
        return _DT[t.dtype]
    except KeyError:
        raise RuntimeError(f"unsupported dtype {t.dtype}") from None


def pad_rows(n: int) -> int:
    return (n + ROW_PAD - 1) // ROW_PAD * ROW_PAD


def pad_k(c: int) -> int:
    return (c + K_PAD - 1) // K_PAD * K_PAD


# --------------------------------------------------------------------------------------------------
@_on_device
def normalize_gather(x0: torch.Tensor, x1: Optional[torch.Tensor], rows: torch.Tensor
                     ) -> Tuple[torch.Tensor, torch.Tensor]:
    """rows (B, n) int32 pool ids -> (operand (B, C_pad/8, 2, n_pad, 4) fp32 k-panels, norms (B, n))."""
    _req(x0, "x0"), _req(rows, "rows")
    B, P0, C = x0.shape
    P1 = 0 if x1 is None else _req(x1, "x1").shape[1]
    n = rows.shape[1]
    n_pad, C_pad = pad_rows(n), pad_k(C)
    out = torch.empty((B, C_pad // 8, 2, n_pad, 4), dtype=torch.float32, device=x0.device)
    norms = torch.empty((B, max(n, 1)), dtype=torch.float32, device=x0.device)
    _check(lib().vtm_normalize_gather(_ptr(x0), P0, _ptr(x1), P1, dtype_code(x0), B, C, _ptr(rows), n,
                                      _ptr(norms), _ptr(out), n_pad, C_pad, _stream()), "vtm_normalize_gather")
    return out, norms


@_on_device
def match(a: torch.Tensor, b: torch.Tensor, Ns: int, Nd: int, align: bool) -> torch.Tensor:
    """Packed (orderable(max) << 32 | ~argmax) per src row: (B, Ns) or (1, Ns) when aligned (int64 bits)."""
    B, G, _, Ns_pad, _ = a.shape
    C_pad, Nd_pad = G * 8, b.shape[3]
    best = torch.empty((1 if align else B, Ns), dtype=torch.int64, device=a.device)
    _check(lib().vtm_match(_ptr(a), _ptr(b), B, Ns, Nd, Ns_pad, Nd_pad, C_pad, int(align), _ptr(best), _stream()),
           "vtm_match")
    return best


MATCH_ONE_LAUNCH, MATCH_SCOUT_RANGE = 0, 1         # include/vidtome_hip.h: VTM_MATCH_*


@_on_device
def match_filtered(x0: torch.Tensor, x1: Optional[torch.Tensor], a_rows: torch.Tensor, b_rows: torch.Tensor,
                   align: bool, want_flag: bool = False, seed=None, mode: int = MATCH_ONE_LAUNCH,
                   stats_host: Optional[torch.Tensor] = None, order=None, scout_steps: int = 0):
    """Same packed result as normalize_gather x2 + match, through the fp16-filter / fp32-refine path.  ``seed`` (optional,
    never changes the result): (tokens per frame N, L = pool rows that are chunk tokens, pos1 (B, P1) int32 positions of the
    x1 rows or None, table (B, N) int32 position -> dst index or None for identity) -- every src row then starts from the
    score of the dst row at its own token position.  ``mode`` = the launch plan (MATCH_ONE_LAUNCH / MATCH_SCOUT_RANGE: a scout
    launch + the filter over the spans of live dst tiles, for levels with position-ordered rows; same bits;
    include/vidtome_hip.h, vtm_match_filtered_plan).  ``stats_host``: a PINNED host tensor of 8 int32 that receives the
    call's counters asynchronously (flags_out of the C ABI) -- what merge.MatchPlanner steers by; ``want_flag`` returns them
    as a device tensor instead.  ``order`` = (a_order, b_order) from `position_order`: a_rows / b_rows are then its SORTED lists
    and the result is reported (and ties are broken) in the original indexing -- the same bits as the call on the unsorted
    lists (vtm_match_filtered_ordered; aligned calls: lists from a ``shared`` position_order).  ``scout_steps`` (scout + range plan): the scout tests after that many
    64-channel steps instead of the filter's own test depth (VTM_MATCH_SCOUT_STEPS; 0 = the filter's depth)."""
    _req(x0, "x0"), _req(a_rows, "a_rows"), _req(b_rows, "b_rows")
    B, P0, C = x0.shape
    P1 = 0 if x1 is None else _req(x1, "x1").shape[1]
    Ns, Nd = a_rows.shape[1], b_rows.shape[1]
    nbytes = lib().vtm_match_filtered_ws_bytes(B, C, Ns, Nd, int(align))
    ws = _workspace("match", nbytes, x0.device)
    best = torch.empty((1 if align else B, Ns), dtype=torch.int64, device=x0.device)
    # whole-call escape, unusable norm seen, escaped rows, refined pairs, blocks tested, blocks alive, (internal), scout's
    # live wave tiles
    flag = torch.zeros((8,), dtype=torch.int32, device=x0.device) if want_flag else None
    flags_out = _ptr(flag)
    if flag is None and stats_host is not None:
        if not (stats_host.is_pinned() and stats_host.dtype == torch.int32 and stats_host.numel() >= 8):
            raise RuntimeError("match_filtered: stats_host must be a pinned int32 tensor of 8 elements")
        flags_out = stats_host.data_ptr()
    mode = int(mode) | ((int(scout_steps) & 0xff) << 8)
    sN = sL = 0
    pos1 = table = None
    if seed is not None and SEED_MATCHER:
        sN, sL, pos1, table = seed
        if pos1 is not None and (pos1.dtype != torch.int32 or tuple(pos1.shape) != (B, P1) or not pos1.is_contiguous()):
            raise RuntimeError("match_filtered: seed positions must be a contiguous (B, P1) int32 tensor")
        if table is not None and (table.dtype != torch.int32 or tuple(table.shape) != (B, sN) or not table.is_contiguous()):
            raise RuntimeError("match_filtered: the seed table must be a contiguous (B, N) int32 tensor")
    if order is not None:
        a_order, b_order = order
        for o, n, name in ((a_order, Ns, "a_order"), (b_order, Nd, "b_order")):
            if o.dtype != torch.int32 or tuple(o.shape) != (B, n) or not o.is_contiguous() or not o.is_cuda:
                raise RuntimeError(f"match_filtered: {name} must be a contiguous (B, {n}) int32 device tensor")
        _check(lib().vtm_match_filtered_ordered(_ptr(x0), P0, _ptr(x1), P1, dtype_code(x0), B, C, _ptr(a_rows), Ns,
                                                _ptr(b_rows), Nd, int(align), _ptr(ws), nbytes, _ptr(best), flags_out, int(sL), int(sN),
                                                _ptr(pos1), _ptr(table), int(mode), _ptr(a_order), _ptr(b_order), _stream()),
               "vtm_match_filtered_ordered")
        return (best, flag) if want_flag else best
    _check(lib().vtm_match_filtered_plan(_ptr(x0), P0, _ptr(x1), P1, dtype_code(x0), B, C, _ptr(a_rows), Ns,
                                         _ptr(b_rows), Nd, int(align), _ptr(ws), nbytes, _ptr(best), flags_out,
                                         int(sL), int(sN), _ptr(pos1), _ptr(table), int(mode), _stream()),
           "vtm_match_filtered_plan")
    return (best, flag) if want_flag else best


POSITION_ORDER_MAX_N = 16360        # include/vidtome_hip.h: VTM_POSITION_ORDER_MAX_N
_ZEROED: dict = {}


def _zeroed_counters(n_ints: int, device: torch.device) -> torch.Tensor:
    """The counter block of vtm_position_order: zero when a call starts, zero again when it has run -- allocated (zeroed)
    once per (device, stream, thread) like the scratch buffers, re-allocated when a larger one is needed."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream, threading.get_ident())
    buf = _ZEROED.get(key)
    if buf is None or buf.numel() < n_ints:
        buf = torch.zeros((max(int(n_ints), 16),), dtype=torch.int32, device=device)
        with _WS_LOCK:
            _ZEROED[key] = buf
    return buf


@_on_device
def position_order(a_rows: torch.Tensor, b_rows: torch.Tensor, L: int, N: int, pos1: Optional[torch.Tensor], P0: int,
                   want_table: bool = True, shared: bool = False):
    """Both row lists of a matcher call sorted by token position (vtm_position_order, include/vidtome_hip.h):
    -> (a_sorted, a_order, b_sorted, b_order, table).  Position of pool row r: r % N below L, pos1[b, r - P0] for the rows of
    x1 (``pos1`` (B, P1) int32 or None).  ``shared`` (aligned matching): ONE order, sample 0's, for every sample's lists."""
    _req(a_rows, "a_rows"), _req(b_rows, "b_rows")
    B, Ns = a_rows.shape
    Nd = b_rows.shape[1]
    P1 = 0
    if pos1 is not None:
        if pos1.dtype != torch.int32 or pos1.dim() != 2 or pos1.shape[0] != B or not pos1.is_contiguous():
            raise RuntimeError("position_order: positions must be a contiguous (B, P1) int32 tensor")
        P1 = pos1.shape[1]
    dev = a_rows.device
    counters = _zeroed_counters(lib().vtm_position_order_counter_ints(B, N), dev)
    nbytes = lib().vtm_position_order_ws_bytes(B, Ns, Nd, N)
    ws = _workspace("order", nbytes, dev)
    i32 = dict(dtype=torch.int32, device=dev)
    a_sorted, a_order = torch.empty((B, Ns), **i32), torch.empty((B, Ns), **i32)
    b_sorted, b_order = torch.empty((B, Nd), **i32), torch.empty((B, Nd), **i32)
    table = torch.empty((B, N), **i32) if want_table else None
    _check(lib().vtm_position_order(_ptr(a_rows), Ns, _ptr(b_rows), Nd, B, int(L), int(N), _ptr(pos1), int(P0), P1,
                                    _ptr(counters), _ptr(ws), nbytes, _ptr(a_sorted), _ptr(a_order), _ptr(b_sorted),
                                    _ptr(b_order), _ptr(table), int(shared), _stream()), "vtm_position_order")
    return a_sorted, a_order, b_sorted, b_order, table


@_on_device
def decode_best(best: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    nm = torch.empty(best.shape, dtype=torch.float32, device=best.device)
    ni = torch.empty(best.shape, dtype=torch.int32, device=best.device)
    _check(lib().vtm_decode_best(_ptr(best), best.numel(), _ptr(nm), _ptr(ni), _stream()), "vtm_decode_best")
    return nm, ni


@_on_device
def sort_desc(best: torch.Tensor) -> torch.Tensor:
    rows, n = best.shape
    perm = torch.empty((rows, n), dtype=torch.int32, device=best.device)
    nbytes = lib().vtm_sort_ws_bytes(rows, n)
    ws = _workspace("sort", nbytes, best.device)
    _check(lib().vtm_sort_desc(_ptr(best), rows, n, _ptr(perm), _ptr(ws), nbytes, _stream()), "vtm_sort_desc")
    return perm


def partition_counts(N_in: int, unm_pre: int, tnum: int, ts: int, randf: int) -> Tuple[int, int]:
    ns, nd = _i64(0), _i64(0)
    _check(lib().vtm_partition_counts(N_in, unm_pre, tnum, ts, randf, ctypes.byref(ns), ctypes.byref(nd)),
           "vtm_partition_counts")
    return int(ns.value), int(nd.value)


@_on_device
def partition_local(cur: Optional[torch.Tensor], B: int, N_in: int, unm_pre: int, tnum: int, ts: int,
                    randf: int, device) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    Ns, Nd = partition_counts(N_in, unm_pre, tnum, ts, randf)
    i32 = dict(dtype=torch.int32, device=device)
    a_pos, b_pos = torch.empty((Ns,), **i32), torch.empty((Nd,), **i32)
    a_rows, b_rows = torch.empty((B, Ns), **i32), torch.empty((B, Nd), **i32)
    _check(lib().vtm_partition_local(_ptr(cur), B, N_in, unm_pre, tnum, ts, randf, _ptr(a_pos), _ptr(b_pos),
                                     _ptr(a_rows), _ptr(b_rows), Ns, Nd, _stream()), "vtm_partition_local")
    return a_pos, b_pos, a_rows, b_rows


@_on_device
def anchor_pos(amap: Optional[torch.Tensor], B: int, M: int, L: int, tokens: int, old_pos: Optional[torch.Tensor],
               device) -> torch.Tensor:
    """Token positions (B, M) int32 of a new anchor set = pool[amap] (amap None: the first M pool rows), pool = [chunk of L
    rows | old anchors with positions old_pos]; -1 = unknown."""
    out = torch.empty((B, M), dtype=torch.int32, device=device)
    Mg = 0 if old_pos is None else old_pos.shape[1]
    _check(lib().vtm_anchor_pos(_ptr(amap), B, M, L, tokens, _ptr(old_pos), Mg, _ptr(out), _stream()), "vtm_anchor_pos")
    return out


@_on_device
def anchor_maps(inv_g: torch.Tensor, off: int, new_cur: torch.Tensor, Ml: int, L: int, tokens: int,
                old_pos: Optional[torch.Tensor], want_pos: bool) -> Tuple[torch.Tensor, torch.Tensor, Optional[torch.Tensor]]:
    """(loc, amap, pos) behind a global level: merged position of every local token, pool row of every new anchor, and
    (``want_pos``) its token position; see include/vidtome_hip.h."""
    _req(inv_g, "inv_g"), _req(new_cur, "new_cur")
    B, N_in = inv_g.shape
    i32 = dict(dtype=torch.int32, device=inv_g.device)
    loc, amap = torch.empty((B, Ml), **i32), torch.empty((B, Ml), **i32)
    pos = torch.empty((B, Ml), **i32) if want_pos else None
    Mg = 0 if old_pos is None else old_pos.shape[1]
    _check(lib().vtm_anchor_maps(_ptr(inv_g), N_in, off, _ptr(new_cur), new_cur.shape[1], B, Ml, L, tokens if want_pos else 0,
                                 _ptr(old_pos), Mg, _ptr(loc), _ptr(amap), _ptr(pos), _stream()), "vtm_anchor_maps")
    return loc, amap, pos


@_on_device
def partition_global(cur_local: torch.Tensor, anchor_base: int, Mg: int, local_is_src: bool, seed_table: Optional[torch.Tensor] = None,
                     tokens: int = 0, anchor_positions: Optional[torch.Tensor] = None
                     ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """``seed_table`` (B, tokens) int32 (optional) is filled with position -> dst index for the matcher's seeds."""
    B, Ml = cur_local.shape
    src_len = Ml if local_is_src else Mg
    Nd = Ml + Mg - src_len
    i32 = dict(dtype=torch.int32, device=cur_local.device)
    a_pos, b_pos = torch.empty((src_len,), **i32), torch.empty((Nd,), **i32)
    a_rows, b_rows = torch.empty((B, src_len), **i32), torch.empty((B, Nd), **i32)
    _check(lib().vtm_partition_global(_ptr(cur_local), B, Ml, anchor_base, Mg, int(local_is_src), _ptr(a_pos),
                                      _ptr(b_pos), _ptr(a_rows), _ptr(b_rows), _ptr(seed_table), int(tokens),
                                      _ptr(anchor_positions), _stream()), "vtm_partition_global")
    return a_pos, b_pos, a_rows, b_rows


@_on_device
def plan_apply(best, perm, a_pos, b_pos, a_rows, b_rows, r: int, align: bool, want_indices: bool):
    B, Ns = a_rows.shape
    Nd = b_rows.shape[1]
    N_in, U = Ns + Nd, Ns - r
    i32 = dict(dtype=torch.int32, device=a_rows.device)
    new_cur = torch.empty((B, U + Nd), **i32)
    inv = torch.empty((B, N_in), **i32)
    unm_idx = torch.empty((B, U), **i32) if want_indices else None
    src_idx = torch.empty((B, r), **i32) if want_indices else None
    dst_idx = torch.empty((B, r), **i32) if want_indices else None
    _check(lib().vtm_plan_apply(_ptr(best), _ptr(perm), _ptr(a_pos), _ptr(b_pos), _ptr(a_rows), _ptr(b_rows), B,
                                N_in, Ns, Nd, r, int(align), _ptr(new_cur), _ptr(inv), _ptr(unm_idx),
                                _ptr(src_idx), _ptr(dst_idx), _stream()), "vtm_plan_apply")
    return new_cur, inv, unm_idx, src_idx, dst_idx


@_on_device
def compose(inv_acc: Optional[torch.Tensor], inv_level: torch.Tensor, n: int, offset: int = 0) -> torch.Tensor:
    B, level_len = inv_level.shape
    out = torch.empty((B, n), dtype=torch.int32, device=inv_level.device)
    _check(lib().vtm_compose(_ptr(inv_acc), _ptr(inv_level), B, n, level_len, offset, _ptr(out), _stream()),
           "vtm_compose")
    return out


@_on_device
def gather_rows(x0: torch.Tensor, x1: Optional[torch.Tensor], idx: torch.Tensor, pad_to: int = 1) -> torch.Tensor:
    """out (B, M_pad, C): rows [0, M) = pool[idx]; rows >= M (padding up to a multiple of pad_to) are zero."""
    _req(x0, "x0"), _req(idx, "map")
    B, P0, C = x0.shape
    P1 = 0 if x1 is None else _req(x1, "x1").shape[1]
    M = idx.shape[1]
    Mp = (M + pad_to - 1) // pad_to * pad_to
    out = (torch.zeros if Mp != M else torch.empty)((B, Mp, C), dtype=x0.dtype, device=x0.device)
    _check(lib().vtm_gather_rows(_ptr(x0), P0, _ptr(x1), P1, dtype_code(x0), B, C, _ptr(idx), M, _ptr(out), Mp,
                                 _stream()), "vtm_gather_rows")
    return out


REDUCE_MODES = {"sum": 0, "prod": 1, "mean": 2, "amax": 3, "amin": 4}      # torch.scatter_reduce's (merge.py:127-131)


@_on_device
def merge_reduce(x: torch.Tensor, src_rows: torch.Tensor, dst_rows: torch.Tensor, dst_idx: torch.Tensor, mode: str,
                 out: torch.Tensor, out_row0: int) -> torch.Tensor:
    """The reference's non-"replace" merge modes: fold row src_rows[b, i] of x into dst row dst_idx[b, i] (whose own row
    of x is dst_rows[b, j]) like ``scatter_reduce(..., reduce=mode, include_self=True)`` on the CPU, writing the Nd
    reduced rows to out[:, out_row0:out_row0 + Nd].  The pairs are sorted by destination with the library's own stable
    radix sort (index order survives inside a destination's segment -- that order is the summation order)."""
    if mode not in REDUCE_MODES:
        raise ValueError(f"merge mode {mode!r}: expected 'replace' or one of {sorted(REDUCE_MODES)}")
    _req(x, "x"), _req(src_rows, "src_rows"), _req(dst_rows, "dst_rows"), _req(dst_idx, "dst_idx"), _req(out, "out")
    B, N, C = x.shape
    r, Nd = src_rows.shape[1], dst_rows.shape[1]
    if r > 0:
        keys = ((-1 - dst_idx.long()) << 32).contiguous()       # high word = 0xffffffff - dst: "descending" = ascending dst
        order = sort_desc(keys)
        seg_dst = torch.gather(dst_idx, 1, order.long()).contiguous()
    else:
        order = seg_dst = dst_idx
    _check(lib().vtm_merge_reduce(_ptr(x), dtype_code(x), B, N, C, _ptr(src_rows), _ptr(dst_rows), _ptr(seg_dst), _ptr(order),
                                  r, Nd, REDUCE_MODES[mode], _ptr(out), out.shape[1], out_row0, _stream()), "vtm_merge_reduce")
    return out


@_on_device
def unmerge_add(y: torch.Tensor, inv: torch.Tensor, resid: Optional[torch.Tensor]) -> torch.Tensor:
    """out[b, i] = y[b, inv[b, i]] (+ resid[b, i]);  y is (B, Mp, C) (only rows < M are referenced)."""
    _req(y, "y"), _req(inv, "inv")
    B, Mp, C = y.shape
    L = inv.shape[1]
    out = torch.empty((B, L, C), dtype=y.dtype, device=y.device)
    if resid is not None:
        _req(resid, "resid")
        if resid.numel() != out.numel() or resid.dtype != y.dtype:
            raise RuntimeError("residual shape/dtype mismatch")
    _check(lib().vtm_unmerge_add(_ptr(y), Mp, _ptr(inv), _ptr(resid), dtype_code(y), B, L, C, _ptr(out), _stream()),
           "vtm_unmerge_add")
    return out


@_on_device
def transpose_cols(x: torch.Tensor, c0: int, C: int) -> torch.Tensor:
    """x (BF, N, ld) 16-bit, contiguous along the last axis -> (BF, C, Np) = columns [c0, c0 + C) channel-major, Np = N
    rounded up to 8 (zero-filled): the V^T operand of the attention core from a fused q | k | v projection."""
    _req(x, "x")
    BF, N, ld = x.shape
    if x.stride(2) != 1 or x.stride(1) != ld or x.stride(0) != N * ld or c0 % 8 or C % 8:
        raise RuntimeError("transpose_cols: dense (BF, N, ld) input, column window aligned to 8")
    Np = (N + 7) // 8 * 8
    out = torch.empty((BF, C, Np), dtype=x.dtype, device=x.device)
    _check(lib().vtm_transpose_cols(x.data_ptr() + c0 * x.element_size(), ld, dtype_code(x), BF, N, C, _ptr(out), Np, _stream()),
           "vtm_transpose_cols")
    return out


# A/B switch: seed the filtered matcher's running maxima from same-position guesses (profiles/r04_seeds.txt); results are
# identical either way
SEED_MATCHER = os.environ.get("VIDTOME_SEED", "1") != "0"

# A/B switch (profiles/r04_attention_split_all.txt): split every work item of a query-bounded attention launch in two
SPLIT_ALL_BOUNDED = os.environ.get("VIDTOME_ATT_SPLIT_ALL", "1") != "0"

# A/B switch: the anchors' exact duplicates enter attn1 as one key each (fold_keys / vtm_attention_kv_folded; d = 40 heads)
FOLD_KEYS = os.environ.get("VIDTOME_FOLD_KEYS", "1") != "0"


def _attention_ws(B: int, heads: int, Mq: int, Mk: int, d: int, device):
    """Workspace for the split last round of an attention launch (None when the shape needs none)."""
    nb = int(lib().vtm_attention_ws_bytes(B, heads, Mq, Mk, d))
    if nb == 0:
        return None, 0
    return _workspace("attention", nb, device), nb


@_on_device
def attention(q: torch.Tensor, k: torch.Tensor, vt: torch.Tensor, heads: int, M: int, scale: float,
              share_groups: int = 1) -> torch.Tensor:
    """q, k: (B, Mp, C) views with arbitrary last-dim-contiguous row stride; vt: (B, C, ldvt) = v transposed.
    Returns out (B, Mp, C) (rows >= M untouched/zero)."""
    B, Mp, C = q.shape
    d = C // heads
    if q.stride(2) != 1 or k.stride(2) != 1 or vt.stride(2) != 1:
        raise RuntimeError("attention operands must be contiguous along their last axis")
    if q.stride(0) != Mp * q.stride(1) or k.stride(0) != Mp * k.stride(1) or vt.stride(0) != C * vt.stride(1):
        raise RuntimeError("attention operands must have dense batch strides")
    out = torch.zeros((B, Mp, C), dtype=q.dtype, device=q.device) if Mp != M else \
        torch.empty((B, Mp, C), dtype=q.dtype, device=q.device)
    ws, nb = _attention_ws(B, heads, M, M, d, q.device)
    _check(lib().vtm_attention(q.data_ptr(), q.stride(1), k.data_ptr(), k.stride(1), vt.data_ptr(), vt.stride(1),
                               out.data_ptr(), C, dtype_code(q), B, heads, M, Mp, d, float(scale),
                               int(share_groups), _ptr(ws), nb, _stream()), "vtm_attention")
    return out


@_on_device
def cfg_ddim(x: Optional[torch.Tensor], eps_uncond: torch.Tensor, eps_cond: Optional[torch.Tensor], guidance: float,
             a: float, b: float, c: float, d: float, want_eps: bool = False):
    """generate.py:276-278 + 281-311 fused: returns x_next (and the guided eps when want_eps)."""
    _req(eps_uncond, "eps_uncond")
    n = eps_uncond.numel()
    # the reference's coefficients are 0-dim fp32 tensors.  torch's CPU kernels (the parity oracle) cast the
    # multipliers b, c, d to the tensor dtype before the op but divide by the ORIGINAL fp32 value of a
    # (tests/golden/ddim.npz pins this for fp16); for fp32 tensors all four are used as they are.
    b, c, d = (float(torch.tensor(v, dtype=torch.float32).to(eps_uncond.dtype)) for v in (b, c, d))
    a = float(torch.tensor(a, dtype=torch.float32))
    x_out = torch.empty_like(eps_uncond) if x is not None else None
    eps_out = torch.empty_like(eps_uncond) if (want_eps or x is None) else None
    _check(lib().vtm_cfg_ddim(_ptr(x), _ptr(eps_uncond), _ptr(eps_cond), dtype_code(eps_uncond), n, float(guidance),
                              float(a), float(b), float(c), float(d), _ptr(eps_out), _ptr(x_out), _stream()),
           "vtm_cfg_ddim")
    return (x_out, eps_out) if want_eps else (x_out if x is not None else eps_out)


@_on_device
def layernorm(x: torch.Tensor, weight: Optional[torch.Tensor], bias: Optional[torch.Tensor], eps: float) -> torch.Tensor:
    """torch.nn.LayerNorm over the last axis (patch.py:139-146 `self.norm1(hidden_states)`)."""
    _req(x, "x")
    C = x.shape[-1]
    xc = x.contiguous()
    for name, p in (("weight", weight), ("bias", bias)):
        if p is not None and (p.dtype != x.dtype or p.numel() != C or p.device != x.device):
            raise RuntimeError(f"vidtome_amd: layernorm {name} must be a ({C},) {x.dtype} tensor on {x.device}")
    out = torch.empty_like(xc)
    _check(lib().vtm_layernorm(_ptr(xc), _ptr(weight.contiguous() if weight is not None else None),
                               _ptr(bias.contiguous() if bias is not None else None), dtype_code(xc),
                               xc.numel() // C, C, float(eps), _ptr(out), _stream()), "vtm_layernorm")
    return out


@_on_device
def attention_kv(q: torch.Tensor, k: torch.Tensor, vt: torch.Tensor, heads: int, Mq: int, Mk: int, scale: float,
                 use_workspace: bool = True, q_count: Optional[torch.Tensor] = None,
                 k_fold: Optional[Tuple[torch.Tensor, torch.Tensor]] = None, share_groups: int = 1) -> torch.Tensor:
    """Cross-attention core (patch.py:178-183): q (B, Mqp, C), k (B, Mkp, C) views contiguous along the last axis,
    vt (B, C, ldvt >= Mk) = v transposed.  Returns (B, Mqp, C).  ``q_count`` (B,) int32 on the device: only the first
    q_count[b] query rows of sample b are meaningful (compact_queries); the other rows of the result are undefined.
    ``k_fold`` = (k_count (B,) int32, k_bias (B, >= Mk) uint32 pairs) from fold_keys: k / vt hold a duplicate-free key
    list, only the first k_count[b] entries are keys, each standing for 2^bias identical ones (head dims 8 and 40).
    ``share_groups`` > 1: the probabilities of the first B / share_groups samples serve every group (pnp_utils.py:57-67);
    with ``q_count`` the caller vouches that every sample of a group has the same live rows (align_batch) -- no key folding."""
    if share_groups != 1 and k_fold is not None:
        raise RuntimeError("attention_kv: shared probabilities do not go with folded keys")
    B, Mqp, C = q.shape
    Mkp = k.shape[1]
    d = C // heads
    if q.stride(2) != 1 or k.stride(2) != 1 or vt.stride(2) != 1:
        raise RuntimeError("attention operands must be contiguous along their last axis")
    if q.stride(0) != Mqp * q.stride(1) or k.stride(0) != Mkp * k.stride(1) or vt.stride(0) != C * vt.stride(1):
        raise RuntimeError("attention operands must have dense batch strides")
    out = torch.zeros((B, Mqp, C), dtype=q.dtype, device=q.device) if Mqp != Mq else \
        torch.empty((B, Mqp, C), dtype=q.dtype, device=q.device)
    ws, nb = _attention_ws(B, heads, Mq, Mk, d, q.device) if use_workspace else (None, 0)
    if q_count is not None and (q_count.dtype != torch.int32 or q_count.numel() != B or not q_count.is_cuda):
        raise RuntimeError("attention_kv: q_count must be a (B,) int32 device tensor")
    if q_count is not None and use_workspace and SPLIT_ALL_BOUNDED:
        nb2 = int(lib().vtm_attention_kv_bounded_ws_bytes(B, heads, Mq, Mk, d))
        if nb2 > nb:
            ws, nb = _workspace("attention", nb2, q.device), nb2
    if share_groups != 1 and q_count is not None:
        _check(lib().vtm_attention_kv_shared_bounded(q.data_ptr(), q.stride(1), k.data_ptr(), k.stride(1), vt.data_ptr(),
                                                     vt.stride(1), out.data_ptr(), C, dtype_code(q), B, heads, Mq, Mqp, Mk, Mkp,
                                                     d, float(scale), int(share_groups), _ptr(q_count), _ptr(ws), nb, _stream()),
               "vtm_attention_kv_shared_bounded")
        return out
    if k_fold is not None:
        k_count, k_bias = k_fold
        if k_count.dtype != torch.int32 or k_count.numel() != B or k_bias.dtype != torch.int32 or k_bias.dim() != 2 \
                or k_bias.shape[0] != B or k_bias.shape[1] < Mk or not k_bias.is_contiguous():
            raise RuntimeError("attention_kv: k_fold must be ((B,) int32 counts, (B, >= Mk) int32 bias words)")
        _check(lib().vtm_attention_kv_folded(q.data_ptr(), q.stride(1), k.data_ptr(), k.stride(1), vt.data_ptr(), vt.stride(1),
                                             out.data_ptr(), C, dtype_code(q), B, heads, Mq, Mqp, Mk, Mkp, d, float(scale),
                                             _ptr(q_count), _ptr(k_count), _ptr(k_bias), k_bias.shape[1], _ptr(ws), nb,
                                             _stream()), "vtm_attention_kv_folded")
        return out
    if q_count is not None:
        _check(lib().vtm_attention_kv_bounded(q.data_ptr(), q.stride(1), k.data_ptr(), k.stride(1), vt.data_ptr(),
                                              vt.stride(1), out.data_ptr(), C, dtype_code(q), B, heads, Mq, Mqp, Mk, Mkp,
                                              d, float(scale), _ptr(q_count), _ptr(ws), nb, _stream()),
               "vtm_attention_kv_bounded")
        return out
    _check(lib().vtm_attention_kv(q.data_ptr(), q.stride(1), k.data_ptr(), k.stride(1), vt.data_ptr(), vt.stride(1),
                                  out.data_ptr(), C, dtype_code(q), B, heads, Mq, Mqp, Mk, Mkp, d, float(scale),
                                  int(share_groups), _ptr(ws), nb, _stream()), "vtm_attention_kv")
    return out


@_on_device
def compact_queries(loc: torch.Tensor, U: int, Nd: int) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """loc (B, Ml) merged position of every local token -> (qc (B, Ml) distinct positions, tmap (B, Ml) row of qc each
    local token reads, count (B,) number of distinct positions); see include/vidtome_hip.h."""
    _req(loc, "loc")
    B, Ml = loc.shape
    i32 = dict(dtype=torch.int32, device=loc.device)
    qc, tmap, count = torch.empty((B, Ml), **i32), torch.empty((B, Ml), **i32), torch.empty((B,), **i32)
    nb = int(lib().vtm_compact_queries_ws_bytes(B, Nd))
    ws = _workspace("compact", nb, loc.device)
    _check(lib().vtm_compact_queries(_ptr(loc), B, Ml, U, Nd, _ptr(ws), nb, _ptr(qc), _ptr(tmap), _ptr(count), _stream()),
           "vtm_compact_queries")
    return qc, tmap, count


@_on_device
def fold_keys(cur: torch.Tensor, L: int, cid: torch.Tensor, n_ids: int, dtype: torch.dtype
              ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """cur (B, M) pool row of every merged position (rows >= L: anchor row cur - L), cid (B, Ma) content id of every anchor
    row (< n_ids; equal ids = identical rows) -> (key_sel (B, M) surviving positions, k_bias (B, Mp) int32 words holding
    log2(copies) as a 16-bit (hi, lo) pair of ``dtype``, k_count (B,)); see include/vidtome_hip.h."""
    _req(cur, "cur")
    _req(cid, "cid")
    B, M = cur.shape
    Mp = (M + 7) // 8 * 8
    i32 = dict(dtype=torch.int32, device=cur.device)
    key_sel, k_bias, k_count = torch.empty((B, M), **i32), torch.empty((B, Mp), **i32), torch.empty((B,), **i32)
    code = {torch.float16: 1, torch.bfloat16: 2}[dtype]
    nb = int(lib().vtm_fold_keys_ws_bytes(B, M, n_ids))
    ws = _workspace("fold", nb, cur.device)
    _check(lib().vtm_fold_keys(_ptr(cur), B, M, L, _ptr(cid), cid.shape[1], n_ids, code, _ptr(ws), nb, _ptr(key_sel),
                               _ptr(k_bias), Mp, _ptr(k_count), _stream()), "vtm_fold_keys")
    return key_sel, k_bias, k_count


@_on_device
def geglu(x: torch.Tensor) -> torch.Tensor:
    """value * gelu(gate) over the two halves of the last axis (the GEGLU feed-forward, patch.py:187-199)."""
    _req(x, "x")
    D = x.shape[-1] // 2
    xc = x.contiguous()
    out = torch.empty(xc.shape[:-1] + (D,), dtype=x.dtype, device=x.device)
    _check(lib().vtm_geglu(_ptr(xc), dtype_code(xc), xc.numel() // (2 * D), D, _ptr(out), _stream()), "vtm_geglu")
    return out


@_on_device
def linear_rows(x0: torch.Tensor, x1: Optional[torch.Tensor], rows: Optional[torch.Tensor],
                rows2: Optional[torch.Tensor], n: int, weight: torch.Tensor, bias: Optional[torch.Tensor],
                transposed: bool = False, pad_to: int = 8, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[b, i] = pool[b, rows[b, rows2[b, i]]] @ weight^T (+ bias) for i < n (either map may be None = identity).
    Returns (B, n_pad, N) token-major, or (B, N, n_pad) channel-major when ``transposed`` (n_pad = n rounded up to
    ``pad_to``; the padding rows / columns are zero when this function allocates the result -- consumers such as a
    gate multiply or a later GEMM then see finite values -- and left alone in a preallocated ``out`` view, last axis
    contiguous)."""
    _req(x0, "x0"), _req(weight, "weight")
    B, P0, K = x0.shape
    P1 = 0 if x1 is None else _req(x1, "x1").shape[1]
    N = weight.shape[0]
    if weight.shape[1] != K or weight.dtype != x0.dtype:
        raise RuntimeError("linear_rows: weight must be (N, K) in the token dtype")
    if rows is not None:
        _req(rows, "rows")
    if rows2 is not None:
        _req(rows2, "rows2")
    n_pad = (n + pad_to - 1) // pad_to * pad_to
    if out is None:
        out = torch.empty((B, N, n_pad) if transposed else (B, n_pad, N), dtype=x0.dtype, device=x0.device)
        if n_pad != n:
            (out[:, :, n:] if transposed else out[:, n:]).zero_()
    if out.stride(2) != 1:
        raise RuntimeError("linear_rows: out must be contiguous along its last axis")
    _check(lib().vtm_linear_rows(_ptr(x0), P0, _ptr(x1), P1, dtype_code(x0), B, K, _ptr(rows),
                                 0 if rows is None else rows.shape[1], _ptr(rows2), n, _ptr(weight),
                                 _ptr(bias.contiguous() if bias is not None else None), N, out.data_ptr(), out.stride(1),
                                 out.stride(0), int(transposed), _stream()), "vtm_linear_rows")
    return out


# ---- panel GEMMs (csrc/ff.hip): the feed-forward and the cross-attention query projection of the patched block ----
def panel_rows(n: int) -> int:
    return (n + ROW_PAD - 1) // ROW_PAD * ROW_PAD


@_on_device
def to_panels(x: torch.Tensor, order: Optional[torch.Tensor] = None, rows: Optional[int] = None) -> torch.Tensor:
    """(rows, C) row-major -> k-panels (C / 8, rows_pad, 8); ``order`` (n,) int32: output row r = x[order[r]] (-1 = zeros)."""
    _req(x, "x")
    C = x.shape[-1]
    x2 = x.reshape(-1, C)
    n = x2.shape[0] if order is None else order.numel()
    n = n if rows is None else rows
    out = torch.empty((C // 8, panel_rows(n), 8), dtype=x.dtype, device=x.device)
    _check(lib().vtm_to_panels(_ptr(x2), dtype_code(x2), n, C, _ptr(order), _ptr(out), out.shape[1], _stream()),
           "vtm_to_panels")
    return out


@_on_device
def layernorm_panels(x: torch.Tensor, weight: Optional[torch.Tensor], bias: Optional[torch.Tensor], eps: float) -> torch.Tensor:
    """torch.nn.LayerNorm over the last axis, result as k-panels (C / 8, rows_pad, 8) (rows = all leading axes flattened)."""
    _req(x, "x")
    C = x.shape[-1]
    xc = x.contiguous()
    rows = xc.numel() // C
    out = torch.empty((C // 8, panel_rows(rows), 8), dtype=x.dtype, device=x.device)
    _check(lib().vtm_layernorm_panels(_ptr(xc), _ptr(weight), _ptr(bias), dtype_code(xc), rows, C, float(eps), _ptr(out),
                                      out.shape[1], _stream()), "vtm_layernorm_panels")
    return out


@_on_device
def ff_geglu(x_panels: torch.Tensor, n: int, w1_panels: torch.Tensor, D: int, bias: Optional[torch.Tensor]) -> torch.Tensor:
    """value * gelu(gate) of the GEGLU projection, panels in, panels (D / 8, n_pad, 8) out; see include/vidtome_hip.h."""
    K = x_panels.shape[0] * 8
    out = torch.empty((D // 8, x_panels.shape[1], 8), dtype=x_panels.dtype, device=x_panels.device)
    _check(lib().vtm_ff_geglu(_ptr(x_panels), n, x_panels.shape[1], _ptr(w1_panels), D, w1_panels.shape[1], K, _ptr(bias),
                              dtype_code(x_panels), _ptr(out), _stream()), "vtm_ff_geglu")
    return out


@_on_device
def linear_panels(x_panels: torch.Tensor, n: int, w_panels: torch.Tensor, N: int, bias: Optional[torch.Tensor],
                  resid: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """(n, N) token rows = x W^T (+ bias) (+ resid), panels in; see include/vidtome_hip.h.  Either operand may be a row
    range of a larger panel tensor (a view ``p[:, r0:r1]``: the panel stride is taken from the view); ``out`` may be a
    preallocated (n, N) view with a row stride >= N."""
    K = x_panels.shape[0] * 8
    if x_panels.stride(2) != 1 or x_panels.stride(1) != 8 or w_panels.stride(2) != 1 or w_panels.stride(1) != 8 \
            or w_panels.shape[0] * 8 != K:
        raise RuntimeError("linear_panels: operands must be (K / 8, rows, 8) panel tensors (or row ranges of one)")
    if out is None:
        out = torch.empty((n, N), dtype=x_panels.dtype, device=x_panels.device)
    if out.stride(1) != 1 or out.shape[0] < n or out.shape[1] < N:
        raise RuntimeError("linear_panels: out must be an (n, N) view, contiguous along its rows")
    if resid is not None:
        _req(resid, "resid")
        if resid.numel() != n * N or resid.dtype != out.dtype or out.stride(0) != N:
            raise RuntimeError("linear_panels: residual shape / dtype mismatch")
    _check(lib().vtm_linear_panels(_ptr(x_panels), n, x_panels.stride(0) // 8, _ptr(w_panels), N, w_panels.stride(0) // 8, K,
                                   _ptr(bias), _ptr(resid), dtype_code(x_panels), out.data_ptr(), out.stride(0), _stream()),
           "vtm_linear_panels")
    return out


@_on_device
def gather_panels(x0: torch.Tensor, x1: Optional[torch.Tensor], rows: Optional[torch.Tensor], rows2: Optional[torch.Tensor],
                  n: int) -> torch.Tensor:
    """Panels (C / 8, B * n_pad, 8) of pool[b, rows[b, rows2[b, i]]], i < n, sample b at rows b * n_pad (n_pad = n rounded up
    to 256, padding rows zero); pool = x0 | x1 as for gather_rows."""
    _req(x0, "x0")
    B, P0, C = x0.shape
    P1 = 0 if x1 is None else _req(x1, "x1").shape[1]
    n_pad = panel_rows(n)
    out = torch.empty((C // 8, B * n_pad, 8), dtype=x0.dtype, device=x0.device)
    _check(lib().vtm_gather_panels(_ptr(x0), P0, _ptr(x1), P1, dtype_code(x0), B, C, _ptr(rows),
                                   0 if rows is None else rows.shape[1], _ptr(rows2), n, _ptr(out), n_pad, _stream()),
           "vtm_gather_panels")
    return out

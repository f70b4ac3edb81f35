/*
 * vidtome_hip.h -- C ABI of libvidtome_hip.so: the MI355X (gfx950) implementation of VidToMe's
 * cross-frame token-merging hot path.
 *
 * Every entry point replaces a piece of the reference's Python hot path (the reference has no
 * native code; "binding" = the ctypes stub in vidtome_amd/_lib.py).
 * Citations are file:line under the reference checkout (/root/reference).
 *
 * Conventions
 *   - plain pointers and sizes only; all pointers are DEVICE pointers unless stated otherwise;
 *   - the caller allocates every input, output and workspace (e.g. through PyTorch's caching
 *     allocator); the library is stateless, never allocates or frees device memory and never
 *     synchronises the device;
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = default stream);
 *   - every call returns 0 on success or a negative VTM_E* code; vtm_last_error() returns a
 *     thread-local description of the last failure;
 *   - index arrays are int32 (the reference uses int64 tensors; values are identical);
 *   - token tensors are row-major (B, rows, C) in the model dtype (VTM_F16 / VTM_BF16 / VTM_F32).
 *
 * Canonical arithmetic of the matching path (bitwise equal to oracle/vtm_oracle.c):
 *   tokens upcast to fp32; norm = sqrtf(k-ascending fmaf chain of x*x); xhat = x / norm (IEEE
 *   divide); score = k-ascending fmaf chain from +0 (v_mfma_f32_32x32x2_f32 is exactly that chain);
 *   row max = first index among equals, first NaN wins; sort = descending, NaN first, ties by
 *   ascending index.
 */
#ifndef VIDTOME_HIP_H
#define VIDTOME_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2 (round 6): `flags_out` of the vtm_match_filtered* family is 8 int32 (it was 4 in version 1: a host built against the
 * version-1 header would see a 16-byte overrun, hence the bump -- check vtm_version() before calling). */
#define VTM_ABI_VERSION 2

enum vtm_dtype { VTM_F32 = 0, VTM_F16 = 1, VTM_BF16 = 2 };

enum vtm_status {
    VTM_OK = 0,
    VTM_EINVAL = -1,   /* bad argument (null pointer, negative size, unsupported dtype/shape) */
    VTM_ELAUNCH = -2,  /* HIP launch / runtime error (message in vtm_last_error) */
    VTM_EWORKSPACE = -3 /* workspace too small */
};

typedef void *vtm_stream_t; /* hipStream_t */

int vtm_version(void);
const char *vtm_last_error(void);
/* Bit mask of the ablation switches (vidtome_amd/csrc/ablate.h: experiment builds that drop loads / barriers / stores of
 * the hand-scheduled kernels and produce WRONG results) the library was compiled with.  0 for every library that may be
 * shipped; a host should refuse to run on anything else. */
int vtm_build_ablations(void);

/* Tile geometry the operand matrices of vtm_match must be padded to (rows to VTM_MATCH_ROW_PAD,
 * channels to VTM_MATCH_K_PAD). */
#define VTM_MATCH_ROW_PAD 256
#define VTM_MATCH_K_PAD 32
int64_t vtm_pad_rows(int64_t n);
int64_t vtm_pad_k(int64_t C);

/* ------------------------------------------------------------------------------------------------
 * vtm_normalize_gather -- replaces `metric / metric.norm(dim=-1, keepdim=True)` fused with `split`
 * (vidtome/merge.py:76-85 and 383-390).
 * The token pool is two row segments: x0 = the joined chunk (B, P0, C) and x1 = the block's global
 * anchor tokens (B, P1, C) (x1 may be NULL when P1 == 0); pool row id p < P0 addresses x0, else x1.
 * rows (B, n) int32 are pool row ids.  out receives the normalised rows in the k-PANEL operand layout of
 * vtm_match, B * C_pad * n_pad floats indexed [b][g = k/8][kh = k%2][row][e = (k%8)/2]: a panel (g, kh)
 * holds, for every row, the 4 channels 8g + kh + {0,2,4,6} as one 16-byte entry, so that (i) lane
 * (row, kh)'s 16-byte read feeds four consecutive v_mfma_f32_32x32x2_f32 k-steps in ascending k order,
 * (ii) a tile's panel slice is one contiguous run of rows (LDS-DMA friendly, fully coalesced);
 * rows >= n and channels >= C are zero.  norms (B, n) fp32 receives the row norms (it doubles
 * as the scratch between the two kernels of the call).
 * ---------------------------------------------------------------------------------------------- */
int vtm_normalize_gather(const void *x0, int64_t P0, const void *x1, int64_t P1, int dtype, int64_t B,
                         int64_t C, const int32_t *rows, int64_t n, float *norms, float *out,
                         int64_t n_pad, int64_t C_pad, vtm_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * vtm_match -- replaces `scores = a @ b.transpose(-1,-2)` + `scores.max(dim=-1)`
 * (vidtome/merge.py:87,109-113 and 392,413-417) and, with align != 0, the aligned variant
 * `torch.cat([*scores], dim=-1).max(dim=-1)` (merge.py:93-97 / 397-401).  The (B, Ns, Nd) score
 * matrix is never materialised.
 * a (src, Ns_pad rows), b (dst, Nd_pad rows): k-panel operands written by vtm_normalize_gather.
 * best: (B, Ns) uint64 when align == 0, (Ns) when align != 0.  Each entry is a packed key
 *     (orderable(node_max) << 32) | ~node_idx
 * whose unsigned maximum implements "largest value, first index, first NaN wins"; node_idx is in
 * [0, Nd) (align == 0) or [0, B*Nd) (align != 0, concatenated dst axis).  The call zero-fills
 * `best` itself.  Decode with vtm_decode_best.
 * ---------------------------------------------------------------------------------------------- */
int vtm_match(const float *a, const float *b, int64_t B, int64_t Ns, int64_t Nd, int64_t Ns_pad,
              int64_t Nd_pad, int64_t C_pad, int align, uint64_t *best, vtm_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * vtm_match_filtered -- the same packed result as vtm_normalize_gather x2 + vtm_match, BIT FOR BIT, several
 * times faster: an fp16-MFMA filter pass (operands hi = fp16(1024*xhat), one product hi_dst * hi_src;
 * the residual lo terms are a build-time option) collects for every src row the dst rows whose approximate score lies within a
 * rigorous error window of the row's running maximum -- skipping the rest of a 32 x 32 block once a Cauchy-Schwarz bound on
 * the channels still to come says that none of its pairs can reach that window -- and an fp32 refine pass evaluates the
 * canonical fmaf chain on those candidates only.  Bounded escapes: a row with more than 64 candidates (flat image
 * regions, massively duplicated dst rows), or whose own norm is not a finite positive number in [2^-100, 2^100], is recomputed
 * by exact fp32-MFMA score tiles against all dst rows; a DST row of that kind (zero token -> NaN xhat, merge.py:84 has no eps)
 * makes that kernel recompute every row of the call (device flag, no host round trip).  Four launches per call (operand
 * preparation, filter, refine, escape).  C > 1280 is rejected (the error budget is derived for C <= 1280; use vtm_match).
 * Inputs are the token pool (x0 | x1, as vtm_normalize_gather) and the gathered pool ids a_rows (B, Ns),
 * b_rows (B, Nd); B * Nd < 2^31.  ws: >= vtm_match_filtered_ws_bytes(...) bytes.  flags_out (optional, 8 int32,
 * device): [0] = 1 if every row was recomputed exactly (a dst row without a usable norm), [1] = 1 if any row without a
 * usable norm was seen, [2] = number of rows recomputed by the escape because their candidate list overflowed or their own
 * norm was unusable, [3] = number of (row, dst) pairs the refine pass evaluated, [4] = 32 x 32 score blocks the filter's
 * partial-sum pruning tested, [5] = blocks still alive after the test (the others skipped their remaining MFMAs; both 0
 * when the rows are too short to prune), [6] = internal (the escape launch's work counter while it runs: unspecified), [7] = 0
 * (vtm_match_filtered_plan: blocks inside the spans of the second launch).  The block counters are only collected when flags_out
 * is given.  flags_out may be device memory or host memory (ABI version 2): memory a kernel can store to -- device, pinned /
 * registered host -- is written by the call's last launch itself, pageable host memory by a 32-byte asynchronous copy behind the
 * call; either way the values are there once the stream has passed the call.  Derivation of the window:
 * vidtome_amd/csrc/match_filter.hip.
 *
 * vtm_match_filtered_seeded -- the same result, usually faster on video tokens: before the filter starts every src row gets
 * a starting maximum from ONE guessed pair, the dst row at the same token position (one more small launch).  seed_N =
 * tokens per frame (0 = no seeds = vtm_match_filtered); pool rows < seed_L are tokens of the joined chunk (position =
 * row % seed_N), the rows of x1 have the positions seed_pos1 (B, P1) or, NULL, none; seed_table (B, seed_N) maps a position to
 * a dst index (vtm_partition_global writes it), NULL = identity (a local level: the first dst frame's rows are dst
 * indices 0 .. seed_N-1).  The score of an actual pair is a valid running maximum, so the result cannot change; a useless
 * guess only fails to help.
 * ---------------------------------------------------------------------------------------------- */
size_t vtm_match_filtered_ws_bytes(int64_t B, int64_t C, int64_t Ns, int64_t Nd, int align);
int vtm_match_filtered(const void *x0, int64_t P0, const void *x1, int64_t P1, int dtype, int64_t B,
                       int64_t C, const int32_t *a_rows, int64_t Ns, const int32_t *b_rows, int64_t Nd,
                       int align, void *ws, size_t ws_bytes, uint64_t *best, int32_t *flags_out,
                       vtm_stream_t stream);
int vtm_match_filtered_seeded(const void *x0, int64_t P0, const void *x1, int64_t P1, int dtype, int64_t B,
                              int64_t C, const int32_t *a_rows, int64_t Ns, const int32_t *b_rows, int64_t Nd,
                              int align, void *ws, size_t ws_bytes, uint64_t *best, int32_t *flags_out,
                              int64_t seed_L, int64_t seed_N, const int32_t *seed_pos1, const int32_t *seed_table,
                              vtm_stream_t stream);
/* vtm_match_filtered_plan -- vtm_match_filtered_seeded with the launch plan chosen by the caller; the RESULT is the same bits
 * for every plan.  VTM_MATCH_ONE_LAUNCH = as above.  VTM_MATCH_SCOUT_RANGE (round 5) is for levels whose src AND dst rows are in
 * (frame, position) order (the first local level): a "scout" launch runs the filter loop over the channels in front of the
 * pruning test only and marks the 256 x 128 (src tile, dst tile) pairs in which a block stays alive; the filter launch proper then
 * shrinks every workgroup's dst range -- one dst frame of seed_N tokens per split -- to the span of the marked tiles.  On
 * frames of one clip 94 % of the tile pairs of such a level are dead and a top level-1 call drops from 0.75 to about 0.5 ms;
 * when most tiles stay alive (uncorrelated tokens, noisy clips at this test depth, rows in similarity-rank order) the scout is
 * pure overhead (+ 40 %).  The host steers by the counters of the previous call of the same level: flags_out may be PINNED HOST
 * memory (the 32-byte copy is asynchronous either way); with this plan flags_out[4] / [5] = blocks tested / alive in the scout
 * and flags_out[7] = blocks inside the spans the second launch processed ([7] / [4] = the fraction of the level it had to stream:
 * the plan pays below about 0.09 -- merge.MatchPlanner's HIGH; round 5's first measurement, "below about 0.45", did not survive
 * the per-level runs of profiles/r05_n_scout_range_plan.txt).  Falls back to one launch when there are no seeds, the rows are
 * shorter than 256 channels or seed_N is not a multiple of 128 >= 256. */
#define VTM_MATCH_ONE_LAUNCH 0
#define VTM_MATCH_SCOUT_RANGE 1
/* mode = VTM_MATCH_SCOUT_RANGE | VTM_MATCH_SCOUT_STEPS(k): the scout tests after k 64-channel steps instead of the filter's own
 * test depth (40 % of the channels: 2 steps at C = 320, 4 at C = 640) -- with rest norms of its own, so the certificate is as
 * rigorous; k = 0 or k >= that depth: the filter's depth.  The scout's cost is its steps (1 step: -34 % at C = 320); a low-noise
 * clip's dead tiles are dead after one step already, on noisier data more tiles stay marked and the spans grow. */
#define VTM_MATCH_SCOUT_STEPS(k) (((k) & 0xff) << 8)
int vtm_match_filtered_plan(const void *x0, int64_t P0, const void *x1, int64_t P1, int dtype, int64_t B,
                            int64_t C, const int32_t *a_rows, int64_t Ns, const int32_t *b_rows, int64_t Nd,
                            int align, void *ws, size_t ws_bytes, uint64_t *best, int32_t *flags_out,
                            int64_t seed_L, int64_t seed_N, const int32_t *seed_pos1, const int32_t *seed_table,
                            int mode, vtm_stream_t stream);

/* vtm_position_order + vtm_match_filtered_ordered (round 5) -- the matcher on POSITION-ORDERED rows, result in the caller's
 * indexing.  The reference's sequence order behind the first level is [unmerged tokens by descending score | dst tokens]
 * (merge.py:98-117): a 32 x 32 score block then holds a same-position pair of two frames with probability 0.25-0.4 and a
 * quarter of the blocks survive the filter's pruning test, against 1 % when the rows lie in position order.  The matcher's
 * result -- per src row the maximal canonical score and the LOWEST dst index attaining it -- does not depend on the order in
 * which the rows are met, so:
 *   vtm_position_order sorts both row lists of a call by token position (position of pool row r: r % N for r < L, the
 *   chunk's (frame, position) rows; pos1[b, r - P0] for the rows of x1, pos1 NULL or out of range = none: such rows go behind
 *   the last position) -- a counting sort, stable (ties by original index), 3 small launches for both operands of all B
 *   samples.  Outputs: a_sorted / b_sorted = the lists in position order, a_order / b_order = the original index of every
 *   sorted entry, table (B, N; optional) = position -> first sorted dst entry holding it, -1 = none (the seed table of the
 *   call).  counters: vtm_position_order_counter_ints(B, N) int32 that must be ZERO on entry and are zero again when the call
 *   has run (allocate zeroed once, reuse); ws: vtm_position_order_ws_bytes bytes of scratch.  shared_order != 0 (aligned
 *   matching): sample 0's positions decide ONE order and every sample's lists (and table) are written in it.
 *   vtm_match_filtered_ordered = vtm_match_filtered_plan on the sorted lists (aligned calls: the lists must come from a
 *   shared_order sort -- entry i is the same original index in every sample -- and sample 0's a_order names the rows), with refine / escape reporting
 *   row and column through a_order / b_order and breaking ties by the ORIGINAL dst index: `best` is bit-identical to
 *   vtm_match_filtered(a_rows, b_rows).  With VTM_MATCH_SCOUT_RANGE the dst axis is one position-major run cut
 *   into the one-launch plan's splits (a src tile's span lies in one or two of them).  N <= VTM_POSITION_ORDER_MAX_N. */
#define VTM_POSITION_ORDER_MAX_N 16360   /* tokens per frame: N + 2 offsets in 64 KB of LDS */
size_t vtm_position_order_counter_ints(int64_t B, int64_t N);
size_t vtm_position_order_ws_bytes(int64_t B, int64_t Ns, int64_t Nd, int64_t N);
int vtm_position_order(const int32_t *a_rows, int64_t Ns, const int32_t *b_rows, int64_t Nd, int64_t B, int64_t L,
                       int64_t N, const int32_t *pos1, int64_t P0, int64_t P1, int32_t *counters, void *ws,
                       size_t ws_bytes, int32_t *a_sorted, int32_t *a_order, int32_t *b_sorted, int32_t *b_order,
                       int32_t *table, int shared_order, vtm_stream_t stream);
int vtm_match_filtered_ordered(const void *x0, int64_t P0, const void *x1, int64_t P1, int dtype, int64_t B,
                               int64_t C, const int32_t *a_sorted, int64_t Ns, const int32_t *b_sorted, int64_t Nd,
                               int align, void *ws, size_t ws_bytes, uint64_t *best, int32_t *flags_out, int64_t seed_L,
                               int64_t seed_N, const int32_t *seed_pos1, const int32_t *seed_table, int mode,
                               const int32_t *a_order, const int32_t *b_order, vtm_stream_t stream);

/* node_max (fp32, -0 canonicalised to +0) and node_idx (int32) out of packed keys; either output may
 * be NULL. */
int vtm_decode_best(const uint64_t *best, int64_t n, float *node_max, int32_t *node_idx,
                    vtm_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * vtm_sort_desc -- replaces `node_max.argsort(dim=-1, descending=True)` (merge.py:98,113 / 402,417)
 * with the canonical stable order (descending value, NaN first, ties by ascending index).
 * best (rows, n) packed keys from vtm_match -> perm (rows, n) int32.
 * ws: workspace of at least vtm_sort_ws_bytes(rows, n) bytes.
 * ---------------------------------------------------------------------------------------------- */
size_t vtm_sort_ws_bytes(int64_t rows, int64_t n);
int vtm_sort_desc(const uint64_t *best, int64_t rows, int64_t n, int32_t *perm, void *ws,
                  size_t ws_bytes, vtm_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Index planning (all tiny, int32, device-resident; no host round trips).
 *
 * vtm_partition_local -- the src/dst partition of bipartite_soft_matching_randframe
 * (merge.py:41-74): tnum = (N_in - unm_pre) / F is passed in; ts = min(target_stride, F);
 * randf is the value of the reference's `torch.randint(0, ts, [1], generator=...)` draw (the host
 * draws it from the same CPU generator).  cur (B, N_in) holds the pool row id of every token of the
 * current sequence (NULL = identity, i.e. the first level).  Outputs: a_pos (Ns) / b_pos (Nd) =
 * a_idx / b_idx of the reference, a_rows (B, Ns) / b_rows (B, Nd) = cur gathered at those positions.
 * Ns / Nd must be the counts vtm_partition_counts returns.
 * ---------------------------------------------------------------------------------------------- */
int vtm_partition_counts(int64_t N_in, int64_t unm_pre, int64_t tnum, int64_t ts, int64_t randf,
                         int64_t *Ns, int64_t *Nd); /* host-only helper */
int vtm_partition_local(const int32_t *cur, int64_t B, int64_t N_in, int64_t unm_pre, int64_t tnum,
                        int64_t ts, int64_t randf, int32_t *a_pos, int32_t *b_pos, int32_t *a_rows,
                        int32_t *b_rows, int64_t Ns, int64_t Nd, vtm_stream_t stream);

/* vtm_partition_global -- bipartite_soft_matching_2s's split (merge.py:374-375) for the sequence
 * `cat([local, global])` (local_is_src != 0, patch.py:63-66) or `cat([global, local])`
 * (patch.py:68-71).  cur_local (B, Ml) pool ids of the local merged tokens; the anchors are pool
 * rows [anchor_base, anchor_base + Mg).  Outputs a_pos/b_pos/a_rows/b_rows as above with
 * Ns = src_len, Nd = Ml + Mg - src_len.
 * seed_table (optional, (B, tokens) int32, tokens = tokens per frame): filled with, for every token position, the index of
 * ONE dst row holding that position (-1: none) -- what vtm_match_filtered_seeded takes; local tokens have position
 * pool id % tokens, anchor j has anchor_pos[b, j] ((B, Mg), optional: without it anchors have no position). */
int vtm_partition_global(const int32_t *cur_local, int64_t B, int64_t Ml, int64_t anchor_base,
                         int64_t Mg, int local_is_src, int32_t *a_pos, int32_t *b_pos,
                         int32_t *a_rows, int32_t *b_rows, int32_t *seed_table, int64_t tokens,
                         const int32_t *anchor_pos, vtm_stream_t stream);

/* vtm_anchor_pos -- token positions of a new anchor set (the host tracks them next to module.global_tokens so that the
 * next global level can be seeded): anchors_out[b, p] = pool[b, amap[b, p]] (patch.py:80; amap == NULL: the first M pool
 * rows, patch.py:82), pool = [joined chunk: L rows, position = row % tokens | old anchors: old_pos (B, Mg) or NULL].
 * out (B, M) int32, -1 where the position is unknown. */
int vtm_anchor_pos(const int32_t *amap, int64_t B, int64_t M, int64_t L, int64_t tokens, const int32_t *old_pos,
                   int64_t Mg, int32_t *out, vtm_stream_t stream);

/* vtm_anchor_maps -- the maps behind a global level in one launch (they are vtm_compose x 2 + vtm_anchor_pos):
 * loc (B, Ml) = merged position of every local token = inv_g[off + t] (merge.py:459: the local slice of the level's unmerge
 * map inv_g (B, N_in)); amap (B, Ml) = new_cur[loc] = the pool row each token of u(merged) is a copy of (patch.py:80 as one
 * gather from [chunk of L rows | old anchors]); pos (B, Ml), optional = the token position of that row (row % tokens for
 * chunk rows, old_pos (B, Mg) for anchor rows, -1 unknown). */
int vtm_anchor_maps(const int32_t *inv_g, int64_t N_in, int64_t off, const int32_t *new_cur, int64_t M, int64_t B,
                    int64_t Ml, int64_t L, int64_t tokens, const int32_t *old_pos, int64_t Mg, int32_t *loc, int32_t *amap,
                    int32_t *pos, vtm_stream_t stream);

/* vtm_plan_apply -- the index split after the sort (merge.py:100-117 / 404-421) and the bookkeeping
 * of the merge / unmerge closures (merge.py:119-155 / 423-460) as composed maps:
 *   unm_idx = perm[r:], src_idx = perm[:r], dst_idx = node_idx[src_idx] (% Nd when align);
 *   new_cur (B, U + Nd), U = Ns - r: pool ids of `cat([unm, dst])`  (the merge closure);
 *   inv (B, N_in): position in the merged sequence each input position is restored from
 *                  (the unmerge closure: dst -> itself, unm -> itself, src -> its dst).
 * best/perm have one row when align != 0.  unm_idx/src_idx/dst_idx (B, U)/(B, r)/(B, r) are optional
 * (NULL to skip) copies of the reference's index tensors. */
int vtm_plan_apply(const uint64_t *best, const int32_t *perm, const int32_t *a_pos,
                   const int32_t *b_pos, const int32_t *a_rows, const int32_t *b_rows, int64_t B,
                   int64_t N_in, int64_t Ns, int64_t Nd, int64_t r, int align, int32_t *new_cur,
                   int32_t *inv, int32_t *unm_idx, int32_t *src_idx, int32_t *dst_idx,
                   vtm_stream_t stream);

/* vtm_compose -- func_warper composition of unmerge closures (vidtome/utils.py:42-48, patch.py:85):
 * out[b, i] = inv_level[b, offset + inv_acc[b, i]] for i < n (inv_acc NULL = identity).  `offset`
 * selects the part bipartite_soft_matching_2s's unmerge returns (merge.py:459). */
int vtm_compose(const int32_t *inv_acc, const int32_t *inv_level, int64_t B, int64_t n,
                int64_t level_len, int64_t offset, int32_t *out, vtm_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * vtm_gather_rows -- the composed `merge` closure in replace mode (merge.py:119-133 / 423-437,
 * patch.py:52,76) and the global-token update `u(merged_tokens)` (patch.py:80):
 * out[b, p, :] = pool[b, map[b, p], :], pool = (x0 | x1) as in vtm_normalize_gather.  out is
 * (B, out_rows, C) with out_rows >= M (rows >= M are not written: callers pad merged sequences to a
 * multiple of 8 rows so the projection GEMMs keep 16-byte aligned leading dimensions).
 * ---------------------------------------------------------------------------------------------- */
int vtm_gather_rows(const void *x0, int64_t P0, const void *x1, int64_t P1, int dtype, int64_t B,
                    int64_t C, const int32_t *map, int64_t M, void *out, int64_t out_rows,
                    vtm_stream_t stream);

/* vtm_merge_reduce -- the `merge` closure's OTHER modes (merge.py:127-131 / 431-435; never reached from
 * compute_merge, part of the closure protocol):
 *     dst = dst.scatter_reduce(-2, dst_idx.expand(n, r, c), gather(src, src_idx), reduce=mode, include_self=True)
 * with torch's CPU arithmetic -- the sources of a destination row are folded into it one by one in index order, in fp32;
 * a 16-bit tensor is rounded once at the end; "mean" divides the rounded sum by (1 + number of sources) and rounds again;
 * amax / amin propagate NaN.  x is (B, N, C); src_rows (B, r) / dst_rows (B, Nd) are the rows of x of the merged src tokens
 * (in src_idx order) and of the dst tokens; seg_dst (B, r) lists the destinations of the r pairs in ASCENDING order and
 * seg_order (B, r) the pair each entry is (a STABLE sort by destination, e.g. vtm_sort_desc on keys whose high word is
 * 0xffffffff - dst_idx).  Writes out[b, out_row0 + j, :] for j < Nd; out is (B, out_ld, C) -- the merged sequence
 * [unm | dst] with out_row0 = number of unmerged tokens. */
enum { VTM_REDUCE_SUM = 0, VTM_REDUCE_PROD = 1, VTM_REDUCE_MEAN = 2, VTM_REDUCE_AMAX = 3, VTM_REDUCE_AMIN = 4 };
int vtm_merge_reduce(const void *x, int dtype, int64_t B, int64_t N, int64_t C, const int32_t *src_rows,
                     const int32_t *dst_rows, const int32_t *seg_dst, const int32_t *seg_order, int64_t r, int64_t Nd,
                     int mode, void *out, int64_t out_ld, int64_t out_row0, vtm_stream_t stream);

/* vtm_unmerge_add -- the composed `unmerge` closure + split_frame + residual
 * (merge.py:135-155 / 439-460, vidtome/utils.py:37-40, patch.py:168-169):
 * out[b, i, :] = y[b, inv[b, i], :] (+ resid[b, i, :] when resid != NULL).  y is (B, M, C). */
int vtm_unmerge_add(const void *y, int64_t M, const int32_t *inv, const void *resid, int dtype,
                    int64_t B, int64_t L, int64_t C, void *out, vtm_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * vtm_attention -- the self-attention core inside `self.attn1(...)` (patch.py:157-162) as stated by
 * the reference's own `sa_forward` (utils/pnp_utils.py:47-95): per head
 *     out = softmax(q k^T * scale) v,      no mask, no dropout.
 * q, k, out: (B, Mp, h*d) buffers of which the first M rows per sample are the sequence (Mp >= M is
 * the per-sample row count of the buffers), with row strides ldq/ldk/ldo elements (heads interleaved on the channel
 * axis = head_to_batch_dim / batch_to_head_dim without the copies); vt: (B, h*d, ldvt) = v
 * TRANSPOSED (channel-major, key-contiguous, ldvt >= M and a multiple of 8) so that the PV operand
 * is read without a transpose.  dtype VTM_F16 or VTM_BF16; fp32 accumulation; d in {40,64,80,160}
 * (any multiple of 8 up to 160).
 * share_groups > 1 = the PnP injection branch (pnp_utils.py:57-67,86-90): probabilities of sample
 * b come from q/k of sample b % (B / share_groups), v stays per sample.
 * ---------------------------------------------------------------------------------------------- */
int vtm_attention(const void *q, int64_t ldq, const void *k, int64_t ldk, const void *vt,
                  int64_t ldvt, void *out, int64_t ldo, int dtype, int64_t B, int64_t h, int64_t M,
                  int64_t Mp, int64_t d, float scale, int share_groups, void *ws, size_t ws_bytes,
                  vtm_stream_t stream);

/* Optional workspace of vtm_attention / vtm_attention_kv (0 = none needed).  All workgroups of an attention launch
 * take the same time; when the last round of workgroups would leave most of the chip idle, those query blocks are
 * split along the key axis into short workgroups (same launch, behind the whole ones) whose partial results (fp32
 * accumulators, running max, denominator) live in this workspace and are merged by a second kernel.  ws may be NULL
 * (no splitting, slower for such shapes). */
size_t vtm_attention_ws_bytes(int64_t B, int64_t h, int64_t Mq, int64_t Mk, int64_t d);

/* The same kernel with separate query / key lengths: the block's cross-attention `self.attn2(...)`
 * (vidtome/patch.py:178-183; SD: Mk = 77 text tokens per frame).  q (B, Mqp, .) / out as above; k (B, Mkp, .),
 * vt (B, h*d, ldvt >= Mk).  vtm_attention is this entry with Mk = Mq, Mkp = Mqp. */
int vtm_attention_kv(const void *q, int64_t ldq, const void *k, int64_t ldk, const void *vt, int64_t ldvt,
                     void *out, int64_t ldo, int dtype, int64_t B, int64_t h, int64_t Mq, int64_t Mqp,
                     int64_t Mk, int64_t Mkp, int64_t d, float scale, int share_groups, void *ws, size_t ws_bytes,
                     vtm_stream_t stream);

/* vtm_attention_kv with a DEVICE-side query bound: sample b only has q_count[b] <= Mq meaningful query rows (the
 * compacted live queries of vtm_compact_queries); query blocks that start at or beyond the count exit at once, rows
 * beyond it are not meaningful.  The launch is sized for the host-known bound Mq -- no host round trip. */
int vtm_attention_kv_bounded(const void *q, int64_t ldq, const void *k, int64_t ldk, const void *vt, int64_t ldvt,
                             void *out, int64_t ldo, int dtype, int64_t B, int64_t h, int64_t Mq, int64_t Mqp,
                             int64_t Mk, int64_t Mkp, int64_t d, float scale, const int32_t *q_count, void *ws,
                             size_t ws_bytes, vtm_stream_t stream);
/* ... and with shared probabilities (pnp_utils.py:57-67, 75-90) on top: the probabilities of sample b come from q / k of sample
 * b % (B / share_groups), v stays per sample, and every sample of a group has the SAME live rows (align_batch: one index set
 * for the batch, merge.py:73-76), so q_count[b % (B / share_groups)] bounds them all.  q_count has B entries; the first
 * B / share_groups are read.  Workspace: vtm_attention_ws_bytes. */
int vtm_attention_kv_shared_bounded(const void *q, int64_t ldq, const void *k, int64_t ldk, const void *vt, int64_t ldvt,
                                    void *out, int64_t ldo, int dtype, int64_t B, int64_t h, int64_t Mq, int64_t Mqp,
                                    int64_t Mk, int64_t Mkp, int64_t d, float scale, int share_groups,
                                    const int32_t *q_count, void *ws, size_t ws_bytes, vtm_stream_t stream);
/* workspace of a vtm_attention_kv_bounded launch: with a device-side query count the launch cannot plan its rounds, so every
 * work item is split in two along the key axis (one partial record per workgroup, merged by a second kernel); with less
 * workspace than this the launch falls back to the plain plan of vtm_attention_ws_bytes */
size_t vtm_attention_kv_bounded_ws_bytes(int64_t B, int64_t h, int64_t Mq, int64_t Mk, int64_t d);

/* vtm_transpose_cols -- V^T of a block that does not merge (patch.py:157-162 at the sites beyond max_downsample: per-frame
 * attention): the q | k | v projection GEMM leaves v as columns of its token-major output, the attention core reads V
 * channel-major.  x (BF, N, ldx) 16-bit, the C columns starting at `x` -> out (BF, C, ldo), tokens N .. ldo zero-filled.
 * C, ldx, ldo multiples of 8, ldo >= N, 16-byte aligned pointers. */
int vtm_transpose_cols(const void *x, int64_t ldx, int dtype, int64_t BF, int64_t N, int64_t C, void *out, int64_t ldo,
                       vtm_stream_t stream);

/* vtm_fold_keys / vtm_attention_kv_folded -- the anchors' exact duplicates as ONE key each.  patch.py:80 stores
 * u(merged_tokens) as the next chunk's global tokens: every local token that merged into an anchor row carries that
 * row's content, so the anchors hold groups of identical rows, and so does the merged sequence they join
 * (patch.py:63-71).  Identical key / value rows weigh in softmax(q k^T) v exactly like one key whose (base-2) score carries
 * + log2(multiplicity).  vtm_fold_keys: cur (B, M) pool row of every merged position (rows >= L are anchor rows cur - L of
 * Ma), cid (B, Ma) a content id in [0, n_ids) per anchor row (equal ids <=> identical rows; vtm_compact_queries' tmap of
 * the block that produced the anchors; an id outside [0, n_ids), e.g. -1, = "no id": the row stands for itself).  Outputs: key_sel (B, M) the surviving merged positions in ascending order (the
 * first copy of every group; entries past the count are 0), k_bias (B, ldkb) per surviving key log2(copies present) as
 * a (hi, lo) pair of the keys' 16-bit type (dtype VTM_F16 / VTM_BF16), k_count (B) the number of surviving keys.
 * vtm_attention_kv_folded is vtm_attention_kv_bounded (q_count may be NULL) over such a key list: k / vt hold the
 * projections of the key_sel rows, only the first k_count[b] (<= Mk, a device value) are keys, and the bias pair rides
 * in the spare k-slots of the contraction -- head dims with d % 16 == 8 only (SD's 40).  Same block output as the
 * unfolded call up to the rounding of log2(m) (2^-21 relative). */
size_t vtm_fold_keys_ws_bytes(int64_t B, int64_t M, int64_t n_ids);
int vtm_fold_keys(const int32_t *cur, int64_t B, int64_t M, int64_t L, const int32_t *cid, int64_t Ma, int64_t n_ids,
                  int dtype, void *ws, size_t ws_bytes, int32_t *key_sel, uint32_t *k_bias, int64_t ldkb,
                  int32_t *k_count, vtm_stream_t stream);
int vtm_attention_kv_folded(const void *q, int64_t ldq, const void *k, int64_t ldk, const void *vt, int64_t ldvt,
                            void *out, int64_t ldo, int dtype, int64_t B, int64_t h, int64_t Mq, int64_t Mqp,
                            int64_t Mk, int64_t Mkp, int64_t d, float scale, const int32_t *q_count,
                            const int32_t *k_count, const uint32_t *k_bias, int64_t ldkb, void *ws, size_t ws_bytes,
                            vtm_stream_t stream);

/* vtm_compact_queries -- which attention outputs a global level with the local chunk on the src side actually needs
 * (bipartite_soft_matching_2s's unmerge, merge.py:439-460, returns for every merged local token the output row of the
 * anchor token it merged into: `src = gather(dst, dst_idx)`; several local tokens may share one).  loc (B, Ml): merged
 * position of every local token (< U: its own row; >= U: anchor row loc - U of Nd).  Outputs: qc (B, Ml) the DISTINCT
 * positions ([0, U) then the matched anchor rows ascending; entries past the count are 0), tmap (B, Ml) the row of
 * that list each local token reads, count (B) the number of distinct positions.  ws: vtm_compact_queries_ws_bytes. */
size_t vtm_compact_queries_ws_bytes(int64_t B, int64_t Nd);
int vtm_compact_queries(const int32_t *loc, int64_t B, int64_t Ml, int64_t U, int64_t Nd, void *ws, size_t ws_bytes,
                        int32_t *qc, int32_t *tmap, int32_t *count, vtm_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Panel GEMMs -- the rest of the patched block around the hot path (vidtome/patch.py:171-199): the feed-forward
 * `self.ff(self.norm3(hidden_states)) + hidden_states` (Diffusers FeedForward of SD blocks: GEGLU(Linear C -> 8C), Linear
 * 4C -> C) and the query projection of the cross-attention `self.attn2(self.norm2(hidden_states), ...)`.
 * Operands are k-panels: [K / 8][rows_pad][8 elements], rows_pad = vtm_panel_rows(rows) (a multiple of 256); padding
 * rows may hold anything.  Token operands come from vtm_layernorm_panels (or vtm_to_panels), weights are packed once
 * with vtm_to_panels (`order` = the row permutation below, or NULL).
 *   vtm_layernorm_panels  torch.nn.LayerNorm with its result written as panels (norm3 / norm2; norm1 of the un-merged sites).
 *   vtm_gather_panels     the composed merge closure (merge.py:119-133 / 423-437) writing panels: row b * rows_per_sample + i
 *                         = pool[b, map[b, map2[b, i]]] (either map may be NULL), padding rows zero -- the token operand of
 *                         attn1's projections (patch.py:157-162) at the sites where the panel GEMM is used.
 *   vtm_ff_geglu          out = value * gelu(gate) of  x W1^T + b1  (erf gelu), written as panels [D / 8][n_pad][8] -- the
 *                         2D-wide projection is never written.  W1 is packed in TILE order: 128-row tile t = the value rows of
 *                         output channels 64 t .. 64 t + 63 followed by their gate rows (rows D + 64 t ..); bias (2 D fp32) in
 *                         the same order.  D % 64 == 0, K % 64 == 0.
 *   vtm_linear_panels     out (n, ldo) token rows = x W^T (+ bias fp32 (N)) (+ resid (n, ldo)), rounded like torch's Linear
 *                         followed by the residual add.  N % 8 == 0, K % 64 == 0.  n_pad / w_rows_pad are the ROW STRIDES of the
 *                         two panel operands (either may be a row range of a larger panel tensor, e.g. one sample's rows:
 *                         V^T = W_v X^T is this call with the weight as "token" operand and the sample's tokens as "weight").
 * ---------------------------------------------------------------------------------------------- */
int64_t vtm_panel_rows(int64_t n);
int vtm_to_panels(const void *x, int dtype, int64_t rows, int64_t C, const int32_t *order, void *out, int64_t rows_pad,
                  vtm_stream_t stream);
int vtm_layernorm_panels(const void *x, const void *gamma, const void *beta, int dtype, int64_t rows, int64_t C, float eps,
                         void *out, int64_t panel_rows, vtm_stream_t stream);
int vtm_gather_panels(const void *x0, int64_t P0, const void *x1, int64_t P1, int dtype, int64_t B, int64_t C,
                      const int32_t *map, int64_t map_ld, const int32_t *map2, int64_t n, void *out, int64_t rows_per_sample,
                      vtm_stream_t stream);
int vtm_ff_geglu(const void *x_panels, int64_t n, int64_t n_pad, const void *w1_panels, int64_t D, int64_t w_rows_pad,
                 int64_t K, const float *bias, int dtype, void *out_panels, vtm_stream_t stream);
int vtm_linear_panels(const void *x_panels, int64_t n, int64_t n_pad, const void *w_panels, int64_t N, int64_t w_rows_pad,
                      int64_t K, const float *bias, const void *resid, int dtype, void *out, int64_t ldo,
                      vtm_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * vtm_cfg_ddim -- the caller-side elementwise tail of a denoising step:
 * classifier-free guidance `eps = uncond + guidance * (cond - uncond)` (generate.py:276-278) fused with the
 * closed-form DDIM update of `pred_next_x` (generate.py:281-311):
 *     pred_x0 = (x - b*eps) / a ;   x_out = c*pred_x0 + d*eps
 * sampling: (a, b, c, d) = (mu, sigma, mu_prev, sigma_prev); inversion: (mu_prev, sigma_prev, mu, sigma).
 * eps_cond may be NULL (no guidance: eps = eps_uncond); eps_out / x_out are optional outputs.  Every
 * operation rounds to the tensor dtype in the reference's order, so fp16/bf16 results are bit-identical to
 * the torch expression.
 * ---------------------------------------------------------------------------------------------- */
int vtm_cfg_ddim(const void *x, const void *eps_uncond, const void *eps_cond, int dtype, int64_t n,
                 float guidance, float a, float b, float c, float d, void *eps_out, void *x_out,
                 vtm_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * vtm_layernorm -- the block's norm1 (vidtome/patch.py:139-146, the plain torch.nn.LayerNorm branch
 * `norm_hidden_states = self.norm1(hidden_states)`), the first operation of the patched segment and the
 * producer of the matching metric; also usable for norm2 / norm3 (patch.py:173-176, 187).
 * x, out: (rows, C) contiguous in `dtype`; gamma, beta: (C) in the same dtype or NULL (no affine part).
 * fp32 statistics (mean, then centred sum of squares; biased variance like torch), y = (x - mean) *
 * rsqrt(var + eps) * gamma + beta evaluated in fp32, rounded once.  C % 8 == 0, C <= 2048.
 * ---------------------------------------------------------------------------------------------- */
int vtm_layernorm(const void *x, const void *gamma, const void *beta, int dtype, int64_t rows, int64_t C,
                  float eps, void *out, vtm_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * vtm_geglu -- the gated activation inside the block's feed-forward `self.ff(...)` (vidtome/patch.py:187-199;
 * SD blocks use the Diffusers GEGLU feed-forward): x (rows, 2 D) = [value | gate] -> out (rows, D) =
 * value * gelu(gate), exact (erf) gelu, the gate rounded to `dtype` before the product like torch's two ops.
 * ---------------------------------------------------------------------------------------------- */
int vtm_geglu(const void *x, int dtype, int64_t rows, int64_t D, void *out, vtm_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * vtm_linear_rows -- `Linear(gather(tokens))`: the attention projections of the patched block
 * (`self.attn1(...)`, vidtome/patch.py:157-162; to_q / to_k / to_v / to_out[0] as in utils/pnp_utils.py:47-95)
 * fed by the composed merge map instead of a materialised merged tensor (the reference's
 * merge closure `cat([gather(src, unm_idx), dst])`, merge.py:119-133 / 423-437, followed by nn.Linear).
 *   out[b, i, :] = pool[b, p(i), :] @ W^T (+ bias),   p(i) = rows[b, rows2[b, i]]   (either map may be NULL:
 *   rows2 NULL -> p(i) = rows[b, i];  rows NULL -> p(i) = rows2[b, i] or i)
 * pool = x0 (B, P0, K) | x1 (B, P1, K) as in vtm_gather_rows; rows: (B, rows_ld) int32 pool ids (the composed merge
 * map); rows2: (B, n) int32 positions in the merged sequence (the live-query rows of a global level).
 * W: (N, K) row-major like torch.nn.Linear.weight, bias: (N) or NULL; fp16 / bf16, fp32 accumulation.
 * transposed == 0: out is (B, >= n, ldo >= N) token-major; transposed != 0: out is (B, N, ldo >= n) channel-major
 * (V^T for vtm_attention).  out_batch_stride in elements.  K % 32 == 0.  Rows >= n of out are not written.
 * ---------------------------------------------------------------------------------------------- */
int vtm_linear_rows(const void *x0, int64_t P0, const void *x1, int64_t P1, int dtype, int64_t B, int64_t K,
                    const int32_t *rows, int64_t rows_ld, const int32_t *rows2, int64_t n, const void *W,
                    const void *bias, int64_t N, void *out, int64_t ldo, int64_t out_batch_stride,
                    int transposed, vtm_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* VIDTOME_HIP_H */

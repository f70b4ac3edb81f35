// Index planning: partition (src/dst split), index split after the sort, composed merge/unmerge maps.
// Everything here is tiny int32 work kept on the device so that a block's whole merge plan is built
// without a host round trip (the reference builds the same index tensors with ~20 small torch ops per
// level: vidtome/merge.py:52-74, 98-117, 119-155; vidtome/patch.py:44-85).
#include "common.h"

#include <algorithm>

namespace {

// number of dst frames among frames [0, f): frames g with g % ts == randf
__device__ __host__ inline int64_t dst_frames_before(int64_t f, int64_t ts, int64_t randf) {
    return f > randf ? (f - randf - 1) / ts + 1 : 0;
}

// merge.py:59-69: position p of the input sequence -> (is_dst, index inside a_idx / b_idx)
__global__ __launch_bounds__(256) void partition_local_kernel(
    const int32_t *__restrict__ cur, int64_t B, int64_t N_in, int64_t unm_pre, int64_t tnum, int64_t ts,
    int64_t randf, int32_t *__restrict__ a_pos, int32_t *__restrict__ b_pos,
    int32_t *__restrict__ a_rows, int32_t *__restrict__ b_rows, int64_t Ns, int64_t Nd) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * N_in) return;
    const int64_t b = idx / N_in, p = idx % N_in;
    const int32_t row = cur ? cur[idx] : (int32_t)p;
    const int64_t n_frame_dst = Nd - unm_pre;  // dst positions that come from frames
    bool is_dst;
    int64_t j;
    if (p < unm_pre) {  // previously unmerged tokens are appended to dst (merge.py:67-69)
        is_dst = true;
        j = n_frame_dst + p;
    } else {
        const int64_t q = p - unm_pre;
        const int64_t f = q / tnum, t = q % tnum;
        const int64_t nb = dst_frames_before(f, ts, randf);
        is_dst = (f % ts) == randf;
        j = is_dst ? nb * tnum + t : q - nb * tnum;
    }
    if (is_dst) {
        if (b == 0) b_pos[j] = (int32_t)p;
        b_rows[b * Nd + j] = row;
    } else {
        if (b == 0) a_pos[j] = (int32_t)p;
        a_rows[b * Ns + j] = row;
    }
}

// merge.py:374-375 on cat([local, global]) / cat([global, local]) (patch.py:63-71)
__global__ __launch_bounds__(256) void partition_global_kernel(
    const int32_t *__restrict__ cur_local, int64_t B, int64_t Ml, int64_t anchor_base, int64_t Mg,
    int local_is_src, int32_t *__restrict__ a_pos, int32_t *__restrict__ b_pos,
    int32_t *__restrict__ a_rows, int32_t *__restrict__ b_rows, int32_t *__restrict__ table, int64_t tokens,
    const int32_t *__restrict__ anchor_pos) {
    const int64_t N = Ml + Mg;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * N) return;
    const int64_t b = idx / N, p = idx % N;
    const int64_t src_len = local_is_src ? Ml : Mg;
    // pool row id of position p of the concatenated sequence
    int32_t row;
    if (local_is_src)
        row = p < Ml ? cur_local[b * Ml + p] : (int32_t)(anchor_base + (p - Ml));
    else
        row = p < Mg ? (int32_t)(anchor_base + p) : cur_local[b * Ml + (p - Mg)];
    if (p < src_len) {
        if (b == 0) a_pos[p] = (int32_t)p;
        a_rows[b * src_len + p] = row;
    } else {
        const int64_t j = p - src_len;
        if (b == 0) b_pos[j] = (int32_t)p;
        b_rows[b * (N - src_len) + j] = row;
        // the matcher's seeds (match_filter.hip, seed_kernel): token position -> ONE dst row holding that position (several
        // frames hold it; whichever write lands last is as good as any other).  Local tokens: pool row % tokens per
        // frame; anchors: the positions the host tracks with them.
        if (table != nullptr) {
            int64_t pos = -1;
            if (row < anchor_base) pos = row % tokens;
            else if (anchor_pos != nullptr) pos = anchor_pos[b * Mg + (row - anchor_base)];
            if (pos >= 0 && pos < tokens) table[b * tokens + pos] = (int32_t)j;
        }
    }
}

// positions of the rows of a new anchor set (patch.py:80,82): anchors_out[b, p] = pool[b, amap[b, p]] with pool =
// [joined chunk (L rows, position = row % tokens) | old anchors (their tracked positions)]; amap == nullptr: rows 0 .. M-1
// of the pool themselves
__global__ __launch_bounds__(256) void anchor_pos_kernel(const int32_t *__restrict__ amap, int64_t B, int64_t M, int64_t L,
                                                         int64_t tokens, const int32_t *__restrict__ old_pos, int64_t Mg,
                                                         int32_t *__restrict__ out) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * M) return;
    const int64_t b = idx / M;
    const int64_t row = amap ? amap[idx] : idx % M;
    int32_t pos = -1;
    if (row < L) pos = (int32_t)(row % tokens);
    else if (old_pos != nullptr && row - L < Mg) pos = old_pos[b * Mg + (row - L)];
    out[idx] = pos;
}

// The three maps patch.py:80 needs behind a global level, in one launch (they were vtm_compose x 2 + vtm_anchor_pos):
//   loc[t]  = inv_g[off + t]      merged position of local token t (merge.py:459: the local part of the level's unmerge map)
//   amap[t] = new_cur[loc[t]]     pool row the new anchor t is a copy of   (u(merged) as ONE gather from [chunk | old anchors])
//   pos[t]  = token position of that row (the matcher's seeds), optional
__global__ __launch_bounds__(256) void anchor_maps_kernel(const int32_t *__restrict__ inv_g, int64_t N_in, int64_t off,
                                                          const int32_t *__restrict__ new_cur, int64_t M, int64_t B, int64_t Ml,
                                                          int64_t L, int64_t tokens, const int32_t *__restrict__ old_pos,
                                                          int64_t Mg, int32_t *__restrict__ loc, int32_t *__restrict__ amap,
                                                          int32_t *__restrict__ pos) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * Ml) return;
    const int64_t b = idx / Ml, t = idx % Ml;
    const int32_t p = inv_g[b * N_in + off + t];
    const int64_t row = new_cur[b * M + p];
    loc[idx] = p;
    amap[idx] = (int32_t)row;
    if (pos != nullptr) {
        int32_t q = -1;
        if (row < L) q = (int32_t)(row % tokens);
        else if (old_pos != nullptr && row - L < Mg) q = old_pos[b * Mg + (row - L)];
        pos[idx] = q;
    }
}

// merge.py:100-117 (index split) + 119-155 (closure bookkeeping as maps).
// One thread per sorted rank e in [0, Ns) and one per dst index j in [0, Nd).
__global__ __launch_bounds__(256) void plan_apply_kernel(
    const uint64_t *__restrict__ best, const int32_t *__restrict__ perm,
    const int32_t *__restrict__ a_pos, const int32_t *__restrict__ b_pos,
    const int32_t *__restrict__ a_rows, const int32_t *__restrict__ b_rows, int64_t B, int64_t N_in,
    int64_t Ns, int64_t Nd, int64_t r, int align, int32_t *__restrict__ new_cur,
    int32_t *__restrict__ inv, int32_t *__restrict__ unm_idx, int32_t *__restrict__ src_idx,
    int32_t *__restrict__ dst_idx) {
    const int64_t per = Ns + Nd;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * per) return;
    const int64_t b = idx / per, e = idx % per;
    const int64_t U = Ns - r, M = U + Nd;
    if (e < Ns) {
        const int64_t bb = align ? 0 : b;  // aligned: one matching shared by all samples (merge.py:106-108)
        const int32_t i = perm[bb * Ns + e];
        if (e < r) {  // merged src token: restored from its dst (merge.py:142,152-153)
            const uint32_t ni = ~(uint32_t)(best[bb * Ns + i] & 0xffffffffull);
            const int32_t dj = (int32_t)(align ? ni % (uint32_t)Nd : ni);  // merge.py:102-103 / 117
            inv[b * N_in + a_pos[i]] = (int32_t)(U + dj);
            if (src_idx) src_idx[b * r + e] = i;
            if (dst_idx) dst_idx[b * r + e] = dj;
        } else {  // unmerged src token: kept, in similarity-rank order (merge.py:124,149-150)
            const int64_t u = e - r;
            new_cur[b * M + u] = a_rows[b * Ns + i];
            inv[b * N_in + a_pos[i]] = (int32_t)u;
            if (unm_idx) unm_idx[b * U + u] = i;
        }
    } else {  // dst token (merge.py:133,147)
        const int64_t j = e - Ns;
        new_cur[b * M + U + j] = b_rows[b * Nd + j];
        inv[b * N_in + b_pos[j]] = (int32_t)(U + j);
    }
}

__global__ __launch_bounds__(256) void compose_kernel(const int32_t *__restrict__ inv_acc,
                                                      const int32_t *__restrict__ inv_level, int64_t B,
                                                      int64_t n, int64_t level_len, int64_t offset,
                                                      int32_t *__restrict__ out) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * n) return;
    const int64_t b = idx / n, i = idx % n;
    const int64_t p = inv_acc ? inv_acc[idx] : i;
    out[idx] = inv_level[b * level_len + offset + p];
}

__global__ __launch_bounds__(256) void decode_best_kernel(const uint64_t *__restrict__ best, int64_t n,
                                                          float *__restrict__ node_max,
                                                          int32_t *__restrict__ node_idx) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const uint64_t k = best[idx];
    if (node_idx) node_idx[idx] = (int32_t)(~(uint32_t)(k & 0xffffffffull));
    if (node_max) {
        const uint32_t o = (uint32_t)(k >> 32);
        uint32_t u;
        if (o == 0xffffffffu)
            u = 0x7fc00000u;  // NaN
        else
            u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
        node_max[idx] = __uint_as_float(u);
    }
}

inline unsigned blocks_for(int64_t n) { return (unsigned)vtm::cdiv(n, 256); }

// Distinct attention queries of a global level whose src side is the local chunk (merge.py:439-460 + patch.py:59-82).
// `loc[b, t]` is the merged position of local token t: an unmerged token has a position of its own in [0, U), a merged
// one the position U + j of the anchor row j it merged into -- and several local tokens may have merged into the same
// anchor row, whose attention output would then be computed once per token.  Outputs
//   qc[b, :]   the DISTINCT merged positions: [0 .. U) then U + j for every matched j, ascending; entries past the
//              count repeat position 0 (valid rows for the projection GEMM that nobody reads)
//   tmap[b, t] the row of that compact list local token t reads its attention output from
//   count[b]   number of distinct queries
// Two wide launches around a presence BITMAP of the anchor rows (Nd bits per sample: 4 KB at the cfg-2 top block): the
// first sets the bits; in the second every workgroup loads its sample's whole bitmap into LDS, scans the word popcounts
// (rank(j) = words before j's word + bits below j inside it) and writes its slice of qc / tmap.  (One workgroup per
// sample walking flags and ranks through global memory took 50-95 us; no host round trip either way.)
constexpr int CQ_THREADS = 1024;
constexpr int CQ_MAX_WORDS = 4096;      // Nd <= 131 072 anchors per sample (cfg-5: 90 319)

__global__ __launch_bounds__(256) void compact_mark_kernel(const int32_t *__restrict__ loc, int64_t B, int64_t Ml, int64_t U,
                                                           int64_t words, uint32_t *__restrict__ bits) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * Ml) return;
    const int64_t b = i / Ml;
    const int32_t p = loc[i];
    if (p >= U) atomicOr(&bits[b * words + ((p - U) >> 5)], 1u << ((p - U) & 31));
}

__global__ __launch_bounds__(CQ_THREADS) void compact_queries_kernel(const int32_t *__restrict__ loc, int64_t Ml, int64_t U,
                                                                    int64_t Nd, int64_t words, const uint32_t *__restrict__ bits,
                                                                    int32_t *__restrict__ qc, int32_t *__restrict__ tmap,
                                                                    int32_t *__restrict__ count) {
    __shared__ uint32_t sbits[CQ_MAX_WORDS];
    __shared__ int32_t spre[CQ_MAX_WORDS];        // matched anchor rows in the words before this one
    __shared__ int32_t wave_sum[CQ_THREADS / 64];
    __shared__ int32_t total_s;
    const int64_t b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t *bb = bits + b * words;
    for (int64_t w = tid; w < words; w += CQ_THREADS) sbits[w] = bb[w];
    __syncthreads();
    // exclusive prefix sum of the word popcounts; thread i owns the contiguous words [i * per, (i + 1) * per)
    const int per = (int)((words + CQ_THREADS - 1) / CQ_THREADS);
    const int64_t w0 = (int64_t)tid * per, w1 = w0 + per < words ? w0 + per : words;
    int32_t mine = 0;
    for (int64_t w = w0; w < w1; ++w) mine += __popc(sbits[w]);
    int32_t incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int32_t v = __shfl_up(incl, off, 64);
        if (lane >= off) incl += v;
    }
    if (lane == 63) wave_sum[wave] = incl;
    __syncthreads();
    if (tid == 0) {
        int32_t run = 0;
        for (int w = 0; w < CQ_THREADS / 64; ++w) {
            const int32_t v = wave_sum[w];
            wave_sum[w] = run;
            run += v;
        }
        total_s = run;
    }
    __syncthreads();
    int32_t run = wave_sum[wave] + incl - mine;
    for (int64_t w = w0; w < w1; ++w) {
        spre[w] = run;
        run += __popc(sbits[w]);
    }
    __syncthreads();
    const int64_t cnt = U + total_s;
    if (blockIdx.x == 0 && tid == 0) count[b] = (int32_t)cnt;
    const int32_t *lb = loc + b * Ml;
    int32_t *qb = qc + b * Ml, *tb = tmap + b * Ml;
    const int64_t gsz = (int64_t)gridDim.x * CQ_THREADS, g0 = (int64_t)blockIdx.x * CQ_THREADS + tid;
    auto rank_of = [&](int64_t j) { return spre[j >> 5] + __popc(sbits[j >> 5] & ((1u << (j & 31)) - 1u)); };
    for (int64_t j = g0; j < Nd; j += gsz)                      // matched anchor rows -> their slot of the compact list
        if (sbits[j >> 5] >> (j & 31) & 1u) qb[U + rank_of(j)] = (int32_t)(U + j);
    for (int64_t t = g0; t < Ml; t += gsz) {
        if (t < U) qb[t] = (int32_t)t;
        else if (t >= cnt) qb[t] = 0;
        const int32_t p = lb[t];
        tb[t] = p < U ? p : (int32_t)(U + rank_of(p - U));
    }
}


// ---- duplicate keys of the merged sequence (the anchors' exact copies) -> one key each + a multiplicity ----
// patch.py:80 stores u(merged) as the next chunk's anchors: every local token that merged into an anchor row carries that
// row's content, so the anchor set holds groups of IDENTICAL rows (cid = a content id per anchor row, equal ids <=> equal
// rows; the host gets it for free: vtm_compact_queries' tmap).  Identical key rows give identical scores and identical
// value rows: m copies weigh exactly like one key whose score carries + log2(m) (base-2 softmax).  Three launches behind one
// memset: mark (first merged position and number of present copies per content id), keep (bitmap of the surviving
// positions), compact (LDS prefix sums of the bitmap words like compact_queries_kernel).
__global__ __launch_bounds__(256) void fold_mark_kernel(const int32_t *__restrict__ cur, int64_t B, int64_t M, int64_t L,
                                                        const int32_t *__restrict__ cid, int64_t Ma, int64_t n_ids,
                                                        uint32_t *__restrict__ firstinv, uint32_t *__restrict__ cnt) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * M) return;
    const int64_t b = i / M, m = i % M;
    const int64_t id = cur[i];
    if (id < L) return;
    const int64_t c = cid[b * Ma + (id - L)];
    if (c < 0 || c >= n_ids) return;                                      // "no id": a row of its own
    atomicMax(&firstinv[b * n_ids + c], 0xffffffffu - (uint32_t)m);      // zero-initialised: the smallest m wins
    atomicAdd(&cnt[b * n_ids + c], 1u);
}

__global__ __launch_bounds__(256) void fold_keep_kernel(const int32_t *__restrict__ cur, int64_t M, int64_t L,
                                                        const int32_t *__restrict__ cid, int64_t Ma, int64_t n_ids,
                                                        const uint32_t *__restrict__ firstinv, int64_t words,
                                                        uint32_t *__restrict__ bits) {
    const int64_t b = blockIdx.y, m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // a wave = 64 consecutive positions
    bool keep = false;
    if (m < M) {
        const int64_t id = cur[b * M + m];
        keep = id < L;
        if (!keep) {
            const int64_t c = cid[b * Ma + (id - L)];
            keep = c < 0 || c >= n_ids || firstinv[b * n_ids + c] == 0xffffffffu - (uint32_t)m;
        }
    }
    const unsigned long long bal = __ballot(keep);
    const int lane = threadIdx.x & 63;
    if ((lane & 31) == 0 && (m >> 5) < words) bits[b * words + (m >> 5)] = (uint32_t)(bal >> (lane & 32));
}

__device__ __forceinline__ uint32_t split16(float v, int dtype) {     // v = hi + lo in the 16-bit type of the keys
    if (dtype == VTM_F16) {
        const _Float16 h = (_Float16)v, l = (_Float16)(v - (float)h);
        return (uint32_t)__builtin_bit_cast(unsigned short, h) | ((uint32_t)__builtin_bit_cast(unsigned short, l) << 16);
    }
    const __bf16 h = (__bf16)v, l = (__bf16)(v - (float)h);
    return (uint32_t)__builtin_bit_cast(unsigned short, h) | ((uint32_t)__builtin_bit_cast(unsigned short, l) << 16);
}

__global__ __launch_bounds__(CQ_THREADS) void fold_compact_kernel(const int32_t *__restrict__ cur, int64_t M, int64_t L,
                                                                 const int32_t *__restrict__ cid, int64_t Ma, int64_t n_ids,
                                                                 const uint32_t *__restrict__ cnt, int64_t words,
                                                                 const uint32_t *__restrict__ bits, int dtype,
                                                                 int32_t *__restrict__ key_sel, uint32_t *__restrict__ k_bias,
                                                                 int64_t ldkb, int32_t *__restrict__ k_count) {
    __shared__ uint32_t sbits[CQ_MAX_WORDS];
    __shared__ int32_t spre[CQ_MAX_WORDS];
    __shared__ int32_t wave_sum[CQ_THREADS / 64];
    __shared__ int32_t total_s;
    const int64_t b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t *bb = bits + b * words;
    for (int64_t w = tid; w < words; w += CQ_THREADS) sbits[w] = bb[w];
    __syncthreads();
    const int per = (int)((words + CQ_THREADS - 1) / CQ_THREADS);
    const int64_t w0 = (int64_t)tid * per, w1 = w0 + per < words ? w0 + per : words;
    int32_t mine = 0;
    for (int64_t w = w0; w < w1; ++w) mine += __popc(sbits[w]);
    int32_t incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int32_t v = __shfl_up(incl, off, 64);
        if (lane >= off) incl += v;
    }
    if (lane == 63) wave_sum[wave] = incl;
    __syncthreads();
    if (tid == 0) {
        int32_t run = 0;
        for (int w = 0; w < CQ_THREADS / 64; ++w) {
            const int32_t v = wave_sum[w];
            wave_sum[w] = run;
            run += v;
        }
        total_s = run;
    }
    __syncthreads();
    int32_t run = wave_sum[wave] + incl - mine;
    for (int64_t w = w0; w < w1; ++w) {
        spre[w] = run;
        run += __popc(sbits[w]);
    }
    __syncthreads();
    const int64_t kept = total_s;
    if (blockIdx.x == 0 && tid == 0) k_count[b] = (int32_t)kept;
    const int64_t gsz = (int64_t)gridDim.x * CQ_THREADS, g0 = (int64_t)blockIdx.x * CQ_THREADS + tid;
    for (int64_t m = g0; m < M; m += gsz) {
        if (sbits[m >> 5] >> (m & 31) & 1u) {
            const int64_t j = spre[m >> 5] + __popc(sbits[m >> 5] & ((1u << (m & 31)) - 1u));
            const int64_t id = cur[b * M + m];
            key_sel[b * M + j] = (int32_t)m;
            uint32_t bias = 0u;                                           // log2(1)
            if (id >= L) {
                const int64_t ci = cid[b * Ma + (id - L)];
                const uint32_t c = (ci < 0 || ci >= n_ids) ? 1u : cnt[b * n_ids + ci];
                if (c > 1u) bias = split16(log2f((float)c), dtype);
            }
            k_bias[b * ldkb + j] = bias;
        }
        if (m >= kept) {                                                  // past the count: valid rows nobody reads
            key_sel[b * M + m] = 0;
            k_bias[b * ldkb + m] = 0u;
        }
    }
}

}  // namespace

VTM_EXPORT size_t vtm_compact_queries_ws_bytes(int64_t B, int64_t Nd) {
    return (size_t)(B > 0 && Nd > 0 ? B * vtm::cdiv(Nd, 32) * 4 : 0);
}

VTM_EXPORT int vtm_compact_queries(const int32_t *loc, int64_t B, int64_t Ml, int64_t U, int64_t Nd, void *ws,
                                   size_t ws_bytes, int32_t *qc, int32_t *tmap, int32_t *count, vtm_stream_t stream) {
    VTM_REQUIRE(loc && ws && qc && tmap && count, "vtm_compact_queries: null pointer");
    VTM_REQUIRE(B > 0 && Ml > 0 && U >= 0 && U <= Ml && Nd > 0, "vtm_compact_queries: bad sizes");
    VTM_REQUIRE(U + Nd < (1ll << 31), "vtm_compact_queries: index space overflow");
    if (ws_bytes < vtm_compact_queries_ws_bytes(B, Nd))
        return vtm::fail(VTM_EWORKSPACE, "vtm_compact_queries: workspace %zu < %zu bytes", ws_bytes,
                         vtm_compact_queries_ws_bytes(B, Nd));
    const int64_t words = vtm::cdiv(Nd, 32);
    VTM_REQUIRE(words <= CQ_MAX_WORDS, "vtm_compact_queries: more than %d anchor rows per sample", CQ_MAX_WORDS * 32);
    hipStream_t s = vtm::as_stream(stream);
    uint32_t *bits = static_cast<uint32_t *>(ws);
    const hipError_t e = hipMemsetAsync(bits, 0, (size_t)(B * words * 4), s);
    if (e != hipSuccess) return vtm::fail(VTM_ELAUNCH, "vtm_compact_queries: memset: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(compact_mark_kernel, dim3((unsigned)vtm::cdiv(B * Ml, 256)), dim3(256), 0, s, loc, B, Ml, U, words, bits);
    const unsigned per_sample = (unsigned)std::min<int64_t>(vtm::cdiv(std::max(Ml, Nd), CQ_THREADS * 2), 64);
    hipLaunchKernelGGL(compact_queries_kernel, dim3(per_sample, (unsigned)B), dim3(CQ_THREADS), 0, s, loc, Ml, U, Nd, words,
                       (const uint32_t *)bits, qc, tmap, count);
    return vtm::launch_status("vtm_compact_queries");
}

VTM_EXPORT size_t vtm_fold_keys_ws_bytes(int64_t B, int64_t M, int64_t n_ids) {
    return (size_t)(B > 0 && M > 0 && n_ids > 0 ? B * (2 * n_ids + vtm::cdiv(M, 32)) * 4 : 0);
}

VTM_EXPORT int vtm_fold_keys(const int32_t *cur, int64_t B, int64_t M, int64_t L, const int32_t *cid, int64_t Ma,
                             int64_t n_ids, int dtype, void *ws, size_t ws_bytes, int32_t *key_sel, uint32_t *k_bias,
                             int64_t ldkb, int32_t *k_count, vtm_stream_t stream) {
    VTM_REQUIRE(cur && cid && ws && key_sel && k_bias && k_count, "vtm_fold_keys: null pointer");
    VTM_REQUIRE(B > 0 && M > 0 && L >= 0 && Ma > 0 && n_ids > 0 && ldkb >= M, "vtm_fold_keys: bad sizes");
    VTM_REQUIRE(dtype == VTM_F16 || dtype == VTM_BF16, "vtm_fold_keys: the keys are fp16 or bf16");
    if (ws_bytes < vtm_fold_keys_ws_bytes(B, M, n_ids))
        return vtm::fail(VTM_EWORKSPACE, "vtm_fold_keys: workspace %zu < %zu bytes", ws_bytes, vtm_fold_keys_ws_bytes(B, M, n_ids));
    const int64_t words = vtm::cdiv(M, 32);
    VTM_REQUIRE(words <= CQ_MAX_WORDS, "vtm_fold_keys: more than %d merged rows per sample", CQ_MAX_WORDS * 32);
    hipStream_t s = vtm::as_stream(stream);
    uint32_t *firstinv = static_cast<uint32_t *>(ws), *cnt = firstinv + B * n_ids, *bits = cnt + B * n_ids;
    const hipError_t e = hipMemsetAsync(ws, 0, vtm_fold_keys_ws_bytes(B, M, n_ids), s);
    if (e != hipSuccess) return vtm::fail(VTM_ELAUNCH, "vtm_fold_keys: memset: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(fold_mark_kernel, dim3(blocks_for(B * M)), dim3(256), 0, s, cur, B, M, L, cid, Ma, n_ids, firstinv, cnt);
    hipLaunchKernelGGL(fold_keep_kernel, dim3(blocks_for(M), (unsigned)B), dim3(256), 0, s, cur, M, L, cid, Ma, n_ids,
                       (const uint32_t *)firstinv, words, bits);
    const unsigned per_sample = (unsigned)std::min<int64_t>(vtm::cdiv(M, CQ_THREADS * 2), 64);
    hipLaunchKernelGGL(fold_compact_kernel, dim3(per_sample, (unsigned)B), dim3(CQ_THREADS), 0, s, cur, M, L, cid, Ma, n_ids,
                       (const uint32_t *)cnt, words, (const uint32_t *)bits, dtype, key_sel, k_bias, ldkb, k_count);
    return vtm::launch_status("vtm_fold_keys");
}

VTM_EXPORT int vtm_partition_counts(int64_t N_in, int64_t unm_pre, int64_t tnum, int64_t ts,
                                    int64_t randf, int64_t *Ns, int64_t *Nd) {
    VTM_REQUIRE(Ns && Nd && N_in >= unm_pre && unm_pre >= 0 && tnum > 0 && ts > 0 && randf >= 0 &&
                    randf < ts,
                "vtm_partition_counts: bad arguments");
    // frames are q / tnum for q in [0, N_in - unm_pre); the last one may be partial (and, when
    // (N_in - unm_pre) % F != 0, has an index >= F exactly as in the reference: merge.py:59-60)
    const int64_t Q = N_in - unm_pre;
    int64_t nd = 0;
    for (int64_t f = 0; f * tnum < Q; ++f)
        if (f % ts == randf) nd += (Q - f * tnum) < tnum ? (Q - f * tnum) : tnum;
    *Nd = nd + unm_pre;
    *Ns = Q - nd;
    return VTM_OK;
}

VTM_EXPORT int vtm_partition_local(const int32_t *cur, int64_t B, int64_t N_in, int64_t unm_pre,
                                   int64_t tnum, int64_t ts, int64_t randf, int32_t *a_pos,
                                   int32_t *b_pos, int32_t *a_rows, int32_t *b_rows, int64_t Ns,
                                   int64_t Nd, vtm_stream_t stream) {
    VTM_REQUIRE(a_pos && b_pos && a_rows && b_rows, "vtm_partition_local: null pointer");
    int64_t ns = 0, nd = 0;
    if (int rc = vtm_partition_counts(N_in, unm_pre, tnum, ts, randf, &ns, &nd)) return rc;
    VTM_REQUIRE(ns == Ns && nd == Nd, "vtm_partition_local: Ns/Nd (%lld,%lld) != expected (%lld,%lld)",
                (long long)Ns, (long long)Nd, (long long)ns, (long long)nd);
    VTM_REQUIRE(B > 0, "vtm_partition_local: B must be positive");
    hipLaunchKernelGGL(partition_local_kernel, dim3(blocks_for(B * N_in)), dim3(256), 0,
                       vtm::as_stream(stream), cur, B, N_in, unm_pre, tnum, ts, randf, a_pos, b_pos,
                       a_rows, b_rows, Ns, Nd);
    return vtm::launch_status("vtm_partition_local");
}

VTM_EXPORT int vtm_partition_global(const int32_t *cur_local, int64_t B, int64_t Ml, int64_t anchor_base,
                                    int64_t Mg, int local_is_src, int32_t *a_pos, int32_t *b_pos,
                                    int32_t *a_rows, int32_t *b_rows, int32_t *seed_table, int64_t tokens,
                                    const int32_t *anchor_pos, vtm_stream_t stream) {
    VTM_REQUIRE(cur_local && a_pos && b_pos && a_rows && b_rows, "vtm_partition_global: null pointer");
    VTM_REQUIRE(B > 0 && Ml > 0 && Mg > 0, "vtm_partition_global: bad sizes");
    VTM_REQUIRE(seed_table == nullptr || tokens > 0, "vtm_partition_global: seed table without a token count");
    if (seed_table) {     // "no dst row holds this position" = -1
        const hipError_t e = hipMemsetAsync(seed_table, 0xff, (size_t)B * tokens * sizeof(int32_t), vtm::as_stream(stream));
        if (e != hipSuccess) return vtm::fail(VTM_ELAUNCH, "vtm_partition_global: memset: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(partition_global_kernel, dim3(blocks_for(B * (Ml + Mg))), dim3(256), 0,
                       vtm::as_stream(stream), cur_local, B, Ml, anchor_base, Mg, local_is_src, a_pos,
                       b_pos, a_rows, b_rows, seed_table, tokens, anchor_pos);
    return vtm::launch_status("vtm_partition_global");
}

VTM_EXPORT int vtm_anchor_pos(const int32_t *amap, int64_t B, int64_t M, int64_t L, int64_t tokens, const int32_t *old_pos,
                              int64_t Mg, int32_t *out, vtm_stream_t stream) {
    VTM_REQUIRE(out && B > 0 && M > 0 && L >= 0 && tokens > 0, "vtm_anchor_pos: bad arguments");
    hipLaunchKernelGGL(anchor_pos_kernel, dim3(blocks_for(B * M)), dim3(256), 0, vtm::as_stream(stream), amap, B, M, L, tokens,
                       old_pos, Mg, out);
    return vtm::launch_status("vtm_anchor_pos");
}

VTM_EXPORT int vtm_anchor_maps(const int32_t *inv_g, int64_t N_in, int64_t off, const int32_t *new_cur, int64_t M, int64_t B,
                               int64_t Ml, int64_t L, int64_t tokens, const int32_t *old_pos, int64_t Mg, int32_t *loc,
                               int32_t *amap, int32_t *pos, vtm_stream_t stream) {
    VTM_REQUIRE(inv_g && new_cur && loc && amap, "vtm_anchor_maps: null pointer");
    VTM_REQUIRE(B > 0 && Ml > 0 && M > 0 && off >= 0 && off + Ml <= N_in && L >= 0 && (pos == nullptr || tokens > 0),
                "vtm_anchor_maps: bad sizes");
    hipLaunchKernelGGL(anchor_maps_kernel, dim3(blocks_for(B * Ml)), dim3(256), 0, vtm::as_stream(stream), inv_g, N_in, off, new_cur,
                       M, B, Ml, L, tokens, old_pos, Mg, loc, amap, pos);
    return vtm::launch_status("vtm_anchor_maps");
}

VTM_EXPORT int vtm_plan_apply(const uint64_t *best, const int32_t *perm, const int32_t *a_pos,
                              const int32_t *b_pos, const int32_t *a_rows, const int32_t *b_rows,
                              int64_t B, int64_t N_in, int64_t Ns, int64_t Nd, int64_t r, int align,
                              int32_t *new_cur, int32_t *inv, int32_t *unm_idx, int32_t *src_idx,
                              int32_t *dst_idx, vtm_stream_t stream) {
    VTM_REQUIRE(best && perm && a_pos && b_pos && a_rows && b_rows && new_cur && inv,
                "vtm_plan_apply: null pointer");
    VTM_REQUIRE(B > 0 && Ns >= 0 && Nd > 0 && r >= 0 && r <= Ns && N_in == Ns + Nd,
                "vtm_plan_apply: bad sizes");
    hipLaunchKernelGGL(plan_apply_kernel, dim3(blocks_for(B * (Ns + Nd))), dim3(256), 0,
                       vtm::as_stream(stream), best, perm, a_pos, b_pos, a_rows, b_rows, B, N_in, Ns, Nd,
                       r, align, new_cur, inv, unm_idx, src_idx, dst_idx);
    return vtm::launch_status("vtm_plan_apply");
}

VTM_EXPORT int vtm_compose(const int32_t *inv_acc, const int32_t *inv_level, int64_t B, int64_t n,
                           int64_t level_len, int64_t offset, int32_t *out, vtm_stream_t stream) {
    VTM_REQUIRE(inv_level && out && B > 0 && n > 0 && offset >= 0, "vtm_compose: bad arguments");
    hipLaunchKernelGGL(compose_kernel, dim3(blocks_for(B * n)), dim3(256), 0, vtm::as_stream(stream),
                       inv_acc, inv_level, B, n, level_len, offset, out);
    return vtm::launch_status("vtm_compose");
}

VTM_EXPORT int vtm_decode_best(const uint64_t *best, int64_t n, float *node_max, int32_t *node_idx,
                               vtm_stream_t stream) {
    VTM_REQUIRE(best && n >= 0, "vtm_decode_best: bad arguments");
    if (n == 0) return VTM_OK;
    hipLaunchKernelGGL(decode_best_kernel, dim3(blocks_for(n)), dim3(256), 0, vtm::as_stream(stream),
                       best, n, node_max, node_idx);
    return vtm::launch_status("vtm_decode_best");
}

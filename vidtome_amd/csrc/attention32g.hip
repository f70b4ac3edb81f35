// attention32g: self-attention for the head dims WITHOUT a spare contraction slot (d = 64: SD-2.x, cfg-5; d = 80: SD-1.5 mid
// blocks) with the tile loop skewed by one tile (round 6).
//
// attention_kernel runs a tile as QK^T -> maximum / shift -> exps -> PV, one phase after the other: at d = 64 that is 512
// matrix cycles and ~510 VALU cycles in a row, at d = 80 704 and ~430.  Here iteration t issues
//     S(t + 1) = K(t + 1) Q^T   and   O^T += V^T(t - 1) P^T(t - 1)                (matrix pipe; all v_mfma_f32_32x32x16, each of
//                                                                                  which leaves the issue port free for ~15 of
//                                                                                  its 32 cycles)
// beside
//     maximum, shift decision, v_fma + v_exp + pack of tile t                     (VALU, from the S(t) of iteration t - 1)
// in one basic block.  One wave owns 32 queries (8 waves per workgroup, 256 registers, 2 waves per SIMD); K runs two tiles ahead
// and V^T one tile ahead in 3-slot LDS rings, one barrier per tile; the fragments are transient (read right before their MFMAs).
// Same arithmetic per query and tile as attention_kernel: base-2 online softmax with the deferred rescale (shift raised only
// past 2^8), P in registers in the S^T accumulator layout, the denominator through a ones-row of V^T where the head dim leaves
// one (d = 80) or as an fp32 sum of the fp16-rounded P (d = 64).  Query-bounded launches use the device-side geometric plan of
// attention16_parts.h.  Reference: vidtome/patch.py:157-162 = `sa_forward`, utils/pnp_utils.py:47-95.
#include "attention16_parts.h"

#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace {

template <int D> constexpr int rec32() { return (D + 31) / 32 * 16 + 2; }   // accumulators, running max, denominator

template <typename T, int D, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void attention32_combine_kernel(
    const float *__restrict__ partial, T *__restrict__ out, int64_t ldo, int64_t H, int64_t M, int64_t Mp, int64_t nqb,
    int64_t id0, int nsplit, int xcd_groups, const int32_t *__restrict__ q_count, const DevPlan *__restrict__ dev_plan) {
    constexpr int NT = WAVES * 64, QB = WAVES * QW, DV = (D + 31) / 32, NA = DV * 16, REC = rec32<D>();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    int64_t rec0 = (int64_t)blockIdx.x * nsplit;
    int64_t pos = id0 + blockIdx.x;
    if (dev_plan != nullptr) {
        if ((int)blockIdx.x >= dev_plan->split_items) return;
        nqb = dev_plan->nqb;
        xcd_groups = nqb >= 32 ? xcd_groups : 0;
        pos = dev_plan->tier[0].items + blockIdx.x;
        int ti = 1;
        while (ti + 1 < dev_plan->ntiers && pos >= dev_plan->tier[ti + 1].item0) ++ti;
        const DevTier tr = dev_plan->tier[ti];
        nsplit = tr.nsplit;
        rec0 = tr.rec0 + (pos - tr.item0) * tr.nsplit;
    }
    const int64_t lin = item_of(pos, nqb, xcd_groups);
    const int64_t b = lin / (nqb * H), h = (lin / nqb) % H;
    const int64_t q0 = (lin % nqb) * QB + wave * QW;
    if (q_count != nullptr && (lin % nqb) * QB >= (int64_t)q_count[b]) return;
    float m = -INFINITY;
    for (int sp = 0; sp < nsplit; ++sp) m = fmaxf(m, partial[((rec0 + sp) * REC + NA) * NT + tid]);
    float acc[NA], den = 0.0f;
#pragma unroll
    for (int r = 0; r < NA; ++r) acc[r] = 0.0f;
    for (int sp = 0; sp < nsplit; ++sp) {
        const float *pp = partial + (rec0 + sp) * REC * NT + tid;
        const float ms = pp[NA * NT];
        const float w = ms == -INFINITY ? 0.0f : __builtin_amdgcn_exp2f(ms - m);   // (a piece that saw no key)
        den = __builtin_fmaf(pp[(NA + 1) * NT], w, den);
#pragma unroll
        for (int r = 0; r < NA; ++r) acc[r] = __builtin_fmaf(pp[r * NT], w, acc[r]);
    }
    f32x16 o[DV];
#pragma unroll
    for (int r = 0; r < NA; ++r) o[r >> 4][r & 15] = acc[r];
    write_output<T, D>(o, den, out, ldo, b, h, q0, M, Mp, l31, hi);
}

template <typename T, int D, int WAVES>
__global__ __launch_bounds__(WAVES * 64, 2) void attention32g_kernel(
    const T *__restrict__ q, int64_t ldq, const T *__restrict__ k, int64_t ldk,
    const T *__restrict__ vt, int64_t ldvt, T *__restrict__ out, int64_t ldo, int64_t H,
    int64_t M, int64_t Mp, int64_t Mk, int64_t Mkp, float scale_log2e, int64_t nqb, int64_t nwhole,
    int nsplit_tail, float *__restrict__ partial_base, int xcd_groups, const int32_t *__restrict__ q_count,
    int64_t split_major_items, const DevPlan *__restrict__ dev_plan) {
    using F = Frag<T>;
    using vec = typename F::vec;
    using elem = typename F::elem;
    static_assert(D % 16 == 0, "head dims with a spare contraction slot belong to attention16*.hip");
    constexpr int KR = 3, VR = 3;
    constexpr int NT = WAVES * 64, QB = WAVES * QW;
    constexpr int DK = D / 16, DV = (D + 31) / 32, VROWS = DV * 32;
    constexpr bool SPARE = (D % 32) != 0;     // O^T row D is free -> the denominator through the MFMA
    constexpr int K_STRIDE = DK * 16 + 8;
    constexpr int DCH = D / 8;
    constexpr int K_CHUNKS = KV * DCH, V_CHUNKS = D * (KV / 8);
    constexpr int K_PER_T = (K_CHUNKS + NT - 1) / NT, V_PER_T = (V_CHUNKS + NT - 1) / NT;
    constexpr int SK_TILE = KV * K_STRIDE, SV_TILE = VROWS * VT_STRIDE;
    constexpr int REC = rec32<D>(), NA = DV * 16;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    elem *sK = reinterpret_cast<elem *>(smem);   // [KR][KV][K_STRIDE]
    elem *sV = sK + KR * SK_TILE;                // [VR][VROWS][VT_STRIDE]

    int64_t tier_item0 = nwhole, tier_wg0 = nwhole, tier_rec0 = 0;
    if (dev_plan != nullptr) {        // query-bounded launch: the roles come from attention16_plan_kernel
        nqb = dev_plan->nqb;
        xcd_groups = nqb >= 32 ? xcd_groups : 0;
        int ti = 0;
        while (ti + 1 < dev_plan->ntiers && (int)blockIdx.x >= dev_plan->tier[ti + 1].wg0) ++ti;
        const DevTier tr = dev_plan->tier[ti];
        if ((int64_t)blockIdx.x >= (int64_t)tr.wg0 + (int64_t)tr.items * tr.nsplit) return;
        nwhole = dev_plan->tier[0].items;
        nsplit_tail = tr.nsplit;
        split_major_items = tr.items;
        tier_item0 = tr.item0;
        tier_wg0 = tr.wg0;
        tier_rec0 = tr.rec0;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const bool tail_wg = (int64_t)blockIdx.x >= nwhole;
    const int64_t tail_id = (int64_t)blockIdx.x - tier_wg0;
    const int nsplit = tail_wg ? nsplit_tail : 1;
    const int64_t tail_item = split_major_items ? tail_id % split_major_items : tail_id / nsplit;
    const int split = !tail_wg ? 0 : split_major_items ? (int)(tail_id / split_major_items) : (int)(tail_id % nsplit);
    const int64_t lin = item_of(tail_wg ? tier_item0 + tail_item : (int64_t)blockIdx.x, nqb, xcd_groups);
    float *partial = tail_wg ? partial_base + (tier_rec0 + tail_item * nsplit + split) * REC * NT : nullptr;
    const int64_t b = lin / (nqb * H), h = (lin / nqb) % H;
    const int64_t qblock0 = (lin % nqb) * QB;
    const int64_t q0 = qblock0 + wave * QW;
    const int64_t C = H * D;
    if (q_count != nullptr && qblock0 >= (int64_t)q_count[b]) return;

    // one-time LDS init: K pad columns = 0 (they meet Q's zero padding), V^T pad rows = 0 except row D = 1 (the denominator
    // row); V^T ring slot 2 is read by the first iteration's all-zero deferred PV before any tile was written there
    for (int i = tid; i < KR * KV * (K_STRIDE - D); i += NT) {
        const int row = i / (K_STRIDE - D), c = D + i % (K_STRIDE - D);
        sK[row * K_STRIDE + c] = (elem)0.0f;
    }
    if constexpr (VROWS > D) {
        for (int i = tid; i < VR * (VROWS - D) * VT_STRIDE; i += NT) {
            const int t1 = i / ((VROWS - D) * VT_STRIDE), rem = i % ((VROWS - D) * VT_STRIDE);
            const int row = D + rem / VT_STRIDE, c = rem % VT_STRIDE;
            sV[t1 * SV_TILE + row * VT_STRIDE + c] = (elem)((row == D) ? 1.0f : 0.0f);
        }
    }
    for (int i = tid; i < D * VT_STRIDE; i += NT) sV[2 * SV_TILE + i] = (elem)0.0f;

    vec qf[DK];
    {
        const int64_t qi = q0 + l31;
        const T *qp = q + (b * Mp + (qi < M ? qi : 0)) * ldq + h * D;
#pragma unroll
        for (int ks = 0; ks < DK; ++ks) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (qi < M) v = *reinterpret_cast<const uint4 *>(qp + ks * 16 + hi * 8);
            qf[ks] = *reinterpret_cast<vec *>(&v);
        }
    }

    uint32_t kgo[K_PER_T], vgo[V_PER_T];
    int koff[K_PER_T], voff[V_PER_T], krow[K_PER_T], vkey[V_PER_T];
    bool kok[K_PER_T], vok[V_PER_T];
#pragma unroll
    for (int i = 0; i < K_PER_T; ++i) {
        const int c = tid + i * NT;
        kok[i] = c < K_CHUNKS;
        krow[i] = c / DCH;
        kgo[i] = kok[i] ? (uint32_t)(krow[i] * (int)ldk + (c % DCH) * 8) * 2u : 0u;
        koff[i] = krow[i] * K_STRIDE + (c % DCH) * 8;
    }
#pragma unroll
    for (int i = 0; i < V_PER_T; ++i) {
        const int c = tid + i * NT;
        vok[i] = c < V_CHUNKS;
        vkey[i] = (c % (KV / 8)) * 8;
        vgo[i] = vok[i] ? (uint32_t)((c / (KV / 8)) * (int)ldvt + vkey[i]) * 2u : 0u;
        // inside every 16-key group the tile is stored as [k0-3 | k8-11 | k4-7 | k12-15] (see attention_kernel)
        voff[i] = (c / (KV / 8)) * VT_STRIDE + (vkey[i] & ~15) + ((vkey[i] >> 3) & 1) * 4;
    }
    const auto rsrc_k = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(k + b * Mkp * ldk + h * D), 0, 0x7fffffff, 0x00020000);
    const auto rsrc_v = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(vt + (b * C + h * D) * ldvt), 0, 0x7fffffff, 0x00020000);
    const uint32_t kstep = (uint32_t)(KV * ldk) * 2u, vstep = (uint32_t)KV * 2u;
    auto fetch = [](const auto &rsrc, uint32_t voff_, uint32_t soff_) {
        return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff_, soff_, 0));
    };

    uint4 rk[K_PER_T], rv[V_PER_T];
    auto issue_k = [&](auto full_tag, int tk) {
        const uint32_t so = (uint32_t)tk * kstep;
        if constexpr (decltype(full_tag)::value) {
#pragma unroll
            for (int i = 0; i < K_PER_T; ++i) rk[i] = fetch(rsrc_k, kgo[i], so);
        } else {
            const int64_t key0 = (int64_t)tk * KV;
#pragma unroll
            for (int i = 0; i < K_PER_T; ++i) {
                uint4 v = make_uint4(0, 0, 0, 0);
                if (kok[i] && key0 + krow[i] < Mk) v = fetch(rsrc_k, kgo[i], so);
                rk[i] = v;
            }
        }
    };
    auto issue_v = [&](auto full_tag, int tv) {
        const uint32_t so = (uint32_t)tv * vstep;
        if constexpr (decltype(full_tag)::value) {
#pragma unroll
            for (int i = 0; i < V_PER_T; ++i) rv[i] = fetch(rsrc_v, vgo[i], so);
        } else {
            const int64_t key0 = (int64_t)tv * KV;
#pragma unroll
            for (int i = 0; i < V_PER_T; ++i) {
                uint4 v = make_uint4(0, 0, 0, 0);
                const int64_t key = key0 + vkey[i];
                if (vok[i] && key < Mk) {
                    v = fetch(rsrc_v, vgo[i], so);
                    mask_keys(v, (int)(Mk - key));
                }
                rv[i] = v;
            }
        }
    };
    auto write_k = [&](int slot) {
        elem *dk = sK + slot * SK_TILE;
#pragma unroll
        for (int i = 0; i < K_PER_T; ++i)
            if (kok[i]) *reinterpret_cast<uint4 *>(dk + koff[i]) = rk[i];
    };
    auto write_v = [&](int slot) {
        elem *dv = sV + slot * SV_TILE;
#pragma unroll
        for (int i = 0; i < V_PER_T; ++i)
            if (vok[i]) {
                uint2 *dst = reinterpret_cast<uint2 *>(dv + voff[i]);
                dst[0] = make_uint2(rv[i].x, rv[i].y);
                dst[2] = make_uint2(rv[i].z, rv[i].w);
            }
    };

    f32x16 o[DV];   // o[dv][r] = row 32 dv + (r & 3) + 8 (r >> 2) + 4 hi of query l31
#pragma unroll
    for (int dv = 0; dv < DV; ++dv)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dv][r] = 0.0f;
    float m_run = -INFINITY, l_run = 0.0f;

    auto qk = [&](f32x16 (&s)[2], int slot) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const elem *kp = sK + slot * SK_TILE + (kb * 32 + l31) * K_STRIDE + hi * 8;
            vec kf[DK];
#pragma unroll
            for (int ks = 0; ks < DK; ++ks) kf[ks] = *reinterpret_cast<const vec *>(kp + ks * 16);
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kb][r] = 0.0f;
#pragma unroll
            for (int ks = 0; ks < DK; ++ks) s[kb] = F::mfma(kf[ks], qf[ks], s[kb]);
        }
    };
    auto mask_s = [&](f32x16 (&s)[2], int lim) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi >= lim) s[kb][r] = -INFINITY;
    };
    auto pv = [&](int vslot, const vec (&pf)[4]) {   // 4 steps of 16 keys; k-slot (hi, e) <-> key 16 st + 8 (e >> 2) + 4 hi + (e & 3)
#pragma unroll
        for (int dv = 0; dv < DV; ++dv) {
            const elem *vp = sV + vslot * SV_TILE + (dv * 32 + l31) * VT_STRIDE + 8 * hi;
#pragma unroll
            for (int st = 0; st < 4; ++st) o[dv] = F::mfma(*reinterpret_cast<const vec *>(vp + st * 16), pf[st], o[dv]);
        }
    };

    using std::false_type;
    using std::true_type;
    const int ntiles = (int)((Mk + KV - 1) / KV), nfull = (int)(Mk / KV);
    const int tps = (ntiles + nsplit - 1) / nsplit;
    const int tb = split * tps, te = tb + tps < ntiles ? tb + tps : ntiles;
    const int fe = te < nfull ? te : nfull;
    if (tb >= te) {
        if (partial) {
            float *pp_ = partial + tid;
#pragma unroll
            for (int r = 0; r < NA; ++r) pp_[r * NT] = 0.0f;
            pp_[NA * NT] = -INFINITY;
            pp_[(NA + 1) * NT] = 0.0f;
        }
        return;
    }

    if (tb < fe) { issue_k(true_type{}, tb); issue_v(true_type{}, tb); } else { issue_k(false_type{}, tb); issue_v(false_type{}, tb); }
    write_k(0);
    write_v(0);
    if (tb + 1 < te) {
        if (tb + 1 < fe) issue_k(true_type{}, tb + 1); else issue_k(false_type{}, tb + 1);
        write_k(1);
    }
    __syncthreads();

    f32x16 sC[2], sN[2];
    vec pfP[4];               // P^T of the previous tile, waiting for its PV
#pragma unroll
    for (int st = 0; st < 4; ++st)
#pragma unroll
        for (int e = 0; e < 8; ++e) pfP[st][e] = (elem)0.0f;
    auto tile_max = [&](const f32x16 (&s)[2]) {   // the tile's largest scaled score per query (both lane halves)
        float mt = fmaxf(s[0][0], s[1][0]);
#pragma unroll
        for (int r = 1; r < 16; ++r) mt = fmaxf(fmaxf(mt, s[0][r]), s[1][r]);
        return fmaxf(mt, __shfl_xor(mt, 32, 64)) * scale_log2e;
    };
    qk(sC, 0);
    if (tb >= fe) mask_s(sC, (int)(Mk - (int64_t)tb * KV));     // (a range that starts on the ragged tile)
    float mtC = tile_max(sC);

    // iteration t.  The shift decision of tile t comes FIRST (its maximum was taken at the end of iteration t - 1, when S(t)'s
    // MFMAs had long finished), so that everything behind it is ONE basic block: S(t + 1) and PV(t - 1) on the matrix pipe, the
    // v_fma / v_exp / pack of tile t and the maximum of tile t + 1 beside them.  A raised shift rescales O^T, the fp32
    // denominator AND the P^T of tile t - 1 that still waits for its PV (one more fp16 rounding of those 2 048 values, on
    // the rare tiles that raise the shift at all).
    auto iteration = [&](auto fast_tag, int t, int kn, int kw, int vp_, int vw) {
        constexpr bool FAST = decltype(fast_tag)::value;
        if constexpr (FAST) {
            issue_k(true_type{}, t + 2);
            issue_v(true_type{}, t + 1);
        } else {
            if (t + 2 < te) { if (t + 2 < fe) issue_k(true_type{}, t + 2); else issue_k(false_type{}, t + 2); }
            if (t + 1 < te) { if (t + 1 < fe) issue_v(true_type{}, t + 1); else issue_v(false_type{}, t + 1); }
        }
        if (!__all(mtC <= m_run + DEFER_THR)) {
            const float m_new = fmaxf(m_run, mtC);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);   // first tile: exp2(-inf) = 0
            m_run = m_new;
            l_run *= alpha;
#pragma unroll
            for (int dv = 0; dv < DV; ++dv)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[dv][r] *= alpha;
#pragma unroll
            for (int st = 0; st < 4; ++st)
#pragma unroll
                for (int e = 0; e < 8; ++e) pfP[st][e] = (elem)((float)pfP[st][e] * alpha);
        }
        __builtin_amdgcn_sched_barrier(0);
        qk(sN, kn);               // (the last tile of the range computes a stale slot's scores: never used)
        pv(vp_, pfP);             // (first iteration: zeros)
        vec pf[4];
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            float p[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                p[e] = __builtin_amdgcn_exp2f(__builtin_fmaf(sC[st >> 1][8 * (st & 1) + e], scale_log2e, -m_run));
                if constexpr (!SPARE) l_run += p[e];
            }
            F::pack8(pf[st], p);
        }
        if constexpr (!FAST) {
            const int lim_next = (int)(Mk - (int64_t)(t + 1) * KV);
            if (t + 1 < te && lim_next < KV) mask_s(sN, lim_next);
        }
        mtC = tile_max(sN);
#pragma unroll
        for (int st = 0; st < 4; ++st) pfP[st] = pf[st];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) sC[kb] = sN[kb];
        // (the LDS writes are predicated per thread -- a branch.  P^T is only consumed by the NEXT iteration, so left alone the
        // compiler sinks the whole v_fma / v_exp / pack sequence behind that branch, out of the MFMAs' basic block: the empty
        // asm makes the packed values due HERE)
        asm volatile("" : "+v"(pfP[0]), "+v"(pfP[1]), "+v"(pfP[2]), "+v"(pfP[3]));
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (FAST) {
            write_k(kw);
            write_v(vw);
        } else {
            if (t + 2 < te) write_k(kw);
            if (t + 1 < te) write_v(vw);
        }
        __syncthreads();
    };

    int t = tb, kc = 0, kn = 1, kw = 2, vprev = 2, vcur = 0, vw = 1;
    auto rotate = [&]() {
        const int k0 = kc;
        kc = kn;
        kn = kw;
        kw = k0;
        const int v0 = vprev;
        vprev = vcur;
        vcur = vw;
        vw = v0;
    };
    for (; t + 2 < fe; ++t) {
        iteration(true_type{}, t, kn, kw, vprev, vw);
        rotate();
    }
    for (; t < te; ++t) {
        iteration(false_type{}, t, kn, kw, vprev, vw);
        rotate();
    }
    pv(vprev, pfP);           // the last tile's PV

    if (partial) {
        float *pp_ = partial + tid;
#pragma unroll
        for (int r = 0; r < NA; ++r) pp_[r * NT] = o[r >> 4][r & 15];
        pp_[NA * NT] = m_run;
        pp_[(NA + 1) * NT] = l_run;   // (d = 80: 0 -- the ones-row's sum rides in the accumulators)
        return;
    }
    write_output<T, D>(o, l_run, out, ldo, b, h, q0, M, Mp, l31, hi);
}

bool att32_enabled() {
    static const bool on = [] {
        const char *e = getenv("VTM_ATT32");      // A/B hook, read once per process
        return e == nullptr || atoi(e) != 0;
    }();
    return on;
}

template <typename T, int D, int WAVES>
int launch32g(const Args16 &a) {
    constexpr int DK = D / 16, NT = WAVES * 64, QB = WAVES * QW, VROWS = (D + 31) / 32 * 32;
    constexpr size_t lds = (size_t)(3 * KV * (DK * 16 + 8) + 3 * VROWS * VT_STRIDE) * 2;
    if (lds > 64 * 1024) {
        static std::atomic<bool> attr_set[vtm::MAX_DEVICES];
        const int dev = vtm::current_device();
        if (!attr_set[dev].load(std::memory_order_acquire)) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(attention32g_kernel<T, D, WAVES>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return vtm::fail(VTM_ELAUNCH, "vtm_attention: LDS attribute: %s", hipGetErrorString(e));
            attr_set[dev].store(true, std::memory_order_release);
        }
    }
    const size_t rec_bytes = (size_t)rec32<D>() * NT * sizeof(float);
    const float scale_log2e = a.scale * 1.4426950408889634f;
    const int64_t nqb_max = vtm::cdiv(a.M, QB);
    const int xcd_pairs = (a.B * a.h) % 8 == 0 ? (int)(a.B * a.h / 8) : 0;
    const int slots = vtm::device_cus();                 // one 8-wave workgroup per CU (256 registers per wave)
    if (a.q_count != nullptr && a.ws != nullptr && a.ws_bytes >= 256 + (size_t)plan_tail_wgs(slots) * rec_bytes) {
        DevPlan *plan = reinterpret_cast<DevPlan *>(a.ws);
        float *records = reinterpret_cast<float *>(static_cast<char *>(a.ws) + 256);
        hipLaunchKernelGGL(attention16_plan_kernel, dim3(1), dim3(64), 0, a.s, a.q_count, (int)a.B, (int)a.h, QB, slots,
                           (int)vtm::cdiv(a.Mk, KV), plan);
        const int64_t total = nqb_max * a.h * a.B, tail_max = plan_tail_wgs(slots);
        VTM_REQUIRE(total + tail_max < (1ll << 31) / 16, "vtm_attention: grid too large");
        hipLaunchKernelGGL((attention32g_kernel<T, D, WAVES>), dim3((unsigned)(total + tail_max)), dim3(NT), lds, a.s,
                           (const T *)a.q, a.ldq, (const T *)a.k, a.ldk, (const T *)a.vt, a.ldvt, (T *)a.out, a.ldo, a.h, a.M, a.Mp,
                           a.Mk, a.Mkp, scale_log2e, nqb_max, total, 1, records, xcd_pairs, a.q_count, (int64_t)0,
                           (const DevPlan *)plan);
        hipLaunchKernelGGL((attention32_combine_kernel<T, D, WAVES>), dim3((unsigned)plan_split_items(slots)), dim3(NT), 0, a.s,
                           (const float *)records, (T *)a.out, a.ldo, a.h, a.M, a.Mp, nqb_max, total, 1, xcd_pairs, a.q_count,
                           (const DevPlan *)plan);
        return vtm::launch_status("vtm_attention");
    }
    TailPlan p = plan_tail16(a.B, a.h, a.M, a.Mk, QB, 1, rec_bytes, false);
    if (p.nsplit > 1 && (!a.ws || a.ws_bytes < p.ws_bytes)) {
        p.nsplit = 1;
        p.full = p.total;
    }
    VTM_REQUIRE(p.total < (1ll << 31) / 16, "vtm_attention: grid too large");
    const int64_t rem = p.total - p.full;
    const int xcd_groups = p.nqb >= 32 ? xcd_pairs : 0;
    hipLaunchKernelGGL((attention32g_kernel<T, D, WAVES>), dim3((unsigned)(p.full + rem * p.nsplit)), dim3(NT), lds, a.s,
                       (const T *)a.q, a.ldq, (const T *)a.k, a.ldk, (const T *)a.vt, a.ldvt, (T *)a.out, a.ldo, a.h, a.M, a.Mp,
                       a.Mk, a.Mkp, scale_log2e, p.nqb, p.full, p.nsplit, (float *)a.ws, xcd_groups, a.q_count, (int64_t)0,
                       (const DevPlan *)nullptr);
    if (p.nsplit > 1)
        hipLaunchKernelGGL((attention32_combine_kernel<T, D, WAVES>), dim3((unsigned)rem), dim3(NT), 0, a.s, (const float *)a.ws,
                           (T *)a.out, a.ldo, a.h, a.M, a.Mp, p.nqb, p.full, p.nsplit, xcd_groups, a.q_count,
                           (const DevPlan *)nullptr);
    return vtm::launch_status("vtm_attention");
}

}  // namespace

namespace vtm_att {

bool shape32g_for(int64_t d, int share_groups, int64_t Mk) {
    // (short key axes -- cross-attention's 77 keys -- have nothing to skew)
    return att32_enabled() && share_groups == 1 && (d == 64 || d == 80) && Mk >= 4 * KV;
}

size_t ws_bytes32g(int64_t d, int64_t B, int64_t h, int64_t Mq, int64_t Mk, bool bounded) {
    constexpr int WAVES = 8, NT = WAVES * 64;
    const size_t rec = (size_t)(d == 64 ? rec32<64>() : rec32<80>()) * NT * sizeof(float);
    size_t n = plan_tail16(B, h, Mq, Mk, (int64_t)WAVES * QW, 1, rec, false).ws_bytes;
    if (bounded) n = std::max(n, (size_t)256 + (size_t)plan_tail_wgs(vtm::device_cus()) * rec);
    return n;
}

int attention32g(const Args16 &a, int64_t d) {
    if (a.dtype == VTM_F16) return d == 64 ? launch32g<__half, 64, 8>(a) : launch32g<__half, 80, 8>(a);
    if (a.dtype == VTM_BF16) return d == 64 ? launch32g<vtm_bf16, 64, 8>(a) : launch32g<vtm_bf16, 80, 8>(a);
    return vtm::fail(VTM_EINVAL, "vtm_attention: dtype must be VTM_F16 or VTM_BF16");
}

}  // namespace vtm_att

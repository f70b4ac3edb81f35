// attention16g: shared-probability attention (utils/pnp_utils.py:57-67, 75-90) at d = 40 with the probabilities computed ONCE and
// the matrix pipe kept busy by a one-tile SKEW (round 6).
//
// attention16_kernel<NQ = 1, NG> already computes one QK^T and one set of exps per source (query block, head) and NG PV
// accumulations; per 64-key tile and wave that is 192 matrix cycles of QK^T, NG x 192 of PV and ~360 of VALU in a row --
// 1 129 cycles at NG = 3 of which the matrix pipe works 768.  Here the tile loop is skewed by one tile: iteration t issues
//     S(t + 1) = K(t + 1) Q^T   and   PV of tile t - 1 for all NG samples        (matrix pipe, independent of each other)
// beside
//     exps / pack / maximum / swaps of tile t                                     (VALU, from the S(t) of iteration t - 1)
// in ONE basic block, so the softmax hides under 768 matrix cycles instead of following them.  K runs two tiles ahead in a
// 3-slot LDS ring (S(t + 1) reads K(t + 1) while a slower wave may still redo tile t exactly from K(t)), the NG V^T tiles run
// one ahead in a 3-slot ring too (tile t - 1 is read while t waits and t + 1 is written); the V^T fragments are read per
// (sample, k-step) right before their MFMAs -- no fragment survives an iteration -- which keeps the 72 accumulator registers
// of three samples, two score tiles and two packed P inside 256.  Arithmetic per query and tile identical to
// attention16_kernel / attention_kernel's PV16 path.
#include "attention16_parts.h"

#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace {

template <typename T, int D, int NG, int WAVES>
__global__ __launch_bounds__(WAVES * 64, 2) void attention16g_kernel(
    const T *__restrict__ q, int64_t ldq, const T *__restrict__ k, int64_t ldk,
    const T *__restrict__ vt, int64_t ldvt, T *__restrict__ out, int64_t ldo, int64_t H,
    int64_t M, int64_t Mp, int64_t Mk, int64_t Mkp, float scale_log2e, int64_t src_batch, int64_t nqb, int64_t nwhole,
    int nsplit_tail, float *__restrict__ partial_base, int xcd_groups, const int32_t *__restrict__ q_count,
    const DevPlan *__restrict__ dev_plan) {
    using F = Frag<T>;
    using vec = typename F::vec;
    using elem = typename F::elem;
    static_assert(pv16_for(D) && (D % 16) != 0, "the 16-row O^T path: a head dim with a spare k-slot and a spare O^T row");
    constexpr int KR = 3, VR = 3;
    constexpr int NT = WAVES * 64, QB = WAVES * QW, NV = NG;
    constexpr int DK = (D + 15) / 16, DV16 = (D + 16) / 16, VROWS = vrows_for(D);
    constexpr int BIAS_HI = (D % 16) / 8, BIAS_E = D % 8;
    constexpr int K_STRIDE = DK * 16 + 8;
    constexpr int DCH = D / 8;
    constexpr int K_CHUNKS = KV * DCH, V_CHUNKS1 = D * (KV / 8), V_CHUNKS = NG * V_CHUNKS1;
    constexpr int K_PER_T = (K_CHUNKS + NT - 1) / NT, V_PER_T = (V_CHUNKS + NT - 1) / NT;
    constexpr int SK_TILE = KV * K_STRIDE, SV_TILE1 = VROWS * VT_STRIDE, SV_TILE = NG * SV_TILE1;
    constexpr int REC = rec16<D>(), NA = DV16 * 8;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    elem *sK = reinterpret_cast<elem *>(smem);   // [KR][KV][K_STRIDE]
    elem *sV = sK + KR * SK_TILE;                // [VR][NG][VROWS][VT_STRIDE]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int l15 = lane & 15, g16 = lane >> 4;
    // query-bounded launch planned on the device (attention16_plan_kernel, see attention16s_kernel): the roles of the workgroups
    // behind the whole items come from the plan's tiers (wave-uniform loads); inside a tier all first pieces, then all second ...
    int64_t tier_item0 = nwhole, tier_wg0 = nwhole, tier_rec0 = 0, split_major_items = 0;
    if (dev_plan != nullptr) {
        nqb = dev_plan->nqb;
        xcd_groups = nqb >= 32 ? xcd_groups : 0;
        int ti = 0;
        while (ti + 1 < dev_plan->ntiers && (int)blockIdx.x >= dev_plan->tier[ti + 1].wg0) ++ti;
        const DevTier tr = dev_plan->tier[ti];
        if ((int64_t)blockIdx.x >= (int64_t)tr.wg0 + (int64_t)tr.items * tr.nsplit) return;   // behind the last tier
        nwhole = dev_plan->tier[0].items;
        nsplit_tail = tr.nsplit;
        split_major_items = tr.items;
        tier_item0 = tr.item0;
        tier_wg0 = tr.wg0;
        tier_rec0 = tr.rec0;
    }
    const bool tail_wg = (int64_t)blockIdx.x >= nwhole;
    const int64_t tail_id = (int64_t)blockIdx.x - tier_wg0;      // (host plan: one tier behind the whole items)
    const int nsplit = tail_wg ? nsplit_tail : 1;
    const int64_t tail_item = split_major_items ? tail_id % split_major_items : tail_id / nsplit;
    const int split = !tail_wg ? 0 : split_major_items ? (int)(tail_id / split_major_items) : (int)(tail_id % nsplit);
    const int64_t lin = item_of(tail_wg ? tier_item0 + tail_item : (int64_t)blockIdx.x, nqb, xcd_groups);
    float *partial = tail_wg ? partial_base + (tier_rec0 + tail_item * nsplit + split) * NV * REC * NT : nullptr;
    const int64_t b = lin / (nqb * H), h = (lin / nqb) % H;   // b: a SOURCE sample; the groups are the samples b + g * src_batch
    const int64_t q0 = (lin % nqb) * QB + wave * QW;
    const int64_t C = H * D;
    // a device-side query bound (compacted live queries, the same rows in every sample of the group): blocks beyond the source
    // sample's count leave at once, the key-split pieces of such a block too (the combine kernel skips their records)
    if (q_count != nullptr && (lin % nqb) * QB >= (int64_t)q_count[b]) return;

    for (int i = tid; i < KR * KV * (K_STRIDE - D); i += NT) {
        const int row = i / (K_STRIDE - D), c = D + i % (K_STRIDE - D);
        sK[row * K_STRIDE + c] = (elem)(c == D ? 1.0f : 0.0f);
    }
    for (int i = tid; i < VR * NG * (VROWS - D) * VT_STRIDE; i += NT) {
        const int t1 = i / ((VROWS - D) * VT_STRIDE), rem = i % ((VROWS - D) * VT_STRIDE);
        const int row = D + rem / VT_STRIDE, c = rem % VT_STRIDE;
        sV[t1 * SV_TILE1 + row * VT_STRIDE + c] = (elem)((row == D) ? 1.0f : 0.0f);
    }

    // V^T ring slot 2 is read by the first iteration's (all-zero) deferred PV before any tile was written there: 0 x garbage may be NaN
    for (int i = tid; i < NG * D * VT_STRIDE; i += NT) {
        const int g = i / (D * VT_STRIDE), rem = i % (D * VT_STRIDE);
        sV[2 * SV_TILE + g * SV_TILE1 + rem] = (elem)0.0f;
    }

    vec qf[DK];
    {
        const int64_t qi = q0 + l31;
        const T *qp = q + (b * Mp + (qi < M ? qi : 0)) * ldq + h * D;
#pragma unroll
        for (int ks = 0; ks < DK; ++ks) {
            const int d0 = ks * 16 + hi * 8;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (d0 < D && qi < M) v = *reinterpret_cast<const uint4 *>(qp + d0);
            qf[ks] = *reinterpret_cast<vec *>(&v);
#pragma unroll
            for (int e = 0; e < 8; ++e) qf[ks][e] = (elem)((float)qf[ks][e] * scale_log2e);
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
        const int gi = vok[i] ? c / V_CHUNKS1 : 0, cc = vok[i] ? c % V_CHUNKS1 : 0;
        vkey[i] = (cc % (KV / 8)) * 8;
        vgo[i] = vok[i] ? (uint32_t)(((int64_t)gi * src_batch * C + cc / (KV / 8)) * ldvt + vkey[i]) * 2u : 0u;
        voff[i] = gi * SV_TILE1 + (cc / (KV / 8)) * VT_STRIDE + (vkey[i] & ~15) + ((vkey[i] >> 3) & 1) * 4;
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

    f32x4 o16[NG][DV16][2];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int dv = 0; dv < DV16; ++dv)
#pragma unroll
            for (int qh = 0; qh < 2; ++qh)
#pragma unroll
                for (int e = 0; e < 4; ++e) o16[g][dv][qh][e] = 0.0f;
    float m_run = -INFINITY, m_bias = 0.0f;

    auto qk = [&](f32x16 (&s)[2], int slot) {   // S = K(slot) Q^T: the K fragments are transient
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
    auto exps = [&](vec (&pf)[4], const f32x16 (&s)[2]) {
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            float p[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) p[e] = __builtin_amdgcn_exp2f(s[st >> 1][8 * (st & 1) + e]);
            F::pack8(pf[st], p);
        }
    };
    auto ptop_of = [&](const vec (&pf)[4]) {
        uint32_t pw[16];
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            const u32x4 w = __builtin_bit_cast(u32x4, pf[st]);
#pragma unroll
            for (int j = 0; j < 4; ++j) pw[4 * st + j] = w[j];
        }
#pragma unroll
        for (int j = 0; j < 5; ++j) pw[j] = F::pmax3(pw[3 * j], pw[3 * j + 1], pw[3 * j + 2]);
        const uint32_t pr = F::pmax3(F::pmax3(pw[0], pw[1], pw[2]), F::pmax3(pw[3], pw[4], pw[15]), pw[15]);
        return max(pr >> 16, pr & 0xffffu);
    };
    auto swaps = [&](vec (&pp)[2][2], const vec (&pf)[4]) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            u32x4 x = __builtin_bit_cast(u32x4, pf[2 * ks]), y = __builtin_bit_cast(u32x4, pf[2 * ks + 1]);
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const auto r = __builtin_amdgcn_permlane16_swap(x[w], y[w], false, false);
                x[w] = r[0];
                y[w] = r[1];
            }
            pp[ks][0] = __builtin_bit_cast(vec, x);
            pp[ks][1] = __builtin_bit_cast(vec, y);
        }
    };
    auto pv = [&](int vslot, const vec (&pp)[2][2]) {   // O^T of every sample += V^T(vslot) P^T; fragments read right before their MFMAs
        const elem *vp = sV + vslot * SV_TILE + l15 * VT_STRIDE + (g16 & 1) * 16 + (g16 >> 1) * 8;
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                vec a[DV16];
#pragma unroll
                for (int dv = 0; dv < DV16; ++dv)
                    a[dv] = *reinterpret_cast<const vec *>(vp + g * SV_TILE1 + ks * 32 + dv * 16 * VT_STRIDE);
#pragma unroll
                for (int dv = 0; dv < DV16; ++dv) {
                    o16[g][dv][0] = F::mfma16(a[dv], pp[ks][0], o16[g][dv][0]);
                    o16[g][dv][1] = F::mfma16(a[dv], pp[ks][1], o16[g][dv][1]);
                }
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
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                float *pp_ = partial + g * REC * NT + tid;
#pragma unroll
                for (int r = 0; r < NA; ++r) pp_[r * NT] = 0.0f;
                pp_[NA * NT] = pp_[(NA + 1) * NT] = -INFINITY;
                pp_[(NA + 2) * NT] = 0.0f;
            }
        }
        return;
    }

    // prologue: K(tb) -> K slot 0, V^T(tb) -> V slot 0, K(tb + 1) -> K slot 1
    if (tb < fe) { issue_k(true_type{}, tb); issue_v(true_type{}, tb); } else { issue_k(false_type{}, tb); issue_v(false_type{}, tb); }
    write_k(0);
    write_v(0);
    if (tb + 1 < te) {
        if (tb + 1 < fe) issue_k(true_type{}, tb + 1); else issue_k(false_type{}, tb + 1);
        write_k(1);
    }
    __syncthreads();

    f32x16 sC[2], sN[2];      // scores of the tile whose softmax runs / of the next tile
    vec ppP[2][2];            // swapped P^T of the previous tile, waiting for its PV
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int qh = 0; qh < 2; ++qh)
#pragma unroll
            for (int e = 0; e < 8; ++e) ppP[ks][qh][e] = (elem)0.0f;
    qk(sC, 0);

    // iteration t.  kc / kn / kw: K ring slots of tiles t, t + 1 and the slot K(t + 2) goes to; vp / vc / vw: V^T ring slots of
    // tiles t - 1 (read by the deferred PV), t, and the slot V^T(t + 1) goes to
    auto iteration = [&](auto fast_tag, int t, int kc, int kn, int kw, int vp_, int vw) {
        constexpr bool FAST = decltype(fast_tag)::value;
        const int lim = FAST ? KV : (int)(Mk - (int64_t)t * KV);
        if constexpr (FAST) {
            issue_k(true_type{}, t + 2);
            issue_v(true_type{}, t + 1);
        } else {
            if (t + 2 < te) { if (t + 2 < fe) issue_k(true_type{}, t + 2); else issue_k(false_type{}, t + 2); }
            if (t + 1 < te) { if (t + 1 < fe) issue_v(true_type{}, t + 1); else issue_v(false_type{}, t + 1); }
        }
        if constexpr (!FAST) {
            if (lim < KV) mask_s(sC, lim);
        }
        // matrix pipe: S(t + 1) and the PV of tile t - 1 (the first iteration multiplies zeros; the last one computes a stale
        // slot's scores: never read) ...
        qk(sN, kn);
        pv(vp_, ppP);
        // ... beside the softmax of tile t
        vec pf[4];
        exps(pf, sC);
        const uint32_t top = ptop_of(pf);
        if (__any(top > F::BITS_256 || m_run == -INFINITY)) {   // first tile, or scores that outgrew the shift by more than 2^8
            qk(sC, kc);
            if constexpr (!FAST) {
                if (lim < KV) mask_s(sC, lim);
            }
            float mt = fmaxf(sC[0][0], sC[1][0]);
#pragma unroll
            for (int r = 1; r < 16; ++r) mt = fmaxf(fmaxf(mt, sC[0][r]), sC[1][r]);
            mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
            if (!__all(m_bias + mt <= m_run + DEFER_THR)) {
                const float m_new = (float)(elem)(fmaxf(m_run, m_bias + mt));
                const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
                const float delta = m_bias - m_new;
                m_run = m_new;
                m_bias = m_new;
#pragma unroll
                for (int qh = 0; qh < 2; ++qh) {
                    const float a_ = __shfl(alpha, 16 * qh + l15, 64);
#pragma unroll
                    for (int g = 0; g < NG; ++g)
#pragma unroll
                        for (int dv = 0; dv < DV16; ++dv)
#pragma unroll
                            for (int e = 0; e < 4; ++e) o16[g][dv][qh][e] *= a_;
                }
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        sC[kb][r] += delta;   // this tile AND the next were computed with the old shift
                        sN[kb][r] += delta;
                    }
                if (hi == BIAS_HI) qf[DK - 1][BIAS_E] = (elem)(-m_new);
            }
            exps(pf, sC);
        }
        swaps(ppP, pf);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) sC[kb] = sN[kb];
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
        iteration(true_type{}, t, kc, kn, kw, vprev, vw);
        rotate();
    }
    for (; t < te; ++t) {
        iteration(false_type{}, t, kc, kn, kw, vprev, vw);
        rotate();
    }
    pv(vprev, ppP);           // the last tile's PV

    if (partial) {
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            float *pp_ = partial + g * REC * NT + tid;
#pragma unroll
            for (int r = 0; r < NA; ++r) pp_[r * NT] = o16[g][r >> 3][(r >> 2) & 1][r & 3];
#pragma unroll
            for (int qh = 0; qh < 2; ++qh) pp_[(NA + qh) * NT] = __shfl(m_run, 16 * qh + l15, 64);
            pp_[(NA + 2) * NT] = 0.0f;
        }
        return;
    }
#pragma unroll
    for (int g = 0; g < NG; ++g) write_output16<T, D>(o16[g], out, ldo, b + g * src_batch, h, q0, M, Mp, lane);
}

template <typename T, int D, int NG, int WAVES>
int launch16g(const Args16 &a) {
    constexpr int DK = (D + 15) / 16, NT = WAVES * 64, QB = WAVES * QW, NV = NG;
    constexpr size_t lds = (size_t)(3 * KV * (DK * 16 + 8) + 3 * NG * vrows_for(D) * VT_STRIDE) * 2;
    if (lds > 64 * 1024) {
        static std::atomic<bool> attr_set[vtm::MAX_DEVICES];
        const int dev = vtm::current_device();
        if (!attr_set[dev].load(std::memory_order_acquire)) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(attention16g_kernel<T, D, NG, WAVES>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return vtm::fail(VTM_ELAUNCH, "vtm_attention: LDS attribute: %s", hipGetErrorString(e));
            attr_set[dev].store(true, std::memory_order_release);
        }
    }
    const int64_t src_batch = a.B / a.share_groups;
    const size_t rec_bytes = (size_t)NV * rec16<D>() * NT * sizeof(float);
    {
        // query-bounded (vtm_attention_kv_shared_bounded): planned on the device like attention16s_kernel's bounded launches --
        // the count is a device value, a host plan for Mq rows leaves the last round of the LIVE items mostly idle
        // (34 816 rows, 0.78 live: 849 items on 256 slots took 4 rounds for 3.3 rounds of work)
        const int slots = vtm::device_cus();       // one workgroup per CU
        const int64_t nqb_max = vtm::cdiv(a.M, QB);
        const size_t need = devplan_ws_bytes(slots, rec_bytes);
        if (a.q_count != nullptr && a.ws != nullptr && a.ws_bytes >= need && devplan_enabled() && nqb_max * a.h * src_batch >= 2 * slots) {
            DevPlan *plan = reinterpret_cast<DevPlan *>(a.ws);
            float *records = reinterpret_cast<float *>(static_cast<char *>(a.ws) + DEVPLAN_HEADER);
            const int xcd_pairs = (src_batch * a.h) % 8 == 0 ? (int)(src_batch * a.h / 8) : 0;
            const float sl2e = a.scale * 1.4426950408889634f;
            hipLaunchKernelGGL(attention16_plan_kernel, dim3(1), dim3(64), 0, a.s, a.q_count, (int)src_batch, (int)a.h, QB, slots,
                               (int)vtm::cdiv(a.Mk, KV), plan);
            const int64_t total = nqb_max * a.h * src_batch, tail_max = plan_tail_wgs(slots);
            VTM_REQUIRE(total + tail_max < (1ll << 31) / 16, "vtm_attention: grid too large");
            hipLaunchKernelGGL((attention16g_kernel<T, D, NG, WAVES>), dim3((unsigned)(total + tail_max)), dim3(NT), lds, a.s,
                               (const T *)a.q, a.ldq, (const T *)a.k, a.ldk, (const T *)a.vt, a.ldvt, (T *)a.out, a.ldo, a.h, a.M, a.Mp,
                               a.Mk, a.Mkp, sl2e, src_batch, nqb_max, total, 1, records, xcd_pairs, a.q_count, (const DevPlan *)plan);
            hipLaunchKernelGGL((attention16_combine_kernel<T, D, 1, NG, WAVES>), dim3((unsigned)plan_split_items(slots), (unsigned)NV),
                               dim3(NT), 0, a.s, (const float *)records, (T *)a.out, a.ldo, a.h, a.M, a.Mp, nqb_max, total, 1, xcd_pairs,
                               a.q_count, src_batch, (const DevPlan *)plan);
            return vtm::launch_status("vtm_attention");
        }
    }
    TailPlan p = plan_tail16(src_batch, a.h, a.M, a.Mk, QB, 1, rec_bytes, false);
    if (p.nsplit > 1 && (!a.ws || a.ws_bytes < p.ws_bytes)) {
        p.nsplit = 1;
        p.full = p.total;
    }
    const float scale_log2e = a.scale * 1.4426950408889634f;
    VTM_REQUIRE(p.total < (1ll << 31) / 16, "vtm_attention: grid too large");
    const int64_t rem = p.total - p.full;
    const int xcd_groups = ((src_batch * a.h) % 8 == 0 && p.nqb >= 32) ? (int)(src_batch * a.h / 8) : 0;
    hipLaunchKernelGGL((attention16g_kernel<T, D, NG, WAVES>), dim3((unsigned)(p.full + rem * p.nsplit)), dim3(NT), lds, a.s,
                       (const T *)a.q, a.ldq, (const T *)a.k, a.ldk, (const T *)a.vt, a.ldvt, (T *)a.out, a.ldo, a.h, a.M, a.Mp,
                       a.Mk, a.Mkp, scale_log2e, src_batch, p.nqb, p.full, p.nsplit, (float *)a.ws, xcd_groups, a.q_count,
                       (const DevPlan *)nullptr);
    if (p.nsplit > 1)
        hipLaunchKernelGGL((attention16_combine_kernel<T, D, 1, NG, WAVES>), dim3((unsigned)rem, (unsigned)NV), dim3(NT), 0, a.s,
                           (const float *)a.ws, (T *)a.out, a.ldo, a.h, a.M, a.Mp, p.nqb, p.full, p.nsplit, xcd_groups,
                           a.q_count, src_batch, (const DevPlan *)nullptr);
    return vtm::launch_status("vtm_attention");
}

}  // namespace

namespace vtm_att {

size_t ws_bytes16g(int ng, int64_t src_batch, int64_t h, int64_t Mq, int64_t Mk, bool bounded) {
    constexpr int WAVES = 8, NT = WAVES * 64;
    const size_t rec = (size_t)ng * rec16<40>() * NT * sizeof(float);
    size_t n = plan_tail16(src_batch, h, Mq, Mk, (int64_t)WAVES * QW, 1, rec, false).ws_bytes;
    if (bounded && devplan_enabled()) n = std::max(n, devplan_ws_bytes(vtm::device_cus(), rec));
    return n;
}

int attention16g(const Args16 &a, int ng) {
    if (a.dtype == VTM_F16) {
        if (ng == 2) return launch16g<__half, 40, 2, 8>(a);
        if (ng == 3) return launch16g<__half, 40, 3, 8>(a);
    } else if (a.dtype == VTM_BF16) {
        if (ng == 2) return launch16g<vtm_bf16, 40, 2, 8>(a);
        if (ng == 3) return launch16g<vtm_bf16, 40, 3, 8>(a);
    }
    return vtm::fail(VTM_EINVAL, "vtm_attention: no shared-probability instantiation for %d groups", ng);
}

}  // namespace vtm_att

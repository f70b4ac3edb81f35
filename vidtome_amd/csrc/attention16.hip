// attention16: the d = 40 (and FOLDed) form of vtm_attention with a WIDER wave tile (round 6).
//
// Same arithmetic as attention_kernel's PV16 path in attention.hip (read its header first: swapped QK^T, the shift riding in
// a spare k-slot, exps first and the maximum afterwards on the packed P, O^T from 16-row blocks, P^T re-laid with
// v_permlane16_swap) -- what changes is how much one wave carries per K / V^T tile:
//
//   NQ  query sub-tiles of 32 rows per wave (1 or 2).  NQ = 2: the K fragments a wave reads from LDS and the V^T fragments
//       serve 64 queries instead of 32 (half the LDS fragment traffic per MFMA), a wave holds 256 registers (two waves per
//       SIMD instead of four, the same queries in flight per SIMD), and the two sub-tiles are independent instruction
//       streams INSIDE one wave: the exps of one run beside the MFMAs of the other without a third LDS buffer (what
//       tools/ubench/attn_tile_model.hip's "software-pipelined" line measured at 6 %).  VERDICT r05 item 1a.
//   NG  value groups per wave (1 or 3).  NG = 3 is the reference's shared-probability attention done the reference's way
//       (utils/pnp_utils.py:57-67, 75-90: `sim` and `softmax` of the SOURCE sample computed once, the probabilities reused for
//       every one of the `num_inputs` samples): one QK^T, one set of exps, NG PV accumulations against the NG samples' V^T
//       tiles.  attention_kernel's share_groups only redirects the q / k pointers and recomputes everything per sample.
//       VERDICT r05 item 2.
//
// Reference: the `self.attn1(...)` call at vidtome/patch.py:157-162 = `sa_forward`, utils/pnp_utils.py:47-95.
#include "attention16_parts.h"

#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace {

template <typename T, int D, bool FOLD, int NQ, int NG, int WAVES>
__global__ __launch_bounds__(WAVES * 64, (NQ * NG > 1 ? 2 : 4)) void attention16_kernel(
    const T *__restrict__ q, int64_t ldq, const T *__restrict__ k, int64_t ldk,
    const T *__restrict__ vt, int64_t ldvt, T *__restrict__ out, int64_t ldo, int64_t H,
    int64_t M, int64_t Mp, int64_t Mk_arg, int64_t Mkp, float scale_log2e, int64_t src_batch, int64_t nqb, int64_t nwhole,
    int nsplit_tail, float *__restrict__ partial_base, int xcd_groups, const int32_t *__restrict__ q_count,
    int64_t split_major_items, const int32_t *__restrict__ k_count, const uint32_t *__restrict__ k_bias, int64_t ldkb) {
    // Arguments as attention_kernel's.  Work item = (query block of QB rows, head, sample); with NG > 1 the samples of the
    // grid are the SOURCE samples [0, src_batch) and a workgroup writes the rows of samples b + g * src_batch, g < NG.
    using F = Frag<T>;
    using vec = typename F::vec;
    using elem = typename F::elem;
    static_assert(pv16_for(D) && (D % 16) != 0, "the 16-row O^T path: a head dim with a spare k-slot and a spare O^T row");
    static_assert(!FOLD || (D % 8 == 0 && D % 16 == 8), "key folding needs the spare k-slots of a d % 16 == 8 head");
    static_assert(!FOLD || NG == 1, "folded keys belong to one sample's key list");
    constexpr int NT = WAVES * 64, QB = WAVES * QW * NQ, NV = NQ * NG;
    constexpr int DK = (D + 15) / 16, DV16 = (D + 16) / 16, VROWS = vrows_for(D);
    constexpr int BIAS_HI = (D % 16) / 8, BIAS_E = D % 8;   // lane half / fragment element holding channel D
    constexpr int K_STRIDE = DK * 16 + 8;
    constexpr int DCH = D / 8;
    constexpr int K_CHUNKS = KV * DCH, V_CHUNKS1 = D * (KV / 8), V_CHUNKS = NG * V_CHUNKS1;
    constexpr int K_PER_T = (K_CHUNKS + NT - 1) / NT, V_PER_T = (V_CHUNKS + NT - 1) / NT;
    constexpr int SK_TILE = KV * K_STRIDE, SV_TILE1 = VROWS * VT_STRIDE, SV_TILE = NG * SV_TILE1;
    constexpr int REC = rec16<D>(), NA = DV16 * 8;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    elem *sK = reinterpret_cast<elem *>(smem);   // [2][KV][K_STRIDE]
    elem *sV = sK + 2 * SK_TILE;                 // [2][NG][VROWS][VT_STRIDE]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int l15 = lane & 15, g16 = lane >> 4;
    const bool tail_wg = (int64_t)blockIdx.x >= nwhole;
    const int64_t tail_id = (int64_t)blockIdx.x - nwhole;
    const int nsplit = tail_wg ? nsplit_tail : 1;
    const int64_t tail_item = split_major_items ? tail_id % split_major_items : tail_id / nsplit;
    const int split = !tail_wg ? 0 : split_major_items ? (int)(tail_id / split_major_items) : (int)(tail_id % nsplit);
    const int64_t lin = item_of(tail_wg ? nwhole + tail_item : (int64_t)blockIdx.x, nqb, xcd_groups);
    float *partial = tail_wg ? partial_base + (tail_item * nsplit + split) * NV * REC * NT : nullptr;
    const int64_t b = lin / (nqb * H), h = (lin / nqb) % H;
    const int64_t bq = b % src_batch;  // PnP injection: q / k of the source sample (pnp_utils.py:57-67)
    const int64_t qblock0 = (lin % nqb) * QB;
    const int64_t C = H * D;
    int64_t Mk = Mk_arg;
    if constexpr (FOLD) {
        const int64_t kc = k_count[b];
        Mk = kc < Mk_arg ? (kc > 0 ? kc : 1) : Mk_arg;
    }
    if (q_count != nullptr && qblock0 >= (int64_t)q_count[b]) return;

    // one-time LDS init: K pad columns = 0 except column D = 1 (it meets the shift in the query), V^T pad rows = 0 except
    // row D = 1 (the denominator row) -- tile loads never touch these
    for (int i = tid; i < 2 * KV * (K_STRIDE - D); i += NT) {
        const int row = i / (K_STRIDE - D), c = D + i % (K_STRIDE - D);
        sK[row * K_STRIDE + c] = (elem)(c == D ? 1.0f : 0.0f);
    }
    for (int i = tid; i < 2 * NG * (VROWS - D) * VT_STRIDE; i += NT) {
        const int t1 = i / ((VROWS - D) * VT_STRIDE), rem = i % ((VROWS - D) * VT_STRIDE);
        const int row = D + rem / VT_STRIDE, c = rem % VT_STRIDE;
        sV[t1 * SV_TILE1 + row * VT_STRIDE + c] = (elem)((row == D) ? 1.0f : 0.0f);
    }

    // Q fragments (B operand of S^T = K Q^T), pre-scaled: lane (query l31, half hi) holds d = 16 ks + 8 hi + 0..7
    vec qf[NQ][DK];
#pragma unroll
    for (int sub = 0; sub < NQ; ++sub) {
        const int64_t qi = qblock0 + (wave * NQ + sub) * QW + l31;
        const T *qp = q + (bq * Mp + (qi < M ? qi : 0)) * ldq + h * D;
#pragma unroll
        for (int ks = 0; ks < DK; ++ks) {
            const int d0 = ks * 16 + hi * 8;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (d0 < D && qi < M) v = *reinterpret_cast<const uint4 *>(qp + d0);
            qf[sub][ks] = *reinterpret_cast<vec *>(&v);
#pragma unroll
            for (int e = 0; e < 8; ++e) qf[sub][ks][e] = (elem)((float)qf[sub][ks][e] * scale_log2e);
        }
        if constexpr (FOLD) {   // channels D + 2, D + 3 meet the key's bias pair
            if (hi == BIAS_HI) {
                qf[sub][DK - 1][BIAS_E + 2] = (elem)1.0f;
                qf[sub][DK - 1][BIAS_E + 3] = (elem)1.0f;
            }
        }
    }

    // hoisted staging addresses (see attention_kernel): chunk c = tid + NT i; K: (row c / DCH, 16-byte piece c % DCH);
    // V^T: (group c / V_CHUNKS1, channel row, key piece)
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
        // (the value groups are the samples b + gi * src_batch of the SAME head: one descriptor, the sample stride in the offset)
        vgo[i] = vok[i] ? (uint32_t)(((int64_t)gi * src_batch * C + cc / (KV / 8)) * ldvt + vkey[i]) * 2u : 0u;
        voff[i] = gi * SV_TILE1 + (cc / (KV / 8)) * VT_STRIDE + (vkey[i] & ~15) + ((vkey[i] >> 3) & 1) * 4;
    }
    const auto rsrc_k = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(k + bq * Mkp * ldk + h * D), 0, 0x7fffffff, 0x00020000);
    const auto rsrc_v = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(vt + (b * C + h * D) * ldvt), 0, 0x7fffffff, 0x00020000);
    const uint32_t kstep = (uint32_t)(KV * ldk) * 2u, vstep = (uint32_t)KV * 2u;
    uint32_t so_k = 0, so_v = 0;
    const auto rsrc_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(FOLD ? k_bias + b * ldkb : nullptr), 0, 0x7fffffff, 0x00020000);
    uint32_t so_b = 0, rbias = 0;
    [[maybe_unused]] const uint32_t bgo = (uint32_t)(tid & (KV - 1)) * 4u;
    auto fetch = [](const auto &rsrc, uint32_t voff_, uint32_t soff_) {
        return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff_, soff_, 0));
    };

    uint4 rk[K_PER_T], rv[V_PER_T];
    auto issue_full = [&]() {
#pragma unroll
        for (int i = 0; i < K_PER_T; ++i) rk[i] = fetch(rsrc_k, kgo[i], so_k);
#pragma unroll
        for (int i = 0; i < V_PER_T; ++i) rv[i] = fetch(rsrc_v, vgo[i], so_v);
        if constexpr (FOLD) {
            rbias = __builtin_amdgcn_raw_buffer_load_b32(rsrc_b, bgo, so_b, 0);
            so_b += (uint32_t)KV * 4u;
        }
        so_k += kstep;
        so_v += vstep;
    };
    auto issue_tail = [&](int64_t key0) {
#pragma unroll
        for (int i = 0; i < K_PER_T; ++i) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (kok[i] && key0 + krow[i] < Mk) v = fetch(rsrc_k, kgo[i], so_k);
            rk[i] = v;
        }
#pragma unroll
        for (int i = 0; i < V_PER_T; ++i) {
            uint4 v = make_uint4(0, 0, 0, 0);
            const int64_t key = key0 + vkey[i];
            if (vok[i] && key < Mk) {
                v = fetch(rsrc_v, vgo[i], so_v);
                mask_keys(v, (int)(Mk - key));
            }
            rv[i] = v;
        }
        if constexpr (FOLD) {
            rbias = 0u;
            if (key0 + (tid & (KV - 1)) < Mk) rbias = __builtin_amdgcn_raw_buffer_load_b32(rsrc_b, bgo, so_b, 0);
        }
    };
    auto write_lds = [&](int buf) {
        elem *dk = sK + buf * SK_TILE, *dv = sV + buf * SV_TILE;
        if constexpr (FOLD) {
            if (wave == 0) *reinterpret_cast<uint32_t *>(dk + tid * K_STRIDE + D + 2) = rbias;
        }
#pragma unroll
        for (int i = 0; i < K_PER_T; ++i)
            if (kok[i]) *reinterpret_cast<uint4 *>(dk + koff[i]) = rk[i];
#pragma unroll
        for (int i = 0; i < V_PER_T; ++i)
            if (vok[i]) {
                uint2 *dst = reinterpret_cast<uint2 *>(dv + voff[i]);
                dst[0] = make_uint2(rv[i].x, rv[i].y);
                dst[2] = make_uint2(rv[i].z, rv[i].w);
            }
    };

    // o16[sub][g][dv][qh][e] = O^T row 16 dv + 4 g16 + e of query 16 qh + l15 of sub-tile `sub`, value group g
    f32x4 o16[NQ][NG][DV16][2];
#pragma unroll
    for (int sub = 0; sub < NQ; ++sub)
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int dv = 0; dv < DV16; ++dv)
#pragma unroll
                for (int qh = 0; qh < 2; ++qh)
#pragma unroll
                    for (int e = 0; e < 4; ++e) o16[sub][g][dv][qh][e] = 0.0f;
    float m_run[NQ], m_bias[NQ];
#pragma unroll
    for (int sub = 0; sub < NQ; ++sub) {
        m_run[sub] = -INFINITY;
        m_bias[sub] = 0.0f;
    }

    auto tile = [&](auto tail_tag, int buf, int64_t key0) {
        constexpr bool TAIL = decltype(tail_tag)::value;
        // K fragments of the tile's two 32-key blocks: read ONCE for all NQ sub-tiles
        vec kf[2][DK];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const elem *kp = sK + buf * SK_TILE + (kb * 32 + l31) * K_STRIDE + hi * 8;
#pragma unroll
            for (int ks = 0; ks < DK; ++ks) kf[kb][ks] = *reinterpret_cast<const vec *>(kp + ks * 16);
        }
        f32x16 s[NQ][2];
        auto compute_s = [&](int sub) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s[sub][kb][r] = 0.0f;
#pragma unroll
                for (int ks = 0; ks < DK; ++ks) s[sub][kb] = F::mfma(kf[kb][ks], qf[sub][ks], s[sub][kb]);
            }
            if constexpr (TAIL) {   // lane (l31, hi) holds keys key0 + 32 kb + (r & 3) + 8 (r >> 2) + 4 hi
                const int lim = (int)(Mk - key0);
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi >= lim) s[sub][kb][r] = -INFINITY;
            }
        };
        auto rescale = [&](int sub, float alpha) {
#pragma unroll
            for (int qh = 0; qh < 2; ++qh) {
                const float a = __shfl(alpha, 16 * qh + l15, 64);
#pragma unroll
                for (int g = 0; g < NG; ++g)
#pragma unroll
                    for (int dv = 0; dv < DV16; ++dv)
#pragma unroll
                        for (int e = 0; e < 4; ++e) o16[sub][g][dv][qh][e] *= a;
            }
        };
        // raises the shift of sub-tile `sub` when this tile's maximum outgrew it (the exact way: fp32 maximum of the scores)
        auto raise_shift = [&](int sub) {
            float mt = fmaxf(s[sub][0][0], s[sub][1][0]);
#pragma unroll
            for (int r = 1; r < 16; ++r) mt = fmaxf(fmaxf(mt, s[sub][0][r]), s[sub][1][r]);
            mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
            if (!__all(m_bias[sub] + mt <= m_run[sub] + DEFER_THR)) {
                const float m_new = (float)(elem)(fmaxf(m_run[sub], m_bias[sub] + mt));   // fp16-representable
                const float alpha = __builtin_amdgcn_exp2f(m_run[sub] - m_new);            // first tile: exp2(-inf) = 0
                const float delta = m_bias[sub] - m_new;                                   // exact: both are fp16 values
                m_run[sub] = m_new;
                m_bias[sub] = m_new;
                rescale(sub, alpha);
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) s[sub][kb][r] += delta;   // this tile was computed with the old shift
                if (hi == BIAS_HI) qf[sub][DK - 1][BIAS_E] = (elem)(-m_new);
            }
        };
        vec pf[NQ][4];
        auto softmax_step = [&](int sub, int st) {   // p of keys 16 st + (e & 3) + 8 (e >> 2) + 4 hi
            float p[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) p[e] = __builtin_amdgcn_exp2f(s[sub][st >> 1][8 * (st & 1) + e]);
            F::pack8(pf[sub][st], p);
        };
        // exps first, the maximum afterwards on the packed P (see attention.hip): `softmax_check` redoes the sub-tile the
        // exact way on the first tile or when a score outgrew the shift by more than 2^8
        auto softmax_exps = [&](int sub) {
#pragma unroll
            for (int st = 0; st < 4; ++st) softmax_step(sub, st);
        };
        auto softmax_check = [&](int sub) {
            uint32_t pw[16];
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const u32x4 w = __builtin_bit_cast(u32x4, pf[sub][st]);
#pragma unroll
                for (int j = 0; j < 4; ++j) pw[4 * st + j] = w[j];
            }
#pragma unroll
            for (int j = 0; j < 5; ++j) pw[j] = F::pmax3(pw[3 * j], pw[3 * j + 1], pw[3 * j + 2]);   // 16 -> 5 + 1
            const uint32_t pr = F::pmax3(F::pmax3(pw[0], pw[1], pw[2]), F::pmax3(pw[3], pw[4], pw[15]), pw[15]);
            const uint32_t ptop = max(pr >> 16, pr & 0xffffu);
            if (__any(ptop > F::BITS_256 || m_run[sub] == -INFINITY)) {
                compute_s(sub);
                raise_shift(sub);
                softmax_exps(sub);
            }
        };
        // ---- O^T += V^T P^T in 16-row blocks: 2 steps of 32 keys; V^T fragments read ONCE for all NQ sub-tiles
        const elem *vp = sV + buf * SV_TILE + l15 * VT_STRIDE + (g16 & 1) * 16 + (g16 >> 1) * 8;
        vec a[NG][2][DV16];
        auto load_v = [&]() {
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int dv = 0; dv < DV16; ++dv)
                        a[g][ks][dv] = *reinterpret_cast<const vec *>(vp + g * SV_TILE1 + ks * 32 + dv * 16 * VT_STRIDE);
        };
        auto pv = [&](int sub) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                u32x4 x = __builtin_bit_cast(u32x4, pf[sub][2 * ks]), y = __builtin_bit_cast(u32x4, pf[sub][2 * ks + 1]);
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const auto r = __builtin_amdgcn_permlane16_swap(x[w], y[w], false, false);
                    x[w] = r[0];
                    y[w] = r[1];
                }
                const vec p0 = __builtin_bit_cast(vec, x), p1 = __builtin_bit_cast(vec, y);
#pragma unroll
                for (int g = 0; g < NG; ++g)
#pragma unroll
                    for (int dv = 0; dv < DV16; ++dv) {
                        o16[sub][g][dv][0] = F::mfma16(a[g][ks][dv], p0, o16[sub][g][dv][0]);
                        o16[sub][g][dv][1] = F::mfma16(a[g][ks][dv], p1, o16[sub][g][dv][1]);
                    }
            }
        };
        // Program order = the in-wave pipeline: S_0 | S_1 beside the exps of 0 | PV_0 beside the exps of 1 | PV_1 (with two
        // sub-tiles half of the matrix work has independent VALU work of the SAME wave next to it in its basic block)
#pragma unroll
        for (int sub = 0; sub < NQ; ++sub) compute_s(sub);
        softmax_exps(0);
        softmax_check(0);
        load_v();
#pragma unroll
        for (int sub = 1; sub < NQ; ++sub) {
            softmax_exps(sub);
            pv(sub - 1);
            softmax_check(sub);
        }
        pv(NQ - 1);
    };
    using std::false_type;
    using std::true_type;

    const int ntiles = (int)((Mk + KV - 1) / KV), nfull = (int)(Mk / KV);
    const int tps = (ntiles + nsplit - 1) / nsplit;
    const int tb = split * tps, te = tb + tps < ntiles ? tb + tps : ntiles;
    const int fe = te < nfull ? te : nfull;
    so_k = (uint32_t)tb * kstep;
    so_v = (uint32_t)tb * vstep;
    so_b = (uint32_t)tb * (uint32_t)KV * 4u;
    if (tb < fe) issue_full(); else issue_tail((int64_t)tb * KV);
    write_lds(0);
    __syncthreads();

    int t = tb;
    int buf = 0;
    for (; t + 1 < fe; ++t) {
        issue_full();
        tile(false_type{}, buf, (int64_t)t * KV);
        write_lds(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    if (t < fe) {
        const bool ragged_next = te > fe;
        if (ragged_next) issue_tail((int64_t)fe * KV);
        tile(false_type{}, buf, (int64_t)t * KV);
        if (ragged_next) write_lds(buf ^ 1);
        __syncthreads();
        ++t;
        buf ^= 1;
    }
    if (t < te) tile(true_type{}, buf, (int64_t)t * KV);

    if (partial) {   // split workgroup: one PV16 record per (sub-tile, value group)
#pragma unroll
        for (int sub = 0; sub < NQ; ++sub)
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                float *pp = partial + (sub * NG + g) * REC * NT + tid;
#pragma unroll
                for (int r = 0; r < NA; ++r) pp[r * NT] = o16[sub][g][r >> 3][(r >> 2) & 1][r & 3];
#pragma unroll
                for (int qh = 0; qh < 2; ++qh) pp[(NA + qh) * NT] = __shfl(m_run[sub], 16 * qh + l15, 64);
                pp[(NA + 2) * NT] = 0.0f;
            }
        return;
    }
#pragma unroll
    for (int sub = 0; sub < NQ; ++sub)
#pragma unroll
        for (int g = 0; g < NG; ++g)
            write_output16<T, D>(o16[sub][g], out, ldo, b + g * src_batch, h, qblock0 + (wave * NQ + sub) * QW, M, Mp, lane);
}

// ---- the skewed form of NQ = 2, NG = 1 (tools/ubench/attn_tile_model.hip `tile_model2`: 620 cycles per 32-query tile-wave
// against 703 for the plain order -- profiles/r06_c_ubench_skewed_pipeline.txt) ----
// With two sub-tiles A, B per wave EVERY matrix phase gets independent VALU work of the same wave into its basic block:
//   phase 1 of tile t:  S_B(t) = K(t) Q_B^T,  PV of B's tile t-1      beside   exps / pack / maximum of A(t), swaps of B(t-1)
//   phase 2 of tile t:  S_A(t+1) = K(t+1) Q_A^T,  PV of A's tile t    beside   exps / pack / maximum of B(t), swaps of A(t)
// (a 32x32x16 MFMA leaves the SIMD's issue port free for ~15 of its 32 cycles: tools/ubench/mfma_valu_overlap.hip.)  What
// it takes: K two tiles ahead in a ring of THREE LDS slots (S_A(t+1) reads K(t+1) while slower waves still read K(t);
// V^T stays double-buffered), the V^T fragments of tile t kept in registers across the barrier for B's deferred PV, one
// barrier per tile as before.  The rare exact redo of a sub-tile (first tile / scores that outgrew the shift) re-reads
// its K fragments from the ring.  Arithmetic per query and tile identical to attention16_kernel<NQ = 2>.
template <typename T, int D, bool FOLD, int WAVES>
__global__ __launch_bounds__(WAVES * 64, 2) void attention16s_kernel(
    const T *__restrict__ q, int64_t ldq, const T *__restrict__ k, int64_t ldk,
    const T *__restrict__ vt, int64_t ldvt, T *__restrict__ out, int64_t ldo, int64_t H,
    int64_t M, int64_t Mp, int64_t Mk_arg, int64_t Mkp, float scale_log2e, int64_t src_batch, int64_t nqb, int64_t nwhole,
    int nsplit_tail, float *__restrict__ partial_base, int xcd_groups, const int32_t *__restrict__ q_count,
    int64_t split_major_items, const int32_t *__restrict__ k_count, const uint32_t *__restrict__ k_bias, int64_t ldkb,
    const DevPlan *__restrict__ dev_plan) {
    using F = Frag<T>;
    using vec = typename F::vec;
    using elem = typename F::elem;
    static_assert(pv16_for(D) && (D % 16) != 0, "the 16-row O^T path: a head dim with a spare k-slot and a spare O^T row");
    static_assert(!FOLD || (D % 8 == 0 && D % 16 == 8), "key folding needs the spare k-slots of a d % 16 == 8 head");
    constexpr int NQ = 2, KR = 3, VR = 2;
    int64_t tier_item0 = nwhole, tier_wg0 = nwhole, tier_rec0 = 0;
    if (dev_plan != nullptr) {        // query-bounded launch: the roles come from attention16_plan_kernel (wave-uniform loads)
        nqb = dev_plan->nqb;
        xcd_groups = nqb >= 32 ? xcd_groups : 0;
        int ti = 0;
        while (ti + 1 < dev_plan->ntiers && (int)blockIdx.x >= dev_plan->tier[ti + 1].wg0) ++ti;
        const DevTier tr = dev_plan->tier[ti];
        if ((int64_t)blockIdx.x >= (int64_t)tr.wg0 + (int64_t)tr.items * tr.nsplit) return;   // behind the last tier
        nwhole = dev_plan->tier[0].items;
        nsplit_tail = tr.nsplit;
        split_major_items = tr.items;       // (inside a tier: all first pieces, then all second pieces ...)
        tier_item0 = tr.item0;
        tier_wg0 = tr.wg0;
        tier_rec0 = tr.rec0;
    }
    constexpr int NT = WAVES * 64, QB = WAVES * QW * NQ, NV = NQ;
    constexpr int DK = (D + 15) / 16, DV16 = (D + 16) / 16, VROWS = vrows_for(D);
    constexpr int BIAS_HI = (D % 16) / 8, BIAS_E = D % 8;
    constexpr int K_STRIDE = DK * 16 + 8;
    constexpr int DCH = D / 8;
    constexpr int K_CHUNKS = KV * DCH, V_CHUNKS = D * (KV / 8);
    constexpr int K_PER_T = (K_CHUNKS + NT - 1) / NT, V_PER_T = (V_CHUNKS + NT - 1) / NT;
    constexpr int SK_TILE = KV * K_STRIDE, SV_TILE = VROWS * VT_STRIDE;
    constexpr int REC = rec16<D>(), NA = DV16 * 8;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    elem *sK = reinterpret_cast<elem *>(smem);   // [KR][KV][K_STRIDE]
    elem *sV = sK + KR * SK_TILE;                // [VR][VROWS][VT_STRIDE]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int l15 = lane & 15, g16 = lane >> 4;
    const bool tail_wg = (int64_t)blockIdx.x >= nwhole;
    const int64_t tail_id = (int64_t)blockIdx.x - tier_wg0;      // (host plan: one tier behind the whole items)
    const int nsplit = tail_wg ? nsplit_tail : 1;
    const int64_t tail_item = split_major_items ? tail_id % split_major_items : tail_id / nsplit;
    const int split = !tail_wg ? 0 : split_major_items ? (int)(tail_id / split_major_items) : (int)(tail_id % nsplit);
    const int64_t lin = item_of(tail_wg ? tier_item0 + tail_item : (int64_t)blockIdx.x, nqb, xcd_groups);
    float *partial = tail_wg ? partial_base + (tier_rec0 + tail_item * nsplit + split) * NV * REC * NT : nullptr;
    const int64_t b = lin / (nqb * H), h = (lin / nqb) % H;
    const int64_t bq = b % src_batch;
    const int64_t qblock0 = (lin % nqb) * QB;
    const int64_t C = H * D;
    int64_t Mk = Mk_arg;
    if constexpr (FOLD) {
        const int64_t kc = k_count[b];
        Mk = kc < Mk_arg ? (kc > 0 ? kc : 1) : Mk_arg;
    }
    if (q_count != nullptr && qblock0 >= (int64_t)q_count[b]) return;

    for (int i = tid; i < KR * KV * (K_STRIDE - D); i += NT) {
        const int row = i / (K_STRIDE - D), c = D + i % (K_STRIDE - D);
        sK[row * K_STRIDE + c] = (elem)(c == D ? 1.0f : 0.0f);
    }
    for (int i = tid; i < VR * (VROWS - D) * VT_STRIDE; i += NT) {
        const int t1 = i / ((VROWS - D) * VT_STRIDE), rem = i % ((VROWS - D) * VT_STRIDE);
        const int row = D + rem / VT_STRIDE, c = rem % VT_STRIDE;
        sV[t1 * SV_TILE + row * VT_STRIDE + c] = (elem)((row == D) ? 1.0f : 0.0f);
    }

    vec qf[NQ][DK];
#pragma unroll
    for (int sub = 0; sub < NQ; ++sub) {
        const int64_t qi = qblock0 + (wave * NQ + sub) * QW + l31;
        const T *qp = q + (bq * Mp + (qi < M ? qi : 0)) * ldq + h * D;
#pragma unroll
        for (int ks = 0; ks < DK; ++ks) {
            const int d0 = ks * 16 + hi * 8;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (d0 < D && qi < M) v = *reinterpret_cast<const uint4 *>(qp + d0);
            qf[sub][ks] = *reinterpret_cast<vec *>(&v);
#pragma unroll
            for (int e = 0; e < 8; ++e) qf[sub][ks][e] = (elem)((float)qf[sub][ks][e] * scale_log2e);
        }
        if constexpr (FOLD) {
            if (hi == BIAS_HI) {
                qf[sub][DK - 1][BIAS_E + 2] = (elem)1.0f;
                qf[sub][DK - 1][BIAS_E + 3] = (elem)1.0f;
            }
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
        voff[i] = (c / (KV / 8)) * VT_STRIDE + (vkey[i] & ~15) + ((vkey[i] >> 3) & 1) * 4;
    }
    const auto rsrc_k = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(k + bq * Mkp * ldk + h * D), 0, 0x7fffffff, 0x00020000);
    const auto rsrc_v = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(vt + (b * C + h * D) * ldvt), 0, 0x7fffffff, 0x00020000);
    const uint32_t kstep = (uint32_t)(KV * ldk) * 2u, vstep = (uint32_t)KV * 2u;
    const auto rsrc_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(FOLD ? k_bias + b * ldkb : nullptr), 0, 0x7fffffff, 0x00020000);
    uint32_t rbias = 0;
    [[maybe_unused]] const uint32_t bgo = (uint32_t)(tid & (KV - 1)) * 4u;
    auto fetch = [](const auto &rsrc, uint32_t voff_, uint32_t soff_) {
        return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff_, soff_, 0));
    };

    uint4 rk[K_PER_T], rv[V_PER_T];
    // tile `tk` of K (and its bias words) / tile `tv` of V^T into the staging registers; FULL: completely inside [0, Mk)
    auto issue_k = [&](auto full_tag, int tk) {
        const uint32_t so = (uint32_t)tk * kstep;
        if constexpr (decltype(full_tag)::value) {
#pragma unroll
            for (int i = 0; i < K_PER_T; ++i) rk[i] = fetch(rsrc_k, kgo[i], so);
            if constexpr (FOLD) rbias = __builtin_amdgcn_raw_buffer_load_b32(rsrc_b, bgo, (uint32_t)tk * (uint32_t)KV * 4u, 0);
        } else {
            const int64_t key0 = (int64_t)tk * KV;
#pragma unroll
            for (int i = 0; i < K_PER_T; ++i) {
                uint4 v = make_uint4(0, 0, 0, 0);
                if (kok[i] && key0 + krow[i] < Mk) v = fetch(rsrc_k, kgo[i], so);
                rk[i] = v;
            }
            if constexpr (FOLD) {
                rbias = 0u;
                if (key0 + (tid & (KV - 1)) < Mk) rbias = __builtin_amdgcn_raw_buffer_load_b32(rsrc_b, bgo, (uint32_t)tk * (uint32_t)KV * 4u, 0);
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
        if constexpr (FOLD) {
            if (wave == 0) *reinterpret_cast<uint32_t *>(dk + tid * K_STRIDE + D + 2) = rbias;
        }
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

    f32x4 o16[NQ][DV16][2];
#pragma unroll
    for (int sub = 0; sub < NQ; ++sub)
#pragma unroll
        for (int dv = 0; dv < DV16; ++dv)
#pragma unroll
            for (int qh = 0; qh < 2; ++qh)
#pragma unroll
                for (int e = 0; e < 4; ++e) o16[sub][dv][qh][e] = 0.0f;
    float m_run[NQ] = {-INFINITY, -INFINITY}, m_bias[NQ] = {0.0f, 0.0f};

    // ---- pieces of a tile ----
    auto load_kf = [&](vec (&kf)[2][DK], int slot) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const elem *kp = sK + slot * SK_TILE + (kb * 32 + l31) * K_STRIDE + hi * 8;
#pragma unroll
            for (int ks = 0; ks < DK; ++ks) kf[kb][ks] = *reinterpret_cast<const vec *>(kp + ks * 16);
        }
    };
    auto qk = [&](f32x16 (&s)[2], const vec (&kf)[2][DK], int sub) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kb][r] = 0.0f;
#pragma unroll
            for (int ks = 0; ks < DK; ++ks) s[kb] = F::mfma(kf[kb][ks], qf[sub][ks], s[kb]);
        }
    };
    auto mask_s = [&](f32x16 (&s)[2], int lim) {   // keys >= lim of the (ragged) tile; lane (l31, hi) holds keys 32 kb + (r & 3) + 8 (r >> 2) + 4 hi
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi >= lim) s[kb][r] = -INFINITY;
    };
    auto rescale = [&](int sub, float alpha) {
#pragma unroll
        for (int qh = 0; qh < 2; ++qh) {
            const float a_ = __shfl(alpha, 16 * qh + l15, 64);
#pragma unroll
            for (int dv = 0; dv < DV16; ++dv)
#pragma unroll
                for (int e = 0; e < 4; ++e) o16[sub][dv][qh][e] *= a_;
        }
    };
    auto raise_shift = [&](f32x16 (&s)[2], int sub) {
        float mt = fmaxf(s[0][0], s[1][0]);
#pragma unroll
        for (int r = 1; r < 16; ++r) mt = fmaxf(fmaxf(mt, s[0][r]), s[1][r]);
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
        if (!__all(m_bias[sub] + mt <= m_run[sub] + DEFER_THR)) {
            const float m_new = (float)(elem)(fmaxf(m_run[sub], m_bias[sub] + mt));
            const float alpha = __builtin_amdgcn_exp2f(m_run[sub] - m_new);
            const float delta = m_bias[sub] - m_new;
            m_run[sub] = m_new;
            m_bias[sub] = m_new;
            rescale(sub, alpha);
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) s[kb][r] += delta;
            if (hi == BIAS_HI) qf[sub][DK - 1][BIAS_E] = (elem)(-m_new);
        }
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
    // P^T of one sub-tile from the QK^T layout to the B operands of the 16-row PV: pp[ks][query half]
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
    auto pv = [&](int sub, const vec (&a)[2][DV16], const vec (&pp)[2][2]) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int dv = 0; dv < DV16; ++dv) {
                o16[sub][dv][0] = F::mfma16(a[ks][dv], pp[ks][0], o16[sub][dv][0]);
                o16[sub][dv][1] = F::mfma16(a[ks][dv], pp[ks][1], o16[sub][dv][1]);
            }
    };

    using std::false_type;
    using std::true_type;
    const int ntiles = (int)((Mk + KV - 1) / KV), nfull = (int)(Mk / KV);
    const int tps = (ntiles + nsplit - 1) / nsplit;
    const int tb = split * tps, te = tb + tps < ntiles ? tb + tps : ntiles;
    const int fe = te < nfull ? te : nfull;                                 // end of the full tiles of this range
    if (tb >= te) {                                                         // (a split behind the end of a short key axis)
        if (partial) {
#pragma unroll
            for (int sub = 0; sub < NQ; ++sub) {
                float *pp_ = partial + sub * REC * NT + tid;
#pragma unroll
                for (int r = 0; r < NA; ++r) pp_[r * NT] = 0.0f;
                pp_[NA * NT] = pp_[(NA + 1) * NT] = -INFINITY;
                pp_[(NA + 2) * NT] = 0.0f;
            }
        }
        return;
    }

    // prologue: K(tb) -> ring slot 0, V^T(tb) -> slot 0, K(tb + 1) -> ring slot 1
    if (tb < fe) { issue_k(true_type{}, tb); issue_v(true_type{}, tb); } else { issue_k(false_type{}, tb); issue_v(false_type{}, tb); }
    write_k(0);
    write_v(0);
    if (tb + 1 < te) {
        if (tb + 1 < fe) issue_k(true_type{}, tb + 1); else issue_k(false_type{}, tb + 1);
        write_k(1);
    }
    __syncthreads();

    vec kf[2][DK];            // K fragments of the tile whose S is computed next (shared by both sub-tiles)
    vec av[2][DV16];          // V^T fragments of the tile whose PV runs (A: this tile's phase 2, B: the next tile's phase 1)
    vec pB[2][2];             // B's swapped P^T of the previous tile, waiting for its PV
    f32x16 sA[2], sB[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int dv = 0; dv < DV16; ++dv)
#pragma unroll
            for (int e = 0; e < 8; ++e) av[ks][dv][e] = (elem)0.0f;
#pragma unroll
        for (int qh = 0; qh < 2; ++qh)
#pragma unroll
            for (int e = 0; e < 8; ++e) pB[ks][qh][e] = (elem)0.0f;
    }
    load_kf(kf, 0);
    qk(sA, kf, 0);            // S_A(tb)

    // one tile.  FAST: tiles t .. t + 2 are full (no bounds logic, no masks); kc / kn / kw = ring slots of K(t), K(t + 1) and
    // the slot K(t + 2) goes to; vc = slot of V^T(t)
    auto iteration = [&](auto fast_tag, int t, int kc, int kn, int kw, int vc) {
        constexpr bool FAST = decltype(fast_tag)::value;
        const int lim = FAST ? KV : (int)(Mk - (int64_t)t * KV);            // valid keys of tile t (>= KV: all)
        if constexpr (FAST) {
            issue_k(true_type{}, t + 2);
            issue_v(true_type{}, t + 1);
        } else {
            if (t + 2 < te) { if (t + 2 < fe) issue_k(true_type{}, t + 2); else issue_k(false_type{}, t + 2); }
            if (t + 1 < te) { if (t + 1 < fe) issue_v(true_type{}, t + 1); else issue_v(false_type{}, t + 1); }
        }
        // ---- phase 1: S_B(t), PV_B(t - 1)  beside  the softmax of A(t)
        if constexpr (!FAST) {
            if (lim < KV) mask_s(sA, lim);
        }
        vec pfA[4], ppA[2][2];
        qk(sB, kf, 1);
        pv(1, av, pB);
        exps(pfA, sA);
        const uint32_t topA = ptop_of(pfA);
        if (__any(topA > F::BITS_256 || m_run[0] == -INFINITY)) {   // first tile, or scores that outgrew the shift by > 2^8
            qk(sA, kf, 0);
            if constexpr (!FAST) {
                if (lim < KV) mask_s(sA, lim);
            }
            raise_shift(sA, 0);
            exps(pfA, sA);
        }
        // ---- phase 2: S_A(t + 1), PV_A(t)  beside  the softmax of B(t)
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!FAST) {
            if (lim < KV) mask_s(sB, lim);
        }
        load_kf(kf, kn);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int dv = 0; dv < DV16; ++dv)
                av[ks][dv] = *reinterpret_cast<const vec *>(sV + vc * SV_TILE + (l15 + dv * 16) * VT_STRIDE + (g16 & 1) * 16 +
                                                            (g16 >> 1) * 8 + ks * 32);
        vec pfB[4];
        qk(sA, kf, 0);                 // (the last tile of the range computes a stale slot's scores: never read)
        swaps(ppA, pfA);
        pv(0, av, ppA);
        exps(pfB, sB);
        const uint32_t topB = ptop_of(pfB);
        if (__any(topB > F::BITS_256 || m_run[1] == -INFINITY)) {
            vec kfb[2][DK];            // kf holds tile t + 1 by now: tile t's fragments again from its ring slot
            load_kf(kfb, kc);
            qk(sB, kfb, 1);
            if constexpr (!FAST) {
                if (lim < KV) mask_s(sB, lim);
            }
            raise_shift(sB, 1);
            exps(pfB, sB);
        }
        swaps(pB, pfB);
        // ---- the tiles in flight go to LDS: K(t + 2) -> the ring slot K(t - 1) left, V^T(t + 1) -> the other V^T slot
        if constexpr (FAST) {
            write_k(kw);
            write_v(vc ^ 1);
        } else {
            if (t + 2 < te) write_k(kw);
            if (t + 1 < te) write_v(vc ^ 1);
        }
        __syncthreads();
    };

    int t = tb, kc = 0, kn = 1, kw = 2, vc = 0;
    auto rotate = [&]() {
        const int k0 = kc;
        kc = kn;
        kn = kw;
        kw = k0;
        vc ^= 1;
    };
    for (; t + 2 < fe; ++t) {
        iteration(true_type{}, t, kc, kn, kw, vc);
        rotate();
    }
    for (; t < te; ++t) {
        iteration(false_type{}, t, kc, kn, kw, vc);
        rotate();
    }
    pv(1, av, pB);            // B's last tile

    if (partial) {
#pragma unroll
        for (int sub = 0; sub < NQ; ++sub) {
            float *pp_ = partial + sub * REC * NT + tid;
#pragma unroll
            for (int r = 0; r < NA; ++r) pp_[r * NT] = o16[sub][r >> 3][(r >> 2) & 1][r & 3];
#pragma unroll
            for (int qh = 0; qh < 2; ++qh) pp_[(NA + qh) * NT] = __shfl(m_run[sub], 16 * qh + l15, 64);
            pp_[(NA + 2) * NT] = 0.0f;
        }
        return;
    }
#pragma unroll
    for (int sub = 0; sub < NQ; ++sub)
        write_output16<T, D>(o16[sub], out, ldo, b, h, qblock0 + (wave * NQ + sub) * QW, M, Mp, lane);
}


template <typename T, int D, bool FOLD, int NQ, int NG, int WAVES, bool SKEW>
int launch16(const Args16 &a) {
    static_assert(!SKEW || (NQ == 2 && NG == 1), "the skewed pipeline is the two-sub-tile, one-group shape");
    constexpr int DK = (D + 15) / 16, NT = WAVES * 64, QB = WAVES * QW * NQ, NV = NQ * NG;
    constexpr size_t lds = SKEW ? (size_t)(3 * KV * (DK * 16 + 8) + 2 * vrows_for(D) * VT_STRIDE) * 2
                                : (size_t)2 * (KV * (DK * 16 + 8) + NG * vrows_for(D) * VT_STRIDE) * 2;
    static_assert(lds <= 64 * 1024, "dynamic LDS beyond 64 KB needs hipFuncSetAttribute");
    auto kernel = [] {
        if constexpr (SKEW) return attention16s_kernel<T, D, FOLD, WAVES>;
        else return attention16_kernel<T, D, FOLD, NQ, NG, WAVES>;
    }();
    const int64_t src_batch = a.B / a.share_groups;
    const int64_t B_items = NG > 1 ? src_batch : a.B;
    const size_t rec_bytes = (size_t)NV * rec16<D>() * NT * sizeof(float);
    const float scale_log2e = a.scale * 1.4426950408889634f;
    const int wg_cu = wg_per_cu16(NQ, NG, WAVES);
    const int64_t nqb_max = vtm::cdiv(a.M, QB);
    const int xcd_pairs = (B_items * a.h) % 8 == 0 ? (int)(B_items * a.h / 8) : 0;
    if constexpr (SKEW) {
        // query-bounded: planned on the device (attention16_plan_kernel) when the workspace holds the plan and its records
        const int slots = vtm::device_cus() * wg_cu;
        const size_t need = devplan_ws_bytes(slots, rec_bytes);
        if (a.q_count != nullptr && a.ws != nullptr && a.ws_bytes >= need && devplan_enabled() &&
            nqb_max * a.h * B_items >= 2 * slots) {      // (at least two rounds: a plan costs 50-70 us of small launches)
            DevPlan *plan = reinterpret_cast<DevPlan *>(a.ws);
            float *records = reinterpret_cast<float *>(static_cast<char *>(a.ws) + DEVPLAN_HEADER);
            const int ntiles = (int)vtm::cdiv(a.Mk, KV);
            hipLaunchKernelGGL(attention16_plan_kernel, dim3(1), dim3(64), 0, a.s, a.q_count, (int)B_items, (int)a.h, QB, slots,
                               ntiles, plan);
            const int64_t total = nqb_max * a.h * B_items, tail_max = plan_tail_wgs(slots);
            VTM_REQUIRE(total + tail_max < (1ll << 31) / 16, "vtm_attention: grid too large");
            hipLaunchKernelGGL(kernel, dim3((unsigned)(total + tail_max)), dim3(NT), lds, a.s, (const T *)a.q, a.ldq,
                               (const T *)a.k, a.ldk, (const T *)a.vt, a.ldvt, (T *)a.out, a.ldo, a.h, a.M, a.Mp, a.Mk, a.Mkp,
                               scale_log2e, src_batch, nqb_max, total, 1, records, xcd_pairs, a.q_count, (int64_t)0, a.k_count,
                               a.k_bias, a.ldkb, (const DevPlan *)plan);
            hipLaunchKernelGGL((attention16_combine_kernel<T, D, NQ, NG, WAVES>), dim3((unsigned)plan_split_items(slots), (unsigned)NV), dim3(NT),
                               0, a.s, (const float *)records, (T *)a.out, a.ldo, a.h, a.M, a.Mp, nqb_max, total, 1, xcd_pairs,
                               a.q_count, src_batch, (const DevPlan *)plan);
            return vtm::launch_status("vtm_attention");
        }
    }
    TailPlan p = plan_tail16(B_items, a.h, a.M, a.Mk, QB, wg_cu, rec_bytes, a.q_count != nullptr);
    if (p.split_all && (!a.ws || a.ws_bytes < p.ws_bytes)) p = plan_tail16(B_items, a.h, a.M, a.Mk, QB, wg_cu, rec_bytes, false);
    if (p.nsplit > 1 && (!a.ws || a.ws_bytes < p.ws_bytes)) {
        p.nsplit = 1;
        p.full = p.total;
        p.split_all = false;
    }
    VTM_REQUIRE(p.total < (1ll << 31) / 16, "vtm_attention: grid too large");
    const int64_t rem = p.total - p.full;
    const int xcd_groups = p.nqb >= 32 ? xcd_pairs : 0;
    if constexpr (SKEW)
        hipLaunchKernelGGL(kernel, dim3((unsigned)(p.full + rem * p.nsplit)), dim3(NT), lds,
                           a.s, (const T *)a.q, a.ldq, (const T *)a.k, a.ldk, (const T *)a.vt, a.ldvt, (T *)a.out, a.ldo, a.h, a.M,
                           a.Mp, a.Mk, a.Mkp, scale_log2e, src_batch, p.nqb, p.full, p.nsplit, (float *)a.ws, xcd_groups, a.q_count,
                           p.split_all ? rem : (int64_t)0, a.k_count, a.k_bias, a.ldkb, (const DevPlan *)nullptr);
    else
        hipLaunchKernelGGL(kernel, dim3((unsigned)(p.full + rem * p.nsplit)), dim3(NT), lds,
                           a.s, (const T *)a.q, a.ldq, (const T *)a.k, a.ldk, (const T *)a.vt, a.ldvt, (T *)a.out, a.ldo, a.h, a.M,
                           a.Mp, a.Mk, a.Mkp, scale_log2e, src_batch, p.nqb, p.full, p.nsplit, (float *)a.ws, xcd_groups, a.q_count,
                           p.split_all ? rem : (int64_t)0, a.k_count, a.k_bias, a.ldkb);
    if (p.nsplit > 1)
        hipLaunchKernelGGL((attention16_combine_kernel<T, D, NQ, NG, WAVES>), dim3((unsigned)rem, (unsigned)NV), dim3(NT), 0, a.s,
                           (const float *)a.ws, (T *)a.out, a.ldo, a.h, a.M, a.Mp, p.nqb, p.full, p.nsplit, xcd_groups,
                           a.q_count, src_batch, (const DevPlan *)nullptr);
    return vtm::launch_status("vtm_attention");
}

template <typename T>
int dispatch16(const Args16 &a, const Shape16 &sh) {
#define VTM_A16(FOLD_, NQ_, NG_, W_, SK_) \
    if (a.fold == FOLD_ && sh.nq == NQ_ && sh.ng == NG_ && sh.waves == W_ && sh.skew == SK_) \
        return launch16<T, 40, FOLD_, NQ_, NG_, W_, SK_>(a)
    VTM_A16(false, 2, 1, 8, true);
    VTM_A16(true, 2, 1, 8, true);
    VTM_A16(false, 2, 1, 8, false);
    VTM_A16(true, 2, 1, 8, false);
    VTM_A16(false, 1, 3, 8, false);
    VTM_A16(false, 1, 2, 8, false);
#undef VTM_A16
    return vtm::fail(VTM_EINVAL, "vtm_attention: no wide-tile instantiation for nq=%d ng=%d waves=%d skew=%d fold=%d", sh.nq, sh.ng,
                     sh.waves, (int)sh.skew, (int)a.fold);
}

}  // namespace

namespace vtm_att {

int wg_per_cu16(int nq, int ng, int waves) {
    // registers: NQ * NG > 1 -> 256 per wave, two waves per SIMD = 8 per CU; else four per SIMD
    const int waves_per_cu = (nq * ng > 1 ? 2 : 4) * 4;
    const int by_regs = waves_per_cu / waves;
    return by_regs < 1 ? 1 : by_regs;
}

TailPlan plan_tail16(int64_t B_items, int64_t h, int64_t Mq, int64_t Mk, int64_t QB, int wg_per_cu, size_t item_rec_bytes,
                     bool bounded) {
    TailPlan p;
    p.nqb = vtm::cdiv(Mq, QB);
    p.total = p.nqb * h * B_items;
    const int64_t slots = (int64_t)vtm::device_cus() * wg_per_cu;
    p.full = p.total / slots * slots;
    const int64_t rem = p.total - p.full, ntiles = vtm::cdiv(Mk, KV);
    p.nsplit = 1;
    p.ws_bytes = 0;
    p.split_all = false;
    if (bounded && p.total >= 2 * slots && ntiles >= 64 && p.total % 8 == 0) {   // see plan_tail in attention.hip
        p.full = 0;
        p.nsplit = 2;
        p.split_all = true;
        p.ws_bytes = (size_t)p.total * 2 * item_rec_bytes;
        return p;
    }
    if (p.full > 0 && rem > 0 && rem * 4 <= slots && ntiles >= 32) {
        int64_t ns = slots / rem;
        if (ns > 16) ns = 16;
        if (ns > ntiles / 8) ns = ntiles / 8;
        if (ns >= 2) {
            p.nsplit = (int)ns;
            p.ws_bytes = (size_t)rem * ns * item_rec_bytes;
        }
    }
    if (p.nsplit == 1) p.full = p.total;
    return p;
}

size_t ws_bytes16(const Shape16 &sh, int64_t B_items, int64_t h, int64_t Mq, int64_t Mk, bool bounded) {
    const int NT = sh.waves * 64;
    const size_t rec = (size_t)sh.nq * sh.ng * rec16<40>() * NT * sizeof(float);
    const int wg_cu = wg_per_cu16(sh.nq, sh.ng, sh.waves);
    if (sh.ng > 1 && sh.skew) return ws_bytes16g(sh.ng, B_items, h, Mq, Mk, bounded);
    size_t n = plan_tail16(B_items, h, Mq, Mk, (int64_t)sh.waves * QW * sh.nq, wg_cu, rec, bounded).ws_bytes;
    if (bounded && sh.skew && devplan_enabled()) n = std::max(n, devplan_ws_bytes(vtm::device_cus() * wg_cu, rec));
    return n;
}

int attention16(const Args16 &a, const Shape16 &sh) {
    if (sh.ng > 1 && sh.skew) return attention16g(a, sh.ng);
    if (a.dtype == VTM_F16) return dispatch16<__half>(a, sh);
    if (a.dtype == VTM_BF16) return dispatch16<vtm_bf16>(a, sh);
    return vtm::fail(VTM_EINVAL, "vtm_attention: dtype must be VTM_F16 or VTM_BF16");
}

}  // namespace vtm_att

// vtm_attention: flash-style self-attention over the MERGED token sequence (MFMA, fp16/bf16 in,
// fp32 accumulate).  Reference: the `self.attn1(...)` call at vidtome/patch.py:157-162; the arithmetic
// is the one the reference states itself in utils/pnp_utils.py:47-95 (`sa_forward`):
//     sim = einsum("b i d, b j d -> b i j", q, k) * scale;  attn = sim.softmax(-1);  out = attn @ v
// including its injection branch (probabilities of the source sample reused for every batch group).
//
// Structure:
//   * workgroup = 8 waves, each wave owns 32 query rows; K / V^T tiles of 64 keys are staged through
//     LDS (register-staged, double-buffered) and shared by the 8 waves; all per-thread global pointers
//     and LDS offsets are hoisted, full tiles run without any bounds logic, the ragged last tile has its
//     own masked path (sequence lengths are 34 816, 52 224, 8 704, 64 513 ...);
//   * "swapped" QK^T: S^T = K Q^T with v_mfma_f32_32x32x16 puts one query per lane (j = lane & 31) and
//     its 2 x 16 keys of the tile in that lane's accumulators -> online softmax is in-register, with ONE
//     cross-half exchange per tile for the running max;
//   * per score: one v_fma (scale folded, base 2) + one v_exp; the running max is only raised when it grew
//     by more than 2^8 (deferred rescale: P <= 256 is exact enough in fp16 and the O-rescale, which would
//     drag the accumulators through VALU every tile, becomes rare).  Head dims with a spare k-slot (d = 40)
//     carry the shift INSIDE the contraction (BIAS below): no v_fma, and the maximum is only checked
//     afterwards, on the packed P;
//   * P stays in registers: the PV contraction's k-slot <-> key assignment is chosen to be exactly the
//     one the S^T accumulator layout already has (keys 4*hi + {0..3} and 8 + 4*hi + {0..3} of every
//     16-key group), and V^T is read from LDS with the same assignment -- no permute / LDS round trip;
//   * V arrives TRANSPOSED (channel-major) from the projection GEMM, so V^T needs no transpose; when the
//     head dim leaves a spare row in the last row block of O^T (d = 40, 80), that V^T row is set to
//     ones, so the MFMA itself accumulates the softmax denominator from the SAME fp16-rounded P;
//   * d = 40 builds O^T from 16-row blocks (v_mfma_f32_16x16x32: 48 rows instead of 64, a quarter of the
//     PV matrix work gone); P^T moves from the 32-query accumulator layout to the two 16-query B operands
//     with v_permlane16_swap -- 8 swaps per tile, still no LDS round trip (PV16 below).
#include "attention16_parts.h"   // (the device-side launch plan of query-bounded launches: DevPlan, attention16_plan_kernel)

#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace {

// merges the `nsplit` partial states of a query block (same thread <-> register mapping as attention_kernel)
template <typename T, int D>
__global__ __launch_bounds__(waves_for(D) * 64) void attention_combine_kernel(
    const float *__restrict__ partial, T *__restrict__ out, int64_t ldo, int64_t H, int64_t M, int64_t Mp, int64_t nqb,
    int64_t id0, int nsplit, int xcd_groups, const int32_t *__restrict__ q_count, const DevPlan *__restrict__ dev_plan) {
    constexpr int WAVES = waves_for(D), NT = WAVES * 64, QB = WAVES * QW, DV = (D + 31) / 32;
    constexpr bool PV16 = pv16_for(D);
    constexpr int NA = acc_floats(D), NM = max_floats(D), REC = rec_floats(D);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    int64_t rec0 = (int64_t)blockIdx.x * nsplit;   // first partial record of this item
    int64_t pos = id0 + blockIdx.x;
    if (dev_plan != nullptr) {        // device-planned launch (attention16_parts.h): the launch is sized for the most items a plan can split
        if ((int)blockIdx.x >= dev_plan->split_items) return;
        nqb = dev_plan->nqb;
        xcd_groups = nqb >= 64 ? xcd_groups : 0;
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
    if (q_count != nullptr && (lin % nqb) * QB >= (int64_t)q_count[b]) return;   // its partial records were never written
    float acc[NA], m[NM], l = 0.0f;
#pragma unroll
    for (int r = 0; r < NA; ++r) acc[r] = 0.0f;
#pragma unroll
    for (int j = 0; j < NM; ++j) m[j] = -INFINITY;
    for (int sp = 0; sp < nsplit; ++sp) {
        const float *pp = partial + (rec0 + sp) * REC * NT + tid;
        float fa[NM], fb[NM];
#pragma unroll
        for (int j = 0; j < NM; ++j) {
            const float ms = pp[(NA + j) * NT];
            const float mn = fmaxf(m[j], ms);
            fa[j] = __builtin_amdgcn_exp2f(m[j] - mn);   // exp2(-inf) = 0
            fb[j] = __builtin_amdgcn_exp2f(ms - mn);
            m[j] = mn;
        }
        l = l * fa[0] + pp[(NA + NM) * NT] * fb[0];
#pragma unroll
        for (int r = 0; r < NA; ++r) {
            const int j = PV16 ? (r >> 2) & 1 : 0;   // PV16: accumulator (dv, qh, e) is register (dv * 2 + qh) * 4 + e
            acc[r] = acc[r] * fa[j] + pp[r * NT] * fb[j];
        }
    }
    if constexpr (PV16) {
        f32x4 o[(D + 16) / 16][2];
#pragma unroll
        for (int r = 0; r < NA; ++r) o[r >> 3][(r >> 2) & 1][r & 3] = acc[r];
        write_output16<T, D>(o, out, ldo, b, h, q0, M, Mp, lane);
    } else {
        f32x16 o[DV];
#pragma unroll
        for (int r = 0; r < NA; ++r) o[r >> 4][r & 15] = acc[r];
        write_output<T, D>(o, l, out, ldo, b, h, q0, M, Mp, l31, hi);
    }
}

// attention_combine_kernel for FEW items (the split tail of a mid block: 16 items x 16 records leave 240 CUs idle and
// every thread walking 16 x 50 floats): blockIdx.y shares an item's accumulator registers out in groups of 8 (= two
// 4-channel groups of one query; 32-row O^T layout only), every part redoing the maxima / denominator bookkeeping.  Two
// passes -- the maxima first, then every record weighted by exp2(its maximum - the overall one) -- so that no load waits
// for arithmetic on an earlier record; the register group is addressed at run time, the 8 accumulators are static.
template <typename T, int D>
__global__ __launch_bounds__(waves_for(D) * 64) void attention_combine_parts_kernel(
    const float *__restrict__ partial, T *__restrict__ out, int64_t ldo, int64_t H, int64_t M, int64_t Mp, int64_t nqb,
    int64_t id0, int nsplit, int xcd_groups, const int32_t *__restrict__ q_count) {
    using elem = typename Frag<T>::elem;
    static_assert(!pv16_for(D), "32-row O^T layout");
    constexpr int WAVES = waves_for(D), NT = WAVES * 64, QB = WAVES * QW;
    constexpr int NA = acc_floats(D), REC = rec_floats(D);
    constexpr bool SPARE = (D % 32) != 0;
    constexpr int LREG_ALL = SPARE ? (D / 32) * 16 + ((D % 32) & 3) + 4 * ((D % 32) >> 3) : 0;
    constexpr int LHI = ((D % 32) >> 2) & 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int64_t lin = item_of(id0 + blockIdx.x, nqb, xcd_groups);
    const int64_t b = lin / (nqb * H), h = (lin / nqb) % H;
    const int64_t q0 = (lin % nqb) * QB + wave * QW;
    if (q_count != nullptr && (lin % nqb) * QB >= (int64_t)q_count[b]) return;
    const int r0 = (int)blockIdx.y * 8;
    const float *p0 = partial + (int64_t)blockIdx.x * nsplit * REC * NT + tid;
    float m = -INFINITY;
#pragma unroll 4
    for (int sp = 0; sp < nsplit; ++sp) m = fmaxf(m, p0[((int64_t)sp * REC + NA) * NT]);
    float acc[8], den = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.0f;
#pragma unroll 4
    for (int sp = 0; sp < nsplit; ++sp) {
        const float *pp = p0 + (int64_t)sp * REC * NT;
        const float ms = pp[NA * NT];
        const float w = ms == -INFINITY ? 0.0f : __builtin_amdgcn_exp2f(ms - m);   // (a split that saw no key)
        den = __builtin_fmaf(pp[(SPARE ? LREG_ALL : NA + 1) * NT], w, den);       // the O^T denominator row / the fp32 sum
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_fmaf(pp[(r0 + i) * NT], w, acc[i]);
    }
    const float l_tot = SPARE ? __shfl(den, l31 + 32 * LHI, 64) : den + __shfl_xor(den, 32, 64);
    const float inv_l = 1.0f / l_tot;
    const int64_t qi = q0 + l31;
    if (qi < M) {
        T *op = out + (b * Mp + qi) * ldo + h * D;
        const int dv = r0 >> 4, g0 = (r0 & 15) >> 2;
#pragma unroll
        for (int gi = 0; gi < 2; ++gi) {
            const int d0 = dv * 32 + 8 * (g0 + gi) + 4 * hi;
            if (d0 < D) {
                elem w4[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) w4[e] = (elem)(acc[gi * 4 + e] * inv_l);
                *reinterpret_cast<uint2 *>(op + d0) = *reinterpret_cast<uint2 *>(w4);
            }
        }
    }
}

template <typename T, int D, bool FOLD>
__global__ __launch_bounds__(waves_for(D) * 64, (D <= 48 ? 4 : D <= 96 ? 4 : 1)) void attention_kernel(
    const T *__restrict__ q, int64_t ldq, const T *__restrict__ k, int64_t ldk,
    const T *__restrict__ vt, int64_t ldvt, T *__restrict__ out, int64_t ldo, int64_t H,
    int64_t M, int64_t Mp, int64_t Mk_arg, int64_t Mkp, float scale_log2e, int64_t src_batch, int64_t nqb, int64_t nwhole,
    int nsplit_tail, float *__restrict__ partial_base, int xcd_groups, const int32_t *__restrict__ q_count,
    int64_t split_major_items, const int32_t *__restrict__ k_count, const uint32_t *__restrict__ k_bias, int64_t ldkb,
    const DevPlan *__restrict__ dev_plan) {
    // M / Mp: queries per sample and their row stride; Mk / Mkp: keys per sample and the row stride of k
    // (self-attention passes the same values; cross-attention, patch.py:178-183, has Mk = 77).
    // Work decomposition: work item = (query block, head, sample), query blocks fastest.  Workgroups [0, nwhole) take
    // one item each and all its key tiles; the workgroups behind them share the remaining items `nsplit_tail` ways
    // along the key axis: each covers the key tiles of one split and leaves its
    // unnormalised accumulators, running max and denominator in `partial` for attention_combine_kernel (used for
    // the query blocks that do not fill a whole round of the chip, see launch()).
    using F = Frag<T>;
    using vec = typename F::vec;
    using elem = typename F::elem;
    constexpr int WAVES = waves_for(D), NT = WAVES * 64, QB = WAVES * QW;
    constexpr int DK = (D + 15) / 16;      // k-steps of the QK^T contraction
    constexpr int DV = (D + 31) / 32;      // 32-row blocks of O^T
    constexpr bool PV16 = pv16_for(D);     // ... or 16-row blocks (see pv16_for)
    constexpr int DV16 = (D + 16) / 16, VROWS = vrows_for(D);
    constexpr bool SPARE = (D % 32) != 0;  // O^T row D is free -> softmax denominator through the MFMA
    // BIAS: the QK^T contraction has a spare k-slot (D % 16 != 0, e.g. d = 40 -> 48).  Channel D of every K row is
    // set to 1 and channel D of the (pre-scaled) query to -m, so the MFMA itself delivers s * scale * log2(e) - m
    // and the softmax needs no v_fma per score (with 40-wide heads the kernel is VALU-bound).  The shift m only
    // has to be THE SAME for all keys of a query -- it cancels in the normalisation -- so an fp16-representable
    // running max is as good as the exact one; the scale is folded into the query fragment (one extra fp16
    // rounding of q, the same size as the rounding the projection GEMM already applied).
    constexpr bool BIAS = (D % 16) != 0;
    constexpr int BIAS_HI = (D % 16) / 8, BIAS_E = D % 8;   // lane half / fragment element holding channel D
    // FOLD (vtm_attention_kv_folded): the key list is duplicate-free, key j stands for 2^bias_j identical keys of the merged
    // sequence (vtm_fold_keys); the bias rides in two more spare k-slots (channels D + 2, D + 3 of the K row = hi + lo of
    // log2(multiplicity), against ones in the query), and the number of keys is a DEVICE value (k_count[b] <= Mk_arg).
    static_assert(!FOLD || (BIAS && D % 8 == 0 && D % 16 == 8), "key folding needs the spare k-slots of a d % 16 == 8 head");
    constexpr int K_STRIDE = DK * 16 + 8;  // elements; (DK*8+4) words = 4 x odd -> conflict-free b128
    constexpr int DCH = D / 8;             // 16-byte chunks per K row
    constexpr int K_CHUNKS = KV * DCH;     // per tile
    constexpr int V_CHUNKS = D * (KV / 8);
    constexpr int K_PER_T = (K_CHUNKS + NT - 1) / NT;
    constexpr int V_PER_T = (V_CHUNKS + NT - 1) / NT;
    constexpr int SK_TILE = KV * K_STRIDE, SV_TILE = VROWS * VT_STRIDE;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    elem *sK = reinterpret_cast<elem *>(smem);   // [2][KV][K_STRIDE]
    elem *sV = sK + 2 * SK_TILE;                 // [2][VROWS][VT_STRIDE]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int l15 = lane & 15, g16 = lane >> 4;   // PV16 operand coordinates
    // (round 6) query-bounded launches: the roles come from the plan one thread made on the device from the live counts --
    // whole items first, then tiers of items split 2, 4, 8, 16 ways (attention16_parts.h); workgroups behind the plan leave
    int64_t tier_item0 = nwhole, tier_wg0 = nwhole, tier_rec0 = 0;
    if (dev_plan != nullptr) {
        nqb = dev_plan->nqb;
        xcd_groups = nqb >= 64 ? xcd_groups : 0;
        int ti = 0;
        while (ti + 1 < dev_plan->ntiers && (int)blockIdx.x >= dev_plan->tier[ti + 1].wg0) ++ti;
        const DevTier tr = dev_plan->tier[ti];
        if ((int64_t)blockIdx.x >= (int64_t)tr.wg0 + (int64_t)tr.items * tr.nsplit) return;
        nwhole = dev_plan->tier[0].items;
        nsplit_tail = tr.nsplit;
        split_major_items = tr.items;       // (inside a tier: all first pieces, then all second pieces ...)
        tier_item0 = tr.item0;
        tier_wg0 = tr.wg0;
        tier_rec0 = tr.rec0;
    }
    const bool tail_wg = (int64_t)blockIdx.x >= nwhole;
    const int64_t tail_id = (int64_t)blockIdx.x - tier_wg0;      // (host plan: one tier behind the whole items)
    const int nsplit = tail_wg ? nsplit_tail : 1;
    // split-minor (a partly filled last round: the splits of an item sit next to each other) or split-MAJOR (launches
    // with a device-side query bound split EVERY item, see launch(): all first halves, then all second halves, so that
    // workgroup p and its item still share p % 8 = the XCD the item's (sample, head) pair is pinned to)
    const int64_t tail_item = split_major_items ? tail_id % split_major_items : tail_id / nsplit;
    const int split = !tail_wg ? 0 : split_major_items ? (int)(tail_id / split_major_items) : (int)(tail_id % nsplit);
    const int64_t lin = item_of(tail_wg ? tier_item0 + tail_item : (int64_t)blockIdx.x, nqb, xcd_groups);
    float *partial = tail_wg ? partial_base + (tier_rec0 + tail_item * nsplit + split) * rec_floats(D) * (waves_for(D) * 64) : nullptr;
    const int64_t b = lin / (nqb * H), h = (lin / nqb) % H;
    const int64_t bq = b % src_batch;  // PnP injection: q/k of the source sample (pnp_utils.py:57-67)
    const int64_t q0 = (lin % nqb) * QB + wave * QW;
    const int64_t C = H * D;
    int64_t Mk = Mk_arg;
    if constexpr (FOLD) {
        const int64_t kc = k_count[b];
        Mk = kc < Mk_arg ? (kc > 0 ? kc : 1) : Mk_arg;
    }
    // device-side query bound (compacted live queries, vtm_compact_queries): the launch is sized for the host-known
    // upper bound M; a query block that starts at or beyond its sample's count has nothing anybody reads
    if (q_count != nullptr && (lin % nqb) * QB >= (int64_t)q_count[b]) return;

    // one-time LDS init: K pad columns = 0 (they meet Q's zero padding; garbage could be NaN), V^T pad rows
    // = 0 except row D = 1 (denominator row) -- tile loads never touch these
    for (int i = tid; i < 2 * KV * (K_STRIDE - D); i += NT) {
        const int row = i / (K_STRIDE - D), c = D + i % (K_STRIDE - D);
        sK[row * K_STRIDE + c] = (elem)((BIAS && c == D) ? 1.0f : 0.0f);
    }
    if constexpr (VROWS > D) {
        for (int i = tid; i < 2 * (VROWS - D) * VT_STRIDE; i += NT) {
            const int bufi = i / ((VROWS - D) * VT_STRIDE), rem = i % ((VROWS - D) * VT_STRIDE);
            const int row = D + rem / VT_STRIDE, c = rem % VT_STRIDE;
            sV[bufi * SV_TILE + row * VT_STRIDE + c] = (elem)((row == D) ? 1.0f : 0.0f);
        }
    }

    // Q fragments (B operand of S^T = K Q^T): lane (query l31, half hi) holds d = 16 ks + 8 hi + 0..7
    vec qf[DK];
    {
        const int64_t qi = q0 + l31;
        const T *qp = q + (bq * Mp + (qi < M ? qi : 0)) * ldq + h * D;
#pragma unroll
        for (int ks = 0; ks < DK; ++ks) {
            const int d0 = ks * 16 + hi * 8;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (d0 < D && qi < M) v = *reinterpret_cast<const uint4 *>(qp + d0);
            qf[ks] = *reinterpret_cast<vec *>(&v);
            if constexpr (BIAS) {
#pragma unroll
                for (int e = 0; e < 8; ++e) qf[ks][e] = (elem)((float)qf[ks][e] * scale_log2e);
            }
        }
        if constexpr (FOLD) {   // channels D + 2, D + 3 (lane half BIAS_HI, elements BIAS_E + 2, + 3) meet the key's bias pair
            if (hi == BIAS_HI) {
                qf[DK - 1][BIAS_E + 2] = (elem)1.0f;
                qf[DK - 1][BIAS_E + 3] = (elem)1.0f;
            }
        }
    }

    // hoisted staging addresses: chunk c = tid + 256 i ; K: (row c / DCH, 16-byte piece c % DCH);
    // V^T: (channel row c / 8, key piece c % 8).  Tiles are fetched with buffer loads: a wave-uniform descriptor of
    // the (sample, head) slice, a per-thread 32-bit byte offset computed once, and a scalar tile offset that advances by
    // 64 rows / 64 keys per tile -- no vector address arithmetic inside the loop (the kernel is VALU-bound at d = 40).
    // The descriptors carry no real bound (ragged tiles are masked explicitly below).
    uint32_t kgo[K_PER_T], vgo[V_PER_T];   // byte offsets relative to the tile base
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
        // inside every 16-key group the tile is stored as [k0-3 | k8-11 | k4-7 | k12-15]: the 8 keys one lane
        // feeds to a PV k-step (4 hi + {0..3} and 8 + 4 hi + {0..3}) are then one contiguous 16-byte read
        voff[i] = (c / (KV / 8)) * VT_STRIDE + (vkey[i] & ~15) + ((vkey[i] >> 3) & 1) * 4;
    }
    const auto rsrc_k = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(k + bq * Mkp * ldk + h * D), 0, 0x7fffffff, 0x00020000);
    const auto rsrc_v = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(vt + (b * C + h * D) * ldvt), 0, 0x7fffffff, 0x00020000);
    const uint32_t kstep = (uint32_t)(KV * ldk) * 2u, vstep = (uint32_t)KV * 2u;   // bytes per tile
    uint32_t so_k = 0, so_v = 0;                                                    // scalar tile offsets (bytes)
    // FOLD: the first wave also stages the tile's 64 bias words (one per key) into the K rows
    const auto rsrc_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(FOLD ? k_bias + b * ldkb : nullptr), 0, 0x7fffffff, 0x00020000);
    uint32_t so_b = 0, rbias = 0;
    [[maybe_unused]] const uint32_t bgo = (uint32_t)(tid & (KV - 1)) * 4u;
    auto fetch = [](const auto &rsrc, uint32_t voff_, uint32_t soff_) {
        return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff_, soff_, 0));
    };

    uint4 rk[K_PER_T], rv[V_PER_T];
    auto issue_full = [&]() {   // tile completely inside [0, M): no bounds logic
        // unconditional: surplus threads re-read chunk 0 (their kgo / vgo is 0) and simply do not store it --
        // a load inside a divergent branch makes the compiler wait for ALL outstanding loads right after it
#pragma unroll
        for (int i = 0; i < K_PER_T; ++i) rk[i] = fetch(rsrc_k, kgo[i], so_k);
#pragma unroll
        for (int i = 0; i < V_PER_T; ++i) rv[i] = fetch(rsrc_v, vgo[i], so_v);
        if constexpr (FOLD) {   // (every wave fetches the 64 words -- no load behind a branch, see above -- the first one stores them)
            rbias = __builtin_amdgcn_raw_buffer_load_b32(rsrc_b, bgo, so_b, 0);
            so_b += (uint32_t)KV * 4u;
        }
        so_k += kstep;
        so_v += vstep;
    };
    auto issue_tail = [&](int64_t key0) {   // ragged last tile: rows / keys >= M read as zero
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
            if (vok[i] && key < Mk) {   // ldvt >= Mk rounded up to 8: the 16-byte piece is inside the row
                v = fetch(rsrc_v, vgo[i], so_v);
                mask_keys(v, (int)(Mk - key));   // p is 0 there, but 0 * garbage may be NaN
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
            if (vok[i]) {   // keys 0-3 and 4-7 of the chunk go to the two halves of the 16-key group
                uint2 *dst = reinterpret_cast<uint2 *>(dv + voff[i]);
                dst[0] = make_uint2(rv[i].x, rv[i].y);
                dst[2] = make_uint2(rv[i].z, rv[i].w);
            }
    };

    f32x16 o[PV16 ? 1 : DV];          // 32-row blocks: o[dv][r] = row 32 dv + (r & 3) + 8 (r >> 2) + 4 hi of query l31
    f32x4 o16[PV16 ? DV16 : 1][2];    // PV16: o16[dv][qh][e] = row 16 dv + 4 g16 + e of query 16 qh + l15
#pragma unroll
    for (int dv = 0; dv < (PV16 ? 1 : DV); ++dv)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dv][r] = 0.0f;
#pragma unroll
    for (int dv = 0; dv < (PV16 ? DV16 : 1); ++dv)
#pragma unroll
        for (int qh = 0; qh < 2; ++qh)
#pragma unroll
            for (int e = 0; e < 4; ++e) o16[dv][qh][e] = 0.0f;
    // O^T *= alpha (alpha is per query, in the S^T layout: lane l31 and l31 + 32 hold the same value)
    auto rescale = [&](float alpha) {
        if constexpr (PV16) {
#pragma unroll
            for (int qh = 0; qh < 2; ++qh) {
                const float a = __shfl(alpha, 16 * qh + l15, 64);
#pragma unroll
                for (int dv = 0; dv < DV16; ++dv)
#pragma unroll
                    for (int e = 0; e < 4; ++e) o16[dv][qh][e] *= a;
            }
        } else {
#pragma unroll
            for (int dv = 0; dv < DV; ++dv)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[dv][r] *= alpha;
        }
    };
    float m_run = -INFINITY;   // running max in scaled (log2) units
    float m_bias = 0.0f;       // BIAS: the fp16-representable shift currently held in channel D of the query
    float l_run = 0.0f;        // only used when there is no spare O^T row

    // one tile: S^T = K Q^T -> online softmax -> O^T += V^T P^T.  TAIL = the ragged last tile (keys >= M masked).
    auto tile = [&](auto tail_tag, int buf, int64_t key0) {
        constexpr bool TAIL = decltype(tail_tag)::value;
        // ---- S^T = K Q^T : 2 blocks of 32 keys
        f32x16 s[2];
        auto compute_s = [&]() {
            if constexpr (PV16) {
                // the two 32-key blocks one after the other: the exps of the first start beside the MFMAs of the second
                vec kf[2][DK];
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    const elem *kp = sK + buf * SK_TILE + (kb * 32 + l31) * K_STRIDE + hi * 8;
#pragma unroll
                    for (int ks = 0; ks < DK; ++ks) kf[kb][ks] = *reinterpret_cast<const vec *>(kp + ks * 16);
                }
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int r = 0; r < 16; ++r) s[kb][r] = 0.0f;
#pragma unroll
                    for (int ks = 0; ks < DK; ++ks) s[kb] = F::mfma(kf[kb][ks], qf[ks], s[kb]);
                }
                __builtin_amdgcn_sched_barrier(0);
            } else {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s[kb][r] = 0.0f;
                const elem *kp = sK + buf * SK_TILE + (kb * 32 + l31) * K_STRIDE + hi * 8;
#pragma unroll
                for (int ks = 0; ks < DK; ++ks)
                    s[kb] = F::mfma(*reinterpret_cast<const vec *>(kp + ks * 16), qf[ks], s[kb]);
            }
            }
            if constexpr (TAIL) {   // lane (l31, hi) holds keys key0 + 32 kb + (r & 3) + 8 (r >> 2) + 4 hi
                const int lim = (int)(Mk - key0);
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi >= lim) s[kb][r] = -INFINITY;
            }
        };

        // ---- online softmax, base 2, deferred rescale: raises the shift when this tile's maximum outgrew it
        auto raise_shift = [&]() {
            float mt = fmaxf(s[0][0], s[1][0]);
#pragma unroll
            for (int r = 1; r < 16; ++r) mt = fmaxf(fmaxf(mt, s[0][r]), s[1][r]);
            if constexpr (BIAS) {
                // scores arrive as s c - m_bias: mt is the growth over the current shift (m_run = -inf only before
                // the first tile, which therefore always takes the branch and installs its own maximum)
                mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
                if (!__all(m_bias + mt <= m_run + DEFER_THR)) {
                    const float m_new = (float)(elem)(fmaxf(m_run, m_bias + mt));   // fp16-representable
                    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);      // first tile: exp2(-inf) = 0
                    const float delta = m_bias - m_new;                             // exact: both are fp16 values
                    m_run = m_new;
                    m_bias = m_new;
                    l_run *= alpha;
                    rescale(alpha);
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                        for (int r = 0; r < 16; ++r) s[kb][r] += delta;   // this tile was computed with the old shift
                    if (hi == BIAS_HI) qf[DK - 1][BIAS_E] = (elem)(-m_new);
                }
            } else {
                mt = fmaxf(mt, __shfl_xor(mt, 32, 64)) * scale_log2e;
                if (!__all(mt <= m_run + DEFER_THR)) {
                    const float m_new = fmaxf(m_run, mt);
                    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);   // first tile: exp2(-inf) = 0
                    m_run = m_new;
                    l_run *= alpha;
                    rescale(alpha);
                }
            }
        };
        vec pf[4];
        auto softmax_step = [&](int st) {   // p of keys 16 st + (e & 3) + 8 (e >> 2) + 4 hi
            float p[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float sv = s[st >> 1][8 * (st & 1) + e];
                p[e] = __builtin_amdgcn_exp2f(BIAS ? sv : __builtin_fmaf(sv, scale_log2e, -m_run));
                if constexpr (!SPARE) l_run += p[e];
            }
            F::pack8(pf[st], p);
        };
        compute_s();
        if constexpr (PV16) {
            static_assert(BIAS, "16-row O^T blocks imply a spare k-slot");
            // The shift already rides in the MFMA, so the exps run FIRST and the maximum is taken afterwards, on the
            // packed P (8 packed 3-input maxima instead of 16 fp32 ones, and no cross-half exchange): P <= 2^8 means
            // the shift still holds.  Otherwise -- the first tile, or scores that outgrew the shift by more than
            // 2^8, possibly up to inf in P -- the tile is redone the exact way: scores again, maximum, new shift.
            const elem *vp = sV + buf * SV_TILE + l15 * VT_STRIDE + (g16 & 1) * 16 + (g16 >> 1) * 8;
#pragma unroll
            for (int st = 0; st < 4; ++st) softmax_step(st);
            uint32_t pw[16];
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const u32x4 w = __builtin_bit_cast(u32x4, pf[st]);
#pragma unroll
                for (int j = 0; j < 4; ++j) pw[4 * st + j] = w[j];
            }
#pragma unroll
            for (int j = 0; j < 5; ++j) pw[j] = F::pmax3(pw[3 * j], pw[3 * j + 1], pw[3 * j + 2]);   // 16 -> 5 + 1
            const uint32_t pr = F::pmax3(F::pmax3(pw[0], pw[1], pw[2]), F::pmax3(pw[3], pw[4], pw[15]), pw[15]);
            const uint32_t ptop = max(pr >> 16, pr & 0xffffu);
            if (__any(ptop > F::BITS_256 || m_run == -INFINITY)) {
                compute_s();
                raise_shift();
#pragma unroll
                for (int st = 0; st < 4; ++st) softmax_step(st);
            }
            // ---- O^T += V^T P^T in 16-row blocks: 2 steps of 32 keys.  After the swaps pf[2 ks] / pf[2 ks + 1] are
            // the B operands of queries 0-15 / 16-31: k-slot group g16 holds the keys of 16-key group
            // 2 ks + (g16 & 1), lane half g16 >> 1 -- in the V^T tile that is ONE 16-byte piece (see voff)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                vec a[DV16];
#pragma unroll
                for (int dv = 0; dv < DV16; ++dv)
                    a[dv] = *reinterpret_cast<const vec *>(vp + ks * 32 + dv * 16 * VT_STRIDE);
                u32x4 x = __builtin_bit_cast(u32x4, pf[2 * ks]), y = __builtin_bit_cast(u32x4, pf[2 * ks + 1]);
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const auto r = __builtin_amdgcn_permlane16_swap(x[w], y[w], false, false);
                    x[w] = r[0];
                    y[w] = r[1];
                }
                const vec p0 = __builtin_bit_cast(vec, x), p1 = __builtin_bit_cast(vec, y);
#pragma unroll
                for (int dv = 0; dv < DV16; ++dv) {
                    o16[dv][0] = F::mfma16(a[dv], p0, o16[dv][0]);
                    o16[dv][1] = F::mfma16(a[dv], p1, o16[dv][1]);
                }
            }
        } else {
            raise_shift();
#pragma unroll
            for (int st = 0; st < 4; ++st) softmax_step(st);
            // ---- O^T += V^T P^T : 4 steps of 16 keys; k-slot (hi, e) <-> key 16 st + 8 (e >> 2) + 4 hi + (e & 3)
#pragma unroll
            for (int dv = 0; dv < DV; ++dv) {
                const elem *vp = sV + buf * SV_TILE + (dv * 32 + l31) * VT_STRIDE + 8 * hi;
#pragma unroll
                for (int st = 0; st < 4; ++st)
                    o[dv] = F::mfma(*reinterpret_cast<const vec *>(vp + st * 16), pf[st], o[dv]);
            }
        }
    };
    using std::false_type;
    using std::true_type;

    const int ntiles = (int)((Mk + KV - 1) / KV), nfull = (int)(Mk / KV);
    const int tps = (ntiles + nsplit - 1) / nsplit;                         // key tiles per split
    const int tb = split * tps, te = tb + tps < ntiles ? tb + tps : ntiles;
    const int fe = te < nfull ? te : nfull;                                 // end of the full tiles of this range
    so_k = (uint32_t)tb * kstep;
    so_v = (uint32_t)tb * vstep;
    so_b = (uint32_t)tb * (uint32_t)KV * 4u;
    if (tb < fe) issue_full(); else issue_tail((int64_t)tb * KV);
    write_lds(0);
    __syncthreads();

    // hot loop: full tiles whose successor is full too -- no bounds logic of any kind inside
    int t = tb;
    int buf = 0;
    for (; t + 1 < fe; ++t) {
        issue_full();
        tile(false_type{}, buf, (int64_t)t * KV);
        write_lds(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    if (t < fe) {               // last full tile; prefetches the ragged tile if it belongs to this range
        const bool ragged_next = te > fe;
        if (ragged_next) issue_tail((int64_t)fe * KV);
        tile(false_type{}, buf, (int64_t)t * KV);
        if (ragged_next) write_lds(buf ^ 1);
        __syncthreads();
        ++t;
        buf ^= 1;
    }
    if (t < te) tile(true_type{}, buf, (int64_t)t * KV);

    if (partial) {   // split workgroup: hand the raw state to attention_combine_kernel
        constexpr int NA = acc_floats(D), NM = max_floats(D);
        float *pp = partial + tid;
        if constexpr (PV16) {
#pragma unroll
            for (int r = 0; r < NA; ++r) pp[r * NT] = o16[r >> 3][(r >> 2) & 1][r & 3];
#pragma unroll
            for (int qh = 0; qh < 2; ++qh) pp[(NA + qh) * NT] = __shfl(m_run, 16 * qh + l15, 64);
        } else {
#pragma unroll
            for (int r = 0; r < NA; ++r) pp[r * NT] = o[r >> 4][r & 15];
            pp[NA * NT] = m_run;
        }
        pp[(NA + NM) * NT] = l_run;
        return;
    }
    if constexpr (PV16)
        write_output16<T, D>(o16, out, ldo, b, h, q0, M, Mp, lane);
    else
        write_output<T, D>(o, l_run, out, ldo, b, h, q0, M, Mp, l31, hi);
}

// Tail plan.  All workgroups of a launch take the same time, so the launch runs in "rounds" of as many workgroups
// as the chip holds (slots); the last, partly filled round leaves most CUs idle for a whole workgroup time (cfg-2
// mid blocks: 272 workgroups on 256 slots -> two rounds for 6 % more work than one; top blocks: 4.25 rounds).  The
// work items of that last round are therefore split along the key axis into `nsplit` shorter workgroups that fill
// the chip -- in the SAME launch, behind the whole ones, so they start as the slots of the last whole round free
// up -- and merged by attention_combine_kernel.
// `bounded`: the launch carries a device-side query count (vtm_attention_kv_bounded: compacted live queries).  How many
// of its workgroups do real work is not known when it is launched -- the cfg-2 top block launches 2 176 for ~1 800 live
// ones, 3.5 rounds of 512 that cost 4 -- so the round structure cannot be planned.  It is made finer instead: EVERY
// work item is split in two along the key axis (split-major order), the live ones then fill 7 half-length rounds
// (profiles/r04_attention_split_all.txt); the price is one partial record per workgroup for attention_combine_kernel.
template <int D>
TailPlan plan_tail(int64_t B, int64_t h, int64_t Mq, int64_t Mk, bool bounded = false) {
    constexpr int WAVES = waves_for(D), QB = WAVES * QW;
    constexpr int wg_per_cu = D <= 48 ? 2 : 1;   // resident workgroups per CU (launch bounds / LDS)
    TailPlan p;
    p.nqb = vtm::cdiv(Mq, QB);
    p.total = p.nqb * h * B;
    const int64_t slots = (int64_t)vtm::device_cus() * wg_per_cu;
    p.full = p.total / slots * slots;
    const int64_t rem = p.total - p.full, ntiles = vtm::cdiv(Mk, KV);
    p.nsplit = 1;
    p.ws_bytes = 0;
    p.split_all = false;
    if (bounded && p.total >= 2 * slots && ntiles >= 64 && p.total % 8 == 0) {
        p.full = 0;
        p.nsplit = 2;
        p.split_all = true;
        p.ws_bytes = (size_t)p.total * 2 * rec_floats(D) * (WAVES * 64) * sizeof(float);
        return p;
    }
    // worth it only behind at least one whole round, for long key axes, and when the last round is at most a
    // quarter full (a workgroup that has its CU to itself already runs about twice as fast as in a full round;
    // measured: 128 of 512 -> -7 %, 16 of 256 -> -18 %, 192 or 256 of 512 -> no gain)
    if (p.full > 0 && rem > 0 && rem * 4 <= slots && ntiles >= 32) {
        int64_t ns = slots / rem;
        if (ns > 16) ns = 16;
        if (ns > ntiles / 8) ns = ntiles / 8;
        if (ns >= 2) {
            p.nsplit = (int)ns;
            p.ws_bytes = (size_t)rem * ns * rec_floats(D) * (WAVES * 64) * sizeof(float);
        }
    }
    if (p.nsplit == 1) p.full = p.total;
    return p;
}

bool devplan_on() {
    static const bool on = [] {
        const char *e = getenv("VTM_ATT_DEVPLAN");      // A/B hook, read once per process
        return e == nullptr || atoi(e) != 0;
    }();
    return on;
}
// workspace of a device-planned query-bounded launch of attention_kernel<D>
template <int D>
size_t devplan_ws(int64_t Mk) {
    if (!devplan_on() || vtm::cdiv(Mk, KV) < 16) return 0;
    const int slots = vtm::device_cus() * (D <= 48 ? 2 : 1);
    return 256 + (size_t)plan_tail_wgs(slots) * rec_floats(D) * (waves_for(D) * 64) * sizeof(float);
}

template <typename T, int D, bool FOLD = false>
int launch(const void *q, int64_t ldq, const void *k, int64_t ldk, const void *vt, int64_t ldvt, void *out,
           int64_t ldo, int64_t B, int64_t h, int64_t M, int64_t Mp, int64_t Mk, int64_t Mkp, float scale, int share_groups,
           void *ws, size_t ws_bytes, const int32_t *q_count, hipStream_t s, const int32_t *k_count = nullptr,
           const uint32_t *k_bias = nullptr, int64_t ldkb = 0) {
    constexpr int DK = (D + 15) / 16;
    constexpr size_t lds = (size_t)2 * (KV * (DK * 16 + 8) + vrows_for(D) * VT_STRIDE) * 2;
    if (lds > 64 * 1024) {   // opt in to > 64 KB of dynamic LDS once per (kernel instantiation, device)
        static std::atomic<bool> attr_set[vtm::MAX_DEVICES];   // (one per instantiation of this function template)
        const int dev = vtm::current_device();
        if (!attr_set[dev].load(std::memory_order_acquire)) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(attention_kernel<T, D, FOLD>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess)
                return vtm::fail(VTM_ELAUNCH, "vtm_attention: LDS attribute: %s", hipGetErrorString(e));
            attr_set[dev].store(true, std::memory_order_release);
        }
    }
    constexpr int WAVES = waves_for(D);
    const float scale_log2e_ = scale * 1.4426950408889634f;
    // (round 6) a query-bounded launch is planned ON THE DEVICE from the live counts when the workspace holds the plan and
    // the records of the largest tail a plan can have (vtm_attention_kv_bounded_ws_bytes): whole items, then tiers split
    // 2 .. 16 ways -- rounds 4-5 split every item in two (VTM_ATT_DEVPLAN=0 keeps that plan)
    if (q_count != nullptr && ws != nullptr && devplan_on()) {
        const int slots = vtm::device_cus() * (D <= 48 ? 2 : 1);
        const size_t rec_bytes = (size_t)rec_floats(D) * (WAVES * 64) * sizeof(float);
        constexpr int QB = WAVES * QW;
        // (launches of at least two rounds: the oversized grid and the two small launches of a plan cost 50-70 us, which a
        // one-round launch does not get back -- profiles/r06_k_devplan_attention_kernel.txt)
        if (ws_bytes >= 256 + (size_t)plan_tail_wgs(slots) * rec_bytes && vtm::cdiv(M, QB) * h * B >= 2 * slots) {
            DevPlan *plan = reinterpret_cast<DevPlan *>(ws);
            float *records = reinterpret_cast<float *>(static_cast<char *>(ws) + 256);
            const int64_t nqb_max = vtm::cdiv(M, QB), total = nqb_max * h * B, tail_max = plan_tail_wgs(slots);
            VTM_REQUIRE(total + tail_max < (1ll << 31) / 16, "vtm_attention: grid too large");
            const int xcd_pairs = (B * h) % 8 == 0 ? (int)(B * h / 8) : 0;
            hipLaunchKernelGGL(attention16_plan_kernel, dim3(1), dim3(64), 0, s, q_count, (int)B, (int)h, QB, slots,
                               (int)vtm::cdiv(Mk, KV), plan);
            hipLaunchKernelGGL((attention_kernel<T, D, FOLD>), dim3((unsigned)(total + tail_max)), dim3(WAVES * 64), lds, s,
                               (const T *)q, ldq, (const T *)k, ldk, (const T *)vt, ldvt, (T *)out, ldo, h, M, Mp, Mk, Mkp,
                               scale_log2e_, B / share_groups, nqb_max, total, 1, records, xcd_pairs, q_count, (int64_t)0, k_count,
                               k_bias, ldkb, (const DevPlan *)plan);
            hipLaunchKernelGGL((attention_combine_kernel<T, D>), dim3((unsigned)plan_split_items(slots)), dim3(WAVES * 64), 0, s,
                               (const float *)records, (T *)out, ldo, h, M, Mp, nqb_max, total, 1, xcd_pairs, q_count,
                               (const DevPlan *)plan);
            return vtm::launch_status("vtm_attention");
        }
    }
    TailPlan p = plan_tail<D>(B, h, M, Mk, q_count != nullptr);
    if (p.split_all && (!ws || ws_bytes < p.ws_bytes)) p = plan_tail<D>(B, h, M, Mk);   // not enough workspace: the plain plan
    if (p.nsplit > 1 && (!ws || ws_bytes < p.ws_bytes)) {   // no workspace: plain single launch
        p.nsplit = 1;
        p.full = p.total;
        p.split_all = false;
    }
    const float scale_log2e = scale * 1.4426950408889634f;
    const int64_t src_batch = B / share_groups;
    VTM_REQUIRE(p.total < (1ll << 31) / 16, "vtm_attention: grid too large");
    // one launch: the whole workgroups first, the key-split ones of the last round behind them (they start as the
    // slots of the last whole round free up -- no launch boundary to drain)
    const int64_t rem = p.total - p.full;
    // (sample, head) pairs pinned to XCDs when they divide evenly and every pair has enough query blocks to keep an
    // XCD's share of the chip busy (see item_of); VTM_ATT_NO_XCD_MAP is an A/B build switch
#ifdef VTM_ATT_NO_XCD_MAP
    const int xcd_groups = 0;
#else
    const int xcd_groups = ((B * h) % 8 == 0 && p.nqb >= 64) ? (int)(B * h / 8) : 0;
#endif
    hipLaunchKernelGGL((attention_kernel<T, D, FOLD>), dim3((unsigned)(p.full + rem * p.nsplit)), dim3(WAVES * 64), lds, s,
                       (const T *)q, ldq, (const T *)k, ldk, (const T *)vt, ldvt, (T *)out, ldo, h, M, Mp, Mk, Mkp,
                       scale_log2e, src_batch, p.nqb, p.full, p.nsplit, (float *)ws, xcd_groups, q_count,
                       p.split_all ? rem : (int64_t)0, k_count, k_bias, ldkb, (const DevPlan *)nullptr);
    // (few items: their accumulator groups are shared out, attention_combine_parts_kernel)
    bool parts = !pv16_for(D) && rem * 4 <= vtm::device_cus() && acc_floats(D) % 8 == 0;
    static const bool no_parts = getenv("VTM_DEBUG_COMBINE_PARTS") != nullptr;   // A/B hook, read once per process
    if (no_parts) parts = false;
    if (p.nsplit > 1) {
        if constexpr (!pv16_for(D)) {
            if (parts)
                hipLaunchKernelGGL((attention_combine_parts_kernel<T, D>), dim3((unsigned)rem, (unsigned)(acc_floats(D) / 8)),
                                   dim3(WAVES * 64), 0, s, (const float *)ws, (T *)out, ldo, h, M, Mp, p.nqb, p.full, p.nsplit,
                                   xcd_groups, q_count);
        }
        if (!parts)
            hipLaunchKernelGGL((attention_combine_kernel<T, D>), dim3((unsigned)rem), dim3(WAVES * 64), 0, s,
                               (const float *)ws, (T *)out, ldo, h, M, Mp, p.nqb, p.full, p.nsplit, xcd_groups, q_count,
                               (const DevPlan *)nullptr);
    }
    return vtm::launch_status("vtm_attention");
}

// Which d = 40 launches go to the wide-tile kernel of attention16.hip, and in which shape (round 6).  VTM_ATT16=0 keeps
// everything on attention_kernel; VTM_ATT16_NQ=1 does the same for the plain (one value group) shape only, VTM_ATT16_SKEW=0
// selects the un-skewed kernels (4-wave workgroups were measured 18 % slower -- register spills -- and are not built).  Read
// once per process.  Both kernels compute the same sums in the same per-tile order for a query (the split plans differ),
// so results agree to the tolerance of the key-split combine, not bit for bit.
struct Policy16 {
    bool on, skew;
    int nq, waves;
};
const Policy16 &policy16() {
    static const Policy16 p = [] {
        Policy16 v{true, true, 2, 8};
        if (const char *e = getenv("VTM_ATT16")) v.on = atoi(e) != 0;
        if (const char *e = getenv("VTM_ATT16_SKEW")) v.skew = atoi(e) != 0;
        if (const char *e = getenv("VTM_ATT16_NQ")) v.nq = atoi(e) == 1 ? 1 : 2;
        if (v.nq == 1) v.waves = 8;
        return v;
    }();
    return p;
}
// -> true and the shape when this launch is the wide kernel's
bool shape16_for(int64_t d, int share_groups, bool fold, Shape16 *sh) {
    const Policy16 &p = policy16();
    if (!p.on || d != 40) return false;
    if (share_groups == 1) {
        if (p.nq == 1) return false;               // (one sub-tile, one group IS attention_kernel)
        *sh = Shape16{p.nq, 1, p.waves, p.skew};
        return true;
    }
    if (!fold && (share_groups == 2 || share_groups == 3)) {   // shared probabilities: P once, one PV per sample
        *sh = Shape16{1, share_groups, 8, p.skew};   // (skew: attention16g.hip, else attention16_kernel<NQ = 1, NG>)
        return true;
    }
    return false;
}

template <typename T>
int dispatch(int64_t d, const void *q, int64_t ldq, const void *k, int64_t ldk, const void *vt, int64_t ldvt,
             void *out, int64_t ldo, int64_t B, int64_t h, int64_t M, int64_t Mp, int64_t Mk, int64_t Mkp, float scale, int sg,
             void *ws, size_t ws_bytes, const int32_t *q_count, hipStream_t s) {
    switch (d) {
        case 40: return launch<T, 40>(q, ldq, k, ldk, vt, ldvt, out, ldo, B, h, M, Mp, Mk, Mkp, scale, sg, ws, ws_bytes, q_count, s);
        case 64: return launch<T, 64>(q, ldq, k, ldk, vt, ldvt, out, ldo, B, h, M, Mp, Mk, Mkp, scale, sg, ws, ws_bytes, q_count, s);
        case 80: return launch<T, 80>(q, ldq, k, ldk, vt, ldvt, out, ldo, B, h, M, Mp, Mk, Mkp, scale, sg, ws, ws_bytes, q_count, s);
        case 160: return launch<T, 160>(q, ldq, k, ldk, vt, ldvt, out, ldo, B, h, M, Mp, Mk, Mkp, scale, sg, ws, ws_bytes, q_count, s);
        case 8: return launch<T, 8>(q, ldq, k, ldk, vt, ldvt, out, ldo, B, h, M, Mp, Mk, Mkp, scale, sg, ws, ws_bytes, q_count, s);
        case 16: return launch<T, 16>(q, ldq, k, ldk, vt, ldvt, out, ldo, B, h, M, Mp, Mk, Mkp, scale, sg, ws, ws_bytes, q_count, s);
        case 32: return launch<T, 32>(q, ldq, k, ldk, vt, ldvt, out, ldo, B, h, M, Mp, Mk, Mkp, scale, sg, ws, ws_bytes, q_count, s);
        case 96: return launch<T, 96>(q, ldq, k, ldk, vt, ldvt, out, ldo, B, h, M, Mp, Mk, Mkp, scale, sg, ws, ws_bytes, q_count, s);
        case 128: return launch<T, 128>(q, ldq, k, ldk, vt, ldvt, out, ldo, B, h, M, Mp, Mk, Mkp, scale, sg, ws, ws_bytes, q_count, s);
    }
    return vtm::fail(VTM_EINVAL, "vtm_attention: unsupported head dim %lld (have 8,16,32,40,64,80,96,128,160)",
                     (long long)d);
}

}  // namespace

// (d = 40: the caller does not say how the launch will share its probabilities -- the largest plan any shape could choose)
static size_t ws16_max(int64_t B, int64_t h, int64_t Mq, int64_t Mk, bool bounded) {
    size_t n = 0;
    Shape16 sh;
    for (int sg = 1; sg <= 3; ++sg)
        if (B % sg == 0 && shape16_for(40, sg, false, &sh)) {
            const size_t w = ws_bytes16(sh, sh.ng > 1 ? B / sg : B, h, Mq, Mk, bounded);
            n = w > n ? w : n;
        }
    return n;
}

VTM_EXPORT size_t vtm_attention_ws_bytes(int64_t B, int64_t h, int64_t Mq, int64_t Mk, int64_t d) {
    if (B <= 0 || h <= 0 || Mq <= 0 || Mk <= 0) return 0;
    switch (d) {
        case 40: return std::max(plan_tail<40>(B, h, Mq, Mk).ws_bytes, ws16_max(B, h, Mq, Mk, false));
        case 64: return plan_tail<64>(B, h, Mq, Mk).ws_bytes;
        case 80: return plan_tail<80>(B, h, Mq, Mk).ws_bytes;
        case 160: return plan_tail<160>(B, h, Mq, Mk).ws_bytes;
        case 8: return plan_tail<8>(B, h, Mq, Mk).ws_bytes;
        case 16: return plan_tail<16>(B, h, Mq, Mk).ws_bytes;
        case 32: return plan_tail<32>(B, h, Mq, Mk).ws_bytes;
        case 96: return plan_tail<96>(B, h, Mq, Mk).ws_bytes;
        case 128: return plan_tail<128>(B, h, Mq, Mk).ws_bytes;
    }
    return 0;
}

VTM_EXPORT size_t vtm_attention_kv_bounded_ws_bytes(int64_t B, int64_t h, int64_t Mq, int64_t Mk, int64_t d) {
    if (B <= 0 || h <= 0 || Mq <= 0 || Mk <= 0) return 0;
    switch (d) {
        case 40: return std::max(std::max(plan_tail<40>(B, h, Mq, Mk, true).ws_bytes, devplan_ws<40>(Mk)), ws16_max(B, h, Mq, Mk, true));
        case 64: return std::max(plan_tail<64>(B, h, Mq, Mk, true).ws_bytes, devplan_ws<64>(Mk));
        case 80: return std::max(plan_tail<80>(B, h, Mq, Mk, true).ws_bytes, devplan_ws<80>(Mk));
        case 160: return std::max(plan_tail<160>(B, h, Mq, Mk, true).ws_bytes, devplan_ws<160>(Mk));
        case 8: return std::max(plan_tail<8>(B, h, Mq, Mk, true).ws_bytes, devplan_ws<8>(Mk));
        case 16: return std::max(plan_tail<16>(B, h, Mq, Mk, true).ws_bytes, devplan_ws<16>(Mk));
        case 32: return std::max(plan_tail<32>(B, h, Mq, Mk, true).ws_bytes, devplan_ws<32>(Mk));
        case 96: return std::max(plan_tail<96>(B, h, Mq, Mk, true).ws_bytes, devplan_ws<96>(Mk));
        case 128: return std::max(plan_tail<128>(B, h, Mq, Mk, true).ws_bytes, devplan_ws<128>(Mk));
    }
    return 0;
}

static int attention_any(const void *q, int64_t ldq, const void *k, int64_t ldk, const void *vt, int64_t ldvt,
                         void *out, int64_t ldo, int dtype, int64_t B, int64_t h, int64_t Mq, int64_t Mqp, int64_t Mk,
                         int64_t Mkp, int64_t d, float scale, int share_groups, void *ws, size_t ws_bytes,
                         const int32_t *q_count, vtm_stream_t stream) {
    VTM_REQUIRE(q && k && vt && out, "vtm_attention: null pointer");
    VTM_REQUIRE(B > 0 && h > 0 && Mq > 0 && Mk > 0 && d > 0 && Mqp >= Mq && Mkp >= Mk, "vtm_attention: bad sizes");
    VTM_REQUIRE(share_groups >= 1 && B % share_groups == 0, "vtm_attention: B %% share_groups != 0");
    VTM_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldvt % 8 == 0 && ldo % 4 == 0 && ldvt >= Mk,
                "vtm_attention: leading dimensions must keep 16-byte alignment (ldvt >= Mk, %% 8)");
    // K / V^T tiles are addressed with 32-bit byte offsets inside one (sample, head) slice (buffer loads)
    VTM_REQUIRE((Mkp * ldk + d) * 2 < (1ll << 31) && (d * ldvt + Mkp) * 2 < (1ll << 31),
                "vtm_attention: a (sample, head) slice of K or V^T must stay below 2 GiB");
    hipStream_t s = vtm::as_stream(stream);
    Shape16 sh;
    if ((dtype == VTM_F16 || dtype == VTM_BF16) && shape16_for(d, share_groups, false, &sh)) {
        // the value groups of one (source sample, head) share a buffer descriptor: the sample stride rides in the offset
        if (sh.ng == 1 || (sh.ng * (B / share_groups) * h * d * ldvt) * 2 < (1ll << 31)) {
            const Args16 a{q, ldq, k, ldk, vt, ldvt, out, ldo, dtype, B, h, Mq, Mqp, Mk, Mkp, scale, share_groups, ws, ws_bytes,
                           q_count, s, false, nullptr, nullptr, 0};
            return attention16(a, sh);
        }
    }
    if (dtype == VTM_F16)
        return dispatch<__half>(d, q, ldq, k, ldk, vt, ldvt, out, ldo, B, h, Mq, Mqp, Mk, Mkp, scale, share_groups, ws,
                                ws_bytes, q_count, s);
    if (dtype == VTM_BF16)
        return dispatch<vtm_bf16>(d, q, ldq, k, ldk, vt, ldvt, out, ldo, B, h, Mq, Mqp, Mk, Mkp, scale, share_groups, ws,
                                  ws_bytes, q_count, s);
    return vtm::fail(VTM_EINVAL, "vtm_attention: dtype must be VTM_F16 or VTM_BF16");
}

VTM_EXPORT int vtm_attention_kv(const void *q, int64_t ldq, const void *k, int64_t ldk, const void *vt, int64_t ldvt,
                                void *out, int64_t ldo, int dtype, int64_t B, int64_t h, int64_t Mq, int64_t Mqp, int64_t Mk,
                                int64_t Mkp, int64_t d, float scale, int share_groups, void *ws, size_t ws_bytes,
                                vtm_stream_t stream) {
    return attention_any(q, ldq, k, ldk, vt, ldvt, out, ldo, dtype, B, h, Mq, Mqp, Mk, Mkp, d, scale, share_groups, ws,
                         ws_bytes, nullptr, stream);
}

VTM_EXPORT int vtm_attention_kv_bounded(const void *q, int64_t ldq, const void *k, int64_t ldk, const void *vt, int64_t ldvt,
                                        void *out, int64_t ldo, int dtype, int64_t B, int64_t h, int64_t Mq, int64_t Mqp,
                                        int64_t Mk, int64_t Mkp, int64_t d, float scale, const int32_t *q_count, void *ws,
                                        size_t ws_bytes, vtm_stream_t stream) {
    VTM_REQUIRE(q_count, "vtm_attention_kv_bounded: null q_count");
    return attention_any(q, ldq, k, ldk, vt, ldvt, out, ldo, dtype, B, h, Mq, Mqp, Mk, Mkp, d, scale, 1, ws, ws_bytes,
                         q_count, stream);
}

VTM_EXPORT int vtm_attention_kv_shared_bounded(const void *q, int64_t ldq, const void *k, int64_t ldk, const void *vt, int64_t ldvt,
                                               void *out, int64_t ldo, int dtype, int64_t B, int64_t h, int64_t Mq, int64_t Mqp,
                                               int64_t Mk, int64_t Mkp, int64_t d, float scale, int share_groups,
                                               const int32_t *q_count, void *ws, size_t ws_bytes, vtm_stream_t stream) {
    VTM_REQUIRE(q_count, "vtm_attention_kv_shared_bounded: null q_count");
    VTM_REQUIRE(share_groups >= 1 && B % share_groups == 0, "vtm_attention_kv_shared_bounded: B %% share_groups != 0");
    // (d = 40, 2 or 3 groups: attention16g computes the probabilities once per group and reads the source sample's count;
    // other shapes: attention_kernel recomputes them per sample and reads every sample's own -- equal -- count)
    return attention_any(q, ldq, k, ldk, vt, ldvt, out, ldo, dtype, B, h, Mq, Mqp, Mk, Mkp, d, scale, share_groups, ws, ws_bytes,
                         q_count, stream);
}

VTM_EXPORT int vtm_attention_kv_folded(const void *q, int64_t ldq, const void *k, int64_t ldk, const void *vt, int64_t ldvt,
                                       void *out, int64_t ldo, int dtype, int64_t B, int64_t h, int64_t Mq, int64_t Mqp,
                                       int64_t Mk, int64_t Mkp, int64_t d, float scale, const int32_t *q_count,
                                       const int32_t *k_count, const uint32_t *k_bias, int64_t ldkb, void *ws, size_t ws_bytes,
                                       vtm_stream_t stream) {
    VTM_REQUIRE(q && k && vt && out && k_count && k_bias, "vtm_attention_kv_folded: null pointer");
    VTM_REQUIRE(B > 0 && h > 0 && Mq > 0 && Mk > 0 && Mqp >= Mq && Mkp >= Mk && ldkb >= Mk, "vtm_attention_kv_folded: bad sizes");
    VTM_REQUIRE(d == 40 || d == 8, "vtm_attention_kv_folded: head dim %lld has no spare k-slots for the bias (d = 8, 40 do)", (long long)d);
    VTM_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldvt % 8 == 0 && ldo % 4 == 0 && ldvt >= Mk,
                "vtm_attention_kv_folded: leading dimensions must keep 16-byte alignment (ldvt >= Mk, %% 8)");
    VTM_REQUIRE((Mkp * ldk + d) * 2 < (1ll << 31) && (d * ldvt + Mkp) * 2 < (1ll << 31) && Mkp * 4 < (1ll << 31),
                "vtm_attention_kv_folded: a (sample, head) slice of K or V^T must stay below 2 GiB");
    hipStream_t s = vtm::as_stream(stream);
    Shape16 sh;
    if ((dtype == VTM_F16 || dtype == VTM_BF16) && shape16_for(d, 1, true, &sh)) {
        const Args16 a{q, ldq, k, ldk, vt, ldvt, out, ldo, dtype, B, h, Mq, Mqp, Mk, Mkp, scale, 1, ws, ws_bytes,
                       q_count, s, true, k_count, k_bias, ldkb};
        return attention16(a, sh);
    }
#define VTM_FOLDED(T, D) launch<T, D, true>(q, ldq, k, ldk, vt, ldvt, out, ldo, B, h, Mq, Mqp, Mk, Mkp, scale, 1, ws, ws_bytes, q_count, s, \
                                            k_count, k_bias, ldkb)
    if (dtype == VTM_F16) return d == 40 ? VTM_FOLDED(__half, 40) : VTM_FOLDED(__half, 8);
    if (dtype == VTM_BF16) return d == 40 ? VTM_FOLDED(vtm_bf16, 40) : VTM_FOLDED(vtm_bf16, 8);
#undef VTM_FOLDED
    return vtm::fail(VTM_EINVAL, "vtm_attention_kv_folded: dtype must be VTM_F16 or VTM_BF16");
}

VTM_EXPORT int vtm_attention(const void *q, int64_t ldq, const void *k, int64_t ldk, const void *vt,
                             int64_t ldvt, void *out, int64_t ldo, int dtype, int64_t B, int64_t h, int64_t M,
                             int64_t Mp, int64_t d, float scale, int share_groups, void *ws, size_t ws_bytes,
                             vtm_stream_t stream) {
    return vtm_attention_kv(q, ldq, k, ldk, vt, ldvt, out, ldo, dtype, B, h, M, Mp, M, Mp, d, scale, share_groups, ws,
                            ws_bytes, stream);
}

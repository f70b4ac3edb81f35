// Pieces shared by attention.hip (the general kernel) and attention16.hip (the wide-tile d = 40 kernel of round 6): fragment
// types, tile constants, the XCD-aware work-item map, the 16-row O^T output path, the tail plan record and the hand-over
// between the two translation units.  Header-only, internal linkage.
#pragma once
#include "common.h"

namespace vtm_att {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 b16x8 __attribute__((ext_vector_type(8)));

// waves per workgroup by head dim (each wave owns 32 queries; all waves share the K / V^T tiles): more waves
// amortise the tile staging, bounded by the register budget of the wider heads
constexpr int waves_for(int D) { return D <= 48 ? 8 : D <= 96 ? 16 : 4; }
constexpr int QW = 32;           // queries per wave
constexpr int KV = 64;           // keys per tile
constexpr int VT_STRIDE = KV + 8;  // 72 elements = 144 B: 16-byte aligned rows, conflict-free ds_read_b128 over 32 rows
constexpr float DEFER_THR = 8.0f;  // log2 units

// PV16: O^T is built from 16-row blocks (v_mfma_f32_16x16x32) instead of 32-row blocks when that needs fewer
// matrix-pipe cycles: d = 40 -> 48 rows (the denominator row included) instead of 64, a quarter of the PV work.
// P^T leaves the QK^T accumulators with one query per lane & 31; two v_permlane16_swap per register pair turn
// the fragments of two 16-key steps into the B operands (query = lane & 15) of the two 16-query halves.
constexpr bool pv16_for(int D) { return (D % 32) != 0 && (D + 16) / 16 * 16 < (D + 31) / 32 * 32; }
constexpr int vrows_for(int D) { return pv16_for(D) ? (D + 16) / 16 * 16 : (D + 31) / 32 * 32; }   // V^T tile rows
// per-thread record a key-split workgroup leaves for attention_combine_kernel: accumulators, running max
// (one per accumulator group), denominator
constexpr int acc_floats(int D) { return pv16_for(D) ? (D + 16) / 16 * 8 : (D + 31) / 32 * 16; }
constexpr int max_floats(int D) { return pv16_for(D) ? 2 : 1; }
constexpr int rec_floats(int D) { return acc_floats(D) + max_floats(D) + 1; }

template <typename T> struct Frag;
template <> struct Frag<__half> {
    using vec = h16x8;
    using elem = _Float16;
    __device__ static f32x16 mfma(vec a, vec b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
    __device__ static f32x4 mfma16(vec a, vec b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
    static constexpr uint32_t BITS_256 = 0x5C00u;   // 256.0
    __device__ static uint32_t pmax3(uint32_t a, uint32_t b, uint32_t c) {   // packed maximum of 3 x 2 values
        uint32_t d;
        asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
        return d;
    }
    __device__ static void pack8(vec &dst, const float (&p)[8]) {
        // round-to-nearest (v_cvt_pk_f16_f32): a truncating pack would bias the numerator against the fp32
        // denominator of the head dims without a spare O^T row
#pragma unroll
        for (int i = 0; i < 8; ++i) dst[i] = (_Float16)p[i];
    }
};
template <> struct Frag<vtm_bf16> {
    using vec = b16x8;
    using elem = __bf16;
    __device__ static f32x16 mfma(vec a, vec b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
    __device__ static f32x4 mfma16(vec a, vec b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
    static constexpr uint32_t BITS_256 = 0x4380u;   // 256.0
    __device__ static uint32_t pmax3(uint32_t a, uint32_t b, uint32_t c) {
        // P >= 0: the bit patterns order like the values (inf and NaN on top), so an integer maximum will do
        const u16x2 m = __builtin_elementwise_max(__builtin_elementwise_max(__builtin_bit_cast(u16x2, a),
                                                                            __builtin_bit_cast(u16x2, b)),
                                                  __builtin_bit_cast(u16x2, c));
        return __builtin_bit_cast(uint32_t, m);
    }
    __device__ static void pack8(vec &dst, const float (&p)[8]) {
#pragma unroll
        for (int i = 0; i < 8; ++i) dst[i] = (__bf16)p[i];
    }
};

// XCD-aware placement of the work items (query block, head, sample).  Workgroups are dispatched round-robin over the 8
// XCDs in index order (observed, not a contract -- this is a speed choice, any placement gives the same result), so
// position p runs on XCD p % 8.  All query blocks of one (sample, head) share its K / V^T stream: with `xcd_groups`
// = (B * H) / 8 > 0 the (sample, head) pairs are dealt to the XCDs -- pair hb runs on XCD hb % 8 only -- so that a
// K / V^T slice is fetched into ONE L2 instead of all eight (9x the algorithmic HBM-side traffic otherwise).
__device__ __forceinline__ int64_t item_of(int64_t pos, int64_t nqb, int xcd_groups) {
    if (xcd_groups == 0) return pos;
    const int64_t xcd = pos & 7, slot = pos >> 3;
    return (xcd + 8 * (slot / nqb)) * nqb + slot % nqb;
}

// zero the 16-bit elements j >= valid of a 16-byte piece (8 elements), on whole dwords so that the staging
// registers stay plain 32-bit values (an element-wise view makes the compiler repack them after every load)
__device__ __forceinline__ void mask_keys(uint4 &v, int valid) {
    uint32_t *w = reinterpret_cast<uint32_t *>(&v);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t keep = (2 * j + 1 < valid) ? 0xffffffffu : (2 * j < valid) ? 0x0000ffffu : 0u;
        w[j] &= keep;
    }
}

// normalise and store one wave's O^T accumulators: row q = q0 + l31, channels dv*32 + (r & 3) + 8 (r >> 2) + 4 hi
template <typename T, int D>
__device__ __forceinline__ void write_output(const f32x16 (&o)[(D + 31) / 32], float l_run, T *__restrict__ out,
                                             int64_t ldo, int64_t b, int64_t h, int64_t q0, int64_t M, int64_t Mp,
                                             int l31, int hi) {
    using elem = typename Frag<T>::elem;
    constexpr int DV = (D + 31) / 32;
    constexpr bool SPARE = (D % 32) != 0;
    float l_tot;
    if constexpr (SPARE) {
        // denominator row D of O^T: block D/32, in-block row D%32 = (r&3) + 8(r>>2) + 4hi
        constexpr int LB = D / 32, LR = D % 32;
        constexpr int LHI = (LR >> 2) & 1, LREG = (LR & 3) + 4 * (LR >> 3);
        l_tot = __shfl(o[LB][LREG], l31 + 32 * LHI, 64);
    } else {
        l_tot = l_run + __shfl_xor(l_run, 32, 64);
    }
    const float inv_l = 1.0f / l_tot;
    const int64_t qi = q0 + l31;
    if (qi < M) {
        T *op = out + (b * Mp + qi) * ldo + h * D;
#pragma unroll
        for (int dv = 0; dv < DV; ++dv)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d0 = dv * 32 + 8 * g + 4 * hi;
                if (d0 < D) {  // D % 8 == 0 and d0 % 4 == 0 -> the 4 channels are all valid
                    elem w[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) w[e] = (elem)(o[dv][g * 4 + e] * inv_l);
                    *reinterpret_cast<uint2 *>(op + d0) = *reinterpret_cast<uint2 *>(w);
                }
            }
    }
}

// PV16 layout: o[dv][qh][e] = O^T row 16 dv + 4 (lane >> 4) + e of query q0 + 16 qh + (lane & 15)
template <typename T, int D>
__device__ __forceinline__ void write_output16(const f32x4 (&o)[(D + 16) / 16][2], T *__restrict__ out, int64_t ldo,
                                               int64_t b, int64_t h, int64_t q0, int64_t M, int64_t Mp, int lane) {
    using elem = typename Frag<T>::elem;
    constexpr int DV16 = (D + 16) / 16;
    constexpr int LB = D / 16, LG = (D % 16) / 4, LE = D % 4;   // where the denominator row D sits
    const int l15 = lane & 15, g = lane >> 4;
#pragma unroll
    for (int qh = 0; qh < 2; ++qh) {
        const float inv_l = 1.0f / __shfl(o[LB][qh][LE], 16 * LG + l15, 64);
        const int64_t qi = q0 + 16 * qh + l15;
        if (qi < M) {
            T *op = out + (b * Mp + qi) * ldo + h * D;
#pragma unroll
            for (int dv = 0; dv < DV16; ++dv) {
                const int d0 = dv * 16 + 4 * g;
                if (d0 < D) {   // D % 4 == 0 -> the 4 channels are all valid
                    elem w[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) w[e] = (elem)(o[dv][qh][e] * inv_l);
                    *reinterpret_cast<uint2 *>(op + d0) = *reinterpret_cast<uint2 *>(w);
                }
            }
        }
    }
}

struct TailPlan {
    int64_t nqb, total, full;   // query blocks per (sample, head), all workgroups, workgroups in whole rounds
    int nsplit;                 // splits of each remaining work item (1 = none)
    size_t ws_bytes;
    bool split_all;             // every item is split (launches with a device-side query bound)
};


// ---- hand-over attention.hip -> attention16.hip ----
struct Args16 {
    const void *q; int64_t ldq; const void *k; int64_t ldk; const void *vt; int64_t ldvt; void *out; int64_t ldo;
    int dtype; int64_t B, h, M, Mp, Mk, Mkp; float scale; int share_groups; void *ws; size_t ws_bytes;
    const int32_t *q_count; hipStream_t s; bool fold; const int32_t *k_count; const uint32_t *k_bias; int64_t ldkb;
};
struct Shape16 {
    int nq, ng, waves;      // query sub-tiles per wave, value groups per wave, waves per workgroup
    bool skew;              // nq = 2, ng = 1: the skewed in-wave pipeline (attention16s_kernel)
};
int wg_per_cu16(int nq, int ng, int waves);
TailPlan plan_tail16(int64_t B_items, int64_t h, int64_t Mq, int64_t Mk, int64_t QB, int wg_per_cu, size_t item_rec_bytes,
                     bool bounded);
size_t ws_bytes16(const Shape16 &sh, int64_t B_items, int64_t h, int64_t Mq, int64_t Mk, bool bounded);
int attention16(const Args16 &a, const Shape16 &sh);
// attention16g.hip: shared probabilities (ng = 2, 3 value groups), one-tile skew
size_t ws_bytes16g(int ng, int64_t src_batch, int64_t h, int64_t Mq, int64_t Mk, bool bounded);
int attention16g(const Args16 &a, int ng);


}  // namespace vtm_att

// vtm_match: fused cosine-similarity scoring + row-wise top-1 (the score matrix never exists).
// Reference: vidtome/merge.py:87 `scores = a @ b.transpose(-1, -2)` followed by
//            merge.py:109-113 `node_max, node_idx = scores.max(dim=-1)` (non-aligned) or
//            merge.py:93-97 (aligned: scores of all samples concatenated on the dst axis);
//            same statements at merge.py:392-417 for the global matcher.
// The reference materialises (B, Ns, Nd) fp32 (6.4 GB at the cfg-2 top block) and spends 90 % of the
// matching time in that bmm; here a 128(dst) x 256(src) score tile lives only in MFMA accumulators.
//
// Arithmetic: v_mfma_f32_32x32x2_f32 -- exact fp32, bit-for-bit the k-ascending fmaf chain of the
// oracle (CDNA4 guide section 3), at the fp32 vector peak (157.3 TFLOP/s), so this kernel is bound by
// the fp32 FMA rate, not by HBM (the operands are 12 kFLOP/byte).
//
// Mapping ("swapped" like a flash-attention QK^T): the MFMA A operand is the DST tile and the B operand
// the SRC tile, so D[i = dst][j = src] puts one src row per lane (j = lane & 31) and 16 dst rows per
// accumulator block in that lane's registers: the row-wise max/argmax is a per-lane running
// (value, index) pair with NO cross-lane traffic inside the loop.  Lanes see their dst indices in
// ascending order, so "strictly greater" keeps the first index (torch max semantics); partial results
// of lanes / waves / workgroups / batch samples (aligned mode) are combined with one 64-bit
// atomicMax on the packed key (orderable(value) << 32 | ~index), which is order-independent.
//
// Data movement (v2): operands are stored as k-panels  [C_pad/8][2][rows][4]  (see vidtome_hip.h):
//   * the dst tile of a K-step (128 rows x 32 channels = 8 panels x 2 KiB) goes HBM/L2 -> LDS directly
//     (global_load_lds_dwordx4, no VGPR round trip, no ds_write), lane-linear = conflict-free for the
//     ds_read_b128 fragment reads, double-buffered, one barrier per K-step;
//   * every wave owns 64 src rows and streams its B fragments straight into registers (two fully
//     coalesced 512-byte segments per load), double-buffered in VGPRs -- no LDS, no sharing needed.
// Per wave and K-step: 128 MFMAs (8192 matrix-pipe cycles) against 4 LDS-DMA issues, 8 global loads and
// 16 LDS reads.
#include "common.h"

namespace {

constexpr int BD = 128;   // dst rows per tile (MFMA A operand, via LDS)
constexpr int BS = 256;   // src rows per workgroup (MFMA B operand, registers): 64 per wave
constexpr int BK = 32;    // channels per pipeline step = 4 groups of 8
constexpr int THREADS = 256;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

__device__ __forceinline__ uint32_t orderable(float f) {
    // monotone map fp32 -> uint32 (NaN largest, -0 == +0)
    if (f != f) return 0xffffffffu;
    const uint32_t u = __float_as_uint(f + 0.0f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// One workgroup: src rows [st*256, st*256+256) of sample `bi`, dst tiles [jt0, jt1).
__global__ __launch_bounds__(THREADS, 2) void match_kernel(
    const float *__restrict__ a, const float *__restrict__ b, int64_t Ns, int64_t Nd, int64_t Ns_pad,
    int64_t Nd_pad, int64_t C_pad, int align, int ns_tiles, int nd_tiles, int nsplit, int tiles_per_split,
    unsigned long long *__restrict__ best) {
    // dst tile of one K-step as 8 panels [g][kh][128 rows][4 floats], double-buffered (2 x 16 KiB)
    __shared__ __attribute__((aligned(16))) float sA[2][8 * BD * 4];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kh = lane >> 5;

    int id = blockIdx.x;
    const int split = id % nsplit;
    id /= nsplit;
    const int st = id % ns_tiles;
    const int bi = id / ns_tiles;
    const int jt0 = split * tiles_per_split;
    const int jt1 = min(jt0 + tiles_per_split, nd_tiles);
    if (jt0 >= jt1) return;

    const int KT = (int)(C_pad / BK);
    const int steps = (jt1 - jt0) * KT;
    const float *srcmat = a + (int64_t)bi * C_pad * Ns_pad;   // panels of the src operand
    const float *dstmat = b + (int64_t)bi * C_pad * Nd_pad;   // panels of the dst operand
    const int64_t srow0 = (int64_t)st * BS + wave * 64;

    // B fragments of one 8-channel group: [src block sb] float4 = k-pairs (8g + 2e + kh), e = 0..3;
    // double-buffered in registers at GROUP granularity (32 MFMAs = 2048 matrix-pipe cycles of cover)
    float4 rb[2][2];
    auto load_b = [&](int gi, float4 (&dst)[2]) {
        const int kt = (gi >> 2) % KT, g = gi & 3;
        const float *p = srcmat + (((int64_t)(kt * 4 + g) * 2 + kh) * Ns_pad + srow0 + l31) * 4;
        dst[0] = *reinterpret_cast<const float4 *>(p);
        dst[1] = *reinterpret_cast<const float4 *>(p + 32 * 4);
    };
    // A tile of one K-step: 16 LDS-DMA wave-instructions of 1 KiB (8 panels x 2 halves of 64 rows); wave w
    // issues 4 of them.  LDS image is lane-linear, i.e. exactly [panel][row][4].
    auto load_a = [&](int s, int buf) {
        const int jt = jt0 + s / KT, kt = s % KT;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int q = wave * 4 + t, p = q >> 1, half = q & 1;
            const float *gp = dstmat + (((int64_t)kt * 8 + p) * Nd_pad + (int64_t)jt * BD + half * 64 + lane) * 4;
            float *lp = &sA[buf][(p * BD + half * 64) * 4];
            __builtin_amdgcn_global_load_lds((glb_void *)gp, (lds_void *)lp, 16, 0, 0);
        }
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int ib = 0; ib < 4; ++ib)
#pragma unroll
        for (int sb = 0; sb < 2; ++sb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ib][sb][r] = 0.0f;

    float bestv[2] = {-INFINITY, -INFINITY};
    uint32_t besti[2] = {0xffffffffu, 0xffffffffu};
    const uint32_t idx_base = align ? (uint32_t)((int64_t)bi * Nd) : 0u;

    load_a(0, 0);
    load_b(0, rb[0]);
    __syncthreads();   // (drains the LDS-DMA: vmcnt(0) + barrier)

    for (int s = 0; s < steps; ++s) {
        const int buf = s & 1;
        if (s + 1 < steps) load_a(s + 1, buf ^ 1);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int gi = s * 4 + g;
            if (gi + 1 < steps * 4) load_b(gi + 1, rb[(g + 1) & 1]);
            float4 af[4];
#pragma unroll
            for (int ib = 0; ib < 4; ++ib)
                af[ib] = *reinterpret_cast<const float4 *>(&sA[buf][((g * 2 + kh) * BD + ib * 32 + l31) * 4]);
            const float4 b0 = rb[g & 1][0], b1 = rb[g & 1][1];
            const float bv[2][4] = {{b0.x, b0.y, b0.z, b0.w}, {b1.x, b1.y, b1.z, b1.w}};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
                for (int ib = 0; ib < 4; ++ib) {
                    const float av = e == 0 ? af[ib].x : e == 1 ? af[ib].y : e == 2 ? af[ib].z : af[ib].w;
#pragma unroll
                    for (int sb = 0; sb < 2; ++sb)
                        acc[ib][sb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[sb][e], acc[ib][sb], 0, 0, 0);
                }
            }
        }
        if ((s + 1) % KT == 0) {
            // dst tile finished: fold the wave's 128 x 64 scores into the per-lane running (max, argmax)
            const int jt = jt0 + s / KT;
            const int dst0 = jt * BD + 4 * kh;
            const bool full = (int64_t)(jt + 1) * BD <= Nd;
#pragma unroll
            for (int sb = 0; sb < 2; ++sb) {
                float bv_ = bestv[sb];
                uint32_t bi_ = besti[sb];
#pragma unroll
                for (int ib = 0; ib < 4; ++ib) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int d = dst0 + ib * 32 + (r & 3) + 8 * (r >> 2);
                        const float sc = acc[ib][sb][r];
                        // s > best, or s is NaN while best is not (first NaN sticks: torch max)
                        bool upd = !(sc <= bv_) && (bv_ == bv_);
                        if (!full) upd = upd && (d < Nd);
                        bv_ = upd ? sc : bv_;
                        bi_ = upd ? (uint32_t)d : bi_;
                        acc[ib][sb][r] = 0.0f;
                    }
                }
                bestv[sb] = bv_;
                besti[sb] = bi_;
            }
        }
        __syncthreads();   // next K-step's dst tile has landed (vmcnt(0)) and this one is fully consumed
    }

    // publish: one atomicMax per (lane, src row); combines lane halves, splits and -- in aligned mode --
    // the samples of the batch (merge.py:96-97)
    const int64_t out_row0 = align ? 0 : (int64_t)bi * Ns;
#pragma unroll
    for (int sb = 0; sb < 2; ++sb) {
        const int64_t srow = srow0 + sb * 32 + l31;
        if (srow < Ns && besti[sb] != 0xffffffffu) {
            const unsigned long long key =
                ((unsigned long long)orderable(bestv[sb]) << 32) | (uint32_t)(~(besti[sb] + idx_base));
            atomicMax(&best[out_row0 + srow], key);
        }
    }
}

}  // namespace

static int launch_match(const float *a, const float *b, int64_t B, int64_t Ns, int64_t Nd, int64_t Ns_pad,
                        int64_t Nd_pad, int64_t C_pad, int align, uint64_t *best, hipStream_t s) {
    const int64_t out_rows = align ? Ns : B * Ns;
    hipError_t e = hipMemsetAsync(best, 0, (size_t)out_rows * sizeof(uint64_t), s);
    if (e != hipSuccess) return vtm::fail(VTM_ELAUNCH, "vtm_match: memset: %s", hipGetErrorString(e));
    const int ns_tiles = (int)(Ns_pad / BS), nd_tiles = (int)(Nd_pad / BD);
    // enough workgroups to fill 256 CUs x 2 resident blocks a few times over, but keep >= 4 dst tiles
    // per block so the running-max epilogue and the atomics stay amortised
    int64_t want = vtm::cdiv(1536, (int64_t)ns_tiles * B);
    int nsplit = (int)(want < 1 ? 1 : want);
    if (nsplit > nd_tiles / 4) nsplit = nd_tiles / 4 > 0 ? nd_tiles / 4 : 1;
    const int tiles_per_split = (int)vtm::cdiv(nd_tiles, nsplit);
    nsplit = (int)vtm::cdiv(nd_tiles, tiles_per_split);
    const int64_t grid = (int64_t)B * ns_tiles * nsplit;
    hipLaunchKernelGGL(match_kernel, dim3((unsigned)grid), dim3(THREADS), 0, s, a, b, Ns, Nd, Ns_pad, Nd_pad,
                       C_pad, align, ns_tiles, nd_tiles, nsplit, tiles_per_split,
                       reinterpret_cast<unsigned long long *>(best));
    return vtm::launch_status("vtm_match");
}

VTM_EXPORT int vtm_match(const float *a, const float *b, int64_t B, int64_t Ns, int64_t Nd,
                         int64_t Ns_pad, int64_t Nd_pad, int64_t C_pad, int align, uint64_t *best,
                         vtm_stream_t stream) {
    VTM_REQUIRE(a && b && best, "vtm_match: null pointer");
    VTM_REQUIRE(B > 0 && Ns > 0 && Nd > 0, "vtm_match: bad sizes");
    VTM_REQUIRE(Ns_pad >= Ns && Ns_pad % BS == 0 && Nd_pad >= Nd && Nd_pad % BD == 0,
                "vtm_match: rows must be padded to a multiple of %d", VTM_MATCH_ROW_PAD);
    VTM_REQUIRE(C_pad > 0 && C_pad % BK == 0, "vtm_match: C_pad must be a multiple of %d", BK);
    VTM_REQUIRE(B * Nd < (1ll << 32) - 1, "vtm_match: index space overflow");
    return launch_match(a, b, B, Ns, Nd, Ns_pad, Nd_pad, C_pad, align, best, vtm::as_stream(stream));
}

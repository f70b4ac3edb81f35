// vtm_match: fused cosine-similarity scoring + row-wise top-1 (the score matrix never exists).
// Reference: vidtome/merge.py:87 `scores = a @ b.transpose(-1, -2)` followed by
//            merge.py:109-113 `node_max, node_idx = scores.max(dim=-1)` (non-aligned) or
//            merge.py:93-97 (aligned: scores of all samples concatenated on the dst axis);
//            same statements at merge.py:392-417 for the global matcher.
// The reference materialises (B, Ns, Nd) fp32 (6.4 GB at the cfg-2 top block) and spends 90 % of the
// matching time in that bmm; here a 128(dst) x 128(src) score tile lives only in MFMA accumulators.
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
#include "common.h"

namespace {

constexpr int BM = 128;  // dst rows per tile (MFMA A operand)
constexpr int BN = 128;  // src rows per tile (MFMA B operand)
constexpr int BK = 32;   // channels per pipeline step
constexpr int LDS_STRIDE = BK + 4;  // 36 floats: 16-byte aligned rows, conflict-free ds_read_b128
constexpr int THREADS = 256;

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ uint32_t orderable(float f) {
    // monotone map fp32 -> uint32 (NaN largest, -0 == +0)
    if (f != f) return 0xffffffffu;
    const uint32_t u = __float_as_uint(f + 0.0f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// One thread block: src tile `st` of sample `bi`, dst tiles [jt0, jt1).
__global__ __launch_bounds__(THREADS, 2) void match_kernel(
    const float *__restrict__ a, const float *__restrict__ b, int64_t Ns, int64_t Nd, int64_t Ns_pad,
    int64_t Nd_pad, int64_t C_pad, int align, int ns_tiles, int nd_tiles, int nsplit, int tiles_per_split,
    unsigned long long *__restrict__ best) {
    __shared__ __attribute__((aligned(16))) float sA[2][BM * LDS_STRIDE];  // dst tile, double-buffered
    __shared__ __attribute__((aligned(16))) float sB[2][BN * LDS_STRIDE];  // src tile

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;  // wave's 64 dst rows / 64 src rows inside the tile
    const int l31 = lane & 31, hi = lane >> 5;

    int id = blockIdx.x;
    const int split = id % nsplit;
    id /= nsplit;
    const int st = id % ns_tiles;
    const int bi = id / ns_tiles;
    const int jt0 = split * tiles_per_split;
    const int jt1 = min(jt0 + tiles_per_split, nd_tiles);
    if (jt0 >= jt1) return;

    const float *srcmat = a + ((int64_t)bi * Ns_pad + (int64_t)st * BN) * C_pad;
    const float *dstmat = b + (int64_t)bi * Nd_pad * C_pad;
    const int KT = (int)(C_pad / BK);
    const int steps = (jt1 - jt0) * KT;

    // staging: thread t moves 4 + 4 float4 per step: rows (t >> 3) + 32 q, 16-byte chunk (t & 7)
    const int ld_row = tid >> 3, ld_chunk = tid & 7;
    float4 ra[4], rb[4];
    auto issue_loads = [&](int s) {
        const int jt = jt0 + s / KT, kt = s % KT;
        const float *pa = dstmat + ((int64_t)jt * BM + ld_row) * C_pad + kt * BK + ld_chunk * 4;
        const float *pb = srcmat + (int64_t)ld_row * C_pad + kt * BK + ld_chunk * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            ra[q] = *reinterpret_cast<const float4 *>(pa + (int64_t)q * 32 * C_pad);
            rb[q] = *reinterpret_cast<const float4 *>(pb + (int64_t)q * 32 * C_pad);
        }
    };
    auto write_lds = [&](int buf) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            *reinterpret_cast<float4 *>(&sA[buf][(ld_row + 32 * q) * LDS_STRIDE + ld_chunk * 4]) = ra[q];
            *reinterpret_cast<float4 *>(&sB[buf][(ld_row + 32 * q) * LDS_STRIDE + ld_chunk * 4]) = rb[q];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ti][tj][r] = 0.0f;

    float bestv[2] = {-INFINITY, -INFINITY};
    uint32_t besti[2] = {0xffffffffu, 0xffffffffu};
    const uint32_t idx_base = align ? (uint32_t)((int64_t)bi * Nd) : 0u;

    issue_loads(0);
    write_lds(0);
    __syncthreads();

    for (int s = 0; s < steps; ++s) {
        const int buf = s & 1;
        if (s + 1 < steps) issue_loads(s + 1);

        const float *pA = &sA[buf][(wr * 64 + l31) * LDS_STRIDE + hi * 4];
        const float *pB = &sB[buf][(wc * 64 + l31) * LDS_STRIDE + hi * 4];
#pragma unroll
        for (int g = 0; g < BK / 8; ++g) {
            // one 16-byte read = this lane's operand for 4 consecutive k-pairs (k-interleaved layout)
            const float4 a0 = *reinterpret_cast<const float4 *>(pA + g * 8);
            const float4 a1 = *reinterpret_cast<const float4 *>(pA + 32 * LDS_STRIDE + g * 8);
            const float4 b0 = *reinterpret_cast<const float4 *>(pB + g * 8);
            const float4 b1 = *reinterpret_cast<const float4 *>(pB + 32 * LDS_STRIDE + g * 8);
            const float av[2][4] = {{a0.x, a0.y, a0.z, a0.w}, {a1.x, a1.y, a1.z, a1.w}};
            const float bv[2][4] = {{b0.x, b0.y, b0.z, b0.w}, {b1.x, b1.y, b1.z, b1.w}};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
                for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                    for (int tj = 0; tj < 2; ++tj)
                        acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ti][e], bv[tj][e],
                                                                          acc[ti][tj], 0, 0, 0);
            }
        }

        if ((s + 1) % KT == 0) {
            // dst tile finished: fold its 64 x 64 wave tile into the per-lane running (max, argmax)
            const int jt = jt0 + s / KT;
            const int dst0 = jt * BM + wr * 64 + 4 * hi;
            const bool full = (int64_t)(jt + 1) * BM <= Nd;
#pragma unroll
            for (int tj = 0; tj < 2; ++tj) {
                float bv_ = bestv[tj];
                uint32_t bi_ = besti[tj];
#pragma unroll
                for (int ti = 0; ti < 2; ++ti) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int d = dst0 + ti * 32 + (r & 3) + 8 * (r >> 2);
                        const float sc = acc[ti][tj][r];
                        // s > best, or s is NaN while best is not (first NaN sticks: torch max)
                        bool upd = !(sc <= bv_) && (bv_ == bv_);
                        if (!full) upd = upd && (d < Nd);
                        bv_ = upd ? sc : bv_;
                        bi_ = upd ? (uint32_t)d : bi_;
                        acc[ti][tj][r] = 0.0f;
                    }
                }
                bestv[tj] = bv_;
                besti[tj] = bi_;
            }
        }

        if (s + 1 < steps) write_lds(buf ^ 1);
        __syncthreads();
    }

    // publish: one atomicMax per (lane, src row); combines lane halves, waves, splits and -- in aligned
    // mode -- the samples of the batch (merge.py:96-97)
    const int64_t out_row0 = align ? 0 : (int64_t)bi * Ns;
#pragma unroll
    for (int tj = 0; tj < 2; ++tj) {
        const int64_t srow = (int64_t)st * BN + wc * 64 + tj * 32 + l31;
        if (srow < Ns && besti[tj] != 0xffffffffu) {
            const unsigned long long key =
                ((unsigned long long)orderable(bestv[tj]) << 32) | (uint32_t)(~(besti[tj] + idx_base));
            atomicMax(&best[out_row0 + srow], key);
        }
    }
}

}  // namespace

VTM_EXPORT int vtm_match(const float *a, const float *b, int64_t B, int64_t Ns, int64_t Nd,
                         int64_t Ns_pad, int64_t Nd_pad, int64_t C_pad, int align, uint64_t *best,
                         vtm_stream_t stream) {
    VTM_REQUIRE(a && b && best, "vtm_match: null pointer");
    VTM_REQUIRE(B > 0 && Ns > 0 && Nd > 0, "vtm_match: bad sizes");
    VTM_REQUIRE(Ns_pad >= Ns && Ns_pad % BN == 0 && Nd_pad >= Nd && Nd_pad % BM == 0,
                "vtm_match: row padding must be a multiple of %d", BM);
    VTM_REQUIRE(C_pad > 0 && C_pad % BK == 0, "vtm_match: C_pad must be a multiple of %d", BK);
    VTM_REQUIRE(B * Nd < (1ll << 32) - 1, "vtm_match: index space overflow");
    hipStream_t s = vtm::as_stream(stream);
    const int64_t out_rows = align ? Ns : B * Ns;
    hipError_t e = hipMemsetAsync(best, 0, (size_t)out_rows * sizeof(uint64_t), s);
    if (e != hipSuccess) return vtm::fail(VTM_ELAUNCH, "vtm_match: memset: %s", hipGetErrorString(e));

    const int ns_tiles = (int)(Ns_pad / BN), nd_tiles = (int)(Nd_pad / BM);
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

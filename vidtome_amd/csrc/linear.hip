// vtm_linear_rows: y = gather(pool, rows) @ W^T (+ bias) -- the projections of the patched block's attention
// (`self.attn1` at vidtome/patch.py:157-162: to_q / to_k / to_v on the merged tokens and to_out[0] on the attention
// output; arithmetic as stated in utils/pnp_utils.py:47-95) fed DIRECTLY by the composed merge map.
//
// The merged tokens are a row selection of the pool [joined chunk | anchor tokens] (all levels merge in "replace"
// mode), so the reference's  merge() -> cat -> Linear  chain is one GEMM whose A rows are fetched through the map:
// the merged tensor is never written, and the live-query rows of a global level are a second, composed map
// (row = rows[rows2[i]]).  V can be produced TRANSPOSED (channel-major), which is what the attention kernel's PV
// contraction reads -- no transpose pass.
//
// GEMM: fp16 / bf16 operands, fp32 accumulation (v_mfma_f32_32x32x16), one rounding at the store.  Workgroup tile
// 128 tokens x 128 output channels, 4 waves of 64 x 64 (2 x 2 accumulator blocks), K walked in steps of 32 through
// a double-buffered LDS ring (register-staged: the loads of step s + 1 are in flight while step s runs on the MFMA).
// Both operands are k-contiguous (token rows, and nn.Linear's (out, in) weight rows), so the same LDS tiles serve
// either MFMA operand slot: the row-major outputs put the WEIGHT rows on the M axis (every lane then owns one token and
// 4 consecutive channels per accumulator group: 8-byte stores along a token row), the transposed output puts the TOKEN
// rows there (a lane owns one channel and 4 consecutive tokens).  K = C is 320 / 640 / 1280: the GEMM is short and
// wide, bound by the A-row gather and the output stream (about one pass over the tokens per 128 output channels, served
// by L2 / MALL), not by the matrix pipe.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 b16x8 __attribute__((ext_vector_type(8)));

constexpr int TM = 128, TN = 128, TK = 32, NT = 256;
constexpr int LDK = TK + 8;   // 80-byte rows: 16-byte aligned, conflict-free ds_read_b128 over 32 rows

template <typename T> struct Mma;
template <> struct Mma<__half> {
    using vec = h16x8;
    __device__ static f32x16 run(vec a, vec b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
    __device__ static __half cvt(float v) { return __float2half_rn(v); }
};
template <> struct Mma<vtm_bf16> {
    using vec = b16x8;
    __device__ static f32x16 run(vec a, vec b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
    __device__ static vtm_bf16 cvt(float v) { return __float2bfloat16(v); }
};

// TRANS = false: out[b][token][channel] (row stride ldo); TRANS = true: out[b][channel][token] (row stride ldo)
template <typename T, bool TRANS>
__global__ __launch_bounds__(NT, 3) void linear_rows_kernel(
    const T *__restrict__ x0, int64_t P0, const T *__restrict__ x1, int64_t P1, int64_t K,
    const int32_t *__restrict__ rows, int64_t rows_ld, const int32_t *__restrict__ rows2, int64_t n,
    const T *__restrict__ W, const T *__restrict__ bias, int64_t N, T *__restrict__ out, int64_t ldo,
    int64_t out_batch_stride) {
    using M = Mma<T>;
    using vec = typename M::vec;
    __shared__ __attribute__((aligned(16))) T sX[2][TM * LDK];
    __shared__ __attribute__((aligned(16))) T sW[2][TN * LDK];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;            // the wave's 64 x 64 quadrant: tokens 64 wm.., channels 64 wn..
    const int64_t m0 = (int64_t)blockIdx.x * TM, n0 = (int64_t)blockIdx.y * TN, b = blockIdx.z;

    // staging: 128 rows x 4 pieces of 16 bytes per operand and K-step = 512 pieces, 2 per thread
    const T *xsrc[2], *wsrc[2];
    int soff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = tid + i * NT, r = c >> 2, piece = c & 3;
        int64_t t = m0 + r;
        if (t >= n) t = n - 1;                            // surplus rows recompute the last one; never stored
        int64_t p = rows2 ? rows2[b * n + t] : t;         // live-query rows: position in the merged sequence ...
        if (rows) p = rows[b * rows_ld + p];              // ... -> pool row id
        xsrc[i] = (p < P0 ? x0 + (b * P0 + p) * K : x1 + (b * P1 + (p - P0)) * K) + piece * 8;
        int64_t ch = n0 + r;
        if (ch >= N) ch = N - 1;
        wsrc[i] = W + ch * K + piece * 8;
        soff[i] = r * LDK + piece * 8;
    }
    uint4 rx[2], rw[2];
    auto issue = [&](int64_t k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            rx[i] = *reinterpret_cast<const uint4 *>(xsrc[i] + k0);
            rw[i] = *reinterpret_cast<const uint4 *>(wsrc[i] + k0);
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *reinterpret_cast<uint4 *>(&sX[buf][soff[i]]) = rx[i];
            *reinterpret_cast<uint4 *>(&sW[buf][soff[i]]) = rw[i];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int steps = (int)(K / TK);
    issue(0);
    stage(0);
    __syncthreads();
    for (int s = 0; s < steps; ++s) {
        const int buf = s & 1;
        if (s + 1 < steps) issue((int64_t)(s + 1) * TK);
        const T *px = &sX[buf][(64 * wm + l31) * LDK + hi * 8];
        const T *pw = &sW[buf][(64 * wn + l31) * LDK + hi * 8];
#pragma unroll
        for (int kk = 0; kk < TK / 16; ++kk) {
            vec fx[2], fw[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                fx[i] = *reinterpret_cast<const vec *>(px + i * 32 * LDK + kk * 16);
                fw[i] = *reinterpret_cast<const vec *>(pw + i * 32 * LDK + kk * 16);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    // acc[i][j]: token block i, channel block j.  MFMA result rows = first operand's rows.
                    if constexpr (TRANS) acc[i][j] = M::run(fx[i], fw[j], acc[i][j]);   // rows = tokens, cols = channels
                    else acc[i][j] = M::run(fw[j], fx[i], acc[i][j]);                   // rows = channels, cols = tokens
                }
        }
        if (s + 1 < steps) {
            stage(buf ^ 1);
            __syncthreads();
        }
    }

    // epilogue.  Accumulator register r of a lane = result row (r & 3) + 8 (r >> 2) + 4 hi, column l31.
    T *ob = out + b * out_batch_stride;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int64_t tok0 = m0 + 64 * wm + 32 * i, ch0 = n0 + 64 * wn + 32 * j;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int rr = 8 * g + 4 * hi;             // first of the 4 consecutive result rows of this group
                T w4[4];
                if constexpr (TRANS) {
                    // rows = tokens tok0 + rr .. + 3, column = channel ch0 + l31: out[ch][tok..tok+3]
                    const int64_t ch = ch0 + l31, tok = tok0 + rr;
                    if (ch < N && tok < n) {
                        const float bv = bias ? vtm::to_f32(bias[ch]) : 0.0f;
#pragma unroll
                        for (int e = 0; e < 4; ++e) w4[e] = M::cvt(acc[i][j][4 * g + e] + bv);
                        T *dst = ob + ch * ldo + tok;
                        if (tok + 3 < n && ((ldo | tok) & 3) == 0) {
                            *reinterpret_cast<uint2 *>(dst) = *reinterpret_cast<const uint2 *>(w4);
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (tok + e < n) dst[e] = w4[e];
                        }
                    }
                } else {
                    // rows = channels ch0 + rr .. + 3, column = token tok0 + l31: out[tok][ch..ch+3]
                    const int64_t tok = tok0 + l31, ch = ch0 + rr;
                    if (tok < n && ch < N) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float bv = (bias && ch + e < N) ? vtm::to_f32(bias[ch + e]) : 0.0f;
                            w4[e] = M::cvt(acc[i][j][4 * g + e] + bv);
                        }
                        T *dst = ob + tok * ldo + ch;
                        if (ch + 3 < N && ((ldo | ch) & 3) == 0) {
                            *reinterpret_cast<uint2 *>(dst) = *reinterpret_cast<const uint2 *>(w4);
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (ch + e < N) dst[e] = w4[e];
                        }
                    }
                }
            }
        }
}

template <typename T>
int launch(const void *x0, int64_t P0, const void *x1, int64_t P1, int64_t B, int64_t K, const int32_t *rows,
           int64_t rows_ld, const int32_t *rows2, int64_t n, const void *W, const void *bias, int64_t N, void *out,
           int64_t ldo, int64_t obs, int transposed, hipStream_t s) {
    const dim3 grid((unsigned)vtm::cdiv(n, TM), (unsigned)vtm::cdiv(N, TN), (unsigned)B), block(NT);
    if (transposed)
        hipLaunchKernelGGL((linear_rows_kernel<T, true>), grid, block, 0, s, (const T *)x0, P0, (const T *)x1, P1, K, rows,
                           rows_ld, rows2, n, (const T *)W, (const T *)bias, N, (T *)out, ldo, obs);
    else
        hipLaunchKernelGGL((linear_rows_kernel<T, false>), grid, block, 0, s, (const T *)x0, P0, (const T *)x1, P1, K, rows,
                           rows_ld, rows2, n, (const T *)W, (const T *)bias, N, (T *)out, ldo, obs);
    return vtm::launch_status("vtm_linear_rows");
}

}  // namespace

VTM_EXPORT int vtm_linear_rows(const void *x0, int64_t P0, const void *x1, int64_t P1, int dtype, int64_t B, int64_t K,
                               const int32_t *rows, int64_t rows_ld, const int32_t *rows2, int64_t n, const void *W,
                               const void *bias, int64_t N, void *out, int64_t ldo, int64_t out_batch_stride,
                               int transposed, vtm_stream_t stream) {
    VTM_REQUIRE(x0 && W && out, "vtm_linear_rows: null pointer");
    VTM_REQUIRE(P1 == 0 || x1, "vtm_linear_rows: x1 is null but P1 > 0");
    VTM_REQUIRE(B > 0 && B < 65536 && K > 0 && N > 0 && n >= 0 && P0 >= 0 && P1 >= 0, "vtm_linear_rows: bad sizes");
    VTM_REQUIRE(K % TK == 0, "vtm_linear_rows: K=%lld must be a multiple of %d", (long long)K, TK);
    VTM_REQUIRE(rows || rows2 || n <= P0 + P1, "vtm_linear_rows: identity rows must lie inside the pool");
    VTM_REQUIRE(!rows || rows_ld > 0, "vtm_linear_rows: rows_ld");
    VTM_REQUIRE(ldo >= (transposed ? n : N), "vtm_linear_rows: ldo too small");
    VTM_REQUIRE(vtm::cdiv(N, TN) < 65536, "vtm_linear_rows: N too large");
    if (n == 0) return VTM_OK;
    hipStream_t s = vtm::as_stream(stream);
    switch (dtype) {
        case VTM_F16:
            return launch<__half>(x0, P0, x1, P1, B, K, rows, rows_ld, rows2, n, W, bias, N, out, ldo, out_batch_stride,
                                  transposed, s);
        case VTM_BF16:
            return launch<vtm_bf16>(x0, P0, x1, P1, B, K, rows, rows_ld, rows2, n, W, bias, N, out, ldo, out_batch_stride,
                                    transposed, s);
    }
    return vtm::fail(VTM_EINVAL, "vtm_linear_rows: dtype must be VTM_F16 or VTM_BF16");
}

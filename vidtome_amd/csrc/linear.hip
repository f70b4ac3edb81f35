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
// GEMM: fp16 / bf16 operands, fp32 accumulation (v_mfma_f32_32x32x16), one rounding at the store.  K = C is only
// 320 / 640 / 1280: short and wide, so the shape of the kernel is set by getting BYTES IN FLIGHT, not by the matrix pipe:
//   * a wave owns 32 token rows and fetches THEIR operand fragments straight from global memory into registers, in the
//     MFMA layout (lane = (row, k-half): 16 bytes per k-step), a whole K-chunk (160 channels = 10 k-steps = 40 VGPRs)
//     at a time and one chunk ahead -- a workgroup keeps 80 KB of token reads outstanding, no LDS round trip, and the
//     gather through the row map costs nothing extra (a lane's row pointer is computed once);
//   * the weight tile (128 output channels x 80 channels of K per step) is the operand all 4 waves share: staged
//     through a double-buffered LDS ring (register-staged, L2-resident source), 20 MFMAs per wave and barrier;
//   * both operands are k-contiguous (token rows, and nn.Linear's (out, in) weight rows), so either can take either
//     MFMA slot: the token-major outputs put the WEIGHT rows on the M axis (a lane then owns one token and 4
//     consecutive channels per accumulator group: 8-byte stores along a token row), the channel-major output puts the
//     TOKEN rows there (a lane owns one channel and 4 consecutive tokens).
// Workgroup tile: 128 tokens x 160 output channels (128 when N is not a multiple of 160), 5 (4) x 16 accumulator
// registers per lane.
#include "common.h"
#include "ablate.h"

#include <atomic>
#include <cstdlib>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 b16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int TM = 128, NT = 256;

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

// TRANS = false: out[b][token][channel] (row stride ldo); TRANS = true: out[b][channel][token] (row stride ldo).
// CH: channels of K a wave holds in registers per chunk; TKW: channels of K per weight-tile step (CH % TKW == 0);
// PAIRS: K / CH is even; TN: output channels per workgroup (160 divides every SD channel count: no ragged last tile).
template <typename T, bool TRANS, int CH, int TKW, bool PAIRS, int TN>
__global__ __launch_bounds__(NT, 2) void linear_rows_kernel(
    const T *__restrict__ x0, int64_t P0, const T *__restrict__ x1, int64_t P1, int64_t K,
    const int32_t *__restrict__ rows, int64_t rows_ld, const int32_t *__restrict__ rows2, int64_t n,
    const T *__restrict__ W, const T *__restrict__ bias, int64_t N, T *__restrict__ out, int64_t ldo,
    int64_t out_batch_stride) {
    using M = Mma<T>;
    using vec = typename M::vec;
    constexpr int LDW = TKW + 8;                       // (TKW / 2 + 4) words = 4 x odd -> conflict-free ds_read_b128
    constexpr int NA = CH / 16;                        // A fragments per chunk
    constexpr int WSTEPS = CH / TKW, KK = TKW / 16;    // weight steps per chunk, k-steps per weight step
    constexpr int WPIECES = TN * (TKW / 8), W_PER_T = (WPIECES + NT - 1) / NT, NJ = TN / 32;
    static_assert(CH % TKW == 0 && TKW % 16 == 0 && ((TKW / 2 + 4) / 4) % 2 == 1, "tile shape");
    __shared__ __attribute__((aligned(16))) T sW[2][TN * LDW];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int64_t m0 = (int64_t)blockIdx.x * TM, n0 = (int64_t)blockIdx.y * TN, b = blockIdx.z;

    // this lane's token row (A operand), fetched through the map(s)
    const T *xrow;
    {
        int64_t t = m0 + 32 * wave + l31;
        if (t >= n) t = n - 1;                            // surplus rows recompute the last one; never stored
        int64_t p = rows2 ? rows2[b * n + t] : t;         // live-query rows: position in the merged sequence ...
        if (rows) p = rows[b * rows_ld + p];              // ... -> pool row id
        xrow = (p < P0 ? x0 + (b * P0 + p) * K : x1 + (b * P1 + (p - P0)) * K) + hi * 8;
    }
    // weight tile staging: TN rows x TKW/8 pieces of 16 bytes per step
    const T *wsrc[W_PER_T];
    int woff[W_PER_T];
    bool wok[W_PER_T];
#pragma unroll
    for (int i = 0; i < W_PER_T; ++i) {
        const int c = tid + i * NT;
        wok[i] = c < WPIECES;
        const int r = wok[i] ? c / (TKW / 8) : 0, piece = wok[i] ? c % (TKW / 8) : 0;
        int64_t ch = n0 + r;
        if (ch >= N) ch = N - 1;
        wsrc[i] = W + ch * K + piece * 8;
        woff[i] = r * LDW + piece * 8;
    }
    u32x4 rw[W_PER_T];
    auto issue_w = [&](int64_t k0) {
#pragma unroll
        for (int i = 0; i < W_PER_T; ++i) {
            ABL_LIN_W(rw[i] = *reinterpret_cast<const u32x4 *>(wsrc[i] + k0);)
            ABL_NO_LIN_W(rw[i] = u32x4{(unsigned)k0, 1u, 2u, 3u};)
        }
    };
    auto stage_w = [&](int buf) {
#pragma unroll
        for (int i = 0; i < W_PER_T; ++i)
            if (wok[i]) *reinterpret_cast<u32x4 *>(&sW[buf][woff[i]]) = rw[i];
    };
    // (ABL_* = the ablation switches of ablate.h: in the shipped build each expands to the code it wraps and nothing else)
    auto load_a = [&](vec (&a)[NA], int64_t k0) {
#pragma unroll
        for (int f = 0; f < NA; ++f) {
            ABL_LIN_A(a[f] = *reinterpret_cast<const vec *>(xrow + k0 + f * 16);)
            ABL_NO_LIN_A(for (int e = 0; e < 8; ++e) a[f][e] = (decltype(a[f][e] + a[f][e]))(k0 + f);)
        }
    };

    f32x16 acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;

    const int nchunks = (int)(K / CH);
    const int64_t k_last = K - TKW;
    int wbuf = 0;
    // one K-chunk: `cur` holds this chunk's A fragments, `nxt` receives the next chunk's while this one runs.  Every
    // load is issued unconditionally (the last chunk / step re-fetches itself, unused): with a fixed issue sequence the
    // compiler's waits are counted ones, a conditional prefetch makes it drain everything right after issuing it.
    auto run_chunk = [&](vec (&cur)[NA], vec (&nxt)[NA], int c) {
#pragma unroll
        for (int ws = 0; ws < WSTEPS; ++ws) {
            const int64_t k_next = (int64_t)c * CH + (int64_t)(ws + 1) * TKW;
            issue_w(k_next < k_last ? k_next : k_last);
            // the token prefetch goes BEHIND the first weight loads of the chunk: vmcnt retires in order, so the wait
            // for this step's weight tile then leaves the (younger) token loads in flight
            if (ws == 0) load_a(nxt, (int64_t)(c + 1 < nchunks ? c + 1 : c) * CH);
            // keep the loads HERE: left alone, the scheduler sinks them below the MFMAs (fewer live registers) and the
            // step then waits out their whole latency
            __builtin_amdgcn_sched_barrier(0);
            const T *pw = &sW[wbuf][l31 * LDW + hi * 8];
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
                vec fw[NJ];
#pragma unroll
                for (int j = 0; j < NJ; ++j) fw[j] = *reinterpret_cast<const vec *>(pw + j * 32 * LDW + kk * 16);
                const vec fx = cur[ws * KK + kk];
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    // MFMA result rows = first operand's rows
                    if constexpr (TRANS) acc[j] = M::run(fx, fw[j], acc[j]);   // rows = tokens, cols = channels
                    else acc[j] = M::run(fw[j], fx, acc[j]);                   // rows = channels, cols = tokens
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            stage_w(wbuf ^ 1);
            __syncthreads();
            wbuf ^= 1;
        }
    };

    vec a0[NA], a1[NA];
    load_a(a0, 0);
    issue_w(0);
    stage_w(0);
    __syncthreads();
    if constexpr (PAIRS) {               // even number of chunks: no tail logic at all
        for (int c = 0; c < nchunks; c += 2) {
            run_chunk(a0, a1, c);
            run_chunk(a1, a0, c + 1);
        }
    } else {
        for (int c = 0; c < nchunks; c += 2) {
            run_chunk(a0, a1, c);
            if (c + 1 < nchunks) run_chunk(a1, a0, c + 1);
        }
    }

    // epilogue.  Accumulator register r of a lane = result row (r & 3) + 8 (r >> 2) + 4 hi, column l31: a lane owns
    // 4 consecutive elements along one output row and 32 lanes sit on 32 DIFFERENT rows -- stored directly that is 8
    // bytes per lane at a row stride, which costs more than the whole GEMM (measured: 107 -> 50 us at the cfg-2 top
    // block without the stores).  The tile therefore goes through LDS (the weight ring is free by now) and leaves as
    // 16-byte pieces of whole rows: 320 contiguous bytes per token (token-major), 256 per channel (channel-major).
    ABL_NO_LIN_STORE(if (acc[0][0] != 12345.678f) return;)
    T *ob = out + b * out_batch_stride;
    const int64_t tok0 = m0 + 32 * wave;
    constexpr bool STAGED = sizeof(sW) >= (size_t)(TRANS ? TN * (TM + 8) : 4 * 32 * (TN + 8)) * sizeof(T);
    const bool vec_ok = ((ldo & 7) == 0) && ((reinterpret_cast<uintptr_t>(ob) & 15) == 0);
    if constexpr (STAGED) {
        __syncthreads();                                   // every wave is done with the weight ring
        T *stage = &sW[0][0];
        if constexpr (!TRANS) {
            constexpr int SO = TN + 8;                     // (TN / 2 + 4) words = 4 x odd: conflict-free b64 writes
            T *my = stage + wave * 32 * SO;                // this wave's 32 tokens x TN channels
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int rr = 8 * g + 4 * hi;
                    const int64_t ch = n0 + 32 * j + rr;
                    T w4[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float bv = (bias && ch + e < N) ? vtm::to_f32(bias[ch + e]) : 0.0f;
                        w4[e] = M::cvt(acc[j][4 * g + e] + bv);
                    }
                    *reinterpret_cast<uint2 *>(my + l31 * SO + 32 * j + rr) = *reinterpret_cast<const uint2 *>(w4);
                }
            __builtin_amdgcn_wave_barrier();               // LDS is FIFO per wave: the reads below see the writes above
            constexpr int PR = TN / 8;                     // 16-byte pieces per token row
            static_assert((32 * PR) % 64 == 0, "whole wave-instructions");
#pragma unroll
            for (int it = 0; it < 32 * PR / 64; ++it) {
                const int p = lane + 64 * it;
                const int row = p / PR, c8 = (p % PR) * 8;
                const int64_t tok = tok0 + row, ch = n0 + c8;
                if (tok < n && ch < N) {
                    const uint4 v = *reinterpret_cast<const uint4 *>(my + row * SO + c8);
                    T *dst = ob + tok * ldo + ch;
                    if (ch + 8 <= N && vec_ok) {
                        *reinterpret_cast<uint4 *>(dst) = v;
                    } else {
                        const T *e8 = reinterpret_cast<const T *>(&v);
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            if (ch + e < N) dst[e] = e8[e];
                    }
                }
            }
        } else {
            constexpr int SO = TM + 8;                     // channel rows of the whole 128-token tile
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int64_t ch = n0 + 32 * j + l31;
                const float bv = (bias && ch < N) ? vtm::to_f32(bias[ch]) : 0.0f;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int rr = 8 * g + 4 * hi;
                    T w4[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) w4[e] = M::cvt(acc[j][4 * g + e] + bv);
                    *reinterpret_cast<uint2 *>(stage + (32 * j + l31) * SO + 32 * wave + rr) = *reinterpret_cast<const uint2 *>(w4);
                }
            }
            __syncthreads();
            constexpr int PR = TM / 8;                     // 16 pieces per channel row
            static_assert((TN * PR) % NT == 0, "whole workgroup passes");
#pragma unroll
            for (int it = 0; it < TN * PR / NT; ++it) {
                const int p = tid + NT * it;
                const int row = p / PR, c8 = (p % PR) * 8;
                const int64_t ch = n0 + row, tok = m0 + c8;
                if (ch < N && tok < n) {
                    const uint4 v = *reinterpret_cast<const uint4 *>(stage + row * SO + c8);
                    T *dst = ob + ch * ldo + tok;
                    if (tok + 8 <= n && vec_ok) {
                        *reinterpret_cast<uint4 *>(dst) = v;
                    } else {
                        const T *e8 = reinterpret_cast<const T *>(&v);
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            if (tok + e < n) dst[e] = e8[e];
                    }
                }
            }
        }
    } else {
        // small-K instantiation (its weight ring is too small to hold the tile): direct 8-byte stores
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int64_t ch0 = n0 + 32 * j;
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
                        for (int e = 0; e < 4; ++e) w4[e] = M::cvt(acc[j][4 * g + e] + bv);
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
                            w4[e] = M::cvt(acc[j][4 * g + e] + bv);
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
}

// ---- K = 320 (the cfg-2 / cfg-5 top blocks: 3 of a step's 4 projection GEMMs per site): weights stay in LDS --------------
// A 128 x 160 tile of the kernel above re-stages its 100 KB weight half for every 128 tokens (4 barrier-separated steps whose
// 400 ns of MFMAs cannot cover a global-memory round trip): 69 us where the traffic floor is 27.  Here a workgroup (8 waves,
// one per CU) loads ITS half of W (160 output channels x 320, 105 KB with padding) once and keeps it; after that its waves
// are independent -- no barrier in the loop: each wave takes 32-token blocks (block w, w + 8, ... of the workgroup's span),
// fetches their operand fragments straight from global memory half a block ahead (2 x 10 k-steps = 80 VGPRs), runs the
// 100 MFMAs of a block against LDS fragments, and writes the block out through a small LDS staging area of its own
// (16-byte pieces of whole rows, as above).  LDS fragment reads and MFMAs are 1 : 1 (32 tokens per wave: the 64-token
// variant does not fit next to 80 accumulator registers), which is what bounds it.
constexpr int WS_K = 320, WS_TN = 160, WS_LDW = WS_K + 8;   // 164 words per row = 4 x odd: conflict-free b128
constexpr int WS_W_BYTES = WS_TN * WS_LDW * 2;
#ifndef VTM_WS_WAVES
#define VTM_WS_WAVES 8
#endif
constexpr int WS_NW = VTM_WS_WAVES;                  // waves per workgroup: 8 (two per SIMD) or 12 (three, <= 168 VGPRs)
constexpr int WS_NT = 64 * WS_NW;
constexpr int WS_KCH = WS_NW == 8 ? 10 : 5;          // k-steps per operand chunk (two chunks in registers)
constexpr int WS_NC = 20 / WS_KCH;
constexpr int WS_TP = WS_NW == 8 ? 16 : 8;           // tokens per staging phase
constexpr int WS_STAGE_TOK = WS_TP * (WS_TN + 8) * 2;   // per wave: WS_TP tokens x 168 elements (token-major phase); the
                                                        // channel-major phase (32 channels x 40 elements = 2560 B) fits inside
static_assert(WS_STAGE_TOK >= 32 * 40 * 2, "channel-major staging");
constexpr int WS_LDS_BYTES = WS_W_BYTES + WS_NW * WS_STAGE_TOK;

template <typename T, bool TRANS>
__global__ __launch_bounds__(WS_NT, 1) void linear_rows_ws_kernel(
    const T *__restrict__ x0, int64_t P0, const T *__restrict__ x1, int64_t P1, const int32_t *__restrict__ rows,
    int64_t rows_ld, const int32_t *__restrict__ rows2, int64_t n, const T *__restrict__ W, const T *__restrict__ bias,
    int64_t N, T *__restrict__ out, int64_t ldo, int64_t out_batch_stride, int blocks_per_wave, int spans, int64_t B) {
    using M = Mma<T>;
    using vec = typename M::vec;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T *sW = reinterpret_cast<T *>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    // Workgroup -> (row span, channel half, sample).  Every half of a row span reads the same token rows: the halves of a span
    // are placed on the SAME XCD (workgroup L runs on XCD L % 8), next to each other in dispatch order, so that the second reader
    // finds the rows in that XCD's L2 instead of fetching them from HBM again (two 160-channel halves = twice the operand bytes).
    const int halves = (int)(N / WS_TN);
    const int64_t slot = blockIdx.x >> 3, q = (slot / halves) * 8 + (blockIdx.x & 7);
    if (q >= (int64_t)spans * B) return;
    const int64_t n0 = (slot % halves) * WS_TN, b = q / spans, span = q % spans;
    T *stage = reinterpret_cast<T *>(smem + WS_W_BYTES + wave * WS_STAGE_TOK);

    // the workgroup's weight half: 160 rows x 40 pieces of 16 bytes
    for (int c = tid; c < WS_TN * (WS_K / 8); c += WS_NT) {
        const int r = c / (WS_K / 8), piece = c % (WS_K / 8);
        *reinterpret_cast<u32x4 *>(&sW[r * WS_LDW + piece * 8]) =
            *reinterpret_cast<const u32x4 *>(W + (n0 + r) * WS_K + piece * 8);
    }
    __syncthreads();

    const int64_t RB = (n + 31) / 32;
    const int64_t blk0 = span * blocks_per_wave * WS_NW + wave;              // this wave: blk0, blk0 + WS_NW, ...
    if (blk0 >= RB) return;                                                  // (no barrier below)
    int nblk = (int)((RB - blk0 + WS_NW - 1) / WS_NW);
    if (nblk > blocks_per_wave) nblk = blocks_per_wave;

    auto row_ptr = [&](int64_t blk) -> const T * {
        int64_t t = blk * 32 + l31;
        if (t >= n) t = n - 1;                            // surplus rows recompute the last one; never stored
        int64_t p = rows2 ? rows2[b * n + t] : t;
        if (rows) p = rows[b * rows_ld + p];
        return (p < P0 ? x0 + (b * P0 + p) * WS_K : x1 + (b * P1 + (p - P0)) * WS_K) + hi * 8;
    };
    vec a[2][WS_KCH];                                     // two operand chunks of WS_KCH k-steps: this one and the next
    auto load_chunk = [&](const T *xr, int c) {           // chunk c of a block into buffer c & 1
#pragma unroll
        for (int f = 0; f < WS_KCH; ++f) {
            ABL_LIN_A(a[c & 1][f] = *reinterpret_cast<const vec *>(xr + (c * WS_KCH + f) * 16);)
            ABL_NO_LIN_A(for (int e = 0; e < 8; ++e) a[c & 1][f][e] = (decltype(a[c & 1][f][e] + a[c & 1][f][e]))(f + c);)
        }
    };
    const T *xr = row_ptr(blk0);
    load_chunk(xr, 0);
    load_chunk(xr, 1);

    T *ob = out + b * out_batch_stride;
    const T *pw = &sW[l31 * WS_LDW + hi * 8];
    for (int i = 0; i < nblk; ++i) {
        const int64_t blk = blk0 + WS_NW * i;
        // the last block re-fetches itself (never used): a fixed issue sequence keeps the compiler's waits counted
        const T *xn = row_ptr(i + 1 < nblk ? blk + WS_NW : blk);
        f32x16 acc[5];
#pragma unroll
        for (int j = 0; j < 5; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
#pragma unroll
        for (int c = 0; c < WS_NC; ++c) {
#pragma unroll
            for (int kk = 0; kk < WS_KCH; ++kk) {
                vec fw[5];
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    ABL_LIN_LDS(fw[j] = *reinterpret_cast<const vec *>(pw + j * 32 * WS_LDW + (c * WS_KCH + kk) * 16);)
                    ABL_NO_LIN_LDS(fw[j] = a[c & 1][(kk + j) % WS_KCH];)
                }
                const vec fx = a[c & 1][kk];
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    if constexpr (TRANS) acc[j] = M::run(fx, fw[j], acc[j]);   // rows = tokens, cols = channels
                    else acc[j] = M::run(fw[j], fx, acc[j]);                   // rows = channels, cols = tokens
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            // two chunks ahead, into the registers this one just freed: the rest of this block, then the next block's
            if (c + 2 < WS_NC) load_chunk(xr, c + 2);
            else load_chunk(xn, c + 2 - WS_NC);
            __builtin_amdgcn_sched_barrier(0);
        }

        const int64_t tok0 = blk * 32;
        ABL_NO_LIN_STORE(if (acc[0][0] != 12345.678f) continue;)
        if constexpr (!TRANS) {
            constexpr int SO = WS_TN + 8;
#pragma unroll
            for (int ph = 0; ph < 32 / WS_TP; ++ph) {     // WS_TP tokens at a time
                if (l31 / WS_TP == ph) {
#pragma unroll
                    for (int j = 0; j < 5; ++j)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int rr = 8 * g + 4 * hi;
                            const int64_t ch = n0 + 32 * j + rr;
                            T w4[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) w4[e] = M::cvt(acc[j][4 * g + e] + (bias ? vtm::to_f32(bias[ch + e]) : 0.0f));
                            *reinterpret_cast<uint2 *>(stage + (l31 % WS_TP) * SO + 32 * j + rr) = *reinterpret_cast<const uint2 *>(w4);
                        }
                }
                __builtin_amdgcn_wave_barrier();          // LDS is FIFO per wave: the reads below see the writes above
                constexpr int PR = WS_TN / 8;             // 20 pieces per token row
#pragma unroll
                for (int it = 0; it < (WS_TP * PR + 63) / 64; ++it) {
                    const int p = lane + 64 * it;
                    const int row = p / PR, c8 = (p % PR) * 8;
                    const int64_t tok = tok0 + WS_TP * ph + row;
                    if (row < WS_TP && tok < n)
                        *reinterpret_cast<uint4 *>(ob + tok * ldo + n0 + c8) = *reinterpret_cast<const uint4 *>(stage + row * SO + c8);
                }
                __builtin_amdgcn_wave_barrier();          // ... and the next phase's writes follow these reads
            }
        } else {
            constexpr int SO = 32 + 8;
#pragma unroll
            for (int j = 0; j < 5; ++j) {                 // 32 channels at a time
                const int64_t ch = n0 + 32 * j + l31;
                const float bv = bias ? vtm::to_f32(bias[ch]) : 0.0f;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    T w4[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) w4[e] = M::cvt(acc[j][4 * g + e] + bv);
                    *reinterpret_cast<uint2 *>(stage + l31 * SO + 8 * g + 4 * hi) = *reinterpret_cast<const uint2 *>(w4);
                }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int it = 0; it < 2; ++it) {          // 32 channel rows x 4 pieces of 8 tokens
                    const int p = lane + 64 * it;
                    const int row = p >> 2, c8 = (p & 3) * 8;
                    const int64_t tok = tok0 + c8;
                    if (tok < n) {
                        const uint4 v = *reinterpret_cast<const uint4 *>(stage + row * SO + c8);
                        T *dst = ob + (n0 + 32 * j + row) * ldo + tok;
                        if (tok + 8 <= n) {
                            *reinterpret_cast<uint4 *>(dst) = v;
                        } else {
                            const T *e8 = reinterpret_cast<const T *>(&v);
#pragma unroll
                            for (int e = 0; e < 8; ++e)
                                if (tok + e < n) dst[e] = e8[e];
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        xr = xn;
    }
}

template <typename T, bool TRANS>
int launch_ws(const void *x0, int64_t P0, const void *x1, int64_t P1, int64_t B, const int32_t *rows, int64_t rows_ld,
              const int32_t *rows2, int64_t n, const void *W, const void *bias, int64_t N, void *out, int64_t ldo, int64_t obs,
              hipStream_t s) {
    static std::atomic<bool> attr_set[vtm::MAX_DEVICES];
    const int dev = vtm::current_device();
    if (!attr_set[dev].load(std::memory_order_acquire)) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(linear_rows_ws_kernel<T, TRANS>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, WS_LDS_BYTES);
        if (e != hipSuccess) return vtm::fail(VTM_ELAUNCH, "vtm_linear_rows: LDS attribute: %s", hipGetErrorString(e));
        attr_set[dev].store(true, std::memory_order_release);
    }
    // every wave gets the same number of 32-token blocks, as few as fill the chip's CUs once
    const int64_t RB = vtm::cdiv(n, 32), halves = N / WS_TN;
    int64_t bpw = vtm::cdiv(RB * halves * B, (int64_t)vtm::device_cus() * WS_NW);
    if (bpw < 1) bpw = 1;
    const int64_t spans = vtm::cdiv(RB, WS_NW * bpw);                  // row spans per sample
    const dim3 grid((unsigned)(8 * vtm::cdiv(spans * B, 8) * halves)), block(WS_NT);
    hipLaunchKernelGGL((linear_rows_ws_kernel<T, TRANS>), grid, block, WS_LDS_BYTES, s, (const T *)x0, P0, (const T *)x1, P1,
                       rows, rows_ld, rows2, n, (const T *)W, (const T *)bias, N, (T *)out, ldo, obs, (int)bpw, (int)spans, B);
    return vtm::launch_status("vtm_linear_rows");
}

template <typename T>
int launch(const void *x0, int64_t P0, const void *x1, int64_t P1, int64_t B, int64_t K, const int32_t *rows,
           int64_t rows_ld, const int32_t *rows2, int64_t n, const void *W, const void *bias, int64_t N, void *out,
           int64_t ldo, int64_t obs, int transposed, hipStream_t s) {
    static const bool force_tiled = getenv("VTM_LINEAR_TILED") != nullptr;   // A/B hook, read once per process
    // K = 320 with whole 160-channel halves and 16-byte-aligned output rows: the weight-stationary kernel
    if (K == WS_K && N % WS_TN == 0 && (ldo & 7) == 0 && (obs & 7) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0 &&
        vtm::cdiv(n, 32) * B * (N / WS_TN) < (1ll << 30) && !force_tiled) {
        return transposed ? launch_ws<T, true>(x0, P0, x1, P1, B, rows, rows_ld, rows2, n, W, bias, N, out, ldo, obs, s)
                          : launch_ws<T, false>(x0, P0, x1, P1, B, rows, rows_ld, rows2, n, W, bias, N, out, ldo, obs, s);
    }
    const int tn = (N % 160 == 0) ? 160 : 128;
    const dim3 grid((unsigned)vtm::cdiv(n, TM), (unsigned)vtm::cdiv(N, tn), (unsigned)B), block(NT);
#define VTM_LIN(TR, CH_, TKW_, PAIRS_, TN_)                                                                            \
    hipLaunchKernelGGL((linear_rows_kernel<T, TR, CH_, TKW_, PAIRS_, TN_>), grid, block, 0, s, (const T *)x0, P0,     \
                       (const T *)x1, P1, K, rows, rows_ld, rows2, n, (const T *)W, (const T *)bias, N, (T *)out, ldo, obs)
#define VTM_LIN_TN(TR, CH_, TKW_, PAIRS_)                                                                             \
    do {                                                                                                              \
        if (tn == 160) VTM_LIN(TR, CH_, TKW_, PAIRS_, 160); else VTM_LIN(TR, CH_, TKW_, PAIRS_, 128);                 \
    } while (0)
    if (K % 320 == 0) {                  // the SD channel counts (320, 640, 1280): chunks of 160, an even number of them
        if (transposed) VTM_LIN_TN(true, 160, 80, true); else VTM_LIN_TN(false, 160, 80, true);
    } else {                             // any other multiple of 32
        if (transposed) VTM_LIN_TN(true, 32, 32, false); else VTM_LIN_TN(false, 32, 32, false);
    }
#undef VTM_LIN_TN
#undef VTM_LIN
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
    VTM_REQUIRE(K % 32 == 0, "vtm_linear_rows: K=%lld must be a multiple of 32", (long long)K);
    VTM_REQUIRE(rows || rows2 || n <= P0 + P1, "vtm_linear_rows: identity rows must lie inside the pool");
    VTM_REQUIRE(!rows || rows_ld > 0, "vtm_linear_rows: rows_ld");
    VTM_REQUIRE(ldo >= (transposed ? n : N), "vtm_linear_rows: ldo too small");
    VTM_REQUIRE(vtm::cdiv(N, 128) < 65536, "vtm_linear_rows: N too large");
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

namespace vtm {
int linear_ablations() { return VTM_ABLATIONS; }
}  // namespace vtm

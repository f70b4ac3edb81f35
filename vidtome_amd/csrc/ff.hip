// vtm_ff_*: the feed-forward of the patched block (vidtome/patch.py:187-199: `norm_hidden_states = self.norm3(hidden_states)`,
// `ff_output = self.ff(norm_hidden_states)`, `hidden_states = ff_output + hidden_states`; the Diffusers FeedForward of SD
// blocks is GEGLU(Linear C -> 8C) -> Linear 4C -> C) and the query projection of its cross-attention (patch.py:171-185), as
// "panel GEMMs".
//
// Both operands of these GEMMs are k-contiguous row sets (token rows; nn.Linear's (out, in) weight rows) and every one of
// them is PRODUCED by a kernel of this library or is a constant: so they are kept in the k-panel layout of the matcher
// (match_filter.hip): [k / 8][row][8 elements], rows padded to 256 -- a wave's 64 fragment loads are 1 KiB contiguous, a
// weight tile is 16 LDS-DMA pieces of 1 KiB, and the epilogue's stores into the NEXT GEMM's panel operand are contiguous
// too (a lane pair holds the 8 channels of one panel entry of one token).  norm3 / norm2 write their output in this
// layout (vtm_layernorm_panels), GEMM 1 applies the gated activation in its epilogue and writes the 4C-wide result as
// panels again -- the 8C-wide projection output, the largest tensor of the block (cfg-2 top site: 671 MB), is never
// written -- and GEMM 2 leaves as token rows with bias and residual added.
//
// The main loop is the one-product loop of match_filter.hip's filter_kernel (workgroup tile 256 token rows, held as
// register fragments fetched two k-groups ahead, x 128 weight rows through a double-buffered LDS-DMA ring; every vector
// memory operation issued by hand between two MFMAs with counted waits; 1.0-1.1 PFLOP/s at K = 320 on the matcher's
// shapes): see the comments there for the pipeline and its rules.  Only the per-tile epilogue differs.
#include "common.h"

#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace {

constexpr int FBD = 128;      // weight rows (output channels) per tile (MFMA A operand, LDS)
constexpr int FBS = 256;      // token rows per workgroup (B operand, registers), 64 per wave
constexpr int FBK = 64;       // channels per pipeline step = 4 MFMA k-steps = 8 panels
constexpr int THREADS = 256;
constexpr int MAX_TILES_PER_WG = 16;   // bias slice kept in LDS

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 b16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

template <typename T> struct Mma;
template <> struct Mma<__half> {
    __device__ static f32x16 run(u32x4 a, u32x4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, a), __builtin_bit_cast(h16x8, b), c, 0, 0, 0);
    }
    __device__ static uint16_t bits(float v) { return __half_as_ushort(__float2half_rn(v)); }
    __device__ static float round(float v) { return __half2float(__float2half_rn(v)); }
};
template <> struct Mma<vtm_bf16> {
    __device__ static f32x16 run(u32x4 a, u32x4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(b16x8, a), __builtin_bit_cast(b16x8, b), c, 0, 0, 0);
    }
    __device__ static uint16_t bits(float v) { return __bfloat16_as_ushort(__float2bfloat16(v)); }
    __device__ static float round(float v) { return __bfloat162float(__float2bfloat16(v)); }
};

// 0.5 g (1 + erf(g / sqrt 2)) with erfc(|x|) = t (a1 + t (a2 + t (a3 + t (a4 + t a5)))) e^{-x^2}, t = 1 / (1 + p |x|)
// (Abramowitz & Stegun 7.1.26, |error| <= 1.5e-7 absolute): no cancellation on the negative side, where the exact value
// is erfc itself; far below the rounding of the 16-bit result (torch's GELU here is the erf form: `approximate="none"`).
__device__ __forceinline__ float gelu_erf(float g) {
    const float x = __builtin_fabsf(g) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, x, 1.0f));
    float p = __builtin_fmaf(1.061405429f, t, -1.453152027f);
    p = __builtin_fmaf(p, t, 1.421413741f);
    p = __builtin_fmaf(p, t, -0.284496736f);
    p = __builtin_fmaf(p, t, 0.254829592f);
    const float e = p * t * __builtin_amdgcn_exp2f(-x * x * 1.4426950408889634f);   // erfc(|x|)
    return 0.5f * g * (g < 0.0f ? e : 2.0f - e);
}

enum { EPI_GEGLU = 0, EPI_ROWS = 1 };

struct Epi {
    const float *bias;     // (weight rows of the panel operand) fp32, or null
    void *out;             // GEGLU: panels [D / 8][out_rows_pad][8];  ROWS: (n, ldo) token rows
    int64_t out_rows_pad;  // GEGLU
    int64_t ldo;           // ROWS
    const void *resid;     // ROWS: (n, ldo) added to the result, or null
    int64_t n;             // valid token rows
    int64_t N;             // ROWS: valid output channels
    int64_t nbias;         // entries of `bias` (the last weight tile may reach beyond them)
};

// Weight-row order of a GEGLU tile (prepared once by the host, vtm_ff_pack_geglu): tile t holds the VALUE rows of output
// channels 64 t .. 64 t + 63 in its first two 32-row blocks and their GATE rows in the last two, so that a lane finds the
// gate of every value it holds in the same register of accumulator block ib + 2.
template <typename T, int EPI>
__global__ __launch_bounds__(THREADS, 2) void panel_gemm_kernel(
    const uint4 *__restrict__ ah, const uint4 *__restrict__ bh, int64_t Ns_pad, int64_t Nd_pad, int64_t C_pad,
    int ns_tiles, int nd_tiles, int nsplit, int tiles_per_split, int total_src_tiles, int patch_tiles, Epi E) {
    __shared__ __attribute__((aligned(16))) uint4 sA[2][8 * FBD];           // weight tile ring: 2 x 16 KiB
    __shared__ float sBias[MAX_TILES_PER_WG * FBD];
    // ROWS: per-wave staging of 32 tokens x 128 channels (+ 8 of padding: conflict-free 8-byte writes)
    __shared__ __attribute__((aligned(16))) uint16_t sOut[EPI == EPI_ROWS ? 4 * 32 * (FBD + 8) : 8];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kh = lane >> 5;

    // XCD-aware work mapping, as in filter_kernel: an XCD works through patches of `patch_tiles` token tiles x all
    // splits of the weight rows, so the token fragments (re-read for every weight tile) stay in that XCD's L2
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int per_group = patch_tiles * nsplit;
    const int grp = (slot / per_group) * 8 + xcd;
    const int q = slot % per_group;
    const int stg = grp * patch_tiles + q % patch_tiles;
    const int split = q / patch_tiles;
    if (stg >= total_src_tiles) return;
    const int st_ = stg % ns_tiles;
    const int jt0 = split * tiles_per_split;
    const int jt1 = min(jt0 + tiles_per_split, nd_tiles);
    if (jt0 >= jt1) return;

    const int KT = (int)(C_pad / FBK);
    const int steps = (jt1 - jt0) * KT;
    const uint4 *srch = ah, *dsth = bh;
    const int64_t srow0 = (int64_t)st_ * FBS + wave * 64;

    // The epilogue's parameters live in LDS, not in SGPRs: the hand-issued loads of the loop take their wave-uniform
    // addresses as "s" operands, and with the epilogue's scalars live across the loop the compiler runs out of SGPRs and
    // moves those addresses to VGPRs (which the asm cannot take).  The bias slice of this workgroup's tiles goes there too
    // (a global load inside the epilogue would make the compiler drain the loop's prefetches).
    __shared__ Epi sE;
    if (tid == 0) sE = E;
    {   // (unrolled and predicated, not a loop with a per-thread trip count: behind such a loop the compiler no longer
        // keeps the wave-uniform addresses of the hand-issued loads below in SGPRs)
        const int cnt = (jt1 - jt0) * FBD;
        const float *bsrc = E.bias != nullptr ? E.bias + (int64_t)jt0 * FBD : nullptr;
#pragma unroll
        for (int k = 0; k < MAX_TILES_PER_WG * FBD / THREADS; ++k) {
            const int i = tid + k * THREADS;
            if (i < cnt) sBias[i] = (bsrc != nullptr && (int64_t)jt0 * FBD + i < E.nbias) ? bsrc[i] : 0.0f;
        }
    }

    constexpr int NB = 2, PG = 2;
    const uint32_t voff_b = (uint32_t)(kh * Ns_pad + srow0 + l31) * 16u;
    const int64_t bgroup = 2 * Ns_pad;
    u32x4 rb[4][2];
    auto await_b = [&](auto count_tag, u32x4 (&r)[2]) {
        constexpr int N = decltype(count_tag)::value;
        asm volatile("s_waitcnt vmcnt(%2)" : "+v"(r[0]), "+v"(r[1]) : "n"(N));
    };
    const uint32_t voff_a = (uint32_t)lane * 16u;
    f32x16 acc[4][2];

    const char *const src_b = reinterpret_cast<const char *>(srch);
    const int64_t kstep_b = bgroup * 16;
    const int64_t panel_b = Nd_pad * 16;
    const int64_t tile_dk = 8 * panel_b;
    const int64_t tile_dwrap = FBD * 16 - (int64_t)(KT - 1) * tile_dk;
    const uint32_t lds_a = (uint32_t)reinterpret_cast<uintptr_t>((lds_void *)&sA[0][wave * 2 * FBD]);
    auto load_b1 = [&](const char *pb_, int ks, int sb, u32x4 &dst) {
        const char *ph = pb_ + ks * kstep_b;
        if (sb == 0) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff_b), "s"(ph));
        else asm volatile("global_load_dwordx4 %0, %1, %2 offset:512" : "=v"(dst) : "v"(voff_b), "s"(ph));
    };
    auto load_a_piece = [&](const char *pa_, int buf_, int t) {
        const char *g = pa_ + (t >> 1) * panel_b + (t & 1) * 1024;
        const uint32_t lds_off = lds_a + (uint32_t)buf_ * (uint32_t)sizeof(sA[0]) + (uint32_t)t * 1024u;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(lds_off), "v"(voff_a), "s"(g) : "memory");
    };
    // the two dword loads per step that the matcher spends on published row maxima keep their slots (the wait counts of
    // the loop are written for this exact issue sequence); here they fetch a word nobody uses
    const uint32_t *const dummy_rows = reinterpret_cast<const uint32_t *>(bh);
    uint32_t am[2] = {0u, 0u};
    const uint32_t voff_m = 0u;
    u32x4 fa[2][4];
    const char *pa1 = reinterpret_cast<const char *>(dsth + (int64_t)wave * 2 * Nd_pad + (int64_t)jt0 * FBD);
    const char *pb = src_b;
    {
        load_a_piece(pa1, 0, 0);
        load_a_piece(pa1, 0, 1);
        load_a_piece(pa1, 0, 2);
        load_a_piece(pa1, 0, 3);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (steps > 1) pa1 += KT == 1 ? tile_dwrap : tile_dk;
        load_b1(pb, 0, 0, rb[0][0]);
        load_b1(pb, 0, 1, rb[0][1]);
        load_b1(pb, 1, 0, rb[1][0]);
        load_b1(pb, 1, 1, rb[1][1]);
        load_a_piece(pa1, 1, 0);
        load_a_piece(pa1, 1, 1);
#pragma unroll
        for (int ib = 0; ib < 4; ++ib) fa[0][ib] = __builtin_bit_cast(u32x4, sA[0][kh * FBD + ib * 32 + l31]);
    }

    // ---- per-tile epilogues ----------------------------------------------------------------------------------------
    // accumulator register r of block (ib, sb): weight row 32 ib + (r & 3) + 8 (r >> 2) + 4 kh of the tile, token
    // srow0 + 32 sb + l31
    auto epilogue = [&](int jt) {
        const float *bt = sBias + (jt - jt0) * FBD;
        if constexpr (EPI == EPI_GEGLU) {
            uint16_t *outp = static_cast<uint16_t *>(sE.out);
            const int64_t out_rows_pad = sE.out_rows_pad;
#pragma unroll
            for (int sb = 0; sb < 2; ++sb) {
                const int64_t tok = srow0 + sb * 32 + l31;
#pragma unroll
                for (int ibp = 0; ibp < 2; ++ibp)
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) {
                        uint16_t w4[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int r = 4 * qq + e, row = 32 * ibp + e + 8 * qq + 4 * kh;
                            const float bv = bt[row], bg = bt[64 + row];
                            // the roundings of the unfused chain: proj output, gelu(gate), product (each to the model dtype)
                            const float v = Mma<T>::round(acc[ibp][sb][r] + bv);
                            const float g = Mma<T>::round(acc[ibp + 2][sb][r] + bg);
                            w4[e] = Mma<T>::bits(v * Mma<T>::round(gelu_erf(g)));
                        }
                        const int64_t g2 = (int64_t)jt * 8 + ibp * 4 + qq;     // output panel: channels 64 jt + 32 ibp + 8 qq ..
                        *reinterpret_cast<uint2 *>(outp + (g2 * out_rows_pad + tok) * 8 + 4 * kh) = *reinterpret_cast<const uint2 *>(w4);
                    }
            }
        } else {
            constexpr int SO = FBD + 8;
            uint16_t *my = sOut + wave * 32 * SO;
            uint16_t *outp = static_cast<uint16_t *>(sE.out);
            const uint16_t *resp = static_cast<const uint16_t *>(sE.resid);
            const int64_t ldo = sE.ldo, n_tok = sE.n, n_ch = sE.N;
#pragma unroll
            for (int sb = 0; sb < 2; ++sb) {
#pragma unroll
                for (int ib = 0; ib < 4; ++ib)
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) {
                        float f4[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int row = 32 * ib + e + 8 * qq + 4 * kh;
                            f4[e] = acc[ib][sb][4 * qq + e] + bt[row];
                        }
                        // staged as the ROUNDED GEMM result: torch rounds the Linear output, then `ff_output + hidden_states`
                        uint16_t w4[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) w4[e] = Mma<T>::bits(f4[e]);
                        *reinterpret_cast<uint2 *>(my + l31 * SO + 32 * ib + 8 * qq + 4 * kh) = *reinterpret_cast<const uint2 *>(w4);
                    }
                __builtin_amdgcn_wave_barrier();
                constexpr int PR = FBD / 8;                  // 16-byte pieces per token row of the tile
#pragma unroll
                for (int it = 0; it < 32 * PR / 64; ++it) {
                    const int p = lane + 64 * it;
                    const int row = p / PR, c8 = (p % PR) * 8;
                    const int64_t tok = srow0 + sb * 32 + row, ch = (int64_t)jt * FBD + c8;
                    if (tok < n_tok && ch < n_ch) {          // N % 8 == 0: the piece is all valid
                        uint4 v = *reinterpret_cast<const uint4 *>(my + row * SO + c8);
                        if (resp != nullptr) {
                            const uint4 rv = *reinterpret_cast<const uint4 *>(resp + tok * ldo + ch);
                            const T *a8 = reinterpret_cast<const T *>(&v), *r8 = reinterpret_cast<const T *>(&rv);
                            uint4 o;
                            uint16_t *o8 = reinterpret_cast<uint16_t *>(&o);
#pragma unroll
                            for (int e = 0; e < 8; ++e) o8[e] = Mma<T>::bits(vtm::to_f32(a8[e]) + vtm::to_f32(r8[e]));
                            v = o;
                        }
                        *reinterpret_cast<uint4 *>(outp + tok * ldo + ch) = v;
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
    };

    int kt = 0, jt = jt0;
    for (int st = 0; st < steps; ++st) {
        const int buf = st & 1;
        const bool wrap = kt + 1 == KT;
        const int kt1 = wrap ? 0 : kt + 1;
        const bool wrap1 = kt1 + 1 == KT;
        const char *const pbn = st + 1 < steps ? (wrap ? src_b : pb + 4 * kstep_b) : pb;
        const char *const pa2 = st + 2 < steps ? pa1 + (wrap1 ? tile_dwrap : tile_dk) : pa1;
        auto group = [&](auto s_tag, auto first_tag) {
            constexpr int s = decltype(s_tag)::value;
            constexpr bool FIRST = decltype(first_tag)::value;
            constexpr int COUNT = s == 1 ? NB + 2 * PG : s == 2 ? NB + PG + 2 : NB + 2;
            if constexpr (s != 0) await_b(std::integral_constant<int, COUNT>{}, rb[s]);
            if constexpr (s == 1) asm volatile("" : : "v"(am[0]), "v"(am[1]));
            u32x4 (&fh)[4] = fa[s & 1];
            u32x4 (&fn)[4] = fa[(s + 1) & 1];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int sb = j >> 2, ib = j & 3;
                f32x16 c = acc[ib][sb];
                if constexpr (FIRST) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) c[r] = 0.0f;
                }
                acc[ib][sb] = Mma<T>::run(fh[ib], rb[s][sb], c);
                __builtin_amdgcn_sched_barrier(0);
                if (j < 4) {
                    if constexpr (s < 3) fn[j] = __builtin_bit_cast(u32x4, sA[buf][((s + 1) * 2 + kh) * FBD + j * 32 + l31]);
                    else fn[j] = __builtin_bit_cast(u32x4, sA[buf ^ 1][kh * FBD + j * 32 + l31]);
                } else if (j < 6) {
                    if constexpr (s < 2) load_b1(pb, s + 2, j - 4, rb[s + 2][j - 4]);
                    else load_b1(pbn, s - 2, j - 4, rb[s - 2][j - 4]);
                } else {
                    if constexpr (s == 0) load_a_piece(pa1, buf ^ 1, j - 4);
                    if constexpr (s == 3) load_a_piece(pa2, buf, j - 6);
                    if constexpr (s == 1)
                        asm volatile("global_load_dword %0, %1, %2" : "=v"(am[j - 6]) : "v"(voff_m), "s"(dummy_rows));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        await_b(std::integral_constant<int, NB + PG>{}, rb[0]);
        if (kt == 0) group(std::integral_constant<int, 0>{}, std::true_type{});
        else group(std::integral_constant<int, 0>{}, std::false_type{});
        group(std::integral_constant<int, 1>{}, std::false_type{});
        group(std::integral_constant<int, 2>{}, std::false_type{});
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * NB + 2) : "memory");
        __syncthreads();
        group(std::integral_constant<int, 3>{}, std::false_type{});
        if (wrap) {
            epilogue(jt);
            ++jt;
        }
        kt = kt1;
        pb = pbn;
        pa1 = pa2;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---- rows -> panels (a pass of its own, for operands no kernel of the library produces) and weight packing ----
template <typename T>
__global__ __launch_bounds__(256) void to_panels_kernel(const T *__restrict__ x, int64_t rows, int64_t C, int64_t rows_pad,
                                                        const int32_t *__restrict__ order, uint4 *__restrict__ out) {
    // thread = (row, panel): 16-byte read along the row (a quarter-wave covers 128 contiguous bytes), 16-byte panel write
    const int64_t G = C / 8;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows_pad * G) return;
    const int64_t g = i % G, r = i / G;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (r < rows) {
        const int64_t src = order ? order[r] : r;
        if (src >= 0) v = *reinterpret_cast<const uint4 *>(x + src * C + g * 8);
    }
    out[g * rows_pad + r] = v;
}

// gather -> panels: the merged tokens of a block (a row selection of the pool [x0 | x1] through the composed merge map,
// optionally through a second map: the live-query rows) written straight in the panel layout, sample b's rows at
// b * rows_per_sample (a multiple of 256), padding rows zero.  thread = (sample, row, panel): 16-byte reads along a token
// row, 16-byte panel writes.
template <typename T>
__global__ __launch_bounds__(256) void gather_panels_kernel(const T *__restrict__ x0, int64_t P0, const T *__restrict__ x1,
                                                            int64_t P1, int64_t B, int64_t C, const int32_t *__restrict__ map,
                                                            int64_t map_ld, const int32_t *__restrict__ map2, int64_t n,
                                                            int64_t rows_per_sample, int64_t stride, uint4 *__restrict__ out) {
    const int64_t G = C / 8;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * rows_per_sample * G) return;
    const int64_t g = i % G, r = (i / G) % rows_per_sample, b = i / (G * rows_per_sample);
    uint4 v = make_uint4(0, 0, 0, 0);
    if (r < n) {
        int64_t p = map2 ? map2[b * n + r] : r;
        if (map) p = map[b * map_ld + p];
        const T *row = p < P0 ? x0 + (b * P0 + p) * C : x1 + (b * P1 + (p - P0)) * C;
        v = *reinterpret_cast<const uint4 *>(row + g * 8);
    }
    out[g * stride + b * rows_per_sample + r] = v;
}

inline int64_t pad256(int64_t n) { return vtm::cdiv(n, FBS) * FBS; }

template <typename T, int EPI>
int launch_panel_gemm(const void *tok, int64_t n, int64_t n_pad, const void *w, int64_t Nw, int64_t Nw_pad, int64_t K,
                      const Epi &E, hipStream_t s) {
    // n_pad / Nw_pad are the ROW STRIDES of the two panel operands (an operand may be a row range of a larger panel
    // tensor); the tiles cover the valid rows only
    // the kernel forms a lane's byte offset inside a k-group in 32 bits: (kh * n_pad + row) * 16 with kh <= 1 (the weight
    // operand goes through wave-uniform 64-bit pointers + a lane offset < 1 KiB)
    VTM_REQUIRE(2 * n_pad * 16 < (1ll << 32), "panel GEMM: %lld token panel rows exceed the 32-bit fragment offset",
                (long long)n_pad);
    const int ns_tiles = (int)vtm::cdiv(n, FBS), nd_tiles = (int)vtm::cdiv(Nw, FBD);
    const int total_src_tiles = ns_tiles;
    const int max_patch = (int64_t)16 * FBS * K * 2 <= (3 << 20) ? 16 : 8;
    const int tiles_per_xcd = (int)vtm::cdiv(total_src_tiles, 8);
    const int patches_per_xcd = (int)vtm::cdiv(tiles_per_xcd, max_patch);
    const int patch_tiles = (int)vtm::cdiv(tiles_per_xcd, patches_per_xcd);
    const int ngroups = (int)vtm::cdiv(total_src_tiles, patch_tiles);
    // splits of the weight rows: about three rounds of workgroups on the chip (2 per CU), at most MAX_TILES_PER_WG tiles each
    const int64_t slots = (int64_t)vtm::device_cus() * 2;
    int64_t nsplit = vtm::cdiv(3 * slots, (int64_t)ns_tiles);
    nsplit = std::max<int64_t>(nsplit, vtm::cdiv(nd_tiles, MAX_TILES_PER_WG));
    nsplit = std::min<int64_t>(std::max<int64_t>(nsplit, 1), nd_tiles);
    const int tiles_per_split = (int)vtm::cdiv(nd_tiles, nsplit);
    nsplit = vtm::cdiv(nd_tiles, tiles_per_split);
    const int64_t grid = (int64_t)8 * vtm::cdiv(ngroups, 8) * patch_tiles * nsplit;
    if (grid >= (1ll << 31)) return vtm::fail(VTM_EINVAL, "vtm_ff: grid too large");
    hipLaunchKernelGGL((panel_gemm_kernel<T, EPI>), dim3((unsigned)grid), dim3(THREADS), 0, s, (const uint4 *)tok,
                       (const uint4 *)w, n_pad, Nw_pad, K, ns_tiles, nd_tiles, (int)nsplit, tiles_per_split, total_src_tiles,
                       patch_tiles, E);
    return vtm::launch_status("vtm_ff");
}

}  // namespace

VTM_EXPORT int64_t vtm_panel_rows(int64_t n) { return n <= 0 ? 0 : pad256(n); }

VTM_EXPORT int vtm_to_panels(const void *x, int dtype, int64_t rows, int64_t C, const int32_t *order, void *out,
                             int64_t rows_pad, vtm_stream_t stream) {
    VTM_REQUIRE(x && out && rows > 0 && C > 0 && C % 8 == 0, "vtm_to_panels: bad arguments");
    VTM_REQUIRE(rows_pad >= rows && rows_pad % FBS == 0, "vtm_to_panels: rows_pad must be vtm_panel_rows(rows) or more");
    VTM_REQUIRE(dtype == VTM_F16 || dtype == VTM_BF16, "vtm_to_panels: dtype must be VTM_F16 or VTM_BF16");
    const int64_t total = rows_pad * (C / 8);
    hipLaunchKernelGGL(to_panels_kernel<__half>, dim3((unsigned)vtm::cdiv(total, 256)), dim3(256), 0, vtm::as_stream(stream),
                       (const __half *)x, rows, C, rows_pad, order, (uint4 *)out);
    return vtm::launch_status("vtm_to_panels");
}

VTM_EXPORT int vtm_gather_panels(const void *x0, int64_t P0, const void *x1, int64_t P1, int dtype, int64_t B, int64_t C,
                                 const int32_t *map, int64_t map_ld, const int32_t *map2, int64_t n, void *out,
                                 int64_t rows_per_sample, vtm_stream_t stream) {
    VTM_REQUIRE(x0 && out && B > 0 && C > 0 && C % 8 == 0 && n > 0, "vtm_gather_panels: bad arguments");
    VTM_REQUIRE(P1 == 0 || x1, "vtm_gather_panels: x1 is null but P1 > 0");
    VTM_REQUIRE(map || map2 || n <= P0 + P1, "vtm_gather_panels: identity rows must lie inside the pool");
    VTM_REQUIRE(!map || map_ld > 0, "vtm_gather_panels: map_ld");
    VTM_REQUIRE(rows_per_sample >= n && rows_per_sample % FBS == 0, "vtm_gather_panels: rows_per_sample must be a multiple of 256 >= n");
    VTM_REQUIRE(dtype == VTM_F16 || dtype == VTM_BF16, "vtm_gather_panels: dtype must be VTM_F16 or VTM_BF16");
    const int64_t total = B * rows_per_sample * (C / 8);
    hipLaunchKernelGGL(gather_panels_kernel<__half>, dim3((unsigned)vtm::cdiv(total, 256)), dim3(256), 0, vtm::as_stream(stream),
                       (const __half *)x0, P0, (const __half *)x1, P1, B, C, map, map_ld, map2, n, rows_per_sample,
                       B * rows_per_sample, (uint4 *)out);
    return vtm::launch_status("vtm_gather_panels");
}

VTM_EXPORT int vtm_ff_geglu(const void *x_panels, int64_t n, int64_t n_pad, const void *w1_panels, int64_t D, int64_t w_rows_pad,
                            int64_t K, const float *bias, int dtype, void *out_panels, vtm_stream_t stream) {
    VTM_REQUIRE(x_panels && w1_panels && out_panels, "vtm_ff_geglu: null pointer");
    VTM_REQUIRE(n > 0 && n_pad >= n && n_pad % FBS == 0, "vtm_ff_geglu: token rows must be padded to 256");
    VTM_REQUIRE(D > 0 && D % 64 == 0 && w_rows_pad >= 2 * D && w_rows_pad % FBS == 0, "vtm_ff_geglu: D %% 64, packed weight rows padded to 256");
    VTM_REQUIRE(K > 0 && K % FBK == 0, "vtm_ff_geglu: K must be a multiple of 64");
    Epi E{bias, out_panels, n_pad, 0, nullptr, n, D, 2 * D};
    hipStream_t s = vtm::as_stream(stream);
    if (dtype == VTM_F16) return launch_panel_gemm<__half, EPI_GEGLU>(x_panels, n, n_pad, w1_panels, 2 * D, w_rows_pad, K, E, s);
    if (dtype == VTM_BF16) return launch_panel_gemm<vtm_bf16, EPI_GEGLU>(x_panels, n, n_pad, w1_panels, 2 * D, w_rows_pad, K, E, s);
    return vtm::fail(VTM_EINVAL, "vtm_ff_geglu: dtype must be VTM_F16 or VTM_BF16");
}

VTM_EXPORT int vtm_linear_panels(const void *x_panels, int64_t n, int64_t n_pad, const void *w_panels, int64_t N, int64_t w_rows_pad,
                                 int64_t K, const float *bias, const void *resid, int dtype, void *out, int64_t ldo,
                                 vtm_stream_t stream) {
    VTM_REQUIRE(x_panels && w_panels && out, "vtm_linear_panels: null pointer");
    VTM_REQUIRE(n > 0 && n_pad >= vtm::cdiv(n, FBS) * FBS, "vtm_linear_panels: token rows must be padded to 256");
    VTM_REQUIRE(N > 0 && N % 8 == 0 && w_rows_pad >= vtm::cdiv(N, FBD) * FBD, "vtm_linear_panels: N %% 8, weight rows padded to 128");
    VTM_REQUIRE(K > 0 && K % FBK == 0, "vtm_linear_panels: K must be a multiple of 64");
    VTM_REQUIRE(ldo >= N && ldo % 8 == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0 &&
                    (reinterpret_cast<uintptr_t>(resid) & 15) == 0, "vtm_linear_panels: output rows must be 16-byte aligned");
    Epi E{bias, out, 0, ldo, resid, n, N, N};
    hipStream_t s = vtm::as_stream(stream);
    if (dtype == VTM_F16) return launch_panel_gemm<__half, EPI_ROWS>(x_panels, n, n_pad, w_panels, N, w_rows_pad, K, E, s);
    if (dtype == VTM_BF16) return launch_panel_gemm<vtm_bf16, EPI_ROWS>(x_panels, n, n_pad, w_panels, N, w_rows_pad, K, E, s);
    return vtm::fail(VTM_EINVAL, "vtm_linear_panels: dtype must be VTM_F16 or VTM_BF16");
}

// vtm_match_filtered: the SAME result as vtm_match (canonical fp32 row max / first argmax, bit for bit),
// obtained ~6x faster: an fp16-MFMA *filter* pass finds, for every src row, the few dst rows that can
// possibly be the fp32 argmax, and an exact fp32 *refine* pass evaluates the canonical fmaf chain only on
// those candidates.  Reference: vidtome/merge.py:87-113 / 392-417 (scores + max), as vtm_match.
//
// Why it is exact.  Let s_ij be the canonical fp32 chain value and t_ij the filter's approximation with
// |t_ij - s_ij| <= EPS for all i, j (bound below).  If j* attains max_j s_ij (including every tied j), then
// t_ij* >= s_ij* - EPS >= s_ij - EPS >= t_ij - 2 EPS for every j, i.e. j* lies within W = 2 EPS of every
// approximate score of the row -- in particular of any running maximum.  So "collect every j with
// t_ij >= running_max - W" collects all true argmax columns; the refine pass computes their exact scores
// and combines them with the same packed atomicMax as vtm_match (largest value, first index).
//
// Approximation.  xhat (fp32) is written as 1024*xhat = hi + lo (+ a dropped 2^-22 tail) with hi, lo fp16, and
// the filter accumulates products of these halves with v_mfma_f32_32x32x16_f16 (exact products, fp32
// accumulation).  The shipped variant keeps hi*hi only; the variants that add the lo cross terms are build-time
// options (SRC_LO / DST_LO below, with the error budget and the window of each).  The 1024 scale keeps the
// halves out of the fp16 subnormal range for every component that matters (a component below 6e-8 is
// rounded with an absolute error <= 3e-11, far inside EPS even summed over 1280 channels).
//
// Escapes (all exact, no host round trip, BOUNDED cost): a row whose candidate list overflows CAP (flat image regions,
// massively duplicated dst rows) is put on a per-sample list and recomputed by exact_rows_kernel: fp32-MFMA score tiles
// [256 listed rows x all Nd] -- the arithmetic of vtm_match (v_mfma_f32_32x32x2_f32 = the canonical k-ascending fmaf
// chain), operands normalised on the fly from the token rows and the canonical norms; any row without a finite positive
// norm (zero token -> 0/0, merge.py:84 has no eps) is recognised by refine_kernel: a src row of that kind joins the list, a
// dst row of that kind raises a device flag that makes exact_rows_kernel recompute EVERY row of the call.  Worst case = the cost of the exact matcher (~8 ms at the cfg-2 top level), not the
// ~1000x of a scalar row pass (rounds 1-3).  A row stops collecting the moment its list overflows (the overflowing lane
// publishes +inf as the row's running maximum, which every other lane / split of the row picks up), so flat regions do not
// flood the filter with candidate pushes either.
#include "common.h"
#include "ablate.h"

#include <cstdlib>

#include <algorithm>
#include <type_traits>

namespace {

// (The ABL_* macros in filter_kernel are the ablation switches of ablate.h: in the shipped build each expands to the code
// it wraps and nothing else.)
constexpr int FBD = 128;      // dst rows per tile (MFMA A operand, LDS)
constexpr int FBS = 256;      // src rows per workgroup (B operand, registers), 64 per wave
constexpr int FBK = 64;       // channels per pipeline step = 4 MFMA k-steps = 8 panels
constexpr int THREADS = 256;
constexpr int CAP = 64;       // candidate slots per src row
// Survivors (candidates inside the window of the row's FINAL approximate maximum) go to the per-pair refine pass however
// many a row has (<= CAP): the anchor tokens of a global level hold exact copies of matched rows (patch.py:80), so a
// src row whose best dst row exists m times has m tied survivors -- m ~ 10 after one local-is-src pass -- and the exact
// row pass (all Nd chains of the row) is ~1000x the cost of its few pairs.  The pair list is sized for the worst case
// (every list full), so its reservation cannot fail.
constexpr float SCALE = 1024.0f;
constexpr float INV_S2 = 1.0f / (1024.0f * 1024.0f);
// Which products the filter accumulates (the refine pass is exact whatever the filter does; fewer products = less
// MFMA work and operand traffic, wider window = more candidates for the refine pass):
//   SRC_LO && DST_LO : 3 products  hi*hi + hi*lo + lo*hi
//   DST_LO only      : 2 products  (hi_dst + lo_dst) * hi_src
//   neither          : 1 product   hi_dst * hi_src                                  <- shipped
// Error budget per score for unit vectors (sum |a_k b_k| <= 1), C <= 1280, fp16 unit roundoff u = 2^-11
// (|lo| <= u |1024 xhat| element-wise):
//   each dropped lo operand                <= u * sum |a_k b_k|               = 4.9e-4   (both: 2u + u^2 = 9.8e-4)
//   hi/lo representation tails             <= 3 * 2^-22                       = 7e-7
//   reciprocal-multiply operands (1 product) <= 2 * 2.4e-7                      = 5e-7
//   fp32 accumulation inside the MFMA      <= (#products) * C * 2^-24         = 7.6e-5 per product
//   the canonical fp32 chain itself        <= C * 2^-24                       = 7.6e-5
// EPS = 3.25e-4 (3 products) / 7.5e-4 (2) / 1.2e-3 (1); observed filter errors are ~1e-6 / ~1e-5 / ~2e-5.
// Measured on the cfg-2 shapes: the wider window adds < 10 % surviving pairs (the refine pass is ~3 % of the call).
#ifndef VTM_FILTER_PRODUCTS
#define VTM_FILTER_PRODUCTS 1   // build-time choice (1, 2 or 3); the shipped library uses 1
#endif
constexpr bool SRC_LO = VTM_FILTER_PRODUCTS >= 3;
constexpr bool DST_LO = VTM_FILTER_PRODUCTS >= 2;
static_assert(DST_LO || !SRC_LO, "the src lo half is only used together with the dst lo half");
constexpr float WINDOW = SRC_LO ? 6.5e-4f : DST_LO ? 1.5e-3f : 2.4e-3f;   // 2 * EPS
constexpr int MAX_C = 1280;                              // the budget above is derived for C <= 1280

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

using vtm::to_f32;

template <typename T>
__device__ __forceinline__ const T *pool_row(const T *x0, int64_t P0, const T *x1, int64_t P1, int64_t b,
                                             int64_t r, int64_t C) {
    return r < P0 ? x0 + (b * P0 + r) * C : x1 + (b * P1 + (r - P0)) * C;
}

template <typename T>
__device__ __forceinline__ void load8(const T *src, float (&f)[8]) {
    if constexpr (sizeof(T) == 4) {
        const float4 v0 = *reinterpret_cast<const float4 *>(src);
        const float4 v1 = *reinterpret_cast<const float4 *>(src + 4);
        f[0] = v0.x; f[1] = v0.y; f[2] = v0.z; f[3] = v0.w;
        f[4] = v1.x; f[5] = v1.y; f[6] = v1.z; f[7] = v1.w;
    } else {
        const uint4 v = *reinterpret_cast<const uint4 *>(src);
        const T *e = reinterpret_cast<const T *>(&v);
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = to_f32(e[j]);
    }
}

// max of three without the canonicalising v_max x, x the compiler puts in front of fmaxf on values it cannot prove
// quiet (every MFMA result): a NaN operand is ignored, like fmaxf
__device__ __forceinline__ float vmax3(float a, float b, float c) {
    float d;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ float vmax2(float a, float b) {
    float d;
    asm("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

// ---- x / n for MANY x and ONE n, bit-identical to the IEEE division ----
// hipcc expands x / n (fp32, denormals on) to  d = div_scale(n), m = div_scale(x), r = rcp(d), e = fma(-d, r, 1), r1 = fma(e, r, r),
// q = m * r1, t = fma(-d, q, m), q1 = fma(t, r1, q), t2 = fma(-d, q1, m), div_fmas(t2, r1, q1), div_fixup -- 11 instructions, the
// reciprocal refinement redone for every x because v_div_scale looks at both operands.  When neither operand needs scaling
// (n in [2^-100, 2^100]: guaranteed on the filtered path, refine_kernel; x = 0, or |x| >= 2^-102 and x / n >= 2^-124: checked,
// `div_ok`) div_scale returns its input, div_fmas is a plain fma and div_fixup passes the value through (a zero quotient may
// lose its sign, which a fused multiply-add chain starting from +0 cannot see), so the SAME operations with the SAME roundings can
// be issued with r1 computed once per row: 5 instructions per quotient.
struct RowDivisor {
    float n, r1, lo;
};
__device__ __forceinline__ RowDivisor row_divisor(float n) {
    const float r = __builtin_amdgcn_rcpf(n);
    const float e = __builtin_fmaf(-n, r, 1.0f);
    return {n, __builtin_fmaf(e, r, r), fmaxf(0x1p-102f, n * 0x1p-124f)};
}
__device__ __forceinline__ bool div_ok(float x, const RowDivisor &d) { return __builtin_fabsf(x) >= d.lo || x == 0.0f; }
__device__ __forceinline__ float div_by_row(float x, const RowDivisor &d) {
    const float q = x * d.r1;
    const float t = __builtin_fmaf(-d.n, q, x);
    const float q1 = __builtin_fmaf(t, d.r1, q);
    const float t2 = __builtin_fmaf(-d.n, q1, x);
    return __builtin_fmaf(t2, d.r1, q1);
}

__device__ __forceinline__ uint32_t orderable(float f) {
    if (f != f) return 0xffffffffu;
    const uint32_t u = __float_as_uint(f + 0.0f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float from_orderable(uint32_t o) {   // 0 (never written) -> -inf
    if (o == 0u) return -INFINITY;
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}

// ---- operand preparation: canonical row norms + hi / lo fp16 panels [b][g = k/8][row][8], ONE launch ----
// A wave owns 64 gathered rows (both operands of the match in the same launch), a lane one of them.  The canonical
// norm is a k-ascending fmaf chain per row (the order is part of the bit-exact contract, so it is not tree-reduced --
// same bits as row_norms in normalize.hip): serial per lane, but its INPUT need not arrive serially.  The rows are
// fetched COOPERATIVELY, 80 channels at a time: the wave reads 64 x 160 contiguous bytes as 10 coalesced pieces per
// lane (one chunk ahead), transposes them through a wave-private LDS slab, and each lane then walks its own row's
// pieces out of LDS.  A second sweep of the same kind divides, scales, splits and stores the panels (lanes are
// consecutive rows, so every panel store of a wave is one contiguous 1 KiB segment).  (A lane fetching its own row
// directly issues 64 different cache lines per load and one dependent batch per 64 channels: 47 us at C = 640 however
// few rows there are -- r02_b.)  The launch also clears the call's counters and zero-fills `best` (nothing in this
// kernel reads them: no ordering needed).
// A row whose norm is not a finite positive number (zero token -> 0/0, merge.py:84 has no eps; inf / NaN inputs) has
// non-finite xhat components; refine_kernel recognises it by the stored norm / the tile's rest-norm mark.
struct SplitArgs {   // one operand: gathered rows, their norms (out), the panel outputs
    const int32_t *rows;
    int64_t n;
    float *norms;
    uint4 *out_hi, *out_lo;
    int64_t n_pad;
    // partial-sum pruning (filter_kernel): the norm of the hi operand's channels >= `cut`, per row (src operand) or as
    // the maximum over every 128-row tile (dst operand); nullptr = not wanted
    float *rest, *tile_rest;
    // the same for a second, EARLIER cut (the shallow scout of the scout + range plan); nullptr = not wanted
    float *rest2, *tile_rest2;
};

#ifndef VTM_PREP_PIECES
#define VTM_PREP_PIECES 10      // (A/B build switches; 4 odd multiples keep the LDS rows conflict-free: 2, 6, 10, 14 ...)
#endif
#ifndef VTM_PREP_WAVES
#define VTM_PREP_WAVES 4
#endif
constexpr int PREP_WAVES = VTM_PREP_WAVES;
constexpr int PREP_PIECES = VTM_PREP_PIECES;           // 16-byte pieces (8 channels) of a row per chunk
constexpr int PREP_STRIDE = PREP_PIECES * 16 + 16;     // bytes per LDS row: 44 words = 4 x odd -> conflict-free b128

template <typename T>
__global__ __launch_bounds__(64 * PREP_WAVES) void prep_operand(const T *__restrict__ x0, int64_t P0,
                                                    const T *__restrict__ x1, int64_t P1, int64_t B, int64_t C,
                                                    SplitArgs A0, SplitArgs A1, int64_t C_pad,
                                                    uint32_t *__restrict__ zero, int64_t zero_words,
                                                    unsigned long long *__restrict__ best, int64_t nbest, int64_t cut,
                                                    int64_t cut2) {
    static_assert(sizeof(T) == 2 || sizeof(T) == 4, "element size");
    static_assert(PREP_WAVES % 2 == 0, "a 128-row tile is two waves of one workgroup");
    __shared__ float wave_rest[PREP_WAVES], wave_rest2[PREP_WAVES];
    constexpr int EPP = 16 / (int)sizeof(T);           // elements per 16-byte piece (8 for the 16-bit types, 4 for fp32)
    __shared__ __attribute__((aligned(16))) char slab[PREP_WAVES][64 * PREP_STRIDE];
    const int64_t G = C_pad / 8;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, gsz = (int64_t)gridDim.x * blockDim.x;
    for (int64_t w = gid; w < zero_words; w += gsz) zero[w] = 0u;      // amax, cnt, flags
    for (int64_t w = gid; w < nbest; w += gsz) best[w] = 0ull;         // packed results start from "nothing found"
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    char *my = slab[wave];

    // this lane's row
    int64_t idx = gid;
    const bool second = idx >= B * A0.n_pad;
    if (second) idx -= B * A0.n_pad;
    const SplitArgs &A = second ? A1 : A0;
    const int64_t n = A.n, n_pad = A.n_pad;
    const bool in_range = idx < B * n_pad;
    const int64_t i = in_range ? idx % n_pad : 0, b = in_range ? idx / n_pad : 0;
    const bool real = in_range && i < n;
    // rows outside the operands / padding rows stream row 0 of x0 (their results are never stored)
    const T *src = real ? pool_row(x0, P0, x1, P1, b, A.rows[b * n + i], C) : x0;

    const int pieces = (int)(C * sizeof(T) / 16);                      // 16-byte pieces per row
    const int nchunks = (pieces + PREP_PIECES - 1) / PREP_PIECES;
    // cooperative fetch of one chunk: piece q = lane + 64 t of the wave's 64 x PREP_PIECES block -> (row q / PP, col q % PP)
    uint4 stage[PREP_PIECES];
    auto issue = [&](int chunk) {
#pragma unroll
        for (int t = 0; t < PREP_PIECES; ++t) {
            const int q = lane + 64 * t, r = q / PREP_PIECES, col = q % PREP_PIECES;
            const T *rp = reinterpret_cast<const T *>(__shfl((unsigned long long)reinterpret_cast<uintptr_t>(src), r, 64));
            const int piece = chunk * PREP_PIECES + col;
            stage[t] = piece < pieces ? *reinterpret_cast<const uint4 *>(rp + (int64_t)piece * EPP) : make_uint4(0, 0, 0, 0);
        }
    };
    auto to_lds = [&]() {
#pragma unroll
        for (int t = 0; t < PREP_PIECES; ++t) {
            const int q = lane + 64 * t, r = q / PREP_PIECES, col = q % PREP_PIECES;
            *reinterpret_cast<uint4 *>(my + r * PREP_STRIDE + col * 16) = stage[t];
        }
    };
    auto my_piece = [&](int col, float (&f)[EPP]) {                    // this lane's row, piece `col` of the chunk in LDS
        const uint4 v = *reinterpret_cast<const uint4 *>(my + lane * PREP_STRIDE + col * 16);
        const T *e = reinterpret_cast<const T *>(&v);
#pragma unroll
        for (int k = 0; k < EPP; ++k) f[k] = to_f32(e[k]);
    };

    // sweep 1: the canonical norm chain
    float acc = 0.0f;
    issue(0);
    for (int c = 0; c < nchunks; ++c) {
        to_lds();                                                       // (LDS is FIFO per wave: no barrier needed)
        if (c + 1 < nchunks) issue(c + 1);
        const int valid = min(PREP_PIECES, pieces - c * PREP_PIECES);
        for (int col = 0; col < valid; ++col) {
            float f[EPP];
            my_piece(col, f);
#pragma unroll
            for (int k = 0; k < EPP; ++k) acc = __builtin_fmaf(f[k], f[k], acc);
        }
    }
    const float nrm = __builtin_sqrtf(acc);
    if (real) A.norms[b * n + i] = nrm;

    // sweep 2: xhat = x / norm, scaled, split into fp16 hi (+ lo), written as panels.  The one-product filter only needs
    // hi = fp16(1024 xhat (1 + d)) with |d| <= 2.4e-7 (reciprocal <= 1 ulp, one product rounding; two such operands add
    // <= 5e-7 to the score error, inside the 7e-5 the window keeps in reserve), so it multiplies by ONE reciprocal per row
    // instead of dividing 320-1280 times (~10 instructions each).  The refine pass has its own IEEE divisions; norms
    // outside [2^-100, 2^100] (where the reciprocal could leave the normal range) send the call down the exact path like
    // non-finite ones (refine_kernel).
    const float rscale = SCALE * __builtin_amdgcn_rcpf(nrm);
    uint4 *__restrict__ out_hi = in_range ? A.out_hi + (b * G) * n_pad + i : nullptr;
    uint4 *__restrict__ out_lo = (in_range && A.out_lo) ? A.out_lo + (b * G) * n_pad + i : nullptr;
    constexpr int PPG = 8 / EPP;                                        // pieces per 8-channel panel group (1 or 2)
    float rest = 0.0f, rest2 = 0.0f;                                    // sum of hi^2 over the channels >= cut / >= cut2
    issue(0);
    for (int c = 0; c < nchunks; ++c) {
        to_lds();
        if (c + 1 < nchunks) issue(c + 1);
        const int valid = min(PREP_PIECES, pieces - c * PREP_PIECES);
        for (int col = 0; col + PPG <= valid; col += PPG) {
            float f[8];
            if constexpr (PPG == 1) {
                my_piece(col, f);
            } else {
                float f0[EPP], f1[EPP];
                my_piece(col, f0);
                my_piece(col + 1, f1);
#pragma unroll
                for (int k = 0; k < EPP; ++k) { f[k] = f0[k]; f[EPP + k] = f1[k]; }
            }
            uint4 vh, vl;
            _Float16 *ph = reinterpret_cast<_Float16 *>(&vh), *pl = reinterpret_cast<_Float16 *>(&vl);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float sc = DST_LO ? (f[e] / nrm) * SCALE : f[e] * rscale;   // 2 / 3 products: the canonical xhat
                const _Float16 h = (_Float16)sc;
                ph[e] = h;
                pl[e] = (_Float16)(sc - (float)h);                     // exact difference, rounded once
            }
            if (!real) vh = vl = make_uint4(0, 0, 0, 0);               // padding rows: all-zero operands
            const int64_t g = ((int64_t)c * PREP_PIECES + col) / PPG;
            if (g * 8 >= cut2) {                                         // (cut2 <= cut)
                float sq = 0.0f;
#pragma unroll
                for (int e = 0; e < 8; ++e) sq = __builtin_fmaf((float)ph[e], (float)ph[e], sq);
                rest2 += sq;
                if (g * 8 >= cut) rest += sq;
            }
            if (in_range) {
                out_hi[g * n_pad] = vh;
                if (out_lo) out_lo[g * n_pad] = vl;
            }
        }
    }
    if (in_range)                                                       // channel padding C .. C_pad: zeros
        for (int64_t g = C / 8; g < G; ++g) {
            out_hi[g * n_pad] = make_uint4(0, 0, 0, 0);
            if (out_lo) out_lo[g * n_pad] = make_uint4(0, 0, 0, 0);
        }
    // rest norms for the filter's partial-sum pruning, rounded UP a little (1 + 2^-16 each: the Cauchy-Schwarz bound must
    // also cover the fp32 rounding of this sum and of the MFMA's own accumulation, ~1e-6 relative).  Non-finite values
    // (bad rows) make every comparison against them false or true-forever; such calls are recomputed exactly anyway.
    // A row whose norm is not a finite number in [2^-100, 2^100] (zero token -> NaN xhat, merge.py:84; overflow; a
    // reciprocal that leaves the normal range) reports +inf: refine_kernel recognises the call / the row by it, and the
    // pruning test can never declare a block dead against it.
    const bool bad_row = real && !(nrm >= 0x1p-100f && nrm <= 0x1p100f);
    const float rnorm = !real ? 0.0f : bad_row ? INFINITY : __builtin_sqrtf(rest) * (1.0f + 0x1p-16f);
    if (in_range && A.rest) A.rest[b * n_pad + i] = rnorm;
    float wm = rnorm;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) wm = fmaxf(wm, __shfl_xor(wm, off, 64));
    if (lane == 0) wave_rest[wave] = wm;
    const float rnorm2 = !real ? 0.0f : bad_row ? INFINITY : __builtin_sqrtf(rest2) * (1.0f + 0x1p-16f);
    if (in_range && A.rest2) A.rest2[b * n_pad + i] = rnorm2;
    float wm2 = rnorm2;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) wm2 = fmaxf(wm2, __shfl_xor(wm2, off, 64));
    if (lane == 0) wave_rest2[wave] = wm2;
    __syncthreads();
    if (in_range && A.tile_rest && (threadIdx.x & 127) == 0)            // n_pad is a multiple of 256: tiles do not straddle
        A.tile_rest[(b * n_pad + i) / 128] = fmaxf(wave_rest[wave], wave_rest[wave + 1]);
    if (in_range && A.tile_rest2 && (threadIdx.x & 127) == 0)
        A.tile_rest2[(b * n_pad + i) / 128] = fmaxf(wave_rest2[wave], wave_rest2[wave + 1]);
}

// ---- filter: approximate scores on the fp16 MFMA, candidate collection ----
// Round 5, "scout + range" plan for levels whose rows are position-ordered on both sides (level 1): on frames of one clip
// 94 % of the 256 x 128 WORKGROUP tiles of such a level are dead as a whole at the pruning test, yet the one-launch kernel
// streams all five channel steps of every tile (the steps behind the test cost 0.22 of the 0.75 ms of a top level-1 call,
// profiles/r05_b_scout_timing.txt).  SCOUT = the same loop over only the KP steps in front of the test: nothing is
// collected, a wave whose test leaves a block alive sets the tile's bit in `tilemap`.  The second launch is THIS kernel,
// unchanged, except that a workgroup shrinks its dst range [jt0, jt1) to the span of the set bits inside it (one dst frame
// per split: the live tiles of a src tile sit together there).  Every tile outside the spans is certified dead (no pair of
// it can come within the window of a valid running maximum, see "partial-sum pruning"), every tile inside is processed in
// full: the result is what the one-launch plan computes.  Only the host knows whether the plan pays (merge.MatchPlanner).
template <bool SCOUT>
__global__ __launch_bounds__(THREADS, 2) void filter_kernel(
    const uint4 *__restrict__ ah, const uint4 *__restrict__ al, const uint4 *__restrict__ bh,
    const uint4 *__restrict__ bl, int64_t Ns, int64_t Nd, int64_t Ns_pad, int64_t Nd_pad, int64_t C_pad, int align,
    int ns_tiles, int nd_tiles, int nsplit, int tiles_per_split, int total_src_tiles, int patch_tiles,
    unsigned int *__restrict__ amax, int *__restrict__ cnt, uint2 *__restrict__ cand, int cand_rows, int *__restrict__ flags,
    const float *__restrict__ rest_a, const float *__restrict__ rest_bt, int KP, int count_blocks,
    unsigned int *__restrict__ tilemap, int map_words) {
    // dst tile of one step: 8 panels x 128 rows x 16 B, hi and lo, double-buffered: 2 x 2 x 16 KiB
    __shared__ __attribute__((aligned(16))) uint4 sA[2][DST_LO ? 2 : 1][8 * FBD];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kh = lane >> 5;

    // XCD-aware work mapping (blocks are dispatched round-robin over the 8 XCDs, in index order): an XCD works
    // through patches of `patch_tiles` consecutive src tiles x all dst splits, so the src operands (the part that
    // is re-read for every dst tile) of its ~64 resident workgroups stay inside that XCD's 4 MiB L2 and each dst
    // stream is shared by a whole patch.  Inside a patch the order is split-major: the later splits of a row
    // start when the earlier ones have published their maximum, so their running maximum starts high and
    // their candidate logic stays on the cheap path.  Purely a speed choice; any placement gives the same result.
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int per_group = patch_tiles * nsplit;
    const int grp = (slot / per_group) * 8 + xcd;
    const int q = slot % per_group;
    const int stg = grp * patch_tiles + q % patch_tiles;   // flattened (sample, src tile)
    const int split = q / patch_tiles;
    if (stg >= total_src_tiles) return;
    const int st_ = stg % ns_tiles;
    const int bi = stg / ns_tiles;
    int jt0 = split * tiles_per_split;
    int jt1 = min(jt0 + tiles_per_split, nd_tiles);
    if (jt0 >= jt1) return;
    if constexpr (!SCOUT) {
        if (tilemap != nullptr) {   // range plan: the span of the tiles the scout left alive inside this split (wave-uniform)
            const unsigned int *row = tilemap + (int64_t)stg * map_words;
            int lo = jt1, hi = jt0;
            for (int w = jt0 >> 5; w <= (jt1 - 1) >> 5; ++w) {
                unsigned int bits = row[w];
                if (w == (jt0 >> 5)) bits &= 0xffffffffu << (jt0 & 31);
                if (w == ((jt1 - 1) >> 5) && (jt1 & 31)) bits &= 0xffffffffu >> (32 - (jt1 & 31));
                if (bits) {
                    lo = min(lo, w * 32 + __ffs(bits) - 1);
                    hi = max(hi, w * 32 + 32 - __clz(bits));
                }
            }
            lo = __builtin_amdgcn_readfirstlane(lo);
            hi = __builtin_amdgcn_readfirstlane(hi);
            if (lo >= hi) return;
            jt0 = lo;
            jt1 = hi;
        }
    }

    // channel steps per dst tile: all of them, or (scout) only the KP in front of the pruning test
    const int KT = SCOUT ? KP : (int)(C_pad / FBK);
    const int64_t G = C_pad / 8;
    const int steps = (jt1 - jt0) * KT;
    int n_tested = 0, n_alive = 0;            // (wave-uniform) pruning statistics, published once per wave
    const uint4 *srch = ah + (int64_t)bi * G * Ns_pad, *srcl = al + (int64_t)bi * G * Ns_pad;
    const uint4 *dsth = bh + (int64_t)bi * G * Nd_pad, *dstl = bl + (int64_t)bi * G * Nd_pad;
    const int64_t srow0 = (int64_t)st_ * FBS + wave * 64;

    // ---- vector-memory pipeline, issued and awaited BY HAND (inline asm) --------------------------------------
    // Why not builtins: while an LDS-DMA (global_load_lds) is pending, hipcc turns EVERY wait on a loaded register
    // or LDS read into "s_waitcnt vmcnt(0) lgkmcnt(0)" (the instruction counts as a flat access to two address
    // spaces), which drains the prefetches the moment they are issued -- measured: the MFMA pipe idles half the
    // time.  vmcnt retires in order, so a counted wait only needs the number of operations issued AFTER the one
    // awaited; both loops below issue a fixed sequence per step (no conditional loads), hence constant counts.
    // B(s) = src fragments of k-step group s (' = next step), always fetched two groups (32 MFMAs) ahead; DMA = LDS-DMA
    // pieces of an upcoming dst tile.  The sequences and their counts are written next to each loop (shipped: below
    // "shipped loop"; the phased one of the 2- / 3-product builds: g0: B(2) x NB, DMA x PG | g1: B(3) x NB, DMA x PG |
    // g2: B(0') x NB | g3: B(1') x NB, the DMA pieces behind the B loads of groups 0 / 1 so that no B wait drags a freshly
    // issued piece along, end-of-step wait with the 2 NB youngest loads in flight).
    // Any additional vector-memory operation the compiler issues (candidate pushes, maximum publishing) is younger than
    // ours and can only make these waits stricter, never laxer.  Two rules keep the scheme sound: every wait statement
    // exists ONCE (a copy behind a branch makes the compiler copy the awaited registers before it -- stale fragments),
    // and the kernel must not spill SGPRs (a spilling build of this loop faults on gfx950 / ROCm 7.2).
    constexpr int NB = SRC_LO ? 4 : 2;
    // addresses = wave-uniform 64-bit base (SGPR pair, advanced with scalar adds) + per-lane 32-bit byte offset
    const uint32_t voff_b = (uint32_t)(kh * Ns_pad + srow0 + l31) * 16u;
    const int64_t bgroup = 2 * Ns_pad;   // uint4 entries per k-step group (2 panels)
    u32x4 rb[4][2][2];
    [[maybe_unused]] auto load_b = [&](int kt, int ks, u32x4 (&dst)[2][2]) {
        const uint4 *ph = srch + (int64_t)ABL_STREAMED(kt * 4 + ks, ks) * bgroup;   // uniform
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst[0][0]) : "v"(voff_b), "s"(ph));
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:512" : "=v"(dst[1][0]) : "v"(voff_b), "s"(ph));
        if constexpr (SRC_LO) {
            const uint4 *pl = srcl + (int64_t)(kt * 4 + ks) * bgroup;
            asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst[0][1]) : "v"(voff_b), "s"(pl));
            asm volatile("global_load_dwordx4 %0, %1, %2 offset:512" : "=v"(dst[1][1]) : "v"(voff_b), "s"(pl));
        }
    };
    // the fragments become usable only through this statement (the "+v" ties order every use behind the wait)
    auto await_b = [&](auto count_tag, u32x4 (&r)[2][2]) {
        constexpr int N = decltype(count_tag)::value;
        if constexpr (SRC_LO)
            asm volatile("s_waitcnt vmcnt(%4)" : "+v"(r[0][0]), "+v"(r[1][0]), "+v"(r[0][1]), "+v"(r[1][1]) : "n"(N));
        else
            asm volatile("s_waitcnt vmcnt(%2)" : "+v"(r[0][0]), "+v"(r[1][0]) : "n"(N));
    };
    // A tile of one step: 2 x 16 LDS-DMA wave-instructions of 1 KiB; wave w issues 8 of them, in two halves.
    // Per piece only the step offset (kt, jt) changes: the rest of the address and the LDS target are wave constants.
    constexpr int NPIECE = DST_LO ? 8 : 4, PG = NPIECE / 2;   // DMA pieces per wave and step / per group 0, 1
    const uint4 *abase[NPIECE];
    uint32_t alds[NPIECE];
#pragma unroll
    for (int t = 0; t < NPIECE; ++t) {
        const int q = wave * NPIECE + t, which = q >> 4, qq = q & 15, p = qq >> 1, half = qq & 1;
        abase[t] = (which ? dstl : dsth) + (int64_t)p * Nd_pad + half * 64;
        alds[t] = (uint32_t)reinterpret_cast<uintptr_t>((lds_void *)&sA[0][which][p * FBD + half * 64]);
    }
    const uint32_t voff_a = (uint32_t)lane * 16u;
    [[maybe_unused]] auto load_a_half = [&](int jt, int kt, int buf, int half_id) {
        const int64_t step_off = ABL_STREAMED((int64_t)kt * 8 * Nd_pad + (int64_t)jt * FBD, (int64_t)0);
#pragma unroll
        for (int t = 0; t < PG; ++t) {
            const uint4 *g = abase[half_id * PG + t] + step_off;
            const uint32_t lds_off = alds[half_id * PG + t] + (uint32_t)buf * (uint32_t)sizeof(sA[0]);
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                         :
                         : "s"(lds_off), "v"(voff_a), "s"(g)
                         : "memory");
        }
    };

    f32x16 acc[4][2];   // (re)started with a zero C operand at the first k-step of every dst tile

    const int64_t out_row0 = align ? 0 : (int64_t)bi * Ns;
    const uint32_t idx_base = align ? (uint32_t)((int64_t)bi * Nd) : 0u;
    // per lane and src block: running max (in units of S^2) and the last 4 scores that came within the
    // window of it (a FIFO in registers; an evicted entry that is still inside the window = overflow)
    constexpr float WS = WINDOW * SCALE * SCALE;
    float runmax[2], cv[2][4];
    uint32_t ci[2][4];
    // ---- partial-sum pruning ----------------------------------------------------------------------------------
    // After KP of the KT channel steps of a dst tile the score of a pair is  partial + sum over the remaining channels
    // <= partial + |a_rest| |b_rest|  (Cauchy-Schwarz on the fp16 hi operands themselves: rest_a per src row, rest_bt =
    // the largest |b_rest| of the tile, both from prep_operand).  A 32 x 32 block none of whose pairs can still come
    // within the window of its row's running maximum will neither produce a candidate nor raise a maximum, so its
    // remaining MFMAs are skipped (`live` bit per accumulator block; the loads and the barriers of the loop go on, the
    // matrix pipe -- and its power -- go to the SIMD's other wave).  Frames of a video are correlated: a src row has a
    // handful of strong matches, and >= 90 % of the blocks die at 60 % depth.  Exact: the skipped blocks hold partial
    // sums that are below the threshold themselves, so collect_tile ignores them like any other low score.
    float rest_l[2];
    uint32_t live = 0xffu;
    // candidate lists are slot-major, [slot][row] (cand_rows rows): the first entries of neighbouring rows -- all that most
    // rows ever have -- share cache lines for the lanes of refine_kernel; 32-bit index (the launcher checks the size)
    // The row's list is full: the row goes to exact_rows_kernel whatever else is found, so it stops collecting -- +inf
    // becomes its published running maximum (nothing is ever within the window of +inf) and every lane / split of the row
    // picks that up with its next re-read (a flat image region would otherwise push thousands of candidates per row).
    // (This lane too: with its re-read at the end of the next tile -- no extra state in the loop, whose register budget
    // has no slack.)
    auto push = [&](int64_t srow, float v_scaled, uint32_t d) -> bool {   // append to the row's global candidate list
        const int slot = atomicAdd(&cnt[out_row0 + srow], 1);     // cnt > CAP marks the row for exact_rows_kernel
        if (slot < CAP)
            cand[(uint32_t)slot * (uint32_t)cand_rows + (uint32_t)(out_row0 + srow)] = make_uint2(__float_as_uint(v_scaled * INV_S2), d + idx_base);
        return slot >= CAP;
    };
#pragma unroll
    for (int sb = 0; sb < 2; ++sb) {
        const int64_t srow = srow0 + sb * 32 + l31;
        // start from what other workgroups already found for this row: fewer early candidates
        runmax[sb] = srow < Ns ? from_orderable(amax[out_row0 + srow]) * (SCALE * SCALE)
                               : INFINITY;   // padding rows (all-zero operands) never collect anything
        rest_l[sb] = rest_a ? rest_a[(int64_t)bi * Ns_pad + srow] : 0.0f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            cv[sb][e] = -INFINITY;
            ci[sb][e] = 0;
        }
    }

    // dst tile finished: every score within the window of the lane's running max becomes a candidate
    // The other dst splits of a row (other workgroups) and the lane holding the row's other 64 dst rows of every tile
    // work on the same maximum: each lane publishes its running maximum in amax[row] whenever a tile raised it and picks
    // up the row's published maximum (fetched two groups earlier, see the loop) before it looks at a tile.  Any published
    // value is an approximate score of the row, hence a valid running maximum; starting every split from -inf instead
    // would multiply the record-breaking tiles (8 splits: 71 % of the blocks trigger the candidate path, shared: 25 %).
    uint32_t am[2] = {0u, 0u};
    [[maybe_unused]] auto collect_tile = [&](int jt, auto share_tag) {
        constexpr bool SHARE = decltype(share_tag)::value;
        const int dst0 = jt * FBD + 4 * kh;
        const bool full = (int64_t)(jt + 1) * FBD <= Nd;
        if constexpr (SHARE)   // the two amax loads of this step: g2's and g3's 6 operations are younger
            asm volatile("s_waitcnt vmcnt(%2)" : "+v"(am[0]), "+v"(am[1]) : "n"(2 * NB + PG));
#pragma unroll
        for (int sb = 0; sb < 2; ++sb) {
            float rm = runmax[sb];
            if constexpr (SHARE) rm = fmaxf(rm, from_orderable(am[sb]) * (SCALE * SCALE));
            const float rm_in = rm;
#pragma unroll
            for (int ib = 0; ib < 4; ++ib) {
                if (!full) {   // the last tile of the dst range: rows beyond Nd never match
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (dst0 + ib * 32 + (r & 3) + 8 * (r >> 2) >= Nd) acc[ib][sb][r] = -INFINITY;
                }
                // The block's 16 scores of this lane, as 4 quarters of 4 (quarter k = dst rows 8 k + 0..3).
                // Everything within the window of the running maximum AFTER this block becomes a candidate
                // (a running maximum is never above the final one, so this collects a superset of what the
                // final maximum requires).  Almost always that is nothing, or exactly the block's maximum:
                // that case is handled without a loop -- locate the maximum by compare/select, check that the
                // second largest score stays outside the window, insert.  Ties and near-ties inside one block
                // (rare) take the element-by-element path below.
                const f32x16 &v = acc[ib][sb];
                float q[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) q[k] = vmax2(vmax3(v[4 * k], v[4 * k + 1], v[4 * k + 2]), v[4 * k + 3]);
                const float gm = vmax2(vmax3(q[0], q[1], q[2]), q[3]);
                const float newmax = fmaxf(rm, gm);
                const float thr = newmax - WS;
                const bool trig = gm >= thr && gm > -INFINITY;   // NaN / masked blocks never pass
                if (__any(trig)) {
                    // quarter and element of the (first) maximum
                    const bool c0 = q[0] == gm, c1 = q[1] == gm, c2 = q[2] == gm;
                    float w[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) w[j] = c0 ? v[j] : c1 ? v[4 + j] : c2 ? v[8 + j] : v[12 + j];
                    const int kq = c0 ? 0 : c1 ? 1 : c2 ? 2 : 3;
                    const bool d0 = w[0] == gm, d1 = w[1] == gm, d2 = w[2] == gm;
                    const int eq = d0 ? 0 : d1 ? 1 : d2 ? 2 : 3;
                    // largest score of the block apart from that element
                    const float qo = fmaxf(fmaxf(fmaxf(c0 ? -INFINITY : q[0], (!c0 && c1) ? -INFINITY : q[1]),
                                                 (!c0 && !c1 && c2) ? -INFINITY : q[2]),
                                           (c0 || c1 || c2) ? q[3] : -INFINITY);
                    const float wo = fmaxf(fmaxf(fmaxf(d0 ? -INFINITY : w[0], (!d0 && d1) ? -INFINITY : w[1]),
                                                 (!d0 && !d1 && d2) ? -INFINITY : w[2]),
                                           (d0 || d1 || d2) ? w[3] : -INFINITY);
                    const float second = fmaxf(qo, wo);
                    if (!__any(trig && second >= thr)) {
                        if (trig) {
                            if (cv[sb][3] >= thr)   // evicted entry still inside the window: spill it
                                if (push(srow0 + sb * 32 + l31, cv[sb][3], ci[sb][3])) rm = INFINITY;
                            cv[sb][3] = cv[sb][2]; ci[sb][3] = ci[sb][2];
                            cv[sb][2] = cv[sb][1]; ci[sb][2] = ci[sb][1];
                            cv[sb][1] = cv[sb][0]; ci[sb][1] = ci[sb][0];
                            cv[sb][0] = gm;
                            ci[sb][0] = (uint32_t)(dst0 + ib * 32 + eq + 8 * kq);
                        }
                    } else {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const float x = v[r];
                            if (x >= rm - WS && x > -INFINITY) {
                                const float nrm_ = fmaxf(rm, x);
                                bool full_list = false;
                                if (cv[sb][3] >= nrm_ - WS) full_list = push(srow0 + sb * 32 + l31, cv[sb][3], ci[sb][3]);
                                cv[sb][3] = cv[sb][2]; ci[sb][3] = ci[sb][2];
                                cv[sb][2] = cv[sb][1]; ci[sb][2] = ci[sb][1];
                                cv[sb][1] = cv[sb][0]; ci[sb][1] = ci[sb][0];
                                cv[sb][0] = x;
                                ci[sb][0] = (uint32_t)(dst0 + ib * 32 + (r & 3) + 8 * (r >> 2));
                                rm = full_list ? INFINITY : nrm_;
                            }
                        }
                    }
                }
                rm = fmaxf(rm, gm);
            }
            runmax[sb] = rm;
            if constexpr (SHARE) {
                if (rm > rm_in) atomicMax(&amax[out_row0 + srow0 + sb * 32 + l31], orderable(rm * INV_S2));   // (+inf: the list is full)
            }
        }
    };

    [[maybe_unused]] auto prune_check = [&](int jt) -> uint32_t {
        const float rb = rest_bt[(int64_t)bi * nd_tiles + jt];             // wave-uniform
        uint32_t mask = 0;
#pragma unroll
        for (int sb = 0; sb < 2; ++sb) {
            // a pair can still matter iff partial + rest_a * rest_b >= running max - window
            const float need = runmax[sb] - WS - rest_l[sb] * rb;
#pragma unroll
            for (int ib = 0; ib < 4; ++ib) {
                const f32x16 &v = acc[ib][sb];
                float q[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) q[k] = vmax2(vmax3(v[4 * k], v[4 * k + 1], v[4 * k + 2]), v[4 * k + 3]);
                const float gm = vmax2(vmax3(q[0], q[1], q[2]), q[3]);
                if (__any(gm >= need)) mask |= 1u << (sb * 4 + ib);
            }
        }
        return mask;
    };

#if VTM_FILTER_PRODUCTS > 1 || defined(VTM_FILTER_PHASED)
    {   // the 2- / 3-product variants (and the A/B switch): load phase, then 8-16 MFMAs, per group
        load_a_half(jt0, 0, 0, 0);
        load_a_half(jt0, 0, 0, 1);
        load_b(0, 0, rb[0]);
        load_b(0, 1, rb[1]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();

        int kt = 0, jt = jt0;
        for (int st = 0; st < steps; ++st) {
            const int buf = st & 1;
            const bool wrap = kt + 1 == KT;
            const int ktn = wrap ? 0 : kt + 1, jtn = wrap ? jt + 1 : jt;
            // the last step prefetches too (its own operands again, never used), so that the issue sequence -- and
            // with it every wait count -- is the same in all steps
            const bool more = st + 1 < steps;
            const int ktp = more ? ktn : kt, jtp = more ? jtn : jt;
            // A fragments (dst rows) come from LDS one half-group ahead: the hi halves of group s + 1 are read while
            // the lo products of group s run, the lo halves of group s while its hi products run -- every read has
            // 8 MFMAs (256 cycles) to land, in the registers the previous fragments just vacated
            auto read_a = [&](int which, int s_, h16x8 (&f)[4]) {
#pragma unroll
                for (int ib = 0; ib < 4; ++ib)
                    f[ib] = __builtin_bit_cast(h16x8, sA[buf][which][(s_ * 2 + kh) * FBD + ib * 32 + l31]);
            };
            h16x8 fa[2][4], fl[4];
            read_a(0, 0, fa[0]);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                if (s < 2) load_b(kt, s + 2, rb[s + 2]);
                else load_b(ktp, s - 2, rb[s - 2]);
                ABL_DMA(if (s < 2) load_a_half(jtp, ktp, buf ^ 1, s);)
                // operations issued after B(s): see the table above
                if (s == 0 || s == 3) await_b(std::integral_constant<int, ABL_AWAIT_COUNT(2 * NB + PG)>{}, rb[s]);
                else await_b(std::integral_constant<int, ABL_AWAIT_COUNT(2 * NB + 2 * PG)>{}, rb[s]);
                if constexpr (DST_LO) read_a(1, s, fl);
                else if (s < 3) read_a(0, s + 1, fa[(s + 1) & 1]);   // one group ahead, alternating register sets
                __builtin_amdgcn_sched_barrier(0);
                h16x8 (&fh)[4] = fa[DST_LO ? 0 : (s & 1)];
                auto hi_products = [&](auto first_tag) {
                    constexpr bool FIRST = decltype(first_tag)::value;
#pragma unroll
                    for (int sb = 0; sb < 2; ++sb) {
                        const h16x8 bhf = __builtin_bit_cast(h16x8, rb[s][sb][0]);
                        const h16x8 blf = __builtin_bit_cast(h16x8, rb[s][sb][1]);
#pragma unroll
                        for (int ib = 0; ib < 4; ++ib) {
                            f32x16 c = acc[ib][sb];
                            if constexpr (FIRST) {
#pragma unroll
                                for (int r = 0; r < 16; ++r) c[r] = 0.0f;   // folds into the MFMA's zero C operand
                            }
                            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[ib], bhf, c, 0, 0, 0);
                            if constexpr (SRC_LO) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[ib], blf, c, 0, 0, 0);
                            acc[ib][sb] = c;
                        }
                    }
                };
                if (s == 0 && kt == 0) hi_products(std::true_type{});
                else hi_products(std::false_type{});
                if constexpr (DST_LO) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (s < 3) read_a(0, s + 1, fa[0]);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int sb = 0; sb < 2; ++sb) {
                        const h16x8 bhf = __builtin_bit_cast(h16x8, rb[s][sb][0]);
#pragma unroll
                        for (int ib = 0; ib < 4; ++ib)
                            acc[ib][sb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fl[ib], bhf, acc[ib][sb], 0, 0, 0);
                    }
                }
            }
            if (ABL_WRAP_COND(wrap)) collect_tile(jt, std::false_type{});
            // every wave's DMA pieces of the next tile must have landed before anybody reads them; they are older
            // than the 2 NB loads of groups 2 and 3
            ABL_BARRIER(asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NB) : "memory"); __syncthreads();)
            kt = ktn;
            jt = jtn;
        }
    }
#else
    // ---- shipped loop (one product): every load / LDS read / DMA piece rides BETWEEN two MFMAs ----------------------
    // A wave that issues its ~25 memory instructions of a group in one clump leaves the matrix pipe without work for
    // 100+ cycles each time (the pipe holds one MFMA, not a queue), and the SIMD's other wave is in its own clump a
    // third of the time.  Here the issue order is pinned instruction by instruction (sched_barrier around each): after
    // MFMA j of group s comes ONE of -- j < 4: the LDS read of fragment j of group s + 1 (group 3: of the NEXT step's
    // group 0, out of the other buffer); j = 4, 5: the two B loads of group s + 2; j = 6, 7: a DMA piece (groups 0 and
    // 3 only).  The per-step barrier sits at the end of group 2: by then every wave has read the last fragments of this
    // step's tile (so group 3 may start overwriting it with the tile after next) and the next tile has had two groups
    // to land (its first half is issued in group 3 of the previous step, its second in group 0).
    // Group 1's last two slots fetch the row maxima the other splits have published (X; used at the end of a tile).
    // Vector-memory issue order of a step:  g0: B(2) x2, D x2 | g1: B(3) x2, X x2 | g2: B(0') x2 | g3: B(1') x2, D x2
    // hence the counts at the START of a group (operations issued after the awaited one):
    //     g0 awaits B(0): 2 + 2     g1 awaits B(1): 2 + 2 + 2     g2 awaits B(2): 2 + 2 + 2     g3 awaits B(3): 2 + 2
    // the barrier (after g2's own loads) needs g0's pieces: 2 + 2 + 2 younger loads, and the tile end needs X: 2 + 2 + 2.
    static_assert(NB == 2 && PG == 2, "the schedule below is written for the one-product filter");
    // Addresses are running wave-uniform byte pointers advanced with scalar adds (nothing is recomputed from (kt, jt)):
    //   B fragments of k-step group ks of step kt:   pb + ks * kstep_b   (sb = 1: + 512 through the offset field)
    //   DMA piece t of the wave (panel 2 wave + t / 2, row half t & 1) of a dst tile: pa + (t / 2) * panel_b + (t & 1) * 1024,
    //   into LDS at lds_a + buffer * 16 KiB + t * 1024
    const char *const src_b = reinterpret_cast<const char *>(srch);
    const int64_t kstep_b = bgroup * 16;                         // bytes per k-step group (2 panels of Ns_pad rows)
    const int64_t panel_b = Nd_pad * 16;                         // bytes per dst panel
    const int64_t tile_dk = 8 * panel_b;                         // next channel step of the same dst tile
    const int64_t tile_dwrap = FBD * 16 - (int64_t)(KT - 1) * tile_dk;   // first channel step of the next dst tile
    const uint32_t lds_a = (uint32_t)reinterpret_cast<uintptr_t>((lds_void *)&sA[0][0][wave * 2 * FBD]);
    auto load_b1 = [&](const char *pb_, int ks, int sb, u32x4 &dst) {
        const char *ph = pb_ + ks * kstep_b;   // uniform
        if (sb == 0) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff_b), "s"(ph));
        else asm volatile("global_load_dwordx4 %0, %1, %2 offset:512" : "=v"(dst) : "v"(voff_b), "s"(ph));
    };
    auto load_a_piece = [&](const char *pa_, int buf_, int t) {
        const char *g = pa_ + (t >> 1) * panel_b + (t & 1) * 1024;
        const uint32_t lds_off = lds_a + (uint32_t)buf_ * (uint32_t)sizeof(sA[0]) + (uint32_t)t * 1024u;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(lds_off), "v"(voff_a), "s"(g) : "memory");
    };
    const unsigned int *const amax_rows = amax + out_row0;
    uint32_t voff_m[2];
#pragma unroll
    for (int sb = 0; sb < 2; ++sb) voff_m[sb] = (uint32_t)min(srow0 + sb * 32 + l31, Ns - 1) * 4u;   // padding rows: any valid row
    h16x8 fa[2][4];
    const char *pa1 = reinterpret_cast<const char *>(dsth + (int64_t)wave * 2 * Nd_pad + (int64_t)jt0 * FBD);   // tile of step 0
    const char *pb = src_b;
    {
        load_a_piece(pa1, 0, 0);
        load_a_piece(pa1, 0, 1);
        load_a_piece(pa1, 0, 2);
        load_a_piece(pa1, 0, 3);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (steps > 1) pa1 += KT == 1 ? tile_dwrap : tile_dk;   // tile of step 1
        load_b1(pb, 0, 0, rb[0][0][0]);
        load_b1(pb, 0, 1, rb[0][1][0]);
        load_b1(pb, 1, 0, rb[1][0][0]);
        load_b1(pb, 1, 1, rb[1][1][0]);
        load_a_piece(pa1, 1, 0);   // what group 3 of a previous step would have issued
        load_a_piece(pa1, 1, 1);
#pragma unroll
        for (int ib = 0; ib < 4; ++ib) fa[0][ib] = __builtin_bit_cast(h16x8, sA[0][0][kh * FBD + ib * 32 + l31]);
    }
    int kt = 0, jt = jt0;
    for (int st = 0; st < steps; ++st) {
        const int buf = st & 1;
        const bool wrap = kt + 1 == KT;
        // the last steps prefetch too (valid addresses, never used), so that the issue sequence -- and with it every
        // wait count -- is the same in all steps
        const int kt1 = wrap ? 0 : kt + 1;                        // channel step of step st + 1 (if there is one)
        const bool wrap1 = kt1 + 1 == KT;
        const char *const pbn = st + 1 < steps ? (wrap ? src_b : pb + 4 * kstep_b) : pb;
        const char *const pa2 = st + 2 < steps ? pa1 + (wrap1 ? tile_dwrap : tile_dk) : pa1;   // tile of step st + 2
        auto group = [&](auto s_tag, auto first_tag) {
            constexpr int s = decltype(s_tag)::value;
            constexpr bool FIRST = decltype(first_tag)::value;
            constexpr int COUNT = s == 1 ? NB + 2 * PG : s == 2 ? NB + PG + 2 : NB + 2;
            if constexpr (s != 0)   // group 0's wait precedes the branch on kt (below)
                await_b(std::integral_constant<int, ABL_AWAIT_COUNT(COUNT)>{}, rb[s]);
            // the maxima fetched one step ago (X) are older than the fragments just awaited: only NOW have they certainly
            // landed.  Keeping their registers "in use" up to this point stops the compiler from handing them out while the
            // load is still in flight on the paths that never read them (every step that does not end a tile)
            if constexpr (s == 1) asm volatile("" : : "v"(am[0]), "v"(am[1]));
            h16x8 (&fh)[4] = fa[s & 1];
            h16x8 (&fn)[4] = fa[(s + 1) & 1];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int sb = j >> 2, ib = j & 3;
                bool alive = true;                   // (wave-uniform: a scalar bit test; see "partial-sum pruning")
                if constexpr (!FIRST) alive = (live & (1u << j)) != 0u;
                if (alive) {
                    f32x16 c = acc[ib][sb];
                    if constexpr (FIRST) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) c[r] = 0.0f;   // folds into the MFMA's zero C operand
                    }
                    acc[ib][sb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[ib], __builtin_bit_cast(h16x8, rb[s][sb][0]), c, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (j < 4) {
                    ABL_LDSREAD(if constexpr (s < 3) fn[j] = __builtin_bit_cast(h16x8, sA[buf][0][((s + 1) * 2 + kh) * FBD + j * 32 + l31]);
                                else fn[j] = __builtin_bit_cast(h16x8, sA[buf ^ 1][0][kh * FBD + j * 32 + l31]);)
                    ABL_NO_LDSREAD(fn[j] = fh[j];)
                } else if (j < 6) {
                    ABL_BLOAD(if constexpr (s < 2) load_b1(pb, s + 2, j - 4, rb[s + 2][j - 4][0]);
                              else load_b1(pbn, s - 2, j - 4, rb[s - 2][j - 4][0]);)
                } else {
                    ABL_DMA(if constexpr (s == 0) load_a_piece(pa1, buf ^ 1, j - 4);   /* pieces 2, 3 of the next tile */
                            if constexpr (s == 3) load_a_piece(pa2, buf, j - 6);)      /* pieces 0, 1 of the one after */
                    if constexpr (s == 1)   // agent scope: from L2, where the other workgroups' atomics land
                        asm volatile("global_load_dword %0, %1, %2 sc1" : "=v"(am[j - 6]) : "v"(voff_m[j - 6]), "s"(amax_rows));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        // ONE wait statement for both variants of group 0: the fragments are usable only through the registers this
        // statement returns, and a second copy of it behind the branch would make the compiler copy them BEFORE it
        await_b(std::integral_constant<int, ABL_AWAIT_COUNT(NB + PG)>{}, rb[0]);
        if (kt == 0) group(std::integral_constant<int, 0>{}, std::true_type{});
        else group(std::integral_constant<int, 0>{}, std::false_type{});
        group(std::integral_constant<int, 1>{}, std::false_type{});
        group(std::integral_constant<int, 2>{}, std::false_type{});
        ABL_BARRIER(asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * NB + 2) : "memory"); __syncthreads();)
        group(std::integral_constant<int, 3>{}, std::false_type{});
        if (kt + 1 == KP) {                            // (KP >= KT: pruning is off)
            live = prune_check(jt);
            n_tested += 8;                             // 32 x 32 blocks tested / still alive (flags_out[4], [5])
            n_alive += __popc(live);
            if constexpr (SCOUT) {
                if (live && lane == 0) atomicOr(&tilemap[(int64_t)stg * map_words + (jt >> 5)], 1u << (jt & 31));
            }
        }
        if constexpr (!SCOUT) {
            if (ABL_WRAP_COND(wrap && live)) collect_tile(jt, std::true_type{});
        }
        if (wrap) {
            ++jt;
            live = 0xffu;
        }
        kt = kt1;
        pb = pbn;
        pa1 = pa2;
    }
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the unused prefetches of the last step
    // one pair of atomics per WAVE: per tile they serialise on two words (a counting call took 4.5 instead of 0.75 ms)
    if (count_blocks && lane == 0 && n_tested) {
        if (!SCOUT && tilemap != nullptr) {
            atomicAdd(&flags[7], n_tested);            // range plan: the blocks inside the spans (the scout counted [4], [5])
        } else {
            atomicAdd(&flags[4], n_tested);
            atomicAdd(&flags[5], n_alive);
        }
    }
    if constexpr (SCOUT) return;

    // flush: the entries still inside the window of this lane's final maximum
#pragma unroll
    for (int sb = 0; sb < 2; ++sb) {
        const int64_t srow = srow0 + sb * 32 + l31;
        if (srow >= Ns) continue;
        float rm = runmax[sb];
        if (rm > -INFINITY) {
            // publish this partition's maximum first and prune against what the other partitions of the row
            // have published so far: only entries within the window of the best known maximum can matter
            const uint32_t mine = orderable(rm * INV_S2);
            const uint32_t prev = atomicMax(&amax[out_row0 + srow], mine);
            rm = fmaxf(rm, from_orderable(prev) * (SCALE * SCALE));
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (cv[sb][e] >= rm - WS && cv[sb][e] > -INFINITY) push(srow, cv[sb][e], ci[sb][e]);
    }
}

// ---- seeds: a good running maximum for every src row BEFORE the filter starts ----
// The filter collects candidates against a row's RUNNING maximum and prunes blocks against it, so what it costs depends on
// how early that maximum is good: with the rows of level 2 / the global level arriving in similarity-sorted order a row's
// true match sits anywhere in the dst range and most of the scan runs against a low maximum (more candidates, no pruning;
// a flat image region seen first floods the row's list).  In a video the best match of a token is almost always a token
// at the SAME spatial position of another frame, and positions are known: pool rows below `seed_L` are tokens of the joined
// chunk (position = row % N), the rest carry their position in `pos1` (the anchors' positions, tracked by the host), and
// `table` maps a position to one dst row holding it (identity when nullptr: the first dst frame of a local level).  A
// workgroup's 8-lane groups each take a src row, fetch it and its guess row (5 + 5 coalesced 16-byte pieces per lane at
// C = 320), and publish  (a . b) / (|a| |b|)  -- the score of a REAL pair, fp32, error ~1e-6 -- as the row's starting
// maximum.  Exactness is untouched: any score of an actual pair is a valid running maximum (the window argument needs
// t_ij* >= runmax - W, and t_ij* >= s_ij* - EPS >= s_ig - EPS); a useless guess only fails to help.
template <typename T>
__global__ __launch_bounds__(256) void seed_kernel(const T *__restrict__ x0, int64_t P0, const T *__restrict__ x1, int64_t P1,
                                                   int64_t B, int64_t C, const int32_t *__restrict__ a_rows, int64_t Ns,
                                                   const int32_t *__restrict__ b_rows, int64_t Nd,
                                                   const float *__restrict__ na, const float *__restrict__ nb, int align,
                                                   int64_t seed_L, int64_t N, const int32_t *__restrict__ pos1,
                                                   const int32_t *__restrict__ table, unsigned int *__restrict__ amax, int dry,
                                                   unsigned int *__restrict__ seed_lb, float lb_margin) {
    constexpr int LPR = 8;                                   // lanes per row
    const int64_t g = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / LPR;     // (sample, src row)
    const int sub = threadIdx.x & (LPR - 1);
    if (g >= B * Ns) return;
    const int64_t b = g / Ns, i = g % Ns;
    const int32_t ra = a_rows[b * Ns + i];
    int64_t pos = -1;
    if (ra < seed_L) pos = ra % N;
    else if (pos1 != nullptr && ra - P0 >= 0 && ra - P0 < P1) pos = pos1[b * P1 + (ra - P0)];
    int64_t j = -1;
    if (pos >= 0 && pos < N) j = table ? (int64_t)table[b * N + pos] : pos;
    if (j < 0 || j >= Nd) return;                            // (uniform over the row's 8 lanes)
    const T *pa = pool_row(x0, P0, x1, P1, b, ra, C);
    const T *pb = pool_row(x0, P0, x1, P1, b, b_rows[b * Nd + j], C);
    float acc = 0.0f;
    for (int64_t k = sub * 8; k < C; k += LPR * 8) {
        float fa[8], fb[8];
        load8(pa + k, fa);
        load8(pb + k, fb);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc = __builtin_fmaf(fa[e], fb[e], acc);
    }
#pragma unroll
    for (int off = LPR / 2; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (sub == 0) {
        // two divisions, not one by the product of the norms: the product of two norms near 2^100 overflows (s = 0 -- or a
        // denormal with a large relative error -- would pass the range check and could sit ABOVE every real score of the
        // row).  The published value is lowered by a margin far above the fp32 error of this dot product (~1e-5 at
        // C = 1280) so that it is provably <= the pair's filter score + EPS.
        const float s0 = (acc / na[b * Ns + i]) / nb[b * Nd + j];
        const float s = s0 - 1e-4f;
        // (round 6) ... and nothing is published when the dot product itself left the normal fp32 range: norms near 2^-100 are
        // still "usable", but the products of such rows underflow, acc arrives as 0 or as a denormal with no relative
        // accuracy, and s0 = 0 would be published as a CERTIFIED bound above a row whose real scores are all negative
        const bool acc_ok = __builtin_fabsf(acc) >= 0x1p-100f || (na[b * Ns + i] >= 0x1p-40f && nb[b * Nd + j] >= 0x1p-40f);
        if (s == s && __builtin_fabsf(s) <= 1.5f && acc_ok && !dry) {  // (a row without a usable norm publishes nothing)
            atomicMax(&amax[align ? i : b * Ns + i], orderable(s));
            // ... and, kept apart from the filter's running maximum (which an overflowing row overwrites with +inf), a
            // CERTIFIED lower bound of the row's exact maximum for exact_rows_kernel's tile pruning: this pair's canonical
            // chain value is >= s0 - lb_margin (|fast fp32 dot - canonical chain| <= (4 C + 32) 2^-24, both evaluate the
            // same cosine with the same two norms)
            atomicMax(&seed_lb[align ? i : b * Ns + i], orderable(s0 - lb_margin));
        }
    }
}

// ---- refine: the candidates inside the window of each row's final approximate maximum, re-evaluated exactly ----
// One launch (rounds 1-3: a compaction kernel + a pair kernel).  A workgroup owns RROWS rows:
//   1. it decides what kind of call this is -- every workgroup scans the same few hundred per-tile values prep_operand left
//      (tile_rest: +inf marks a dst tile holding a row without a finite positive norm in [2^-100, 2^100]; zero token -> NaN
//      xhat, merge.py:84 has no eps) and reaches the same answer without a grid-wide exchange: such a call is recomputed as a
//      whole by exact_rows_kernel (flags[0]), the filter's error window means nothing for it;
//   2. per row: the list entries inside the window of the row's final maximum are counted (the first 16 in ONE batch of
//      loads); rows whose list overflowed -- or whose OWN norm is out of range -- go on their sample's list for
//      exact_rows_kernel (one atomic per wave and list);
//   3. the workgroup reserves ONE contiguous slice of the pair list (one atomic), writes its (row, column) pairs there
//      and, behind a barrier, works them off itself, a pair per thread and round: the canonical fp32 chain (x / norm with
//      the operations and roundings of the IEEE expansion, the reciprocal refinement hoisted out of the channel loop),
//      combined with the packed atomicMax of vtm_match.
//   (round 5) A workgroup owns RROWS = 64 rows, not 256: with 4-17 candidates per row (frames of one clip at low noise:
//      5.6 pairs per row over a step, 16 at the global levels) a 256-row workgroup walks 16 rounds of pairs while the
//      launch has one wave per SIMD; four times as many workgroups put four waves on every SIMD.
//   (A screen that scored every pair with a fast, coalesced fp32 dot product first -- 8 lanes per pair, window 2 EPS2 =
//   2 (4 C + 32) 2^-24 -- and ran the canonical chain only on the pairs within that window of the row's best was written
//   twice, exact both times, and SLOWER both times: top global level, 16 pairs per row, 1.51 vs 1.35 ms per call; it adds a
//   pass over the same rows, and the chain pass is not limited by its instruction count.  profiles/r05_d_refine_ab.txt.)
#ifndef VTM_RROWS
#define VTM_RROWS 64        // (A/B build switch: 256 = rounds 1-4)
#endif
constexpr int RROWS = VTM_RROWS;   // rows per refine workgroup (256 threads)
static_assert(RROWS == 32 || RROWS == 64 || RROWS == 128 || RROWS == 256, "refine rows per workgroup");
template <typename T>
__global__ __launch_bounds__(256) void refine_kernel(const T *__restrict__ x0, int64_t P0, const T *__restrict__ x1,
                                                     int64_t P1, int64_t B, int64_t C,
                                                     const int32_t *__restrict__ a_rows, int64_t Ns,
                                                     const int32_t *__restrict__ b_rows, int64_t Nd,
                                                     const float *__restrict__ na, const float *__restrict__ nb,
                                                     int align, int *__restrict__ flags, int64_t rows_out,
                                                     const unsigned int *__restrict__ amax, const int *__restrict__ cnt,
                                                     const uint2 *__restrict__ cand, int *__restrict__ ovf_cnt,
                                                     int *__restrict__ ovf_rows, uint2 *__restrict__ pairs,
                                                     const float *__restrict__ tile_rest, int64_t n_tile_rest,
                                                     unsigned long long *__restrict__ best,
                                                     const int32_t *__restrict__ src_order,
                                                     const int32_t *__restrict__ dst_order) {
    __shared__ int wave_tot[4];
    __shared__ int s_base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // ---- 1. a dst row without a usable norm anywhere?
    {
        bool bad = false;
        for (int64_t t = tid; t < n_tile_rest; t += 256) bad |= !(tile_rest[t] < INFINITY);
        if (__syncthreads_or(bad)) {
            if (blockIdx.x == 0 && tid == 0) {
                flags[0] = 1;
                flags[1] = 1;
            }
            return;
        }
    }
    // ---- 2. per row
    const int64_t row = (int64_t)blockIdx.x * RROWS + tid;
    const bool live = tid < RROWS && row < rows_out;      // (the rows sit in wave 0; all four waves work the pairs off)
    const int n = live ? cnt[row] : 0;
    const float thr = live ? from_orderable(amax[row]) - WINDOW : 0.0f;
    bool bad_src = false;
    if (live) {
        if (align) {
            for (int64_t bi = 0; bi < B; ++bi) bad_src |= !(na[bi * Ns + row] >= 0x1p-100f && na[bi * Ns + row] <= 0x1p100f);
        } else {
            bad_src = !(na[row] >= 0x1p-100f && na[row] <= 0x1p100f);
        }
    }
    const bool ovf = live && (n > CAP || bad_src);
    constexpr int PRE = 16;
    uint2 pre[PRE];
#pragma unroll
    for (int c = 0; c < PRE; ++c) pre[c] = (c < n && !ovf) ? cand[(int64_t)c * rows_out + row] : make_uint2(0x7fc00000u, 0u);
    int ns = 0;
#pragma unroll
    for (int c = 0; c < PRE; ++c) ns += __uint_as_float(pre[c].x) >= thr;   // NaN (not fetched) never counts
    if (!ovf)
        for (int c = PRE; c < n; ++c) ns += __uint_as_float(cand[(int64_t)c * rows_out + row].x) >= thr;
    if (bad_src) flags[1] = 1;
    {
        const int l = (ovf && !align) ? (int)(row / Ns) : 0;
        unsigned long long todo = __ballot(ovf);
        while (todo) {
            const int lead = __ffsll(todo) - 1;
            const int l0 = __shfl(l, lead, 64);
            const unsigned long long grp = __ballot(ovf && l == l0);
            int base = 0;
            if (lane == lead) {
                base = atomicAdd(&ovf_cnt[l0], __popcll(grp));
                atomicAdd(&flags[2], __popcll(grp));
            }
            base = __shfl(base, lead, 64);
            if (ovf && l == l0)
                ovf_rows[(int64_t)l0 * Ns + base + __popcll(grp & ((1ull << lane) - 1ull))] = (int)(row - (int64_t)l0 * Ns);
            todo &= ~grp;
        }
    }
    // ---- 3. the workgroup's slice of the pair list
    int incl = ns;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    int before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        before += w < wave ? wave_tot[w] : 0;
        total += wave_tot[w];
    }
    if (tid == 0) s_base = total > 0 ? atomicAdd(&flags[3], total) : 0;
    __syncthreads();
    const int base = s_base;
    int at = base + before + incl - ns;
    if (ns > 0) {
#pragma unroll
        for (int c = 0; c < PRE; ++c)
            if (__uint_as_float(pre[c].x) >= thr) pairs[at++] = make_uint2((uint32_t)row, pre[c].y);
        for (int c = PRE; c < n; ++c) {
            const uint2 cd = cand[(int64_t)c * rows_out + row];
            if (__uint_as_float(cd.x) >= thr) pairs[at++] = make_uint2((uint32_t)row, cd.y);
        }
    }
    __threadfence_block();
    __syncthreads();
    for (int p = base + tid; p < base + total; p += 256) {
        const uint2 pr = pairs[p];
        const int64_t prow = pr.x;
        const uint32_t col = pr.y;
        const int64_t i = align ? prow : prow % Ns;
        const int64_t bi = align ? (int64_t)(col / (uint32_t)Nd) : prow / Ns;
        const int64_t j = align ? (int64_t)(col % (uint32_t)Nd) : (int64_t)col;
        const T *pa = pool_row(x0, P0, x1, P1, bi, a_rows[bi * Ns + i], C);
        const T *pb = pool_row(x0, P0, x1, P1, bi, b_rows[bi * Nd + j], C);
        const float nrm_a = na[bi * Ns + i], nrm_b = nb[bi * Nd + j];
        const RowDivisor da = row_divisor(nrm_a), db = row_divisor(nrm_b);
        float acc = 0.0f;
        auto chain8 = [&](const float (&fa)[8], const float (&fb)[8]) {
            bool ok = true;
            if constexpr (!std::is_same<T, __half>::value) {   // fp16 tokens: 2^-24 <= |x| and norms < 2^23, always in range
#pragma unroll
                for (int e = 0; e < 8; ++e) ok = ok && div_ok(fa[e], da) && div_ok(fb[e], db);
            }
            if (ok) {
#pragma unroll
                for (int e = 0; e < 8; ++e) acc = __builtin_fmaf(div_by_row(fa[e], da), div_by_row(fb[e], db), acc);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) acc = __builtin_fmaf(fa[e] / nrm_a, fb[e] / nrm_b, acc);
            }
        };
        // 32 channels per step: 4 + 4 independent 16-byte loads in flight per lane (a lane walks its own two rows; with one
        // load per row and step every step costs a cache round trip).  (Fetching the next step's pieces behind this step's
        // chain -- 16 more loads in flight, 192 VGPRs -- measured slower, 53 vs 48 us per call; 64 channels per step, 210 VGPRs: 46.5 vs
        // 47.8 us, inside the noise of a step: neither shipped.)
        int64_t k = 0;
        for (; k + 32 <= C; k += 32) {
            float fa[4][8], fb[4][8];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                load8(pa + k + 8 * u, fa[u]);
                load8(pb + k + 8 * u, fb[u]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) chain8(fa[u], fb[u]);
        }
        for (; k < C; k += 8) {
            float fa[8], fb[8];
            load8(pa + k, fa);
            load8(pb + k, fb);
            chain8(fa, fb);
        }
        // position-ordered call (vtm_match_filtered_ordered; never aligned): row and column in the caller's ORIGINAL indexing,
        // so that the packed maximum's tie rule -- lowest dst index -- is the original one
        // (aligned calls: one result row per src index over all samples' dst rows; the samples' lists are the same rows in the
        // same order, so sample 0's inverse map names the row and the column keeps its sample offset)
        const int64_t orow = src_order ? (align ? (int64_t)src_order[i] : bi * Ns + src_order[prow]) : prow;
        const uint32_t ocol = dst_order ? (uint32_t)dst_order[bi * Nd + j] + (align ? (uint32_t)(bi * Nd) : 0u) : col;
        atomicMax(&best[orow], ((unsigned long long)orderable(acc) << 32) | (uint32_t)(~ocol));
    }
}

// ---- escape: exact fp32-MFMA score tiles for the listed rows (or, flags[0], for every row of the call) ----
// Same arithmetic as match_kernel (match.hip): v_mfma_f32_32x32x2_f32 accumulates in ascending k = the canonical fmaf chain,
// MFMA A operand = dst tile, B operand = src tile, so a lane owns one src row and keeps a running (max, first argmax) pair
// with torch.max's NaN rule.  What differs is where the operands come from: there are no fp32 operand panels on the
// filtered path (they would double its workspace for a pass that normally has nothing to do), so a workgroup normalises
// its tiles ON THE FLY -- token rows through the gather lists, divided by the canonical norms of prep_operand with the
// IEEE-identical per-row reciprocal form of refine_kernel (plain IEEE division when flags[0] says that a norm is outside
// the range where that form is valid) -- into LDS as the k-panel image [g][kh][row][4] the MFMA fragments are read from.
// Per 32-channel step a thread fetches 2 + 2 row pieces (next step's, in flight behind this step's 64 MFMAs), and the
// dst operand is normalised once per 128 listed rows (the scalar row pass this replaces normalised it once per ROW).
// Work items = (list, XS-row tile of the list, sample [aligned: every sample's dst set], dst split); the counts live on
// the device, so a fixed grid strides over the items and leaves at once when there are none.
#ifndef VTM_XS
#define VTM_XS 128          // (A/B build switch: 256 = 64 src rows per wave, one workgroup per CU: profiles/r04_escape_tile.txt)
#endif
constexpr int XS = VTM_XS;   // listed src rows per workgroup: 32 (or two 32-row MFMA blocks) per wave
constexpr int XD = 128;   // dst rows per tile
constexpr int XK = 32;    // channels per step
constexpr int XPD = XD + 1, XPS = XS + 1;   // rows per LDS panel (+1: the 4 pieces of a row land in different bank groups)
constexpr int XSB = XS / 128;                // 32-row src blocks per wave
constexpr int XSU = XS / 64;                 // src rows a thread stages per step
static_assert(XS == 128 || XS == 256, "src tile");

__device__ __forceinline__ int64_t xcdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

template <typename T, bool ORD>
__global__ __launch_bounds__(256, XS == 128 ? 2 : 1) void exact_rows_kernel(
    const T *__restrict__ x0, int64_t P0, const T *__restrict__ x1, int64_t P1, int64_t B, int64_t C,
    const int32_t *__restrict__ a_rows, int64_t Ns, const int32_t *__restrict__ b_rows, int64_t Nd,
    const float *__restrict__ na, const float *__restrict__ nb, int align, const int *__restrict__ flags,
    const int *__restrict__ ovf_cnt, const int *__restrict__ ovf_rows, unsigned long long *__restrict__ best, int nsplit,
    int tiles_per_split, const float *__restrict__ rest_a, const float *__restrict__ rest_bt, int KX, int64_t Ns_pad,
    int64_t rest_tiles, const unsigned int *__restrict__ seed_lb, int *__restrict__ work,
    const int32_t *__restrict__ src_order, const int32_t *__restrict__ dst_order, int32_t *__restrict__ flags_pub) {
    // (round 6) the call's counters for the host -- final since refine_kernel finished, except the work counter [6] this launch
    // hands out -- are published from here when `flags_out` is device-accessible memory (device or pinned host): the last
    // launch of the call does it instead of a 32-byte copy of its own (one dispatch less per matcher call)
    if (flags_pub != nullptr && blockIdx.x == 0 && threadIdx.x < 8) flags_pub[threadIdx.x] = flags[threadIdx.x];
    __shared__ __attribute__((aligned(16))) float sD[8 * XPD * 4];
    // position-ordered call: the ORIGINAL index of every dst row of the tile being scored (double-buffered like the row
    // pointers: the next tile's are staged behind the current tile's last step) -- a lane's running argmax is kept and
    // tie-broken in the original indexing
    __shared__ uint32_t sOrd[ORD ? 2 : 1][ORD ? XD : 1];
    __shared__ __attribute__((aligned(16))) float sS[8 * XPS * 4];
    constexpr int RAW = sizeof(T) == 4 ? 2 : 1;      // 16-byte loads per 8-channel piece

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kh = lane >> 5;
    const bool all = flags[0] != 0;
    const int64_t nlists = align ? 1 : B, nsel = align ? B : 1;
    const int KT = (int)xcdiv(C, XK);
    const int nd_tiles = (int)xcdiv(Nd, XD);
    const int prow = tid >> 2, pg = tid & 3;          // staging role: rows prow, prow + 64; 8-channel group pg of the step

    // Work distribution: the first item of a workgroup is its block index (the normal case -- empty lists -- costs nothing),
    // every further one comes off a device-side counter.  (A static stride hands a workgroup the SAME dst split every
    // time when the split count divides the grid size; with tile pruning the splits inside a flat region cost 2.5x the
    // others and a quarter of the workgroups did all the work: profiles/r05_e_xprune_ab.txt.)
    __shared__ int64_t s_item;
    for (int64_t item = blockIdx.x;;) {
        // ---- which (list, tile, sample, split) ----
        int64_t rem = item, l = 0, cnt_l = 0;
        for (; l < nlists; ++l) {
            cnt_l = all ? Ns : (int64_t)ovf_cnt[l];
            const int64_t items_l = xcdiv(cnt_l, XS) * nsel * nsplit;
            if (rem < items_l) break;
            rem -= items_l;
        }
        if (l == nlists) break;
        const int split = (int)(rem % nsplit);
        rem /= nsplit;
        const int64_t bi = align ? rem % nsel : l;
        const int64_t tile = rem / nsel;
        const int jt0 = split * tiles_per_split, jt1 = min(jt0 + tiles_per_split, nd_tiles);
        auto next_item = [&]() {                      // (all threads; two barriers: s_item is re-used)
            __syncthreads();
            if (tid == 0) s_item = (int64_t)gridDim.x + atomicAdd(work, 1);
            __syncthreads();
            return s_item;
        };
        if (jt0 >= jt1) {
            item = next_item();
            continue;
        }
        auto listed = [&](int r) -> int64_t {         // row (within its sample) of tile entry r, -1 = past the list
            const int64_t at = tile * XS + r;
            if (at >= cnt_l) return -1;
            return all ? at : (int64_t)ovf_rows[l * Ns + at];
        };

        // ---- this thread's src rows (fixed for the item) ----
        const T *ps[XSU];
        RowDivisor ds[XSU];
        bool vs[XSU];
#pragma unroll
        for (int u = 0; u < XSU; ++u) {
            const int64_t i = listed(prow + 64 * u);
            vs[u] = i >= 0;
            const int64_t ii = vs[u] ? i : 0;
            ps[u] = pool_row(x0, P0, x1, P1, bi, a_rows[bi * Ns + ii], C);
            ds[u] = row_divisor(na[bi * Ns + ii]);
        }
        const T *pd[2];
        RowDivisor dd[2];
        bool vd[2];
        auto dst_rows = [&](int jt) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int64_t j = (int64_t)jt * XD + prow + 64 * u;
                vd[u] = j < Nd;
                const int64_t jj = vd[u] ? j : 0;
                pd[u] = pool_row(x0, P0, x1, P1, bi, b_rows[bi * Nd + jj], C);
                dd[u] = row_divisor(nb[bi * Nd + jj]);
                if constexpr (ORD)
                    if (pg == 0) sOrd[jt & 1][prow + 64 * u] = vd[u] ? (uint32_t)dst_order[bi * Nd + jj] : 0xffffffffu;
            }
        };
        uint4 raw[2 + XSU][RAW];                       // pieces in flight: dst rows 0 / 1, then the src rows
        auto fetch = [&](int kt) {
            const int64_t k = (int64_t)kt * XK + pg * 8;
            const bool kin = k < C;                    // C % 8 == 0: a piece is inside or outside as a whole
#pragma unroll
            for (int h = 0; h < RAW; ++h) {
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    raw[u][h] = (kin && vd[u]) ? *reinterpret_cast<const uint4 *>(pd[u] + k + h * (8 / RAW)) : make_uint4(0, 0, 0, 0);
#pragma unroll
                for (int u = 0; u < XSU; ++u)
                    raw[2 + u][h] = (kin && vs[u]) ? *reinterpret_cast<const uint4 *>(ps[u] + k + h * (8 / RAW)) : make_uint4(0, 0, 0, 0);
            }
        };
        // one piece: 8 token values -> xhat (bit-identical to x / norm) -> the two k-halves of its panel group
        auto stage = [&](const uint4 (&rw)[RAW], const RowDivisor &d, bool valid, float *panel, int XP, int row) {
            float f[8], o[8];
            if constexpr (sizeof(T) == 4) {
                f[0] = __uint_as_float(rw[0].x); f[1] = __uint_as_float(rw[0].y); f[2] = __uint_as_float(rw[0].z); f[3] = __uint_as_float(rw[0].w);
                f[4] = __uint_as_float(rw[1].x); f[5] = __uint_as_float(rw[1].y); f[6] = __uint_as_float(rw[1].z); f[7] = __uint_as_float(rw[1].w);
            } else {
                const T *e = reinterpret_cast<const T *>(&rw[0]);
#pragma unroll
                for (int j = 0; j < 8; ++j) f[j] = to_f32(e[j]);
            }
            if (!valid) {                              // rows past the list / past Nd: zero operands (never published)
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = 0.0f;
            } else if (all || !(d.n >= 0x1p-100f && d.n <= 0x1p100f)) {
                // a norm outside the range where the per-row reciprocal form is the IEEE quotient (whole-call mode, or a
                // listed row that is here BECAUSE of its norm): IEEE division
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = f[j] / d.n;
            } else {
                bool ok = true;
                if constexpr (!std::is_same<T, __half>::value) {   // fp16 tokens are always in range (refine_kernel)
#pragma unroll
                    for (int j = 0; j < 8; ++j) ok = ok && div_ok(f[j], d);
                }
                if (ok) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] = div_by_row(f[j], d);
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] = f[j] / d.n;
                }
            }
            // panel (pg, kh) holds channels 8 pg + kh + {0, 2, 4, 6} of a row: ascending k for the 4 MFMAs that read it
            *reinterpret_cast<float4 *>(&panel[((pg * 2 + 0) * XP + row) * 4]) = make_float4(o[0], o[2], o[4], o[6]);
            *reinterpret_cast<float4 *>(&panel[((pg * 2 + 1) * XP + row) * 4]) = make_float4(o[1], o[3], o[5], o[7]);
        };

        f32x16 acc[4][XSB];
        float bestv[XSB];
        uint32_t besti[XSB];
        // Tile pruning (round 5): after the KX 32-channel steps in front of the filter's pruning cut a pair's exact score is
        // at most  partial + |xhat_a rest| |xhat_b rest|  (Cauchy-Schwarz; prep_operand's rest norms of the fp16 operands,
        // 1024 xhat rounded, cover the exact ones with a factor 1 + 2^-9; + 1e-4 for the fp32 chain's own rounding).  If
        // that is below a CERTIFIED lower bound of the row's maximum -- the lane's running exact maximum, or the seed pair's
        // score -- for every row of the tile, no score of the tile can be (or tie) a row maximum and the remaining steps are
        // skipped.  Flat regions are what fills this kernel's lists: a flat src row scores ~1 against the flat dst rows and
        // ~0 against everything else, so the tiles outside the flat region die at 40 % depth.  NaN scores (a row without
        // a usable norm) only occur in whole-call mode or for rows whose rest norm is +inf: no pruning there.
        float floor_[XSB], rest_l[XSB];
        const bool can_prune = !all && rest_a != nullptr && KX > 0 && KX < KT;
#pragma unroll
        for (int sb = 0; sb < XSB; ++sb) {
            bestv[sb] = -INFINITY;
            besti[sb] = 0xffffffffu;
            const int64_t i = listed(wave * (32 * XSB) + sb * 32 + l31);
            floor_[sb] = (can_prune && i >= 0) ? from_orderable(seed_lb[align ? i : l * Ns + i]) : -INFINITY;
            rest_l[sb] = (can_prune && i >= 0) ? rest_a[bi * Ns_pad + i] : 0.0f;
            if (i < 0) floor_[sb] = INFINITY;            // rows past the list never keep a tile alive
        }
        dst_rows(jt0);
        fetch(0);
        int kt = 0, jt = jt0;
        while (jt < jt1) {
            __syncthreads();                           // everybody has read the previous step's panels
#pragma unroll
            for (int u = 0; u < 2; ++u) stage(raw[u], dd[u], vd[u], sD, XPD, prow + 64 * u);
#pragma unroll
            for (int u = 0; u < XSU; ++u) stage(raw[2 + u], ds[u], vs[u], sS, XPS, prow + 64 * u);
            __syncthreads();
            const bool wrap = kt + 1 == KT;
            if (!wrap || jt + 1 < jt1) {               // next step's pieces fly behind this step's MFMAs
                if (wrap) dst_rows(jt + 1);
                fetch(wrap ? 0 : kt + 1);
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float4 af[4];
#pragma unroll
                for (int ib = 0; ib < 4; ++ib) af[ib] = *reinterpret_cast<const float4 *>(&sD[((g * 2 + kh) * XPD + ib * 32 + l31) * 4]);
                float bv[XSB][4];
#pragma unroll
                for (int sb = 0; sb < XSB; ++sb) {
                    const float4 bf = *reinterpret_cast<const float4 *>(&sS[((g * 2 + kh) * XPS + wave * (32 * XSB) + sb * 32 + l31) * 4]);
                    bv[sb][0] = bf.x; bv[sb][1] = bf.y; bv[sb][2] = bf.z; bv[sb][3] = bf.w;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
#pragma unroll
                    for (int ib = 0; ib < 4; ++ib) {
                        const float av = e == 0 ? af[ib].x : e == 1 ? af[ib].y : e == 2 ? af[ib].z : af[ib].w;
#pragma unroll
                        for (int sb = 0; sb < XSB; ++sb) {
                            f32x16 c = acc[ib][sb];
                            if (kt == 0 && g == 0 && e == 0) {
#pragma unroll
                                for (int r = 0; r < 16; ++r) c[r] = 0.0f;
                            }
                            acc[ib][sb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[sb][e], c, 0, 0, 0);
                        }
                    }
                }
            }
            if (wrap) {   // dst tile finished: fold this lane's 64 scores into its running (max, first argmax)
                const int dst0 = jt * XD + 4 * kh;
                const bool full = (int64_t)(jt + 1) * XD <= Nd;
#pragma unroll
                for (int sb = 0; sb < XSB; ++sb) {
                    float bv_ = bestv[sb];
                    uint32_t bi_ = besti[sb];
#pragma unroll
                    for (int ib = 0; ib < 4; ++ib) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int d = dst0 + ib * 32 + (r & 3) + 8 * (r >> 2);
                            const float sc = acc[ib][sb][r];
                            bool upd = !(sc <= bv_) && (bv_ == bv_);   // greater, or the first NaN (torch.max)
                            uint32_t od = (uint32_t)d;
                            if constexpr (ORD) {
                                // rows arrive in position order: "first" means lowest ORIGINAL index -- among equal
                                // scores, and among NaNs
                                od = sOrd[jt & 1][d - jt * XD];
                                upd = upd || (((sc == bv_) || (sc != sc && bv_ != bv_)) && od < bi_);
                            }
                            if (!full) upd = upd && (d < Nd);
                            bv_ = upd ? sc : bv_;
                            bi_ = upd ? od : bi_;
                        }
                    }
                    bestv[sb] = bv_;
                    besti[sb] = bi_;
                }
                ++jt;
                kt = 0;
            } else if (can_prune && kt + 1 == KX) {
                // can any pair of this tile still reach a row maximum?
                const float rb = rest_bt[bi * rest_tiles + jt] * (1.0f + 0x1p-9f) * INV_S2;
                bool dead = true;
#pragma unroll
                for (int sb = 0; sb < XSB; ++sb) {
                    float pm = -INFINITY;
#pragma unroll
                    for (int ib = 0; ib < 4; ++ib)
#pragma unroll
                        for (int r = 0; r < 16; ++r) pm = fmaxf(pm, acc[ib][sb][r]);
                    dead = dead && (pm + rest_l[sb] * rb + 1e-4f < fmaxf(bestv[sb], floor_[sb]));
                }
                if (__syncthreads_and(dead)) {         // (workgroup-uniform) -> next tile; the fetch in flight is replaced
                    ++jt;
                    kt = 0;
                    if (jt < jt1) {
                        dst_rows(jt);
                        fetch(0);
                    }
                } else {
                    ++kt;
                }
            } else {
                ++kt;
            }
        }
#pragma unroll
        for (int sb = 0; sb < XSB; ++sb) {
            const int64_t i = listed(wave * (32 * XSB) + sb * 32 + l31);
            if (i >= 0 && besti[sb] != 0xffffffffu) {
                const uint32_t col = besti[sb] + (align ? (uint32_t)(bi * Nd) : 0u);
                const int64_t orow = align ? (ORD ? (int64_t)src_order[i] : i) : l * Ns + (ORD ? (int64_t)src_order[l * Ns + i] : i);
                atomicMax(&best[orow], ((unsigned long long)orderable(bestv[sb]) << 32) | (uint32_t)(~col));
            }
        }
        item = next_item();
    }
}

inline size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

struct Layout {
    size_t na, nb, ah, al, bh, bl, amax, cnt, seedlb, tilemap, cand, flags, ovf_cnt, ovf, pairs, rest_a, rest_bt, rest_a2, rest_bt2, total;
    int64_t Ns_pad, Nd_pad, C64;
};

Layout make_layout(int64_t B, int64_t C, int64_t Ns, int64_t Nd, int align) {
    Layout L;
    L.Ns_pad = vtm::cdiv(Ns, FBS) * FBS;
    L.Nd_pad = vtm::cdiv(Nd, FBS) * FBS;
    L.C64 = vtm::cdiv(C, 64) * 64;
    const int64_t rows_out = align ? Ns : B * Ns;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o += align_up(bytes); return at; };
    L.na = take((size_t)B * Ns * 4);
    L.nb = take((size_t)B * Nd * 4);
    L.ah = take((size_t)B * L.C64 * L.Ns_pad * 2);
    L.al = take(SRC_LO ? (size_t)B * L.C64 * L.Ns_pad * 2 : 0);
    L.bh = take((size_t)B * L.C64 * L.Nd_pad * 2);
    L.bl = take(DST_LO ? (size_t)B * L.C64 * L.Nd_pad * 2 : 0);
    L.amax = take((size_t)rows_out * 4);      // amax, cnt and flags are contiguous: cleared together
    L.cnt = take((size_t)rows_out * 4);
    L.seedlb = take((size_t)rows_out * 4);    // certified lower bounds of the rows' exact maxima (seed_kernel -> exact_rows_kernel)
    // the scout's map of live (src tile, dst tile) pairs: one bit per dst tile (inside the cleared range)
    L.tilemap = take((size_t)B * (L.Ns_pad / FBS) * ((L.Nd_pad / FBD + 31) / 32) * 4);
    L.flags = take(256);
    L.ovf_cnt = take((size_t)B * 4);          // per-sample overflow-list lengths (inside the cleared range)
    L.cand = take((size_t)rows_out * CAP * 8);
    L.ovf = take((size_t)rows_out * 4);
    L.pairs = take((size_t)rows_out * CAP * 8);
    L.rest_a = take((size_t)B * L.Ns_pad * 4);
    L.rest_bt = take((size_t)B * (L.Nd_pad / FBD) * 4);
    L.rest_a2 = take((size_t)B * L.Ns_pad * 4);              // ... for the shallow scout's cut
    L.rest_bt2 = take((size_t)B * (L.Nd_pad / FBD) * 4);
    L.total = o;
    return L;
}

// Tuning / A-B hooks of the host side, read from the environment ONCE per process (they were getenv calls on
// every matcher call).  Every one of them is result-preserving: they move launch shapes and pruning depths, never what
// `best` holds.  (A hook that changes results belongs behind a VTM_EXP_* build switch of ablate.h, where it sets a bit of
// VTM_ABLATIONS and the Python binding refuses to load the library.)
struct DebugEnv {
    int kp5, kp, nsplit;              // -1 = unset
    bool seed_dry, nsplit_r4, no_xprune;
};
const DebugEnv &debug_env() {
    static const DebugEnv e = [] {
        auto num = [](const char *name) {
            const char *v = getenv(name);
            return v ? atoi(v) : -1;
        };
        DebugEnv d;
        d.kp5 = num("VTM_DEBUG_KP5");
        d.kp = num("VTM_DEBUG_KP");
        d.nsplit = num("VTM_DEBUG_NSPLIT");
        d.seed_dry = getenv("VTM_DEBUG_SEED_DRY") != nullptr;
        d.nsplit_r4 = getenv("VTM_DEBUG_NSPLIT_R4") != nullptr;
        d.no_xprune = getenv("VTM_DEBUG_NOXPRUNE") != nullptr;
        return d;
    }();
    return e;
}

}  // namespace

VTM_EXPORT size_t vtm_match_filtered_ws_bytes(int64_t B, int64_t C, int64_t Ns, int64_t Nd, int align) {
    if (B <= 0 || C <= 0 || Ns <= 0 || Nd <= 0) return 0;
    return make_layout(B, C, Ns, Nd, align).total;
}

static int match_filtered_impl(const void *x0, int64_t P0, const void *x1, int64_t P1, int dtype, int64_t B,
                               int64_t C, const int32_t *a_rows, int64_t Ns, const int32_t *b_rows, int64_t Nd,
                               int align, void *ws, size_t ws_bytes, uint64_t *best, int32_t *flags_out,
                               int64_t seed_L, int64_t seed_N, const int32_t *seed_pos1, const int32_t *seed_table,
                               int mode, const int32_t *src_order, const int32_t *dst_order, vtm_stream_t stream) {
    VTM_REQUIRE(x0 && a_rows && b_rows && ws && best, "vtm_match_filtered: null pointer");
    const int scout_steps = (mode >> 8) & 0xff;      // VTM_MATCH_SCOUT_STEPS(k): the scout tests after k pipeline steps
    mode &= 0xff;
    VTM_REQUIRE((src_order == nullptr) == (dst_order == nullptr), "vtm_match_filtered_ordered: both inverse maps or none");
    VTM_REQUIRE(mode == VTM_MATCH_ONE_LAUNCH || mode == VTM_MATCH_SCOUT_RANGE, "vtm_match_filtered: bad mode %d", mode);
    VTM_REQUIRE(B > 0 && C > 0 && C % 8 == 0 && Ns > 0 && Nd > 0, "vtm_match_filtered: bad sizes");
    VTM_REQUIRE(P1 == 0 || x1, "vtm_match_filtered: x1 is null but P1 > 0");
    VTM_REQUIRE(B * Nd < (1ll << 32) - 1, "vtm_match_filtered: index space overflow");
    VTM_REQUIRE(C <= MAX_C, "vtm_match_filtered: C=%lld > %d (error window derived for C <= %d; use vtm_match)",
                (long long)C, MAX_C, MAX_C);
    const Layout L = make_layout(B, C, Ns, Nd, align);
    if (ws_bytes < L.total)
        return vtm::fail(VTM_EWORKSPACE, "vtm_match_filtered: workspace %zu < %zu bytes", ws_bytes, L.total);
    hipStream_t s = vtm::as_stream(stream);
    char *w = static_cast<char *>(ws);
    float *na = (float *)(w + L.na), *nb = (float *)(w + L.nb);
    uint4 *ah = (uint4 *)(w + L.ah), *al = (uint4 *)(w + L.al), *bh = (uint4 *)(w + L.bh), *bl = (uint4 *)(w + L.bl);
    unsigned int *amax = (unsigned int *)(w + L.amax);
    int *cnt = (int *)(w + L.cnt), *flags = (int *)(w + L.flags);
    uint2 *cand = (uint2 *)(w + L.cand);
    int *ovf_rows = (int *)(w + L.ovf), *ovf_cnt = (int *)(w + L.ovf_cnt);
    uint2 *pairs = (uint2 *)(w + L.pairs);
    const int64_t rows_out = align ? Ns : B * Ns;
    VTM_REQUIRE(rows_out * CAP < (1ll << 31), "vtm_match_filtered: too many rows for the 32-bit candidate index");

    VTM_REQUIRE(dtype == VTM_F32 || dtype == VTM_F16 || dtype == VTM_BF16, "vtm_match_filtered: bad dtype");
    // partial-sum pruning: check after KP of the KT channel steps; off for short rows and for the 2- / 3-product
    // builds (their lo products are not covered by the hi rest norms)
    const int KT = (int)(L.C64 / FBK);
    int KP = (VTM_FILTER_PRODUCTS == 1 && KT >= 4) ? (2 * KT + 2) / 5 : 0;   // 40 % depth (profiles/r04_kp_sweep.txt)
    const DebugEnv &dbg = debug_env();   // tuning / A-B hooks, read ONCE per process (all of them leave the results unchanged)
    if (VTM_FILTER_PRODUCTS == 1 && KT == 5 && dbg.kp5 >= 0 && dbg.kp5 < KT) KP = dbg.kp5;   // the C = 320 levels alone (KT = 5)
    if (VTM_FILTER_PRODUCTS == 1 && dbg.kp >= 0 && dbg.kp < KT) KP = dbg.kp;                 // 0 = off, else the step after which blocks are tested
    const bool prune = KP > 0 && KP < KT;
    // (the per-tile values are written for every call: refine_kernel reads the "row without a usable norm" mark from them)
    float *rest_a = (float *)(w + L.rest_a), *rest_bt = (float *)(w + L.rest_bt);
    const int64_t cut = prune ? (int64_t)KP * FBK : L.C64;
    // the scout of the scout + range plan may test EARLIER than the filter proper (its own rest norms): a low-noise clip's
    // dead tiles are dead after one step already, and the scout's cost is its steps
    const int KPS = (prune && mode == VTM_MATCH_SCOUT_RANGE && scout_steps > 0 && scout_steps < KP) ? scout_steps : KP;
    float *rest_a2 = (float *)(w + L.rest_a2), *rest_bt2 = (float *)(w + L.rest_bt2);
    const int64_t cut2 = KPS < KP ? (int64_t)KPS * FBK : cut;
    {
        // one launch: canonical norms + fp16 panels of both operands; it also clears amax / cnt / flags (contiguous)
        // and zero-fills `best`
        const SplitArgs A0{a_rows, Ns, na, ah, SRC_LO ? al : nullptr, L.Ns_pad, rest_a, nullptr, KPS < KP ? rest_a2 : nullptr, nullptr};
        const SplitArgs A1{b_rows, Nd, nb, bh, DST_LO ? bl : nullptr, L.Nd_pad, nullptr, rest_bt, nullptr, KPS < KP ? rest_bt2 : nullptr};
        const int64_t total = B * (L.Ns_pad + L.Nd_pad);
        unsigned long long *bp0 = reinterpret_cast<unsigned long long *>(best);
        uint32_t *zp = reinterpret_cast<uint32_t *>(w + L.amax);
        const int64_t zw = (int64_t)((L.cand - L.amax) / 4);
        const dim3 grid((unsigned)vtm::cdiv(total, 64 * PREP_WAVES)), block(64 * PREP_WAVES);
        switch (dtype) {
            case VTM_F32:
                hipLaunchKernelGGL(prep_operand<float>, grid, block, 0, s, (const float *)x0, P0, (const float *)x1, P1,
                                   B, C, A0, A1, L.C64, zp, zw, bp0, rows_out, cut, cut2);
                break;
            case VTM_F16:
                hipLaunchKernelGGL(prep_operand<__half>, grid, block, 0, s, (const __half *)x0, P0, (const __half *)x1,
                                   P1, B, C, A0, A1, L.C64, zp, zw, bp0, rows_out, cut, cut2);
                break;
            default:
                hipLaunchKernelGGL(prep_operand<vtm_bf16>, grid, block, 0, s, (const vtm_bf16 *)x0, P0,
                                   (const vtm_bf16 *)x1, P1, B, C, A0, A1, L.C64, zp, zw, bp0, rows_out, cut, cut2);
        }
    }

    if (seed_N > 0) {   // starting maxima from same-position guesses (a kernel boundary behind prep_operand: norms, cleared amax)
        const int dry = dbg.seed_dry;   // A/B hook: the seeds are computed and thrown away
        unsigned int *seedlb = (unsigned int *)(w + L.seedlb);
        const float lb_margin = (4.0f * (float)C + 32.0f) * 0x1p-24f + 1e-6f;   // see seed_kernel
        const dim3 grid((unsigned)vtm::cdiv(B * Ns * 8, 256)), block(256);
        switch (dtype) {
            case VTM_F32:
                hipLaunchKernelGGL(seed_kernel<float>, grid, block, 0, s, (const float *)x0, P0, (const float *)x1, P1, B, C, a_rows,
                                   Ns, b_rows, Nd, (const float *)na, (const float *)nb, align, seed_L, seed_N, seed_pos1, seed_table, amax, dry, seedlb, lb_margin);
                break;
            case VTM_F16:
                hipLaunchKernelGGL(seed_kernel<__half>, grid, block, 0, s, (const __half *)x0, P0, (const __half *)x1, P1, B, C,
                                   a_rows, Ns, b_rows, Nd, (const float *)na, (const float *)nb, align, seed_L, seed_N, seed_pos1, seed_table, amax, dry, seedlb, lb_margin);
                break;
            default:
                hipLaunchKernelGGL(seed_kernel<vtm_bf16>, grid, block, 0, s, (const vtm_bf16 *)x0, P0, (const vtm_bf16 *)x1, P1, B,
                                   C, a_rows, Ns, b_rows, Nd, (const float *)na, (const float *)nb, align, seed_L, seed_N, seed_pos1, seed_table, amax, dry, seedlb, lb_margin);
        }
    }
    {
        const int ns_tiles = (int)(L.Ns_pad / FBS), nd_tiles = (int)(L.Nd_pad / FBD);
        const int total_src_tiles = (int)(B * ns_tiles);
        // patch size: at most 16 src tiles while their (hi) operands fit comfortably in one L2 (16 x 256 rows x C x
        // 2 B <= 3 MiB), else 8 -- and balanced: patch g runs on XCD g % 8, so the tiles are cut into equal patches
        // whose number per XCD is the same for all XCDs (17 patches of 16 tiles would keep one XCD busy for three
        // patches while the other seven idle after two)
        const int max_patch = (int64_t)16 * FBS * L.C64 * 2 <= (3 << 20) ? 16 : 8;
        const int tiles_per_xcd = (int)vtm::cdiv(total_src_tiles, 8);
        const int patches_per_xcd = (int)vtm::cdiv(tiles_per_xcd, max_patch);
        const int patch_tiles = (int)vtm::cdiv(tiles_per_xcd, patches_per_xcd);
        const int ngroups = (int)vtm::cdiv(total_src_tiles, patch_tiles);
        // workgroups one XCD has to run (2 per CU at a time) for a split count
        auto wgs_per_xcd = [&](int ns) {
            const int tps = (int)vtm::cdiv(nd_tiles, ns);
            return patches_per_xcd * patch_tiles * (int)vtm::cdiv(nd_tiles, tps);
        };
        int nsplit;
        if (dbg.nsplit_r4) {
            // rounds 1-4: enough workgroups to fill the chip, >= 4 dst tiles each, at most 8 (every split contributes >= 1
            // candidate per row), 8 whenever the dst axis is long enough, 7 or 6 when that saves a round (A/B hook)
            int64_t want = vtm::cdiv(1536, (int64_t)ns_tiles * B);
            nsplit = (int)(want < 1 ? 1 : want);
            if (nd_tiles >= 32) nsplit = 8;
            if (nsplit > nd_tiles / 4) nsplit = nd_tiles / 4 > 0 ? nd_tiles / 4 : 1;
            if (nsplit > 8) nsplit = 8;
            if (nsplit == 8) {
                const int slots = vtm::device_cus() / 8 * 2;
                for (int ns = 7; ns >= 6; --ns)
                    if (wgs_per_xcd(8) > slots && wgs_per_xcd(ns) <= slots) {
                        nsplit = ns;
                        break;
                    }
            }
        } else {
            // Round 5: the split count that minimises  rounds x (tiles per split + 1.5) + 0.25 splits  -- an XCD runs its
            // workgroups in rounds of its CUs x 2, a
            // workgroup costs its dst tiles plus ~1.5 tiles of prologue / flush, and every split adds candidates to the rows'
            // lists.  Reproduces the best of tools/sweep_nsplit.py's 2..14 sweep on all six cfg-2 shapes
            // (profiles/r05_h_sweeps.txt): level 2 takes 5 splits instead of 8 (768 workgroups = 1.5 rounds -> 480 = one round:
            // -6 %), the mid levels 5 / 14 instead of 8 (-7 % / -10 %), the top global level 7.
            const int slots = vtm::device_cus() / 8 * 2;
            double best_cost = 1e30;
            nsplit = 1;
            for (int ns = 1; ns <= 16 && ns <= nd_tiles; ++ns) {
                const int tps = (int)vtm::cdiv(nd_tiles, ns);
                if ((int)vtm::cdiv(nd_tiles, tps) != ns) continue;          // same workgroups as a smaller count
                if (tps < 4 && ns > 1) continue;                            // >= 4 dst tiles per split (rounds 1-4's rule: every split
                                                                            // starts its rows' running maxima and lists anew)
                const int wgs = wgs_per_xcd(ns);
                // (a single round must leave a few slots free: 63 workgroups on 64 slots measured as two rounds)
                const double rounds = (wgs <= slots && wgs > slots - slots / 16) ? 2.0 : (double)vtm::cdiv(wgs, slots);
                const double cost = rounds * (tps + 1.5) + 0.25 * ns;
                if (cost < best_cost - 1e-9) {
                    best_cost = cost;
                    nsplit = ns;
                }
            }
        }
        if (dbg.nsplit >= 1 && dbg.nsplit <= 16 && dbg.nsplit <= nd_tiles) nsplit = dbg.nsplit;   // tuning hook (tools/sweep_nsplit.py)
        const int tiles_per_split = (int)vtm::cdiv(nd_tiles, nsplit);
        nsplit = (int)vtm::cdiv(nd_tiles, tiles_per_split);
        const int64_t grid = (int64_t)8 * vtm::cdiv(ngroups, 8) * patch_tiles * nsplit;
        const int64_t c_run = L.C64;   // (rounds 4-5 had a timing hook here that cut the channel loop short -- WRONG results from a
                                       // shipped library by an environment variable: removed in round 6)
        // scout + range plan (see filter_kernel): needs the pruning test, the seeds (without a starting maximum nothing is
        // dead) and dst frames of whole tiles -- one split per dst frame, so that a span is the live tiles of ONE frame
        const int map_words = (nd_tiles + 31) / 32;
        unsigned int *tilemap = (unsigned int *)(w + L.tilemap);
        // (position-ordered call: the whole dst axis is ONE position-major run; the splits stay the one-launch plan's -- a src
        // tile's span then lies in one or two of them and the other workgroups leave at once, while a level whose spans are long
        // keeps its parallelism: with a single split corr05's top global level took 2.9 ms instead of 1.3)
        const bool ordered = src_order != nullptr;
        const bool range_plan = mode == VTM_MATCH_SCOUT_RANGE && prune && c_run == L.C64 &&
                                (ordered ? seed_N > 0 && nd_tiles >= 2 : seed_N >= 2 * FBD && seed_N % FBD == 0);
        if (range_plan) {
            hipLaunchKernelGGL(filter_kernel<true>, dim3((unsigned)grid), dim3(THREADS), 0, s, ah, al, bh, bl, Ns, Nd, L.Ns_pad,
                               L.Nd_pad, L.C64, align, ns_tiles, nd_tiles, nsplit, tiles_per_split, total_src_tiles, patch_tiles,
                               amax, cnt, cand, (int)rows_out, flags, (const float *)(KPS < KP ? rest_a2 : rest_a),
                               (const float *)(KPS < KP ? rest_bt2 : rest_bt), KPS, flags_out != nullptr ? 1 : 0, tilemap, map_words);
            const int tps_r = ordered ? tiles_per_split : (int)(seed_N / FBD);
            const int nsplit_r = (int)vtm::cdiv(nd_tiles, tps_r);
            const int64_t grid_r = (int64_t)8 * vtm::cdiv(ngroups, 8) * patch_tiles * nsplit_r;
            hipLaunchKernelGGL(filter_kernel<false>, dim3((unsigned)grid_r), dim3(THREADS), 0, s, ah, al, bh, bl, Ns, Nd,
                               L.Ns_pad, L.Nd_pad, L.C64, align, ns_tiles, nd_tiles, nsplit_r, tps_r, total_src_tiles, patch_tiles,
                               amax, cnt, cand, (int)rows_out, flags, (const float *)rest_a, (const float *)rest_bt, KP,
                               flags_out != nullptr ? 1 : 0, tilemap, map_words);
        } else {
            hipLaunchKernelGGL(filter_kernel<false>, dim3((unsigned)grid), dim3(THREADS), 0, s, ah, al, bh, bl, Ns, Nd, L.Ns_pad,
                               L.Nd_pad, c_run, align, ns_tiles, nd_tiles, nsplit, tiles_per_split, total_src_tiles, patch_tiles,
                               amax, cnt, cand, (int)rows_out, flags, prune ? (const float *)rest_a : nullptr,
                               prune ? (const float *)rest_bt : nullptr, prune ? KP : 0x7fffffff, flags_out != nullptr ? 1 : 0,
                               (unsigned int *)nullptr, map_words);
        }
    }
    bool flags_direct = false;
    {
        const dim3 grid((unsigned)vtm::cdiv(rows_out, RROWS)), block(256);
        unsigned long long *bp = reinterpret_cast<unsigned long long *>(best);
        const int64_t n_tile_rest = B * (L.Nd_pad / FBD);
#define VTM_REFINE_ARGS a_rows, Ns, b_rows, Nd, na, nb, align, flags, rows_out, amax, cnt, cand, ovf_cnt, ovf_rows, pairs, \
                        (const float *)rest_bt, n_tile_rest, bp, src_order, dst_order
        // the escape: a fixed grid strides over the (device-side) lists of overflowed rows -- normally empty, then the
        // workgroups leave at once; dst splits of >= 8 tiles so that a short list still spreads over the chip
        const int xd_tiles = (int)vtm::cdiv(Nd, XD);
        int xsplit = xd_tiles / 8;
        xsplit = xsplit < 1 ? 1 : xsplit > 16 ? 16 : xsplit;
        const int xtps = (int)vtm::cdiv(xd_tiles, xsplit);
        xsplit = (int)vtm::cdiv(xd_tiles, xtps);
        const dim3 xgrid((unsigned)((XS == 128 ? 2 : 1) * vtm::device_cus()));
        // exact_rows_kernel's tile pruning: the rest norms prep_operand wrote for the filter's cut, in 32-channel steps
        const int KX = dbg.no_xprune ? 0 : (int)(cut / XK);
        // where the counters go: written by exact_rows_kernel itself when the destination is memory a kernel can store to
        // (device, or host memory the runtime has pinned / registered), copied behind the call otherwise (pageable host memory)
        int32_t *flags_pub = nullptr;
        if (flags_out) {
            hipPointerAttribute_t attr;
            if (hipPointerGetAttributes(&attr, flags_out) == hipSuccess &&
                (attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeHost || attr.type == hipMemoryTypeManaged))
                flags_pub = attr.type == hipMemoryTypeHost && attr.devicePointer ? (int32_t *)attr.devicePointer : flags_out;
            else
                (void)hipGetLastError();          // (an unregistered pointer is an "error" of the query, not of the call)
        }
        flags_direct = flags_pub != nullptr;
#define VTM_XPRUNE_ARGS (const float *)rest_a, (const float *)rest_bt, KX, L.Ns_pad, L.Nd_pad / FBD, (const unsigned int *)(w + L.seedlb), \
                        flags + 6, src_order, dst_order, flags_pub
        switch (dtype) {
            case VTM_F32:
                hipLaunchKernelGGL(refine_kernel<float>, grid, block, 0, s, (const float *)x0, P0, (const float *)x1, P1,
                                   B, C, VTM_REFINE_ARGS);
                hipLaunchKernelGGL((src_order ? exact_rows_kernel<float, true> : exact_rows_kernel<float, false>), xgrid, block, 0, s, (const float *)x0, P0, (const float *)x1, P1,
                                   B, C, a_rows, Ns, b_rows, Nd, na, nb, align, flags, ovf_cnt, ovf_rows, bp, xsplit, xtps, VTM_XPRUNE_ARGS);
                break;
            case VTM_F16:
                hipLaunchKernelGGL(refine_kernel<__half>, grid, block, 0, s, (const __half *)x0, P0, (const __half *)x1,
                                   P1, B, C, VTM_REFINE_ARGS);
                hipLaunchKernelGGL((src_order ? exact_rows_kernel<__half, true> : exact_rows_kernel<__half, false>), xgrid, block, 0, s, (const __half *)x0, P0, (const __half *)x1,
                                   P1, B, C, a_rows, Ns, b_rows, Nd, na, nb, align, flags, ovf_cnt, ovf_rows, bp, xsplit, xtps, VTM_XPRUNE_ARGS);
                break;
            default:
                hipLaunchKernelGGL(refine_kernel<vtm_bf16>, grid, block, 0, s, (const vtm_bf16 *)x0, P0,
                                   (const vtm_bf16 *)x1, P1, B, C, VTM_REFINE_ARGS);
                hipLaunchKernelGGL((src_order ? exact_rows_kernel<vtm_bf16, true> : exact_rows_kernel<vtm_bf16, false>), xgrid, block, 0, s, (const vtm_bf16 *)x0, P0,
                                   (const vtm_bf16 *)x1, P1, B, C, a_rows, Ns, b_rows, Nd, na, nb, align, flags, ovf_cnt,
                                   ovf_rows, bp, xsplit, xtps, VTM_XPRUNE_ARGS);
        }
    }
    if (int rc = vtm::launch_status("vtm_match_filtered")) return rc;

    if (flags_out && !flags_direct) {
        // (pageable host memory: the copy direction is taken from the pointers)
        const hipError_t e = hipMemcpyAsync(flags_out, flags, 8 * sizeof(int), hipMemcpyDefault, s);
        if (e != hipSuccess) return vtm::fail(VTM_ELAUNCH, "vtm_match_filtered: copy: %s", hipGetErrorString(e));
    }
    return VTM_OK;
}

VTM_EXPORT int vtm_match_filtered(const void *x0, int64_t P0, const void *x1, int64_t P1, int dtype, int64_t B,
                                  int64_t C, const int32_t *a_rows, int64_t Ns, const int32_t *b_rows, int64_t Nd,
                                  int align, void *ws, size_t ws_bytes, uint64_t *best, int32_t *flags_out,
                                  vtm_stream_t stream) {
    return match_filtered_impl(x0, P0, x1, P1, dtype, B, C, a_rows, Ns, b_rows, Nd, align, ws, ws_bytes, best, flags_out, 0, 0,
                               nullptr, nullptr, VTM_MATCH_ONE_LAUNCH, nullptr, nullptr, stream);
}

VTM_EXPORT int vtm_match_filtered_seeded(const void *x0, int64_t P0, const void *x1, int64_t P1, int dtype, int64_t B,
                                         int64_t C, const int32_t *a_rows, int64_t Ns, const int32_t *b_rows, int64_t Nd,
                                         int align, void *ws, size_t ws_bytes, uint64_t *best, int32_t *flags_out,
                                         int64_t seed_L, int64_t seed_N, const int32_t *seed_pos1, const int32_t *seed_table,
                                         vtm_stream_t stream) {
    VTM_REQUIRE(seed_N >= 0 && seed_L >= 0, "vtm_match_filtered_seeded: bad seed description");
    return match_filtered_impl(x0, P0, x1, P1, dtype, B, C, a_rows, Ns, b_rows, Nd, align, ws, ws_bytes, best, flags_out, seed_L,
                               seed_N, seed_pos1, seed_table, VTM_MATCH_ONE_LAUNCH, nullptr, nullptr, stream);
}

VTM_EXPORT int vtm_match_filtered_plan(const void *x0, int64_t P0, const void *x1, int64_t P1, int dtype, int64_t B,
                                       int64_t C, const int32_t *a_rows, int64_t Ns, const int32_t *b_rows, int64_t Nd,
                                       int align, void *ws, size_t ws_bytes, uint64_t *best, int32_t *flags_out,
                                       int64_t seed_L, int64_t seed_N, const int32_t *seed_pos1, const int32_t *seed_table,
                                       int mode, vtm_stream_t stream) {
    VTM_REQUIRE(seed_N >= 0 && seed_L >= 0, "vtm_match_filtered_plan: bad seed description");
    return match_filtered_impl(x0, P0, x1, P1, dtype, B, C, a_rows, Ns, b_rows, Nd, align, ws, ws_bytes, best, flags_out, seed_L,
                               seed_N, seed_pos1, seed_table, mode, nullptr, nullptr, stream);
}

VTM_EXPORT int vtm_match_filtered_ordered(const void *x0, int64_t P0, const void *x1, int64_t P1, int dtype, int64_t B,
                                          int64_t C, const int32_t *a_sorted, int64_t Ns, const int32_t *b_sorted, int64_t Nd,
                                          int align, void *ws, size_t ws_bytes, uint64_t *best, int32_t *flags_out, int64_t seed_L,
                                          int64_t seed_N, const int32_t *seed_pos1, const int32_t *seed_table, int mode,
                                          const int32_t *a_order, const int32_t *b_order, vtm_stream_t stream) {
    VTM_REQUIRE(seed_N >= 0 && seed_L >= 0, "vtm_match_filtered_ordered: bad seed description");
    VTM_REQUIRE(a_order && b_order, "vtm_match_filtered_ordered: null inverse map");
    return match_filtered_impl(x0, P0, x1, P1, dtype, B, C, a_sorted, Ns, b_sorted, Nd, align, ws, ws_bytes, best, flags_out, seed_L,
                               seed_N, seed_pos1, seed_table, mode, a_order, b_order, stream);
}

namespace vtm {
int filter_ablations() { return VTM_ABLATIONS; }
}  // namespace vtm

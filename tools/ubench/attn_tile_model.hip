// Micro-benchmark: the INSTRUCTION MIX of attention_kernel<half,40>'s tile (per wave and 64 keys: 6 x v_mfma_f32_32x32x16_f16 in two
// dependent chains of 3, 32 v_exp_f32, 16 v_cvt_pk_f16_f32, 8 v_pk_maximum3_f16 + compare, 8 v_permlane16_swap, 12 x
// v_mfma_f32_16x16x32_f16) with every operand in registers -- no LDS, no global memory, no barrier.  What it takes per tile on a SIMD
// shared by 4 waves is the floor the real kernel can approach by scheduling alone; the difference to the kernel is what LDS latency,
// the per-tile barrier and the staging cost.
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form=1 tools/ubench/attn_tile_model.hip -o tools/ubench/attn_tile_model
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

// Round 6, VERDICT r05 item 1b: an exponent path without v_exp_f32.  The score arrives from the MFMA as z = (x + 15) / 64
// (scale and shift ride in the contraction's spare k-slots), ONE v_cvt_pknorm_u16_f32 per pair clamps, scales and packs it
// to the fp16 BIT PATTERN [exponent = floor(x) + 15 | mantissa = frac(x)], i.e. Schraudolph's (1 + f) 2^floor(x); the rest
// is packed fp16: g = 1 + f (v_and_or_b32), a cubic in g (3 v_pk_fma_f16) and the exponent put back --
//   POLY 7: p = 2^floor(x) * P3(g),  P3 ~ 2^(g - 1)  (v_and_b32 + v_pk_mul_f16): 7 instructions per PAIR of scores,
//            max relative error 1.3e-3, rms 4.3e-4 (exp + round-to-fp16: 4.8e-4 / 2.1e-4);
//   POLY 6: p = t * C3(g),  C3 ~ 2^(g - 1) / g  (t itself as the fp16 value): 6 instructions, max 3.1e-3, rms 1.1e-3
// against 2 v_exp_f32 + 1 v_cvt_pk_f16_f32 = 3 instructions but 2 of them quarter rate.
template <int POLY>
__device__ __forceinline__ uint32_t exp_pair_poly(float z0, float z1) {
    const u16x2 u2 = __builtin_amdgcn_cvt_pknorm_u16(z0, z1);
    const uint32_t t = __builtin_bit_cast(uint32_t, u2);
    uint32_t gb;   // (t & 0x03ff03ff) | 0x3c003c00 in ONE instruction (the compiler emits v_and + v_or); one scalar operand per VOP3
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(gb) : "v"(t), "s"(0x03ff03ffu), "v"(0x3c003c00u));
    const h16x2 g = __builtin_bit_cast(h16x2, gb);
    if constexpr (POLY == 7) {
        const h16x2 c3 = {(_Float16)0.07923143f, (_Float16)0.07923143f}, c2 = {(_Float16)-0.0130719f, (_Float16)-0.0130719f};
        const h16x2 c1 = {(_Float16)0.48468515f, (_Float16)0.48468515f}, c0 = {(_Float16)0.44906588f, (_Float16)0.44906588f};
        h16x2 acc = __builtin_elementwise_fma(g, c3, c2);
        acc = __builtin_elementwise_fma(g, acc, c1);
        acc = __builtin_elementwise_fma(g, acc, c0);
        const h16x2 e = __builtin_bit_cast(h16x2, t & 0x7c007c00u);
        return __builtin_bit_cast(uint32_t, (h16x2)(e * acc));
    } else {
        const h16x2 c3 = {(_Float16)-0.10246134f, (_Float16)-0.10246134f}, c2 = {(_Float16)0.69064004f, (_Float16)0.69064004f};
        const h16x2 c1 = {(_Float16)-1.35417387f, (_Float16)-1.35417387f}, c0 = {(_Float16)1.76527665f, (_Float16)1.76527665f};
        h16x2 acc = __builtin_elementwise_fma(g, c3, c2);
        acc = __builtin_elementwise_fma(g, acc, c1);
        acc = __builtin_elementwise_fma(g, acc, c0);
        return __builtin_bit_cast(uint32_t, (h16x2)(__builtin_bit_cast(h16x2, t) * acc));
    }
}

__device__ __forceinline__ uint32_t pmax3(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t d;
    asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

// MODE 0: the kernel's order (QK^T, exps, check, PV).  MODE 1: no exps (v_mov instead).  MODE 2: no MFMAs at all (VALU only).
// MODE 3: the PV MFMAs only (the unused QK^T is eliminated).  MODE 4: as 0 with PV from 32-row blocks (8 x 32x32x16, no swaps).
// MODE 6 / 7: as 0 with the packed-fp16 polynomial exponent (exp_pair_poly<7> / <6>) instead of v_exp_f32 + v_cvt_pk.
// MODE 8: as 0 with HALF the pairs through exp_pair_poly<7> (is there a second issue port to win?  r01: no).
template <int MODE, int NT>
__global__ __launch_bounds__(NT) void tile_model(float *out, int iters, float seed, unsigned long long *cyc) {
    const unsigned long long c_begin = clock64();   // s_memtime: shader-clock cycles
    h16x8 q[3], kf[2][3], vf[2][3];
    for (int i = 0; i < 3; ++i)
        for (int e = 0; e < 8; ++e) {
            q[i][e] = (_Float16)(seed * 0.01f + 0.001f * e);
            for (int b = 0; b < 2; ++b) {
                kf[b][i][e] = (_Float16)(seed * 0.02f + 0.001f * (e + b));
                vf[b][i][e] = (_Float16)(seed * 0.03f + 0.002f * (e + b));
            }
        }
    f32x4 o[3][2];
    for (int d = 0; d < 3; ++d)
        for (int h = 0; h < 2; ++h)
            for (int e = 0; e < 4; ++e) o[d][h][e] = 0.0f;
    f32x16 o32[2];
    for (int d = 0; d < 2; ++d)
        for (int r = 0; r < 16; ++r) o32[d][r] = 0.0f;
    uint32_t flag = 0;
    f32x16 s_next[2];
    for (int kb = 0; kb < 2; ++kb)
        for (int r = 0; r < 16; ++r) s_next[kb][r] = seed;
    for (int it = 0; it < iters; ++it) {
        // keep the loop body from being hoisted: the K fragments "change" every tile
        asm volatile("" : "+v"(kf[0][0]), "+v"(kf[1][0]));
        f32x16 s[2];
        if constexpr (MODE == 5) {
            // software pipeline: this iteration's scores were computed LAST iteration; issue the next tile's QK^T now, so
            // that its MFMAs run beside this tile's exps (independent work inside ONE wave)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) s[kb] = s_next[kb];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s_next[kb][r] = 0.0f;
#pragma unroll
                for (int ks = 0; ks < 3; ++ks) s_next[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[kb][ks], q[ks], s_next[kb], 0, 0, 0);
            }
        } else if constexpr (MODE != 2) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < 16; ++r) s[kb][r] = 0.0f;
#pragma unroll
                for (int ks = 0; ks < 3; ++ks) s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[kb][ks], q[ks], s[kb], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        } else {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) s[kb][r] = seed * 1e-3f * (r + it);
            asm volatile("" : "+v"(s[0]), "+v"(s[1]));
        }
        h16x8 pf[4];
        if constexpr (MODE != 3) {
#pragma unroll
            for (int st = 0; st < 4; ++st)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float sv = s[st >> 1][8 * (st & 1) + e];
                    float p;
                    if constexpr (MODE == 1) asm volatile("v_mov_b32 %0, %1" : "=v"(p) : "v"(sv));
                    else p = __builtin_amdgcn_exp2f(sv);
                    pf[st][e] = (_Float16)p;
                }
            if constexpr (MODE == 6 || MODE == 7 || MODE == 8) {
#pragma unroll
                for (int st = 0; st < 4; ++st) {
                    if (MODE == 8 && (st & 1)) continue;      // (half / half: steps 1 and 3 keep the v_exp path above)
                    u32x4 w;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        w[j] = exp_pair_poly<(MODE == 7 ? 6 : 7)>(s[st >> 1][8 * (st & 1) + 2 * j], s[st >> 1][8 * (st & 1) + 2 * j + 1]);
                    pf[st] = __builtin_bit_cast(h16x8, w);    // (the exps of this step above are dead code now)
                }
            }
            uint32_t pw[16];
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const u32x4 w = __builtin_bit_cast(u32x4, pf[st]);
#pragma unroll
                for (int j = 0; j < 4; ++j) pw[4 * st + j] = w[j];
            }
#pragma unroll
            for (int j = 0; j < 5; ++j) pw[j] = pmax3(pw[3 * j], pw[3 * j + 1], pw[3 * j + 2]);
            const uint32_t pr = pmax3(pmax3(pw[0], pw[1], pw[2]), pmax3(pw[3], pw[4], pw[15]), pw[15]);
            flag |= (max(pr >> 16, pr & 0xffffu) > 0x5C00u) ? 1u : 0u;
        } else {
#pragma unroll
            for (int st = 0; st < 4; ++st) pf[st] = q[st % 3];
        }
        if constexpr (MODE == 4) {   // PV from 32-row blocks: 2 blocks x 4 k-steps of 16 keys, P straight from the packing
#pragma unroll
            for (int dv = 0; dv < 2; ++dv)
#pragma unroll
                for (int st = 0; st < 4; ++st) o32[dv] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[dv][st % 3], pf[st], o32[dv], 0, 0, 0);
        } else
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            u32x4 x = __builtin_bit_cast(u32x4, pf[2 * ks]), y = __builtin_bit_cast(u32x4, pf[2 * ks + 1]);
            if constexpr (MODE != 3) {
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const auto r = __builtin_amdgcn_permlane16_swap(x[w], y[w], false, false);
                    x[w] = r[0];
                    y[w] = r[1];
                }
            }
            const h16x8 p0 = __builtin_bit_cast(h16x8, x), p1 = __builtin_bit_cast(h16x8, y);
            if constexpr (MODE != 2) {
#pragma unroll
                for (int dv = 0; dv < 3; ++dv) {
                    o[dv][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf[ks][dv], p0, o[dv][0], 0, 0, 0);
                    o[dv][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf[ks][dv], p1, o[dv][1], 0, 0, 0);
                }
            } else {
                o[0][0][0] += (float)p0[0] + (float)p1[0];
            }
        }
    }
    float acc = (float)flag;
    for (int d = 0; d < 2; ++d)
        for (int r = 0; r < 16; ++r) acc += o32[d][r];
    for (int d = 0; d < 3; ++d)
        for (int h = 0; h < 2; ++h)
            for (int e = 0; e < 4; ++e) acc += o[d][h][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    // wave 0's own duration: the shader clock of the run when every SIMD holds ONE wave (with more, the oldest wave is
    // served first and finishes early, so its duration says nothing about the launch)
    if (cyc && blockIdx.x == 0 && threadIdx.x == 0) *cyc = clock64() - c_begin;
}

// Round 6: the 64-queries-per-wave tile (two 32-query sub-tiles A, B sharing the K / V^T fragments) as a SKEWED in-wave
// pipeline -- every matrix phase has independent VALU work of the same wave in its basic block:
//   phase 1:  S_B = K Q_B^T (6 MFMA) + PV of B's PREVIOUS tile (12 MFMA)   beside   exps / pack / check / swaps of A
//   phase 2:  PV of A (12 MFMA) + S_A of the NEXT tile (6 MFMA)            beside   exps / pack / check / swaps of B
// SGB = 0: plain program order inside a phase (compiler's schedule); SGB = 1: one MFMA, then 5 (32x32) / 3 (16x16) VALU
// (sched_group_barrier).  Per iteration = TWO tile-waves of the other modes.
template <int SGB, int NT>
__global__ __launch_bounds__(NT, 2) void tile_model2(float *out, int iters, float seed, unsigned long long *cyc) {
    const unsigned long long c_begin = clock64();
    h16x8 q[2][3], kf[2][3], vf[2][3];
    for (int i = 0; i < 3; ++i)
        for (int e = 0; e < 8; ++e)
            for (int b = 0; b < 2; ++b) {
                q[b][i][e] = (_Float16)(seed * 0.01f + 0.001f * (e + 3 * b));
                kf[b][i][e] = (_Float16)(seed * 0.02f + 0.001f * (e + b));
                vf[b][i][e] = (_Float16)(seed * 0.03f + 0.002f * (e + b));
            }
    f32x4 o[2][3][2];
    for (int a = 0; a < 2; ++a)
        for (int d = 0; d < 3; ++d)
            for (int h = 0; h < 2; ++h)
                for (int e = 0; e < 4; ++e) o[a][d][h][e] = 0.0f;
    uint32_t flag = 0;
    f32x16 sA[2], sB[2];
    for (int kb = 0; kb < 2; ++kb)
        for (int r = 0; r < 16; ++r) sA[kb][r] = seed, sB[kb][r] = seed;
    h16x8 pB[2][2];     // swapped P^T operands of B's previous tile: [k-step][query half]
    for (int ks = 0; ks < 2; ++ks)
        for (int h = 0; h < 2; ++h) pB[ks][h] = q[0][ks];
    auto softmax = [&](const f32x16 (&s)[2], h16x8 (&pout)[2][2]) {      // exps, pack, check, swaps
        h16x8 pf[4];
#pragma unroll
        for (int st = 0; st < 4; ++st)
#pragma unroll
            for (int e = 0; e < 8; ++e) pf[st][e] = (_Float16)__builtin_amdgcn_exp2f(s[st >> 1][8 * (st & 1) + e]);
        uint32_t pw[16];
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            const u32x4 w = __builtin_bit_cast(u32x4, pf[st]);
#pragma unroll
            for (int j = 0; j < 4; ++j) pw[4 * st + j] = w[j];
        }
#pragma unroll
        for (int j = 0; j < 5; ++j) pw[j] = pmax3(pw[3 * j], pw[3 * j + 1], pw[3 * j + 2]);
        const uint32_t pr = pmax3(pmax3(pw[0], pw[1], pw[2]), pmax3(pw[3], pw[4], pw[15]), pw[15]);
        flag |= (max(pr >> 16, pr & 0xffffu) > 0x5C00u) ? 1u : 0u;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            u32x4 x = __builtin_bit_cast(u32x4, pf[2 * ks]), y = __builtin_bit_cast(u32x4, pf[2 * ks + 1]);
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const auto r = __builtin_amdgcn_permlane16_swap(x[w], y[w], false, false);
                x[w] = r[0];
                y[w] = r[1];
            }
            pout[ks][0] = __builtin_bit_cast(h16x8, x);
            pout[ks][1] = __builtin_bit_cast(h16x8, y);
        }
    };
    auto qk = [&](f32x16 (&s)[2], int sub) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kb][r] = 0.0f;
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[kb][ks], q[sub][ks], s[kb], 0, 0, 0);
        }
    };
    auto pv = [&](int sub, const h16x8 (&p)[2][2]) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int dv = 0; dv < 3; ++dv) {
                o[sub][dv][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf[ks][dv], p[ks][0], o[sub][dv][0], 0, 0, 0);
                o[sub][dv][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf[ks][dv], p[ks][1], o[sub][dv][1], 0, 0, 0);
            }
    };
    auto interleave = [&]() {
        if constexpr (SGB == 1) {
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x2, 5, 0);
            }
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x2, 3, 0);
            }
        }
    };
    for (int it = 0; it < iters; ++it) {
        asm volatile("" : "+v"(kf[0][0]), "+v"(kf[1][0]));
        h16x8 pA[2][2];
        // ---- phase 1
        __builtin_amdgcn_sched_barrier(0);
        qk(sB, 1);
        pv(1, pB);
        softmax(sA, pA);
        interleave();
        __builtin_amdgcn_sched_barrier(0);
        // ---- phase 2
        asm volatile("" : "+v"(kf[0][1]), "+v"(kf[1][1]));     // (the next tile's K fragments)
        pv(0, pA);
        qk(sA, 0);
        softmax(sB, pB);
        interleave();
        __builtin_amdgcn_sched_barrier(0);
    }
    float acc = (float)flag;
    for (int a = 0; a < 2; ++a)
        for (int d = 0; d < 3; ++d)
            for (int h = 0; h < 2; ++h)
                for (int e = 0; e < 4; ++e) acc += o[a][d][h][e];
    for (int kb = 0; kb < 2; ++kb) acc += sA[kb][0] + sB[kb][0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (cyc && blockIdx.x == 0 && threadIdx.x == 0) *cyc = clock64() - c_begin;
}

template <int SGB, int NT>
void run2(const char *name, int wgs_per_cu) {
    const int cus = 256, iters = 2000;
    float *out;
    (void)hipMalloc(&out, sizeof(float) * cus * wgs_per_cu * NT);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    unsigned long long *cyc;
    (void)hipMalloc(&cyc, sizeof(*cyc));
    hipLaunchKernelGGL((tile_model2<SGB, NT>), dim3(cus * wgs_per_cu), dim3(NT), 0, 0, out, 10, 1.0f, (unsigned long long *)nullptr);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((tile_model2<SGB, NT>), dim3(cus * wgs_per_cu), dim3(NT), 0, 0, out, iters, 1.0f, cyc);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const int wps = NT / 256 * wgs_per_cu;
    const double ns = ms * 1e6 / ((double)iters * 2 * wps);   // per 32-query tile-wave, like the other modes
    unsigned long long hc = 0;
    (void)hipMemcpy(&hc, cyc, sizeof(hc), hipMemcpyDeviceToHost);
    printf("%-48s %d waves/SIMD: %.3f ms -> %.1f ns per tile-wave per SIMD (= %.0f cycles @2.4 GHz; matrix pipe alone: 384)", name,
           wps, ms, ns, ns * 2.4);
    if (wps == 1) printf("  [measured: %.0f shader cycles per tile-wave, clock %.2f GHz]", (double)hc / iters / 2, (double)hc / (ms * 1e6));
    printf("\n");
    (void)hipFree(out);
    (void)hipFree(cyc);
}

template <int MODE, int NT>
void run(const char *name, int wgs_per_cu) {
    const int cus = 256, iters = 4000;
    float *out;
    (void)hipMalloc(&out, sizeof(float) * cus * wgs_per_cu * NT);
    dim3 grid(cus * wgs_per_cu), block(NT);   // NT / 256 waves of a workgroup per SIMD
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    unsigned long long *cyc;
    (void)hipMalloc(&cyc, sizeof(*cyc));
    hipLaunchKernelGGL((tile_model<MODE, NT>), grid, block, 0, 0, out, 10, 1.0f, (unsigned long long *)nullptr);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((tile_model<MODE, NT>), grid, block, 0, 0, out, iters, 1.0f, cyc);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const int wps = NT / 256 * wgs_per_cu;
    const double ns = ms * 1e6 / ((double)iters * wps);   // per tile-wave one SIMD executed
    unsigned long long hc = 0;
    (void)hipMemcpy(&hc, cyc, sizeof(hc), hipMemcpyDeviceToHost);
    printf("%-48s %d waves/SIMD: %.3f ms -> %.1f ns per tile-wave per SIMD (= %.0f cycles @2.4 GHz; matrix pipe alone: 384)", name,
           wps, ms, ns, ns * 2.4);
    if (wps == 1) printf("  [measured: %.0f shader cycles per tile-wave, clock %.2f GHz]", (double)hc / iters, (double)hc / (ms * 1e6));
    printf("\n");
    (void)hipFree(out);
    (void)hipFree(cyc);
}

// accuracy of exp_pair_poly on the hardware (v_cvt_pknorm's rounding included): x on a 2^-12 grid over [-14, 8)
template <int POLY>
__global__ void poly_accuracy(float *out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = -14.0f + (float)i * (22.0f / n);
    const uint32_t w = exp_pair_poly<POLY>((x + 15.0f) / 64.0f, (x + 15.0f) / 64.0f);
    const h16x2 p = __builtin_bit_cast(h16x2, w);
    out[i] = (float)p[0] / exp2f(x) - 1.0f;
}

template <int POLY>
void accuracy() {
    const int n = 22 * 4096;
    float *d, *h = (float *)malloc(sizeof(float) * n);
    (void)hipMalloc(&d, sizeof(float) * n);
    hipLaunchKernelGGL(poly_accuracy<POLY>, dim3((n + 255) / 256), dim3(256), 0, 0, d, n);
    (void)hipMemcpy(h, d, sizeof(float) * n, hipMemcpyDeviceToHost);
    double mx = 0, sq = 0, mean = 0;
    for (int i = 0; i < n; ++i) {
        mx = fabs(h[i]) > mx ? fabs(h[i]) : mx;
        sq += (double)h[i] * h[i];
        mean += h[i];
    }
    printf("polynomial exponent, %d instr / pair: relative error of p over x in [-14, 8): max %.2e  rms %.2e  mean %.2e"
           "   (v_exp_f32 + round to fp16: max 4.8e-04 rms 2.1e-04)\n", POLY, mx, sqrt(sq / n), mean / n);
    (void)hipFree(d);
    free(h);
}

int main() {
    accuracy<7>();
    accuracy<6>();
    for (int w : {1, 2}) {
        run<0, 512>("kernel mix (QK^T, exp, check, PV16)", w);
        run<1, 512>("same without the exps (v_mov)", w);
        run<2, 512>("VALU part only", w);
        run<3, 512>("PV MFMAs only (12 x 16x16x32)", w);
        run<4, 512>("PV from 32-row blocks (8 x 32x32x16, no swaps)", w);
        run<5, 512>("software-pipelined (QK^T of t+1 beside exps of t)", w);
        run<6, 512>("polynomial exponent, 7 instr / pair (no v_exp)", w);
        run<7, 512>("polynomial exponent, 6 instr / pair (no v_exp)", w);
        run<8, 512>("half v_exp, half polynomial (7 instr / pair)", w);
    }
    run2<0, 256>("64 q / wave, skewed pipeline, compiler order", 1);
    run2<1, 256>("64 q / wave, skewed pipeline, MFMA : VALU groups", 1);
    run2<0, 512>("64 q / wave, skewed pipeline, compiler order", 1);
    run2<1, 512>("64 q / wave, skewed pipeline, MFMA : VALU groups", 1);
    run<0, 768>("kernel mix, 12-wave workgroup", 1);
    run<5, 768>("software-pipelined, 12-wave workgroup", 1);
    run<0, 256>("kernel mix, 4-wave workgroup", 1);
    run<5, 256>("software-pipelined, 4-wave workgroup", 1);
    run<5, 256>("software-pipelined, 4-wave workgroups", 3);
    return 0;
}

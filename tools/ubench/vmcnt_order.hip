// Probe: does s_waitcnt vmcnt(N) see vector-memory LOADS of different kinds retire in issue order on gfx950?
// The hand-counted waits of filter_kernel (match_filter.hip) rest on it: "vmcnt(k) has passed => every load older than the
// k youngest has landed".  A wave issues a SLOW load A (a cold line far away: HBM), then a FAST load B (one hot line: L2),
// then `s_waitcnt vmcnt(1)` -- by the in-order rule A has landed now -- and looks whether A's data is there:
//     A kind 0: LDS-DMA (global_load_lds_dwordx4)      checked by reading the LDS words back (pre-filled with a sentinel)
//     A kind 1: global_load_dwordx4 into registers      checked by copying the (sentinel-initialised) registers
//     B kind 0: global_load_dword        1: global_load_dword sc1        2: global_load_dword sc0 sc1      3: nt
// Prints the number of trials in which A had NOT landed (0 everywhere = in order).
// hipcc --offload-arch=gfx950 -O3 tools/ubench/vmcnt_order.hip -o tools/ubench/vmcnt_order
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

template <int AK, int BK, int W = 1>
__global__ __launch_bounds__(64) void probe(const uint4 *cold, size_t cold_n, const unsigned *hot, int trials, unsigned *bad,
                                            unsigned seed) {
    __shared__ __attribute__((aligned(16))) uint4 s[64];
    const int lane = threadIdx.x;
    unsigned state = seed + blockIdx.x * 7919u;
    unsigned nbad = 0;
    const uint32_t lds0 = (uint32_t)reinterpret_cast<uintptr_t>((lds_void *)&s[0]);
    for (int t = 0; t < trials; ++t) {
        state = state * 1664525u + 1013904223u;                      // wave-uniform pseudo-random cold line
        const size_t at = ((size_t)(state >> 4) * 64) % (cold_n - 64);
        const uint4 *pa = cold + at;                                  // 1 KiB the wave has (almost certainly) never touched
        s[lane] = make_uint4(0xdeadbeefu, 0xdeadbeefu, 0xdeadbeefu, 0xdeadbeefu);
        __syncthreads();
        unsigned hotv = hot[0];                                       // keep the hot line hot
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        u32x4 ra = {0xdeadbeefu, 0xdeadbeefu, 0xdeadbeefu, 0xdeadbeefu};
        unsigned rbv = 0;
        const uint32_t voff = (uint32_t)lane * 16u;
        if constexpr (AK == 0)
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(lds0), "v"(voff), "s"(pa) : "memory");
        else
            asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(ra) : "v"(voff), "s"(pa) : "memory");
        const uint32_t zoff = 0;
        if constexpr (BK == 0) asm volatile("global_load_dword %0, %1, %2" : "=v"(rbv) : "v"(zoff), "s"(hot) : "memory");
        if constexpr (BK == 1) asm volatile("global_load_dword %0, %1, %2 sc1" : "=v"(rbv) : "v"(zoff), "s"(hot) : "memory");
        if constexpr (BK == 2) asm volatile("global_load_dword %0, %1, %2 sc0 sc1" : "=v"(rbv) : "v"(zoff), "s"(hot) : "memory");
        if constexpr (BK == 3) asm volatile("global_load_dword %0, %1, %2 nt" : "=v"(rbv) : "v"(zoff), "s"(hot) : "memory");
        // by the in-order rule: at most the youngest (B) is still pending => A has landed
        u32x4 seen;
        if constexpr (AK == 0) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(W) : "memory");
            asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(seen) : "v"(lds0 + voff) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(%8)\n\tv_mov_b32 %0, %4\n\tv_mov_b32 %1, %5\n\tv_mov_b32 %2, %6\n\tv_mov_b32 %3, %7"
                         : "=&v"(seen[0]), "=&v"(seen[1]), "=&v"(seen[2]), "=&v"(seen[3])
                         : "v"(ra[0]), "v"(ra[1]), "v"(ra[2]), "v"(ra[3]), "n"(W) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(rbv), "+v"(ra) : : "memory");
        const uint4 want = pa[lane];
        const bool ok = seen[0] == want.x && seen[1] == want.y && seen[2] == want.z && seen[3] == want.w;
        if (__any(!ok)) ++nbad;
        if (rbv + hotv == 0x12345u) ++nbad;                           // (keeps the loads alive)
        __syncthreads();
    }
    if (lane == 0) atomicAdd(bad, nbad);
}

template <int AK, int BK, int W = 1>
static void run(const char *name, const uint4 *cold, size_t n, const unsigned *hot, unsigned *bad) {
    hipMemset(bad, 0, 4);
    const int blocks = 2048, trials = 400;
    hipLaunchKernelGGL((probe<AK, BK, W>), dim3(blocks), dim3(64), 0, 0, cold, n, hot, trials, bad, 12345u + AK * 17 + BK);
    hipDeviceSynchronize();
    unsigned h = 0;
    hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
    printf("%-52s A not landed after vmcnt(%d): %8u of %d trials\n", name, W, h, blocks * trials);
}

int main() {
    const size_t n = (size_t)3 << 26;   // 3 GiB of uint4: far beyond the 256 MiB MALL
    uint4 *cold;
    unsigned *hot, *bad;
    if (hipMalloc(&cold, n * 16) != hipSuccess) return 1;
    hipMalloc(&hot, 256);
    hipMalloc(&bad, 4);
    // contents: words that never equal the sentinel
    std::vector<uint4> h((size_t)1 << 20);
    for (size_t i = 0; i < h.size(); ++i) h[i] = make_uint4((unsigned)i * 4 + 1, (unsigned)i * 4 + 2, (unsigned)i * 4 + 3, (unsigned)i * 4 + 4);
    for (size_t o = 0; o < n; o += h.size()) hipMemcpy(cold + o, h.data(), h.size() * 16, hipMemcpyHostToDevice);
    hipMemset(hot, 0, 256);
    run<0, 0>("A = LDS-DMA (cold), B = global_load_dword (hot)", cold, n, hot, bad);
    run<0, 1>("A = LDS-DMA (cold), B = global_load_dword sc1 (hot)", cold, n, hot, bad);
    run<0, 2>("A = LDS-DMA (cold), B = global_load_dword sc0 sc1", cold, n, hot, bad);
    run<0, 3>("A = LDS-DMA (cold), B = global_load_dword nt", cold, n, hot, bad);
    run<1, 0>("A = load to VGPRs (cold), B = global_load_dword", cold, n, hot, bad);
    run<1, 1>("A = load to VGPRs (cold), B = global_load_dword sc1", cold, n, hot, bad);
    run<1, 2>("A = load to VGPRs (cold), B = global_load_dword sc0 sc1", cold, n, hot, bad);
    run<1, 3>("A = load to VGPRs (cold), B = global_load_dword nt", cold, n, hot, bad);
    // positive controls: no wait at all for A (vmcnt(2)) -- the probe must see the sentinel
    run<0, 1, 2>("control: A = LDS-DMA, B = sc1, NO wait for A", cold, n, hot, bad);
    run<1, 1, 2>("control: A = load to VGPRs, B = sc1, NO wait for A", cold, n, hot, bad);
    return 0;
}

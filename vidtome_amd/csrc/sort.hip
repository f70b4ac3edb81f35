// vtm_sort_desc: stable descending argsort of the row maxima.
// Reference: `edge_idx = node_max.argsort(dim=-1, descending=True)` (vidtome/merge.py:98,113 / 402,417).
// Canonical order (see include/vidtome_hip.h): descending value, NaN first, ties by ascending index.
// The sort key is the high word of vtm_match's packed result (already an order-preserving integer
// image of node_max), inverted so that an ASCENDING stable LSD radix sort yields the DESCENDING order.
//
// v1 design: one 1024-thread workgroup per row of keys, 8 passes of 4-bit digits.  Each thread owns a
// contiguous chunk of the row, so "thread order" == "index order" and the per-(digit, thread) counters
// (16 x 1024 ints = 64 KiB of LDS) give a stable scatter without atomics.  The rows are <= ~110k keys
// (<= 0.9 MB of key+index state), i.e. L2-resident; the kernel is latency-, not bandwidth-bound.
#include "common.h"

namespace {

constexpr int T = 1024;      // threads per workgroup
constexpr int RADIX = 16;    // 4-bit digits
constexpr int PASSES = 8;

__device__ __forceinline__ int wave_inclusive_scan(int v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(v, off, 64);
        if (lane >= off) v += t;
    }
    return v;
}

__global__ __launch_bounds__(T) void sort_desc_kernel(const uint64_t *__restrict__ best, int64_t n,
                                                      int32_t *__restrict__ perm,
                                                      uint32_t *__restrict__ ws) {
    __shared__ int hist[RADIX * T];
    __shared__ int tot[RADIX];
    __shared__ int base[RADIX];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t row = blockIdx.x;
    const uint64_t *kin = best + row * n;
    uint32_t *k0 = ws + row * 4 * n, *k1 = k0 + n;
    int32_t *p0 = reinterpret_cast<int32_t *>(k1 + n), *p1 = p0 + n;
    const int64_t chunk = (n + T - 1) / T;
    const int64_t lo = (int64_t)tid * chunk;
    const int64_t hi = lo + chunk < n ? lo + chunk : n;

    for (int pass = 0; pass < PASSES; ++pass) {
        const int sh = pass * 4;
        const uint32_t *ksrc = (pass & 1) ? k1 : k0;
        const int32_t *psrc = (pass & 1) ? p1 : p0;
        uint32_t *kdst = (pass & 1) ? k0 : k1;
        int32_t *pdst = (pass & 1) ? p0 : p1;
        if (pass == PASSES - 1) pdst = perm + row * n;  // last scatter writes the result directly

#pragma unroll
        for (int d = 0; d < RADIX; ++d) hist[d * T + tid] = 0;
        // every thread only touches its own column hist[.][tid]: no sync needed before counting
        for (int64_t i = lo; i < hi; ++i) {
            const uint32_t key = pass == 0 ? ~(uint32_t)(kin[i] >> 32) : ksrc[i];
            hist[((key >> sh) & 15u) * T + tid] += 1;
        }
        __syncthreads();
        {  // wave `wave` scans digit `wave`'s 1024 counters (index order == thread order)
            int carry = 0;
#pragma unroll 4
            for (int j = 0; j < T / 64; ++j) {
                const int idx = wave * T + j * 64 + lane;
                const int v = hist[idx];
                const int inc = wave_inclusive_scan(v, lane);
                hist[idx] = carry + inc - v;
                carry += __shfl(inc, 63, 64);
            }
            if (lane == 0) tot[wave] = carry;
        }
        __syncthreads();
        if (tid == 0) {
            int acc = 0;
            for (int d = 0; d < RADIX; ++d) {
                base[d] = acc;
                acc += tot[d];
            }
        }
        __syncthreads();
        for (int64_t i = lo; i < hi; ++i) {
            uint32_t key;
            int32_t id;
            if (pass == 0) {
                key = ~(uint32_t)(kin[i] >> 32);
                id = (int32_t)i;
            } else {
                key = ksrc[i];
                id = psrc[i];
            }
            const int d = (key >> sh) & 15u;
            const int pos = base[d] + hist[d * T + tid];
            hist[d * T + tid] += 1;
            if (pass != PASSES - 1) kdst[pos] = key;
            pdst[pos] = id;
        }
        __syncthreads();  // workgroup-scope: the next pass reads what this workgroup just wrote
    }
}

}  // namespace

VTM_EXPORT size_t vtm_sort_ws_bytes(int64_t rows, int64_t n) {
    if (rows <= 0 || n <= 0) return 0;
    return (size_t)rows * (size_t)n * 16u;
}

VTM_EXPORT int vtm_sort_desc(const uint64_t *best, int64_t rows, int64_t n, int32_t *perm, void *ws,
                             size_t ws_bytes, vtm_stream_t stream) {
    VTM_REQUIRE(best && perm && ws, "vtm_sort_desc: null pointer");
    VTM_REQUIRE(rows > 0 && n >= 0 && n < (1ll << 31), "vtm_sort_desc: bad sizes");
    if (ws_bytes < vtm_sort_ws_bytes(rows, n))
        return vtm::fail(VTM_EWORKSPACE, "vtm_sort_desc: workspace %zu < %zu bytes", ws_bytes,
                         vtm_sort_ws_bytes(rows, n));
    if (n == 0) return VTM_OK;
    hipLaunchKernelGGL(sort_desc_kernel, dim3((unsigned)rows), dim3(T), 0, vtm::as_stream(stream), best, n,
                       perm, reinterpret_cast<uint32_t *>(ws));
    return vtm::launch_status("vtm_sort_desc");
}

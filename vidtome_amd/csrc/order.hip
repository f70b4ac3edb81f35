// Position-major order of a matcher call's row lists (round 5): vtm_position_order.
//
// The filter prunes 32 x 32 score blocks whose partial sums cannot reach the rows' running maxima, and whole 256 x 128
// workgroup tiles with the scout + range plan -- which pays when the pairs that DO score high sit together.  In a video a
// token's good matches are the tokens at (and around) its own spatial position in other frames.  At the first local
// level both row lists are in (frame, position) order and 99 % of the blocks die; at level 2 and the global level the
// reference's sequence order is [unmerged tokens by descending score | dst tokens], a 32 x 32 block then holds a
// same-position pair with probability ~0.25-0.4, and a quarter of the blocks stay alive (profiles/r05_q_position_order.txt: the
// same rows in random order cost 1.70 ms per top global call, in position order 1.33, with the range plan 1.09).
//
// The matcher does not care in which order it meets the rows: its result is, per src row, the maximum of the canonical
// scores and the LOWEST dst index attaining it.  So the host hands it both lists sorted by token position (this file) and
// the inverse maps; refine_kernel / exact_rows_kernel report and tie-break in the ORIGINAL indexing
// (vtm_match_filtered_ordered), everything behind the matcher -- argsort, index planning, the reference's token order --
// is untouched.
//
// Counting sort by position, three small launches over both operands of all samples at once:
//   count_kernel    one thread per list entry: atomic increment of its position's counter
//   scatter_kernel  a workgroup serves 1024 entries of ONE (sample, operand): it scans that operand's N + 1 counters into
//                   offsets in LDS (bucket N = rows without a position: behind the last position; every workgroup of the
//                   operand repeats the scan -- 16 KB from L2 at N = 4096 -- rather than wait for a launch that does it
//                   once), then every entry takes slot = offset + arrival rank (a second counter array) in a staging copy.
//                   The operand's first workgroup also publishes the offsets, and for the dst operand the seed table of the
//                   call (position -> first dst entry holding it, -1 = none)
//   settle_kernel   one thread per staged entry: its rank inside its position group by ORIGINAL index (groups are ~F
//                   entries) -> the final, deterministic lists; the group's first thread re-zeroes the two counters, so the
//                   counter block is all zero again when the call ends (the caller provides it zeroed ONCE).
// N + 2 offsets must fit the 64 KB of static LDS: N <= VTM_POSITION_ORDER_MAX_N = 16 360 tokens per frame (a 1024 x 1024
// image has 16 384: the host then leaves the level in the reference's order).
#include "common.h"

namespace {

constexpr int OT = 256;          // threads per workgroup of the per-entry kernels
constexpr int ST = 1024;         // threads of a scan workgroup
constexpr int MAX_N = VTM_POSITION_ORDER_MAX_N;
static_assert((MAX_N + 2 + ST / 64) * 4 <= 64 * 1024, "offsets live in static LDS");
constexpr int SETTLE_MAX = 512;  // larger position groups keep their arrival order (only bucket N of a call whose rows
                                 // mostly lack positions gets there; any order is a correct one)

struct Lists {
    const int32_t *rows[2];      // a_rows (B, n[0]), b_rows (B, n[1])
    int64_t n[2];
    int32_t *sorted[2];          // outputs
    int32_t *order[2];
};

__device__ __forceinline__ int64_t dcdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

__device__ __forceinline__ int64_t position_of(int32_t r, int64_t b, int64_t L, int64_t N, const int32_t *pos1, int64_t P0,
                                               int64_t P1) {
    int64_t pos = -1;
    if (r >= 0 && r < L) pos = r % N;                                     // a token of the joined chunk: (frame, position) rows
    else if (pos1 != nullptr && r - P0 >= 0 && r - P0 < P1) pos = pos1[b * P1 + (r - P0)];
    return (pos >= 0 && pos < N) ? pos : N;
}

// entry g of the concatenated lists -> (sample, operand, index)
__device__ __forceinline__ bool entry_of(int64_t g, const Lists &ls, int64_t B, int64_t &b, int &op, int64_t &i) {
    const int64_t per = ls.n[0] + ls.n[1];
    if (g >= B * per) return false;
    b = g / per;
    i = g % per;
    op = i >= ls.n[0];
    if (op) i -= ls.n[0];
    return true;
}

__global__ __launch_bounds__(OT) void count_kernel(Lists ls, int64_t B, int64_t L, int64_t N, const int32_t *__restrict__ pos1,
                                                   int64_t P0, int64_t P1, int32_t *__restrict__ cnt) {
    int64_t b, i;
    int op;
    if (!entry_of((int64_t)blockIdx.x * OT + threadIdx.x, ls, B, b, op, i)) return;
    const int64_t p = position_of(ls.rows[op][b * ls.n[op] + i], b, L, N, pos1, P0, P1);
    atomicAdd(&cnt[(b * 2 + op) * (N + 1) + p], 1);
}

__global__ __launch_bounds__(ST) void scatter_kernel(Lists ls, int64_t B, int64_t L, int64_t N, const int32_t *__restrict__ pos1,
                                                     int64_t P0, int64_t P1, const int32_t *__restrict__ cnt,
                                                     int32_t *__restrict__ off, int32_t *__restrict__ table,
                                                     int32_t *__restrict__ fill, int32_t *__restrict__ stage_rows,
                                                     int32_t *__restrict__ stage_order, int64_t n_out) {
    __shared__ int s_off[MAX_N + 2];
    __shared__ int wave_tot[ST / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // which (sample, operand), which 1024 entries of it
    const int64_t wg0 = dcdiv(ls.n[0], ST), wg1 = dcdiv(ls.n[1], ST);
    const int64_t b = blockIdx.x / (wg0 + wg1);
    int64_t w = blockIdx.x % (wg0 + wg1);
    const int op = w >= wg0;
    if (op) w -= wg0;
    const int64_t bo = b * 2 + op;
    // ---- offsets: thread t sums the counters [t per, (t + 1) per), the workgroup scans the 1024 sums
    const int per = (int)dcdiv(N + 1, ST);
    const int32_t *c = cnt + bo * (N + 1);
    const int64_t p0 = (int64_t)tid * per;
    int sum = 0;
    for (int k = 0; k < per; ++k) sum += (p0 + k < N + 1) ? c[p0 + k] : 0;
    int incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(incl, d, 64);
        if (lane >= d) incl += t;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    int run = incl - sum;
    for (int ww = 0; ww < wave; ++ww) run += wave_tot[ww];
    int32_t *tb = (w == 0 && op == 1 && table != nullptr) ? table + b * N : nullptr;      // (shared order: replicated below)
    int32_t *og = w == 0 ? off + bo * (N + 2) : nullptr;
    for (int k = 0; k < per; ++k) {
        const int64_t p = p0 + k;
        if (p < N + 1) {
            const int v = c[p];
            s_off[p] = run;
            if (og) og[p] = run;
            if (tb && p < N)
                for (int64_t bb = 0; bb < n_out; ++bb) tb[bb * N + p] = v > 0 ? run : -1;
            run += v;
        }
    }
    if (og && p0 <= N && N < p0 + per) og[N + 1] = run;          // (the thread that served bucket N holds the total)
    __syncthreads();
    // ---- this workgroup's entries
    const int64_t i = w * ST + tid;
    if (i >= ls.n[op]) return;
    const int32_t r = ls.rows[op][b * ls.n[op] + i];
    const int64_t p = position_of(r, b, L, N, pos1, P0, P1);
    const int64_t slot = s_off[p] + atomicAdd(&fill[bo * (N + 1) + p], 1);
    if (slot >= ls.n[op]) return;        // (only a counter block that was NOT zero on entry gets here: stay inside the lists)
    const int64_t at = b * (ls.n[0] + ls.n[1]) + (op ? ls.n[0] : 0) + slot;
    stage_rows[at] = r;
    stage_order[at] = (int32_t)i;
}

__global__ __launch_bounds__(OT) void settle_kernel(Lists ls, int64_t B, int64_t L, int64_t N, const int32_t *__restrict__ pos1,
                                                    int64_t P0, int64_t P1, const int32_t *__restrict__ off,
                                                    int32_t *__restrict__ cnt, int32_t *__restrict__ fill,
                                                    const int32_t *__restrict__ stage_rows,
                                                    const int32_t *__restrict__ stage_order, int64_t n_out) {
    int64_t b, slot;
    int op;
    if (!entry_of((int64_t)blockIdx.x * OT + threadIdx.x, ls, B, b, op, slot)) return;
    const int64_t base = b * (ls.n[0] + ls.n[1]) + (op ? ls.n[0] : 0);
    const int32_t r = stage_rows[base + slot];
    const int32_t mine = stage_order[base + slot];
    const int64_t p = position_of(r, b, L, N, pos1, P0, P1);
    const int32_t *o = off + (b * 2 + op) * (N + 2);
    const int64_t g0 = o[p], g1 = o[p + 1];
    int64_t rank = slot - g0;
    if (g1 - g0 <= SETTLE_MAX) {
        rank = 0;
        for (int64_t q = g0; q < g1; ++q) rank += stage_order[base + q] < mine;
    }
    if (g0 + rank >= ls.n[op]) return;
    ls.sorted[op][b * ls.n[op] + g0 + rank] = r;
    ls.order[op][b * ls.n[op] + g0 + rank] = mine;
    for (int64_t bb = 1; bb < n_out; ++bb) {          // shared order (b == 0): every sample's list in sample 0's order
        ls.sorted[op][bb * ls.n[op] + g0 + rank] = ls.rows[op][bb * ls.n[op] + mine];
        ls.order[op][bb * ls.n[op] + g0 + rank] = mine;
    }
    if (slot == g0) {                                   // nobody reads the counters any more: leave them zeroed
        cnt[(b * 2 + op) * (N + 1) + p] = 0;
        fill[(b * 2 + op) * (N + 1) + p] = 0;
    }
}

}  // namespace

VTM_EXPORT size_t vtm_position_order_counter_ints(int64_t B, int64_t N) {
    if (B <= 0 || N <= 0) return 0;
    return (size_t)(2 * B * 2 * (N + 1));
}

VTM_EXPORT size_t vtm_position_order_ws_bytes(int64_t B, int64_t Ns, int64_t Nd, int64_t N) {
    if (B <= 0 || N <= 0 || Ns <= 0 || Nd <= 0) return 0;
    return (size_t)(B * 2 * (N + 2) + 2 * B * (Ns + Nd)) * 4;
}

VTM_EXPORT int vtm_position_order(const int32_t *a_rows, int64_t Ns, const int32_t *b_rows, int64_t Nd, int64_t B, int64_t L,
                                  int64_t N, const int32_t *pos1, int64_t P0, int64_t P1, int32_t *counters, void *ws,
                                  size_t ws_bytes, int32_t *a_sorted, int32_t *a_order, int32_t *b_sorted, int32_t *b_order,
                                  int32_t *table, int shared_order, vtm_stream_t stream) {
    VTM_REQUIRE(a_rows && b_rows && counters && ws && a_sorted && a_order && b_sorted && b_order, "vtm_position_order: null pointer");
    VTM_REQUIRE(B > 0 && Ns > 0 && Nd > 0 && N > 0 && L >= 0 && P0 >= 0 && P1 >= 0, "vtm_position_order: bad sizes");
    VTM_REQUIRE(B * (Ns + Nd) < (1ll << 31), "vtm_position_order: too many rows");
    VTM_REQUIRE(N <= MAX_N, "vtm_position_order: N=%lld > %d tokens per frame", (long long)N, MAX_N);
    if (ws_bytes < vtm_position_order_ws_bytes(B, Ns, Nd, N))
        return vtm::fail(VTM_EWORKSPACE, "vtm_position_order: workspace %zu < %zu bytes", ws_bytes,
                         vtm_position_order_ws_bytes(B, Ns, Nd, N));
    hipStream_t s = vtm::as_stream(stream);
    int32_t *cnt = counters, *fill = counters + B * 2 * (N + 1);
    int32_t *off = static_cast<int32_t *>(ws);
    int32_t *stage_rows = off + B * 2 * (N + 2), *stage_order = stage_rows + B * (Ns + Nd);
    const Lists ls{{a_rows, b_rows}, {Ns, Nd}, {a_sorted, b_sorted}, {a_order, b_order}};
    // shared order (aligned matching: one result row per src index over all samples): sample 0's positions decide, every
    // sample's lists are written in that order -- whatever the other samples' positions are, entry i is the same original index
    // in all of them
    const int64_t Bs = shared_order ? 1 : B, n_out = shared_order ? B : 1;
    const dim3 grid((unsigned)vtm::cdiv(Bs * (Ns + Nd), OT)), block(OT);
    hipLaunchKernelGGL(count_kernel, grid, block, 0, s, ls, Bs, L, N, pos1, P0, P1, cnt);
    const dim3 sgrid((unsigned)(Bs * (vtm::cdiv(Ns, ST) + vtm::cdiv(Nd, ST))));
    hipLaunchKernelGGL(scatter_kernel, sgrid, dim3(ST), 0, s, ls, Bs, L, N, pos1, P0, P1, (const int32_t *)cnt, off, table, fill,
                       stage_rows, stage_order, n_out);
    hipLaunchKernelGGL(settle_kernel, grid, block, 0, s, ls, Bs, L, N, pos1, P0, P1, (const int32_t *)off, cnt, fill,
                       (const int32_t *)stage_rows, (const int32_t *)stage_order, n_out);
    return vtm::launch_status("vtm_position_order");
}

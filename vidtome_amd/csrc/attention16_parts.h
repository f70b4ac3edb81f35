// Internal pieces shared by the two wide-tile d = 40 translation units (attention16.hip, attention16g.hip): the partial record of a
// key-split workgroup, the device-side launch plan of query-bounded launches and the kernel that combines the records.
#pragma once
#include <cstdlib>
#include "attention_common.h"

namespace {

using namespace vtm_att;

// per-(query sub-tile, value group) record a key-split workgroup leaves for attention16_combine_kernel: the PV16 record of
// attention.hip (24 accumulators at d = 40, one running max per 16-query half, one unused denominator slot)
template <int D> constexpr int rec16() { return (D + 16) / 16 * 8 + 2 + 1; }

// ---- device-side launch plan for QUERY-BOUNDED launches (vtm_attention_kv_bounded: compacted live queries) ----
// How many query blocks are live is a device value (q_count), so the host cannot cut the launch into whole rounds plus a
// key-split tail the way plan_tail16 does for a known length; rounds 4-5 split EVERY item in two instead (finer rounds).
// Measured in round 6 (profiles/r06_d_attention16_ab.txt): 912 live items on 256 slots are 3.56 rounds of work and took the
// time of 4 (4.81 ms against 4.31).  One thread now plans on the device, in front of the launch, from the counts -- a
// GEOMETRIC tail: workgroups are dispatched in index order as slots free up, so pieces that shrink towards the end of the
// grid pack like longest-first list scheduling and the idle tail is one SMALLEST piece long:
//   L live items (the longest sample's blocks x heads x samples), S slots;
//   tier 0: the items of the whole rounds, L - L % S of them, one workgroup each (a last round >= 0.8 full, or the only round of a
//   launch that fills more than half of the chip, counts as whole);
//   then, while items remain: the next tier takes S / n of them (or what is left), n = the smallest power of two >= 2 with
//   S / n <= remaining (at most MAX_SPLIT), each split n ways along the key axis -- S pieces, one short round.
// 912 items on 256 slots: 768 whole, 128 in halves, 16 in sixteenths = 3 + 0.5 + 1/16 rounds -- the work there is.  The
// launch itself is sized for the host-known upper bound; workgroups the plan has no role for leave at once.
constexpr int PLAN_TIERS = 8;
constexpr int PLAN_MAX_SPLIT = 16;
struct DevTier {
    int wg0, item0, items, nsplit, rec0;   // first workgroup, first item, items, pieces per item, first partial record
};
struct DevPlan {
    int nqb, ntiers, split_items, pad;     // live query blocks per (sample, head); tiers in use; items behind tier 0
    DevTier tier[PLAN_TIERS];
};
// upper bounds of a plan on S slots: workgroups behind the whole items / partial records, items that are split
constexpr int64_t plan_tail_wgs(int slots) { return (int64_t)(PLAN_TIERS - 1) * slots; }
constexpr int64_t plan_split_items(int slots) { return slots; }

__global__ void attention16_plan_kernel(const int32_t *__restrict__ q_count, int B, int H, int QB, int slots, int ntiles,
                                        DevPlan *__restrict__ plan) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int nqb = 1;
    for (int b = 0; b < B; ++b) {
        const int n = (q_count[b] + QB - 1) / QB;
        nqb = n > nqb ? n : nqb;
    }
    const int L = nqb * H * B, S = slots;
    int max_ns = ntiles / 8;                     // a piece keeps >= 8 key tiles
    max_ns = max_ns > PLAN_MAX_SPLIT ? PLAN_MAX_SPLIT : max_ns < 1 ? 1 : max_ns;
    DevPlan p;
    p.nqb = nqb;
    p.pad = 0;
    int nt = 0, wg = 0, item = 0, rec = 0;
    // what runs whole: the full rounds; ALSO a last round that is at least 0.8 full, or the one round of a launch that fills
    // more than half of the chip (splitting those moves more partial records than the idle slots are worth)
    int whole = L - L % S;
    if (max_ns < 2 || (L % S) * 5 >= S * 4 || (L < S && L * 2 > S)) whole = L;
    p.tier[nt++] = DevTier{0, 0, whole, 1, 0};
    wg = item = whole;
    while (item < L && nt < PLAN_TIERS) {
        const int rem = L - item;
        int n = 2;
        while (n < max_ns && S / n > rem) n *= 2;
        if (n > max_ns) n = max_ns;
        int take = S / n < rem ? S / n : rem;
        if (nt == PLAN_TIERS - 1) take = rem;    // (never with S = 256: 2, 4, 8, 16, 16 ...: the last tier takes what is left)
        p.tier[nt++] = DevTier{wg, item, take, n, rec};
        wg += take * n;
        rec += take * n;
        item += take;
    }
    p.ntiers = nt;
    p.split_items = L - whole;
    for (int i = nt; i < PLAN_TIERS; ++i) p.tier[i] = DevTier{wg, item, 0, 1, rec};
    *plan = p;
}

template <typename T, int D, int NQ, int NG, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void attention16_combine_kernel(
    const float *__restrict__ partial, T *__restrict__ out, int64_t ldo, int64_t H, int64_t M, int64_t Mp, int64_t nqb,
    int64_t id0, int nsplit, int xcd_groups, const int32_t *__restrict__ q_count, int64_t src_batch,
    const DevPlan *__restrict__ dev_plan) {
    constexpr int NT = WAVES * 64, QB = WAVES * QW * NQ, NV = NQ * NG;
    constexpr int NA = (D + 16) / 16 * 8, REC = rec16<D>();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int v = blockIdx.y, sub = v / NG, g = v % NG;
    int64_t rec0 = (int64_t)blockIdx.x * nsplit;   // first partial record of this item
    int64_t pos = id0 + blockIdx.x;
    if (dev_plan != nullptr) {        // (the launch is sized for the most items a plan can split)
        if ((int)blockIdx.x >= dev_plan->split_items) return;
        nqb = dev_plan->nqb;
        xcd_groups = nqb >= 32 ? xcd_groups : 0;
        pos = dev_plan->tier[0].items + blockIdx.x;
        int ti = 1;
        while (ti + 1 < dev_plan->ntiers && pos >= dev_plan->tier[ti + 1].item0) ++ti;
        const DevTier tr = dev_plan->tier[ti];
        nsplit = tr.nsplit;
        rec0 = tr.rec0 + (pos - tr.item0) * tr.nsplit;
    }
    const int64_t lin = item_of(pos, nqb, xcd_groups);
    const int64_t b = lin / (nqb * H), h = (lin / nqb) % H;
    const int64_t q0 = (lin % nqb) * QB + (wave * NQ + sub) * QW;
    if (q_count != nullptr && (lin % nqb) * QB >= (int64_t)q_count[b]) return;   // its partial records were never written
    float acc[NA], m[2] = {-INFINITY, -INFINITY};
#pragma unroll
    for (int r = 0; r < NA; ++r) acc[r] = 0.0f;
    for (int sp = 0; sp < nsplit; ++sp) {
        const float *pp = partial + ((rec0 + sp) * NV + v) * REC * NT + tid;
        float fa[2], fb[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float ms = pp[(NA + j) * NT];
            const float mn = fmaxf(m[j], ms);
            fa[j] = mn == -INFINITY ? 0.0f : __builtin_amdgcn_exp2f(m[j] - mn);   // exp2(-inf) = 0; (-inf) - (-inf) is not
            fb[j] = mn == -INFINITY ? 0.0f : __builtin_amdgcn_exp2f(ms - mn);
            m[j] = mn;
        }
#pragma unroll
        for (int r = 0; r < NA; ++r) {
            const int j = (r >> 2) & 1;   // accumulator (dv, qh, e) is register (dv * 2 + qh) * 4 + e
            acc[r] = acc[r] * fa[j] + pp[r * NT] * fb[j];
        }
    }
    f32x4 o[(D + 16) / 16][2];
#pragma unroll
    for (int r = 0; r < NA; ++r) o[r >> 3][(r >> 2) & 1][r & 3] = acc[r];
    write_output16<T, D>(o, out, ldo, b + g * src_batch, h, q0, M, Mp, lane);
}

// workspace of a device-planned launch: the plan (one cache line) + the records of the largest tail a plan can have
constexpr size_t DEVPLAN_HEADER = 256;
static_assert(sizeof(DevPlan) <= 256, "the plan lives in the workspace header");
inline size_t devplan_ws_bytes(int slots, size_t rec_bytes) { return DEVPLAN_HEADER + (size_t)plan_tail_wgs(slots) * rec_bytes; }
inline bool devplan_enabled() {
    static const bool on = [] {
        const char *e = getenv("VTM_ATT_DEVPLAN");        // A/B hook, read once per process and translation unit
        return e == nullptr || atoi(e) != 0;
    }();
    return on;
}

}  // namespace

// gx_kernels.hip -- robot-independent kernels (layout rejection sampler, probes) and the dispatch of
// the per-robot launchers (gx_robot_kernels.inl, instantiated in gx_kernels_<robot>.hip).
//
// Reference lines: /root/reference/safe_rl_envs/safe_rl_envs/envs/engine.py.
#include "gx_kernels.h"
#include "gx_robot.h"
#include "gx_policy.h"
#include <cstdlib>

namespace gx {

// ---------------------------------------------------------------------------
// layout rejection sampler: sample_layout (engine.py:546-572) for candidate j,
// key_j = split(key, M)[j] (:263).  Three phases with global compactions between:
//
//  phase 0 (every candidate, 24 blocks): the goal; reject when no robot position can be
//    3.0 away from it.
//  phase 1 (the rest, 220 more blocks; 244 of 600 so far): walk the 10*(H+2)
//    links of the `rng, rng1 = split(rng)` chain, draw only the goal (its 10 tries
//    are all valid because nothing is placed yet, so the 10th wins) and the 10
//    robot tries.  A layout can only succeed if its final robot position is >= 3.0
//    from the goal (:570-571), and the final position is one of the 10 tries, so a
//    candidate none of whose tries is that far is rejected here -- exactly, not
//    heuristically (~75 % of all candidates for the default arena).
//  phase 2 (survivors only): place the hazards (chain restarted from the saved
//    key after the goal), validate the saved robot tries, decide success.
// ---------------------------------------------------------------------------
constexpr int kSampleBlock = 256;
constexpr int kCompactPerThread = 16;
constexpr int kCompactTile = kSampleBlock * kCompactPerThread; // candidates per block of scan_compact_kernel (4096)
constexpr int kSurvWords = 32; // j, rng(2), goal(2), robot tries(20), pad

// squared distance with the operation order of sqrt(sum(square(a - b))) (engine.py:553);
// `sqrtf(d2) < thr` is evaluated as `d2 < thr_sq` where thr_sq is the exact cutoff
// min{x : fl(sqrt(x)) >= thr} computed on the host (SampleParams), so no sqrt is needed
GX_D float dsq(float ax, float ay, float bx, float by)
{
    const float dx = ax - bx, dy = ay - by;
    return dx * dx + dy * dy;
}

GX_D void draw_xy(uint32_t g0, uint32_t g1, float lox, float hix, float loy, float hiy, float& x, float& y)
{
    uint32_t u0, u1, v0, v1;
    split2(g0, g1, u0, u1, v0, v1); // draw_placement :618
    x = uniform_f(u0, u1, lox, hix);
    y = uniform_f(v0, v1, loy, hiy);
}

// workgroup-aggregated slot allocation in a compacted list: ONE returning atomic per 256-thread
// workgroup (a single counter word sustains only ~88 returning atomics/us chip-wide, which at one
// atomic per wave -- 15,625 of them -- would cost more than phase 0 itself).  Must be reached by
// every thread of the workgroup.
template <int BLOCK = kSampleBlock>
GX_D int alloc_slot(bool want, int* __restrict__ counter)
{
    __shared__ int wcnt[BLOCK / 64];
    __shared__ int bbase;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const unsigned long long m = __ballot(want);
    if (lane == 0) wcnt[w] = __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        int tot = 0;
#pragma unroll
        for (int k = 0; k < BLOCK / 64; ++k) tot += wcnt[k];
        bbase = tot ? atomicAdd(counter, tot) : 0;
    }
    __syncthreads();
    int off = bbase;
    for (int k = 0; k < w; ++k) off += wcnt[k];
    const int slot = want ? off + __popcll(m & ((1ull << lane) - 1ull)) : -1;
    __syncthreads();
    return slot;
}

// phase 0 (every candidate, 24 blocks): the goal.  If even the farthest corner of the robot's
// placement rectangle is closer than 3.0 (with a safety margin for rounding) the candidate
// cannot succeed, whatever the robot draws are (~22 % of the candidates).
__global__ __launch_bounds__(kSampleBlock) void sample_phase0_kernel(SampleParams sp,
                                                                     uint8_t* __restrict__ ok,
                                                                     int* __restrict__ n_surv0,
                                                                     uint32_t* __restrict__ surv0,
                                                                     int* __restrict__ blk_cnt)
{
    const int tid = threadIdx.x;
    const int j = blockIdx.x * kSampleBlock + tid;
    const bool live = j < sp.M;
    uint32_t r0, r1;
    split_at(sp.k0, sp.k1, (uint32_t)sp.Mtot, (uint32_t)(sp.c0 + (live ? j : 0)), r0, r1);
    uint32_t n0, n1, g0 = 0, g1 = 0;
    for (int t = 0; t < 10; ++t) { split2(r0, r1, n0, n1, g0, g1); r0 = n0; r1 = n1; }
    float gx, gy;
    draw_xy(g0, g1, sp.lo_x[0], sp.hi_x[0], sp.lo_y[0], sp.hi_y[0], gx, gy);
    const float fx = fmaxf(fabsf(sp.lo_x[2] - gx), fabsf(sp.hi_x[2] - gx));
    const float fy = fmaxf(fabsf(sp.lo_y[2] - gy), fabsf(sp.hi_y[2] - gy));
    const bool feasible = !((fx * fx + fy * fy) * 1.0001f < sp.min_rg_sq);
    if (live) ok[j] = 0;
    if (live && (j & (kCompactTile - 1)) == 0) blk_cnt[j / kCompactTile] = 0; // per-4096 success counts (phase 2)
    const int slot = alloc_slot(live && feasible, n_surv0);
    if (slot >= 0) {
        uint4* rec = reinterpret_cast<uint4*>(surv0) + (size_t)slot * 2;
        rec[0] = make_uint4((uint32_t)j, r0, r1, f2u(gx));
        rec[1] = make_uint4(f2u(gy), 0u, 0u, 0u);
    }
}

// phase 1 (goal-feasible candidates, 220 blocks): hazard links, the 10 robot tries
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void sample_phase1_kernel(SampleParams sp,
                                                                     const int* __restrict__ n_surv0,
                                                                     const uint32_t* __restrict__ surv0,
                                                                     int* __restrict__ n_surv,
                                                                     uint32_t* __restrict__ surv)
{
    const int tid = threadIdx.x;
    const int S0 = *n_surv0;
    const int nh = 10 * (sp.nobj_total - 2);
    // grid-stride with whole waves active until the last one (ballot-based allocation below)
    for (int base_i = blockIdx.x * BLOCK; base_i < S0; base_i += gridDim.x * BLOCK) {
        const int i = base_i + tid;
        const bool live = i < S0;
        const uint4* rin = reinterpret_cast<const uint4*>(surv0) + (size_t)(live ? i : 0) * 2;
        const uint4 h0 = rin[0];
        const uint4 h1 = rin[1];
        const int j = (int)h0.x;
        uint32_t r0 = h0.y, r1 = h0.z;
        const float gx = u2f(h0.w), gy = u2f(h1.x);
        const uint32_t s0 = r0, s1 = r1; // chain state after the goal
        uint32_t n0, n1, g0 = 0, g1 = 0;
        for (int t = 0; t < nh; ++t) { split2(r0, r1, n0, n1, g0, g1); r0 = n0; r1 = n1; }
        float rx[10], ry[10];
        bool any_far = false;
#pragma unroll
        for (int t = 0; t < 10; ++t) {
            split2(r0, r1, n0, n1, g0, g1); r0 = n0; r1 = n1;
            draw_xy(g0, g1, sp.lo_x[2], sp.hi_x[2], sp.lo_y[2], sp.hi_y[2], rx[t], ry[t]);
            if (!(dsq(rx[t], ry[t], gx, gy) < sp.min_rg_sq)) any_far = true;
        }
        const int slot = alloc_slot<BLOCK>(live && any_far, n_surv);
        if (slot >= 0) {
            uint32_t* rec = surv + (size_t)slot * kSurvWords;
            rec[0] = (uint32_t)j; rec[1] = s0; rec[2] = s1; rec[3] = f2u(gx); rec[4] = f2u(gy);
#pragma unroll
            for (int t = 0; t < 10; ++t) { rec[5 + 2 * t] = f2u(rx[t]); rec[6 + 2 * t] = f2u(ry[t]); }
        }
    }
}

// phase 2 (survivors): hazards (+ pillars), the saved robot tries, success.  One wave per workgroup, lane = survivor.
//
// draw_placement (:579-621) keeps the LAST valid of an object's 10 tries, so a try only has to be drawn when every
// later one conflicts -- per candidate 1.1 (first hazard) to ~2.2 (last hazard) draws of 4 Threefry blocks each.  Evaluated
// lane-by-lane that laziness is lost to divergence: a wave keeps drawing while ANY of its 64 candidates is still
// conflicted, ~9 of the 10 tries for the later hazards (round 2: 36 k instructions per wave, half of them draws
// for a handful of lanes).  Here the draws are work items handed to whichever lanes are free: in each round the c
// still-conflicted candidates ("owners") get m = 64/c tries each (a power of two), evaluated by m consecutive lanes
// from the owner's keys and placed objects in LDS; the highest valid try wins (ds_max on the try index).  A round
// costs one draw; an object needs ~3 rounds (64 -> ~25 -> ~6 -> 0 owners) instead of ~9.
// The `rng, rng1 = split(rng)` chain itself (20 blocks per object) is sequential per candidate and stays per lane.
constexpr int kP2Block = 64;  // survivors per wave (the unit of work)
constexpr int kP2Waves = 4;   // independent waves per workgroup: one per SIMD of the CU it lands on
struct P2Lds {
    uint2 keys[9][kP2Block];   // rng1 of the object's tries 0..8, per owner lane (try 9 is drawn by the owner itself);
                               // a helper overwrites the key it consumed with its valid draw
    int best[kP2Block];        // per owner lane: highest valid try of the round (-1: none)
    int owner[kP2Block];       // compacted list of the owners' lanes
};
// 9.5 KB per wave with the default 10 objects: 16 waves per CU, i.e. the ~3900 waves of the default arena (15.2 per
// CU) are resident at once.  Workgroups of FOUR independent waves, because the waves of a workgroup go to the four
// SIMDs of its CU: VALU issue is arbitrated oldest-first, a SIMD with n waves finishes after T_alone + (n-1) T_issue
// (stamps: 160k + (n-1) 85k ticks), and with one-wave workgroups the hardware put 5 waves on 5 % of the SIMDs and 3 on
// 25 % -- the kernel took the 5-wave time.
// Order this wave's LDS traffic (lanes exchange data through LDS; a wave's LDS operations execute in program order, so
// no instruction is needed -- only the compiler must not move them).  LDS-only fences: a fence over all address spaces
// also makes the compiler wait for the wave's outstanding global loads and stores (s_waitcnt vmcnt(0)).
GX_D void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
}

// kFused (round 5; sparse arenas): the survivors of phase 0 come straight here, phase 1 is not run.  Phase 1 exists to
// reject, before any hazard is placed, the candidates none of whose ten robot tries can be 3.0 away from the goal -- 75 %
// of them in the reference's 4 m arena -- at the price of walking the whole `rng, rng1 = split(rng)` chain (20 Threefry
// blocks per object) once just to reach the robot's keys, and again here for the objects' try keys.  In a sparse arena
// (the synthetic config 5: 6 m, 18 objects) it rejects 10 % and the chain is 320 blocks per walk: the second walk costs
// more than the rejection saves.  This form walks the chain ONCE: objects first (no pruning by the robot's tries, which
// are not known yet: a candidate only dies when an object fails all ten tries), then the robot's ten links and tries,
// validated against everything placed (draw_placement :579-621: the last valid try wins; :570-571).  Same candidates
// succeed with the same rows -- the sampler's result does not depend on the form (launch_sample picks by geometry).
template <bool kFused>
__global__ __launch_bounds__(kP2Block * kP2Waves) void sample_phase2_kernel(SampleParams sp,
                                                                 const int* __restrict__ n_surv,
                                                                 const uint32_t* __restrict__ surv,
                                                                 uint8_t* __restrict__ ok,
                                                                 float2* __restrict__ cand_xy,
                                                                 int* __restrict__ blk_cnt)
{
    extern __shared__ float4 smem4[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int nobj = sp.nobj_total;
    const size_t per_wave = sizeof(P2Lds) + (size_t)(nobj - 1) * kP2Block * sizeof(float2);
    char* mine_lds = reinterpret_cast<char*>(smem4) + per_wave * wv; // nothing is shared between the waves
    P2Lds& S = *reinterpret_cast<P2Lds*>(mine_lds);
    float2* placed = reinterpret_cast<float2*>(mine_lds + sizeof(P2Lds)); // [nobj_total - 1][64]
    const unsigned long long below = (1ull << lane) - 1ull;
    const int NS = *n_surv;
    const int wpb = blockDim.x >> 6; // kP2Waves, or 1 when the objects of four waves do not fit 64 KB of LDS
    const int wave0 = blockIdx.x * wpb + wv, nwaves = gridDim.x * wpb;
    if (sp.dbg && lane == 0 && wave0 * kP2Block < NS) {
        unsigned long long* d = sp.dbg + (size_t)wave0 * 4;
        d[0] = __builtin_amdgcn_s_memtime();
        d[2] = __builtin_amdgcn_s_getreg((31 << 11) | 4);  // HW_REG_HW_ID
        d[3] = __builtin_amdgcn_s_getreg((31 << 11) | 20); // HW_REG_XCC_ID
    }
    for (int base = wave0 * kP2Block; base < NS; base += nwaves * kP2Block) { // wave-uniform
        const int i = base + lane;
        const bool live = i < NS;
        const uint32_t* rec = surv + (size_t)(live ? i : 0) * (kFused ? 8 : kSurvWords); // phase-0 / phase-1 record
        const int j = (int)rec[0];
        uint32_t r0 = rec[1], r1 = rec[2];
        const float gx = u2f(rec[3]), gy = u2f(rec[4]);
        wave_sync(); // the previous batch's reads of `placed` are done
        placed[lane] = make_float2(gx, gy);
        bool alive = live; // placed everything so far
        // The robot's ten tries were drawn in phase 1.  The layout succeeds iff its HIGHEST valid try (valid: clear of
        // the goal and of every hazard / pillar, draw_placement :579-621) is >= 3.0 from the goal (:570-571).  Placed
        // objects only ever invalidate tries, so as soon as no far try is valid any more the candidate cannot succeed
        // and the lane stops owning draws (it keeps helping): exact, and ~10 % of the remaining candidates leave per
        // hazard -- the late hazards, whose draws cost the most rounds, see a fifth of them.
        float cx[10], cy[10];
        unsigned rvalid = 0u, rfar = 0u; // bit t: try t conflicts with nothing placed so far / is far from the goal
        if (kFused) {
#pragma unroll
            for (int t = 0; t < 10; ++t) { cx[t] = 0.f; cy[t] = 0.f; }
            rvalid = rfar = 1u; // nothing is known about the robot yet: no pruning
        } else {
            const float tgr = sp.thr_sq[0][2];
#pragma unroll
            for (int t = 0; t < 10; ++t) {
                cx[t] = u2f(rec[5 + 2 * t]); cy[t] = u2f(rec[6 + 2 * t]);
                const float d2 = dsq(cx[t], cy[t], gx, gy);
                if (!(d2 < tgr)) rvalid |= 1u << t;
                if (!(d2 < sp.min_rg_sq)) rfar |= 1u << t;
            }
        }
        if (!(rvalid & rfar)) alive = false;
        for (int o = 1; o < nobj - 1; ++o) { // hazards, then pillars
            const int tn = o <= sp.H ? 1 : 3;
            const float4 hb = sp.haz_bounds ? sp.haz_bounds[o - 1]
                                            : make_float4(sp.lo_x[tn], sp.hi_x[tn], sp.lo_y[tn], sp.hi_y[tn]);
            // cutoffs against a placed goal / hazard / pillar (in registers: no scalar loads in the loops below)
            const float tg = sp.thr_sq[0][tn], th = sp.thr_sq[1][tn], tp = sp.thr_sq[3][tn];
            uint32_t k9a = 0, k9b = 0;
#pragma unroll
            for (int t = 0; t < 10; ++t) {
                uint32_t n0, n1, g0, g1;
                split2(r0, r1, n0, n1, g0, g1); r0 = n0; r1 = n1;
                if (t < 9) S.keys[t][lane] = make_uint2(g0, g1);
                else { k9a = g0; k9b = g1; }
            }
            bool conflicted = alive;
            float px = -__builtin_inff(), py = -__builtin_inff();
            // round 0: every live candidate draws its own try 9; later rounds: m = 1 << sh tries per owner
            int nt = 9, sh = 0; // nt: the highest try not yet evaluated (the same for every owner)
            bool first = true;
            unsigned long long mask = __ballot(conflicted);
            while (mask != 0ull) {
                const int c = __popcll(mask);
                const int rank = __popcll(mask & below);
                if (!first) {
                    sh = c > 32 ? 0 : (c > 16 ? 1 : (c > 8 ? 2 : 3));
                    if (conflicted) { S.owner[rank] = lane; S.best[lane] = -1; }
                }
                wave_sync();
                const int oi = lane >> sh, ht = nt - (lane & ((1 << sh) - 1));
                const bool work = first ? conflicted : (oi < c && ht >= 0);
                bool mine = false; // round 0: my own try 9 is valid
                float cx = 0.f, cy = 0.f;
                if (work) {
                    const int ow = first ? lane : S.owner[oi];
                    uint2 k = make_uint2(k9a, k9b);
                    if (!first) k = S.keys[ht][ow];
                    draw_xy(k.x, k.y, hb.x, hb.y, hb.z, hb.w, cx, cy);
                    bool flag = true;
                    for (int q0 = 0; q0 < o; q0 += 4) { // placement_is_valid :549-555, four LDS reads in flight
                        float2 pq[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) pq[u] = placed[(q0 + u < o ? q0 + u : o - 1) * kP2Block + ow];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int q = q0 + u;
                            if (q < o && dsq(cx, cy, pq[u].x, pq[u].y) < (q == 0 ? tg : (q <= sp.H ? th : tp))) flag = false;
                        }
                    }
                    if (flag) {
                        if (first) mine = true;
                        else { atomicMax(&S.best[ow], ht); S.keys[ht][ow] = make_uint2(f2u(cx), f2u(cy)); }
                    }
                }
                wave_sync();
                if (conflicted) {
                    if (first) {
                        if (mine) { px = cx; py = cy; conflicted = false; }
                    } else {
                        const int b = S.best[lane];
                        if (b >= 0) {
                            const uint2 w = S.keys[b][lane];
                            px = u2f(w.x); py = u2f(w.y); conflicted = false;
                        } else if (nt - (1 << sh) < 0) { // all ten tries conflict: the candidate fails (:565-566)
                            conflicted = false; alive = false;
                        }
                    }
                }
                nt -= 1 << sh;
                first = false;
                mask = __ballot(conflicted);
                wave_sync(); // owner / best are rewritten by the next round
            }
            placed[o * kP2Block + lane] = make_float2(px, py);
            if (!kFused && alive) { // the robot tries this object rules out
                const float thr = o <= sp.H ? sp.thr_sq[1][2] : sp.thr_sq[3][2];
#pragma unroll
                for (int t = 0; t < 10; ++t)
                    if (dsq(cx[t], cy[t], px, py) < thr) rvalid &= ~(1u << t);
                if (!(rvalid & rfar)) alive = false;
            }
            wave_sync();
        }
        if (kFused && alive) { // the robot's ten links and tries (phase 1's), against everything placed
            rvalid = 0u; rfar = 0u;
            const float tgr = sp.thr_sq[0][2], thz = sp.thr_sq[1][2], tpl = sp.thr_sq[3][2];
#pragma unroll 1
            for (int t = 0; t < 10; ++t) {
                uint32_t n0, n1, g0, g1;
                split2(r0, r1, n0, n1, g0, g1); r0 = n0; r1 = n1;
                float tx, ty;
                draw_xy(g0, g1, sp.lo_x[2], sp.hi_x[2], sp.lo_y[2], sp.hi_y[2], tx, ty);
                const float d2 = dsq(tx, ty, gx, gy);
                bool ok_t = !(d2 < tgr);
                for (int q = 1; q < nobj - 1; ++q) {
                    const float2 pq = placed[q * kP2Block + lane];
                    if (dsq(tx, ty, pq.x, pq.y) < (q <= sp.H ? thz : tpl)) ok_t = false;
                }
                // (cx / cy are indexed by compile-time constants below: select instead of a dynamic index)
#pragma unroll
                for (int u = 0; u < 10; ++u) if (u == t) { cx[u] = tx; cy[u] = ty; }
                if (ok_t) rvalid |= 1u << t;
                if (!(d2 < sp.min_rg_sq)) rfar |= 1u << t;
            }
        }
        bool success = alive;
        float px = -__builtin_inff(), py = -__builtin_inff();
        if (alive) { // robot: the last valid try wins
#pragma unroll
            for (int t = 0; t < 10; ++t)
                if (rvalid & (1u << t)) { px = cx[t]; py = cy[t]; }
            if (!rvalid) success = false;
            if (dsq(px, py, gx, gy) < sp.min_rg_sq) success = false; // :570-571
        }
        if (success) {
            ok[j] = 1;
            atomicAdd(&blk_cnt[j / kCompactTile], 1);
            for (int o = 0; o < nobj - 1; ++o) cand_xy[(size_t)j * nobj + o] = placed[o * kP2Block + lane];
            cand_xy[(size_t)j * nobj + nobj - 1] = make_float2(px, py);
        }
    }
    if (sp.dbg && lane == 0 && wave0 * kP2Block < NS) sp.dbg[(size_t)wave0 * 4 + 1] = __builtin_amdgcn_s_memtime();
}

// idx = where(success > 0)[0]  (engine.py:436): ordered compaction of the valid candidates, ONE launch.
// blk_cnt[b] = number of valid candidates among the 4096 of block b: zeroed by phase 0 (which visits every candidate),
// incremented by phase 2 once per success.  Block b sums the counts of the blocks in front of it itself (245 ints for
// 1e6 candidates), so no block waits for another; each of its 256 threads takes the flags of 16 consecutive candidates
// with ONE 16-byte load (the flags are 0 / 1 bytes), the block scans the 256 per-thread counts, and the threads write
// their cand_of entries; the last block also writes layout_size and re-arms the survivor counters for the next
// launch_sample on this pool.  Replaces memset + count + one-block scan + compact (four launches, ~20 us).
__global__ __launch_bounds__(kSampleBlock) void scan_compact_kernel(int M, const uint8_t* __restrict__ ok,
                                                                    const int* __restrict__ blk_cnt,
                                                                    int* __restrict__ cand_of,
                                                                    int* __restrict__ layout_size,
                                                                    int* __restrict__ n_surv)
{
    __shared__ int part[kSampleBlock / 64];
    __shared__ int wsum[kSampleBlock / 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // this thread's 16 flags (cand_ok is padded to whole tiles and zero beyond M)
    const int j0 = blockIdx.x * kCompactTile + tid * kCompactPerThread;
    const uint4 f = *reinterpret_cast<const uint4*>(ok + j0);
    const int mine = __popc(f.x) + __popc(f.y) + __popc(f.z) + __popc(f.w);
    // valid candidates in front of this block: the per-block counts of the blocks before it
    int acc = 0;
    for (int k = tid; k < (int)blockIdx.x; k += kSampleBlock) acc += blk_cnt[k];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
    // inclusive scan of the per-thread counts across the wave
    int inc = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int up = __shfl_up(inc, d, 64);
        if (lane >= d) inc += up;
    }
    if (lane == 0) part[wv] = acc;
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    int off = 0;
#pragma unroll
    for (int k = 0; k < kSampleBlock / 64; ++k) {
        off += part[k];
        if (k < wv) off += wsum[k];
    }
    off += inc - mine;
    const uint32_t w[4] = {f.x, f.y, f.z, f.w};
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if ((w[q] >> (8 * c)) & 0xffu) cand_of[off++] = j0 + 4 * q + c;
    if (blockIdx.x == gridDim.x - 1 && tid == kSampleBlock - 1) {
        *layout_size = off;
        n_surv[0] = 0; n_surv[1] = 0; // consumed by phases 1 and 2 of THIS launch_sample, re-armed for the next one
    }
}

// ---------------------------------------------------------------------------
// Sharded layout sampling (optional, multi-GPU; gx_sample_shard / gx_reset_from_shards): a rank samples a contiguous
// range of the 1e6 candidates, EXPORTS its valid layouts in candidate order, the ranks all-gather the exports, and every
// rank INSTALLS the concatenation -- shard after shard, i.e. in candidate order -- as its pool: the same rows in the
// same order as the unsharded sampler's compacted list, so layout_size and every randint draw agree.
// ---------------------------------------------------------------------------
// `hdr`: null, or the 4-word header of a piggy-backed export block (gx_sample_shard_ahead): count, the key the shard was
// sampled for, and (shard | n_shards << 16) -- the installer checks all of it
__global__ void pool_export_kernel(int nobj_total, const int* __restrict__ layout_size, const int* __restrict__ cand_of,
                                   const float2* __restrict__ cand_xy, float2* __restrict__ rows, int cap,
                                   int* __restrict__ count, uint32_t* __restrict__ hdr, uint32_t k0, uint32_t k1,
                                   uint32_t tag)
{
    const int L = *layout_size;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        *count = L; // may exceed cap: the installer reports the overflow
        if (hdr) { hdr[1] = k0; hdr[2] = k1; hdr[3] = tag; }
    }
    const int n = (L < cap ? L : cap) * nobj_total;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
        const int row = k / nobj_total, o = k - row * nobj_total;
        rows[k] = cand_xy[(size_t)cand_of[row] * nobj_total + o];
    }
}

// Shard s: rows at rows_all + s * row_stride (float2 units), count at counts[s * cnt_stride].  `check`: the three words
// behind each count must be (k0, k1, s | n_shards << 16) -- a block that was sampled for another key, by another rank
// or for another world size is reported like an overflow (layout_size < 0), never installed.
__global__ void pool_install_kernel(int nobj_total, int n_shards, int cap, const float2* __restrict__ rows_all,
                                    long long row_stride, const int* __restrict__ counts, long long cnt_stride, int check,
                                    uint32_t k0, uint32_t k1, int M, float2* __restrict__ cand_xy,
                                    int* __restrict__ cand_of, int* __restrict__ layout_size)
{
    // rows in front of shard s: the counts of the shards before it (a handful of them)
    const int s = blockIdx.y;
    int off = 0, total = 0, bad = 0;
    for (int q = 0; q < n_shards; ++q) {
        const int* h = counts + (size_t)q * cnt_stride;
        const int c = h[0];
        if (c > cap || c < 0) bad = q + 1;
        else if (check && ((uint32_t)h[1] != k0 || (uint32_t)h[2] != k1 || (uint32_t)h[3] != ((uint32_t)q | ((uint32_t)n_shards << 16))))
            bad = 1000 + q;
        if (q < s) off += c;
        total += c;
    }
    if (!bad && total > M) bad = n_shards + 1;
    if (blockIdx.x == 0 && s == 0 && threadIdx.x == 0) *layout_size = bad ? -bad : total; // < 0: an export overflowed / is foreign
    if (bad) return;
    const int n = counts[(size_t)s * cnt_stride] * nobj_total;
    const float2* src = rows_all + (size_t)s * row_stride;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
        cand_xy[(size_t)off * nobj_total + k] = src[k];
        if (k % nobj_total == 0) cand_of[off + k / nobj_total] = off + k / nobj_total;
    }
}

void launch_pool_export(const Pool& pl, int nobj_total, float2* rows, int cap, int* count, hipStream_t s, uint32_t* hdr,
                        uint32_t k0, uint32_t k1, uint32_t tag)
{
    hipLaunchKernelGGL(pool_export_kernel, dim3(256), dim3(256), 0, s, nobj_total, pl.layout_size, pl.cand_of, pl.cand_xy, rows, cap,
                       count, hdr, k0, k1, tag);
}
void launch_pool_install(const Pool& pl, int nobj_total, int n_shards, int cap, const float2* rows_all, const int* counts,
                         int M, hipStream_t s)
{
    hipLaunchKernelGGL(pool_install_kernel, dim3(64, n_shards), dim3(256), 0, s, nobj_total, n_shards, cap, rows_all,
                       (long long)cap * nobj_total, counts, 1LL, 0, 0u, 0u, M, pl.cand_xy, pl.cand_of, pl.layout_size);
}
void launch_pool_install_blocks(const Pool& pl, int nobj_total, int n_shards, int cap, const float* blocks,
                                long long stride_floats, uint32_t k0, uint32_t k1, int M, hipStream_t s)
{
    // block = [count, k0, k1, tag | rows cap x nobj_total x 2]: rows start 4 floats (2 float2) into the block
    hipLaunchKernelGGL(pool_install_kernel, dim3(64, n_shards), dim3(256), 0, s, nobj_total, n_shards, cap,
                       reinterpret_cast<const float2*>(blocks) + 2, stride_floats / 2, reinterpret_cast<const int*>(blocks),
                       stride_floats, 1, k0, k1, M, pl.cand_xy, pl.cand_of, pl.layout_size);
}

// ---------------------------------------------------------------------------
// probes
// ---------------------------------------------------------------------------
__global__ void math_probe_kernel(int n, const float* x, const float* y, float* s, float* c,
                                  float* at2, float* ex)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float ss, cc;
    sincos_f(x[i], ss, cc);
    s[i] = ss; c[i] = cc;
    at2[i] = atan2_f(y[i], x[i]);
    ex[i] = exp_f(x[i]);
}

__global__ void split_probe_kernel(uint32_t k0, uint32_t k1, int n, uint32_t* out)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    uint32_t a, b;
    split_at(k0, k1, (uint32_t)n, (uint32_t)j, a, b);
    out[2 * j] = a; out[2 * j + 1] = b;
}

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
size_t step_lds_bytes(const Params& p, int block) { return (size_t)block * p.D * sizeof(float); }

int pick_block(const Params& p)
{
    static const int forced = [] { const char* e = getenv("GX_BLOCK"); return e ? atoi(e) : 0; }();
    if (forced == 64 || forced == 256) return forced; // tuning experiments only
    // one wave per workgroup: the obs tile is private to the wave, so the tile barrier costs
    // nothing, and small batches spread over more CUs.  Measured at 2^22 envs: 64-thread workgroups
    // 5.27 TB/s vs 256-thread 5.05 TB/s (same box, back to back).
    (void)p;
    return 64;
}

hipError_t launch_sample(const SampleParams& sp, const Pool& pl, hipStream_t s, hipEvent_t after_phase1)
{
    const int M = sp.M;
    const int grid = (M + kSampleBlock - 1) / kSampleBlock;
    // the survivor counters n_surv[0] (phase 1), n_surv[1] (phase 0) are zero here: zero-filled at allocation and
    // re-armed by the previous launch's scan_compact_kernel (launches on one pool are ordered by the engine's events)
    hipLaunchKernelGGL(sample_phase0_kernel, dim3(grid), dim3(kSampleBlock), 0, s, sp, pl.cand_ok, pl.n_surv + 1,
                       pl.surv0, pl.blk_cnt);
    // GX_SAMPLE_GRID_CAP (tests): a small cap makes phases 1 and 2 take many grid-stride iterations at small M
    int cap = 1 << 30;
    if (const char* ev = getenv("GX_SAMPLE_GRID_CAP")) cap = atoi(ev) > 0 ? atoi(ev) : cap;
    const int grid1 = grid < (cap < 3072 ? cap : 3072) ? grid : (cap < 3072 ? cap : 3072);
    const bool fused = sp.fused != 0; // sparse arena: one walk of the chain (sample_phase2_kernel<true>), no phase 1
    if (!fused)
        hipLaunchKernelGGL(sample_phase1_kernel<kSampleBlock>, dim3(grid1), dim3(kSampleBlock), 0, s, sp, pl.n_surv + 1,
                           pl.surv0, pl.n_surv, pl.surv);
    if (after_phase1) {
        const hipError_t st = hipEventRecord(after_phase1, s);
        if (st != hipSuccess) return st;
    }
    const size_t lds_wave = sizeof(P2Lds) + (size_t)(sp.nobj_total - 1) * kP2Block * sizeof(float2);
    const int wpb = kP2Waves * lds_wave <= 65536 ? kP2Waves : 1;
    const int wgs = (M + kP2Block * wpb - 1) / (kP2Block * wpb);
    const int cap2 = cap < 8192 / wpb ? cap : 8192 / wpb;
    const int grid2 = wgs < cap2 ? wgs : cap2;
    const size_t lds2 = wpb * lds_wave;
    if (fused)
        hipLaunchKernelGGL(sample_phase2_kernel<true>, dim3(grid2), dim3(kP2Block * wpb), lds2, s, sp, pl.n_surv + 1, pl.surv0,
                           pl.cand_ok, pl.cand_xy, pl.blk_cnt);
    else
        hipLaunchKernelGGL(sample_phase2_kernel<false>, dim3(grid2), dim3(kP2Block * wpb), lds2, s, sp, pl.n_surv, pl.surv,
                           pl.cand_ok, pl.cand_xy, pl.blk_cnt);
    hipLaunchKernelGGL(scan_compact_kernel, dim3((M + kCompactTile - 1) / kCompactTile), dim3(kSampleBlock), 0, s, M,
                       pl.cand_ok, pl.blk_cnt, pl.cand_of, pl.layout_size, pl.n_surv);
    return hipSuccess;
}

int sample_compact_tile() { return kCompactTile; }


#define GX_ROBOT_DISPATCH(CALL)                                              \
    do {                                                                     \
        if (p.robot == SwimmerRobot::kId) RobotLaunch<SwimmerRobot>::CALL;   \
        else if (p.robot == AntRobot::kId) RobotLaunch<AntRobot>::CALL;      \
        else if (p.robot == WalkerRobot::kId) RobotLaunch<WalkerRobot>::CALL; \
        else if (p.robot == PointBareRobot::kId) RobotLaunch<PointBareRobot>::CALL; \
        else RobotLaunch<PointRobot>::CALL;                                  \
    } while (0)

void launch_step(const Params& p, const DevBuffers& b, const float* act, float* obs, float* rew,
                 float* cost, float* done, float* qacc, hipStream_t s)
{
    GX_ROBOT_DISPATCH(step(p, b, act, obs, rew, cost, done, qacc, s));
}

void launch_reset_apply(const Params& p, const DevBuffers& b, int nobj_total, uint32_t k10,
                        uint32_t k11, uint32_t k20, uint32_t k21, float* obs, int* host_ls, hipStream_t s)
{
    GX_ROBOT_DISPATCH(reset_apply(p, b, nobj_total, k10, k11, k20, k21, obs, host_ls, s));
}

void launch_reset_done(const Params& p, const DevBuffers& b, int nobj_total, uint32_t k10,
                       uint32_t k11, uint32_t k20, uint32_t k21, const float* obs_in, float* obs_out,
                       hipStream_t s)
{
    GX_ROBOT_DISPATCH(reset_done(p, b, nobj_total, k10, k11, k20, k21, obs_in, obs_out, s));
}

void launch_group_rollout(const Params& p, const RolloutArgs& r, const DevBuffers& b, hipStream_t s)
{
    GX_ROBOT_DISPATCH(group(p, r, b, s));
}

void launch_thread_rollout(const Params& p, const RolloutArgs& r, const DevBuffers& b, hipStream_t s)
{
    GX_ROBOT_DISPATCH(thread_rollout(p, r, b, s));
}

bool split_rollout_supported(const Params& p)
{
    // no pose history in the observation; one physics step per control step (the tape carries qpos, and the pose a step
    // returns is the kinematics of the qpos before its LAST substep).  Every robot: the reset_done observation of the
    // Ant / Walker (which needs a physics step) comes from the pool's fake-step table
    return !p.hist_on && p.physics_steps == 1;
}
int split_tape_width(const Params& p)
{
    int w = 0;
    if (p.robot == SwimmerRobot::kId) w = RobotLaunch<SwimmerRobot>::split_width();
    else if (p.robot == PointBareRobot::kId) w = RobotLaunch<PointBareRobot>::split_width();
    else if (p.robot == PointRobot::kId) w = RobotLaunch<PointRobot>::split_width();
    else if (p.robot == AntRobot::kId) w = RobotLaunch<AntRobot>::split_width();
    else if (p.robot == WalkerRobot::kId) w = RobotLaunch<WalkerRobot>::split_width();
    return w;
}
int split_entry_width(const Params& p)
{
    if (p.robot == SwimmerRobot::kId) return RobotLaunch<SwimmerRobot>::split_entry_width();
    if (p.robot == PointBareRobot::kId) return RobotLaunch<PointBareRobot>::split_entry_width();
    if (p.robot == PointRobot::kId) return RobotLaunch<PointRobot>::split_entry_width();
    if (p.robot == AntRobot::kId) return RobotLaunch<AntRobot>::split_entry_width();
    if (p.robot == WalkerRobot::kId) return RobotLaunch<WalkerRobot>::split_entry_width();
    return 0;
}
hipError_t launch_split_rollout(const Params& p, const RolloutArgs& r, float* tape, float4* obj0, float* entry,
                                const DevBuffers& b, hipStream_t s, hipEvent_t hold, int which, int lanes, int n_shards,
                                long long shard_stride, long long out_stride)
{
    if (p.robot == SwimmerRobot::kId) return RobotLaunch<SwimmerRobot>::split(p, r, tape, obj0, entry, b, s, hold, which, lanes, n_shards, shard_stride, out_stride);
    if (p.robot == PointBareRobot::kId) return RobotLaunch<PointBareRobot>::split(p, r, tape, obj0, entry, b, s, hold, which, lanes, n_shards, shard_stride, out_stride);
    if (p.robot == PointRobot::kId) return RobotLaunch<PointRobot>::split(p, r, tape, obj0, entry, b, s, hold, which, lanes, n_shards, shard_stride, out_stride);
    if (p.robot == AntRobot::kId) return RobotLaunch<AntRobot>::split(p, r, tape, obj0, entry, b, s, hold, which, lanes, n_shards, shard_stride, out_stride);
    if (p.robot == WalkerRobot::kId) return RobotLaunch<WalkerRobot>::split(p, r, tape, obj0, entry, b, s, hold, which, lanes, n_shards, shard_stride, out_stride);
    return hipErrorNotSupported;
}

void launch_commit_pending(const Params& p, const DevBuffers& b, int nobj_total, int n_rows, hipStream_t s)
{
    GX_ROBOT_DISPATCH(commit_pending(p, b, nobj_total, n_rows, s));
}

void launch_fake_table(const Params& p, const Pool& pl, int nobj_total, int M, hipStream_t s)
{
    if (!pl.fake) return;
    if (p.robot == AntRobot::kId) RobotLaunch<AntRobot>::fake_table(p, pl, nobj_total, M, s);
    else if (p.robot == WalkerRobot::kId) RobotLaunch<WalkerRobot>::fake_table(p, pl, nobj_total, M, s);
}
int fake_table_width(const Params& p)
{
    if (p.robot == AntRobot::kId) return AntRobot::NQ + AntRobot::NV + 4;
    if (p.robot == WalkerRobot::kId) return WalkerRobot::NQ + WalkerRobot::NV + 4;
    return 0;
}

bool policy_rollout_supported(const Params& p) { return p.nobj <= 16 && p.bins <= 16; }
// width 128 in one launch (group_rollout_kernel<.., 3>): the light robots, observation width = the default task's (padded to
// fours: the first layer's k-steps are a compile-time constant there)
bool policy_fused128_supported(const Params& p)
{
    int ddef = 0;
    if (p.robot == PointRobot::kId || p.robot == PointBareRobot::kId) ddef = PointRobot::NQ + PointRobot::NV + PointRobot::NU + 34;
    else if (p.robot == SwimmerRobot::kId) ddef = SwimmerRobot::NQ + SwimmerRobot::NV + SwimmerRobot::NU + 34;
    else return false;
    return policy_rollout_supported(p) && pad4(p.D) == pad4(ddef);
}
size_t policy_lds_bytes(const Params& p, int impl)
{
    return sizeof(float) * (size_t)policy_lds_floats(p.D, p.robot == AntRobot::kId ? AntRobot::NA : (p.robot == WalkerRobot::kId ? WalkerRobot::NA : 2), impl);
}

// impl: 1 = VALU fmaf chains (one wave per workgroup), 2 = fp32 MFMA tiles (16 envs per workgroup)
void launch_policy_rollout(const Params& p, const RolloutArgs& r, const PolicyArgs& pol, const DevBuffers& b,
                           int impl, hipStream_t s)
{
    GX_ROBOT_DISPATCH(policy(p, r, pol, b, impl, s));
}

__global__ void math_probe2_kernel(int n, const float* x, float* lg, float* th)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    lg[i] = log_f(x[i]);
    th[i] = tanh_f(x[i]);
}

void launch_math_probe2(int n, const float* x, float* lg, float* th, hipStream_t s)
{
    hipLaunchKernelGGL(math_probe2_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, x, lg, th);
}

void launch_math_probe(int n, const float* x, const float* y, float* s_, float* c, float* at2,
                       float* ex, hipStream_t s)
{
    hipLaunchKernelGGL(math_probe_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, x, y, s_, c, at2, ex);
}

void launch_split_probe(uint32_t k0, uint32_t k1, int n, uint32_t* out, hipStream_t s)
{
    hipLaunchKernelGGL(split_probe_kernel, dim3((n + 255) / 256), dim3(256), 0, s, k0, k1, n, out);
}

} // namespace gx

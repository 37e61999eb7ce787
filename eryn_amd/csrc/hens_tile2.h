// hens_tile2.h - k_stretch2: the first launch of an iteration for shapes of MORE than one round of workgroups (round 6).
//
// 8 x 16384 x 64 - what each GPU of BASELINE config 3 runs - launches 1 024 tiles of 64 walkers on 512 workgroup slots.  k_stretch_fast
// runs them as two rounds of workgroups that each walk A -> E alone: record / row-table round trip, row gathers, likelihood on the
// matrix pipe, accept chain of one wave, stores - three dependent memory round trips and two single-wave phases per tile, nothing of
// tile n + 1 in flight while tile n computes (LABNOTES 11.2: re-phasing the two resident workgroups of a CU buys nothing, the chain
// itself has to get shorter).  Here a workgroup is PERSISTENT over its tiles (grid = tiles / tiles_per_wg) and software-pipelined:
//
//   * what phase A of tile n + 1 needs from memory - its walkers' records, its complements' row-table entries - is requested, and its
//     Philox draws are computed, in the shadow of tile n's row gathers, by the waves that will need them (the accept wave and the
//     complement wave ALTERNATE between the tiles: waves 0 / 2 for even tiles, 4 / 6 for odd ones, so a tile's {L, P, log u, factors}
//     stay in its accept wave's registers from A to D as they do in k_stretch_fast);
//   * the first half of tile n + 1's row gathers is issued in front of tile n's accept phase (one wave's latency chain: the other
//     seven have nothing else to do) and consumed behind it;
//   * tile n's accepted rows are stored in the shadow of tile n + 1's gathers;
//   * the proposal tile in LDS is double-buffered (2 x 33 kB + 9 kB: two workgroups per CU as before);
//   * the folded ladder adaptation runs once per workgroup, not once per tile.
//
// Same arithmetic per walker, same order, same draws as k_stretch_fast<DT, LIKE, MODE_STRETCH, 8, false, false> in its in-place,
// column-ordered-records form (the only form this kernel has): bit-identical chains (tests/test_hip_records.py, HENS_NO_TILE2=1).
// The ladder-adaptation blocks below are k_stretch_fast's own, token for token (hens_kernels.h: adapt_part1 / adapt_part2 / the ADX
// wave / adapt_early / the ADW block) - keep them in step.
#pragma once
#include "hens_kernels.h"
// passes of tile n + 1's first gather half that go out in FRONT of tile n's likelihood phase (behind an extra barrier), the rest in
// front of its accept phase.  Measured (8 x 16384 x 64 dense, first launch, two alternations; k_stretch_fast's rounds 21.8 - 22.8 us):
// 0 passes 20.65 / 20.77, 1 pass 19.98 / 19.99, 2 passes 21.38 / 21.39 (18 spilled registers: the dense instantiation sits at the
// 128-register bound of two workgroups per CU); diagonal likelihood 13.72 / 13.56 / 13.78 against 16.35.
#ifndef HENS_T2_PREC
#define HENS_T2_PREC 1
#endif
// a pipeline rank's lead workgroup: its adaptation chain behind the first barrier, in the shadow of tile 0's gathers (1), or in front
// of it as in k_stretch_fast<64> (0)
#ifndef HENS_T2_PIPE_SHADOW
#define HENS_T2_PIPE_SHADOW 0
#endif
namespace hens {

__host__ __device__ constexpr size_t tile2_lds_bytes(int D, int like) {
    return ((size_t)2 * TILE * (D + 2) + 8 * TILE + 128 + 2 * TILE) * 8 + (size_t)2 * 4 * TILE * 4 + mf_lds_extra(D, like);
}

// PIPE: the context is a rank of the ladder pipeline stepping with the two in-place launches (pipe_fused_iteration) - k_stretch_fast's
// hooks for it, ported: the head of the launch (rows-complete flag to the cold neighbour, the last sweep's swap counts to every rank's
// mailbox, the wait for the hot neighbour's rows), the lead workgroup's adaptation chain and the ring the others read their rung's
// beta from, guest rows (a walker that arrived through the pipeline sits in the mailbox and goes home here, accepted or not),
// system-scope stores of rows a peer may pull.  The host keeps k_stretch_fast<PIPE> for what is not ported: the separate-launch
// pipeline's publishing of (L, P) (pub_lp), counts pushed through the reduction machinery (cnt_push == 1: adaptation_delay > 0),
// every-workgroup adaptation (ad_on == 1), the latency-injection hook.  cnt_push == 3 (this kernel only; the host's translation of
// cnt_push == 1 on the delayed schedule): the publishing wave pushes the last sweep's counts as with 2, but the adapting wave takes
// every pair - its own rank's too - from the mailbox (the ladder lags a sweep: adaptation_delay = 1).
template <int DT, int LIKE, bool PIPE = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4))) void k_stretch2(const StretchArgs A) {
    static_assert(DT == 64 || DT == 128, "row widths whose tile does not stay centred (phase E reads the proposal from the tile)");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int NW = 8, D = DT, RS = DT + 2, NT = NW * 64, LPR = DT / 2, RPP = NT / LPR, NPASS = (TILE + RPP - 1) / RPP, HP = NPASS / 2;
    constexpr bool CEN = false;
    static_assert(!like_centred(LIKE, DT) && HP >= 1, "");
    double* const qt = reinterpret_cast<double*>(smem_raw);              // [2][TILE][RS]
    double* const s_part = qt + 2 * TILE * RS;                           // [NW][TILE]
    double* const s_beta = s_part + NW * TILE;                           // [128]
    double* const s_zz2 = s_beta + 128;                                  // [2][TILE]
    int32_t* const s_i = reinterpret_cast<int32_t*>(s_zz2 + 2 * TILE);   // [2][4][TILE]: row of the walker, row of its complement, destination row, flags
    double* const s_mu = reinterpret_cast<double*>(s_i + 2 * 4 * TILE);  // [D] dense: mu for the matrix-pipe phase C
    auto S_RS = [&](int b) { return s_i + (b * 4 + 0) * TILE; };
    auto S_RC = [&](int b) { return s_i + (b * 4 + 1) * TILE; };
    auto S_DST = [&](int b) { return s_i + (b * 4 + 2) * TILE; };
    auto S_FLAG = [&](int b) { return s_i + (b * 4 + 3) * TILE; };

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    if constexpr (like_mf<DT, LIKE, NW>()) {
        if (tid >= NW * 64 - DT / 2) *reinterpret_cast<double2*>(s_mu + 2 * (tid - (NW * 64 - DT / 2))) = *reinterpret_cast<const double2*>(A.mu + 2 * (tid - (NW * 64 - DT / 2)));
    }
    MfRegs mfr{};
    constexpr int ADW = 1;
    // the workgroup's tiles: rung tl, tile columns bx, bx + GX, ... (GX = gridDim.x); the XCD-affine numbering of k_stretch_fast over
    // the smaller grid (an XCD still works on whole rungs, and a workgroup's tiles are all of one rung)
    int bx = blockIdx.x, tl = blockIdx.y;
    if (A.xcd_shift > 0) {
        const int sh = A.xcd_shift - 1, L = bx + (tl << sh), g = (L & 7) * ((int)(gridDim.x * gridDim.y) >> 3) + (L >> 3);
        tl = g >> sh; bx = g & ((1 << sh) - 1);
    }
    // Two tiles per workgroup, the loop below fully unrolled: as a rolled loop over a run-time count the same code takes 173 VGPRs
    // (one workgroup per CU), unrolled 130 -> 128 with a handful of spills under the two-workgroups-per-CU bound; larger grids run
    // as rounds of such pairs.
    constexpr int TP = 2;
    const int GX = (int)gridDim.x;
    const int W = A.W;
    const int Ns = A.split == 0 ? A.N0 : W - A.N0;
    const int s_off = A.split == 0 ? 0 : A.N0;
    const bool ad_on = A.ad_on != 0;
    // (one GPU: every workgroup adapts for itself, ad_on == 1; a pipeline rank: workgroup (0,0) adapts and publishes the ring,
    //  ad_on == 2 - the host launches this kernel with nothing else)
    const bool ad_lead = PIPE && ad_on;
    const bool ad_here = ad_on && (!ad_lead || (blockIdx.x == 0 && blockIdx.y == 0));
    const bool ad_early = ad_here && A.ad.nblocks <= 8 * A.ad.row_groups;      // (the host launches this kernel only then)
    const bool ad_defer = ad_early && !PIPE;
    constexpr int PUSHW = 3;                    // pipeline rank, workgroup (0,0): the wave that publishes the last sweep's swap counts
    if (PIPE && blockIdx.x == 0 && blockIdx.y == 0) {          // (k_stretch_fast's head of a rank's launch)
        if (A.rt_flag && tid == 0) pipe_raise(A.rt_flag, A.rt_value);
        if (A.cnt_push >= 2 && wv == PUSHW) pipe_push_counts(A.cp_rows, A.cp_nblocks, A.cp_np, A.cp_boxes, A.cp_nranks, A.cp_rank, A.cp_T,
                                                             A.rung_begin, W, DT, A.cp_sweep, lane, false);
    }
    const bool cnt_wait = PIPE && ad_here && ad_lead && (A.wmask >> PF_CNT0) != 0ull;
    // the hot neighbour's rows of the previous sweep: wave 0 (tile 0's accept wave) waits, the others meet it at the first barrier -
    // nothing in front of that barrier touches a row
    if (PIPE && A.wmask) {
        if (wv == 0 && ((A.wmask >> lane) & 1ull) && lane < PF_CNT0)
            pipe_spin(A.wflags + lane, A.wtarget, A.wbudget, A.flags, A.wstats ? A.wstats : nullptr);
    }
    const bool ad_x = ad_defer && NW >= 4 && A.ad.T <= 64 && A.ad.moving;
    double ad_c0 = 0.0, ad_c1 = 0.0, ad_dT0 = 0.0, ad_dT1 = 0.0, ad_b0n = 1.0, ad_b1n = 1.0, ad_bb0 = 1.0, ad_bb1 = 1.0, ad_inv0 = 1.0;
    auto adapt_part1 = [&](const double cnt0, const double cnt1, const double ad_b, const double ad_b1, const bool exp_elsewhere = false) {
        const int T = A.ad.T;
        const int e0 = lane, e1 = lane + 64;
        ad_c0 = cnt0; ad_c1 = cnt1; ad_bb0 = ad_b; ad_bb1 = ad_b1;
        if (!A.ad.moving) return;
        if (exp_elsewhere) {                 // (T <= 64) the ratio chain - cnt / W, dS, exp - runs on wave ADX; here: 1 / beta differences
            const double inv0v = 1.0 / ad_b;
            ad_inv0 = inv0v;
            ad_b0n = __shfl_down(ad_b, 1);
            const double inv0n = __shfl_down(inv0v, 1);
            ad_dT0 = (e0 + 2 < T) ? inv0n - inv0v : 0.0;                           // :578; times exp(dS) after the barrier
            ad_dT1 = 0.0;
            return;
        }
        const bool two = T > 64;                                                   // wave-uniform: rungs 64.. exist
        const double r0 = cnt0 / (double)A.ad.W, r1 = two ? cnt1 / (double)A.ad.W : 0.0;   // :587
        const double kappa = A.ad.kappa;                                           // :571-572 (host)
        // ONE reciprocal per rung: 1 / beta of the next rung is the next lane's (round 3: the chain had 1 / b twice per lane, and
        // 1 / b[0] once more in the second part - on the path of every launch since the first barrier comes earlier)
        const double inv0v = 1.0 / ad_b, inv1v = two ? 1.0 / ad_b1 : 1.0;
        ad_inv0 = inv0v;
        // the value of the NEXT rung (e + 1): lane 63's successor is rung 64 = lane 0's second element
        // (lane exchanges are LDS crossbar trips: the second rung set's only for ladders above 64 rungs - wave-uniform)
        const double r0d = __shfl_down(r0, 1), b0d = __shfl_down(ad_b, 1), i0d = __shfl_down(inv0v, 1);
        double r0n = r0d, b0n = b0d, inv0n = i0d, r1n = 0.0, b1n = 1.0, inv1n = 1.0;
        if (two) {
            const double r1first = __shfl(r1, 0), b1first = __shfl(ad_b1, 0), i1first = __shfl(inv1v, 0);
            if (lane == 63) { r0n = r1first; b0n = b1first; inv0n = i1first; }
            r1n = __shfl_down(r1, 1); b1n = __shfl_down(ad_b1, 1); inv1n = __shfl_down(inv1v, 1);
        }
        ad_b0n = b0n; ad_b1n = b1n;
        double dT0 = 0.0, dT1 = 0.0;
        if (e0 + 2 < T) {
            const double dS = kappa * (r0 - r0n);                                  // :575
            dT0 = inv0n - inv0v;                                                   // :578  1 / b[e+1] - 1 / b[e]
            dT0 *= exp(dS);
        }
        if (two && e1 + 2 < T) {
            const double dS = kappa * (r1 - r1n);
            dT1 = inv1n - inv1v;
            dT1 *= exp(dS);
        }
        ad_dT0 = dT0; ad_dT1 = dT1;
    };
    auto adapt_part2 = [&]() {
        const int T = A.ad.T;
        const int e0 = lane, e1 = lane + 64;
        const double cnt0 = ad_c0, cnt1 = ad_c1, ad_b = ad_bb0, ad_b1 = ad_bb1;
        double bnew0 = ad_b, bnew1 = ad_b1;
        if (A.ad.moving) {
            const double dT0 = ad_dT0, dT1 = ad_dT1, b0n = ad_b0n, b1n = ad_b1n;
            double cs0 = 0.0, cs1 = 0.0;                                           // np.cumsum: left-to-right
            for (int i = 0; i + 2 < T; ++i) {
                const double v = i < 64 ? readlane_f64(dT0, i) : readlane_f64(dT1, i - 64);
                if (i == 0) { cs0 = v; cs1 = v; }
                else {
                    if (i <= e0) cs0 = cs0 + v;
                    if (i <= e1) cs1 = cs1 + v;
                }
            }
            const double inv0 = readlane_f64(ad_inv0, 0);                          // 1 / b[0]
            const double bn0 = 1.0 / (cs0 + inv0), bn1 = T > 64 ? 1.0 / (cs1 + inv0) : 0.0;   // :580, belong to rungs e + 1
            const double upd0 = b0n + (bn0 - b0n), upd1 = b1n + (bn1 - b1n);      // :583,:593
            const double up0 = __shfl_up(upd0, 1);
            if (e0 >= 1 && e0 + 1 < T) bnew0 = up0;
            if (T > 64) {                                                          // (wave-uniform: the second rung set)
                const double up1 = __shfl_up(upd1, 1), upd0last = readlane_f64(upd0, 63);
                if (e1 + 1 < T) bnew1 = lane >= 1 ? up1 : upd0last;
            }
        }
        if (e0 < T) s_beta[e0] = bnew0;
        if (e1 < T) s_beta[e1] = bnew1;
        if (ad_lead) {       // publish: agent-scope stores (other XCDs read them with agent-scope loads); retire the slot after next
            double* slot = A.ad_ring + (size_t)(A.ad_serial & 3u) * T;
            double* clear = A.ad_ring + (size_t)((A.ad_serial + 2u) & 3u) * T;
            if (e0 < T) {
                __hip_atomic_store(clear + e0, -1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(slot + e0, bnew0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (e1 < T) {
                __hip_atomic_store(clear + e1, -1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(slot + e1, bnew1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (blockIdx.x == 0 && blockIdx.y == 0) {
            // (workgroup (0,0)'s books: the three pointers are read from the kernarg segment HERE on one GPU - as by-value arguments
            //  they sit in SGPRs from the entry of every workgroup's every wave, late_kernarg)
            constexpr size_t AD = offsetof(StretchArgs, ad);
            double* const betas_out_l = PIPE ? A.ad.betas_out : late_kernarg<double*>(AD + offsetof(AdaptArgs, betas_out));
            double* const swaps_last_l = PIPE ? A.ad.swaps_last : late_kernarg<double*>(AD + offsetof(AdaptArgs, swaps_last));
            double* const swaps_total_l = PIPE ? A.ad.swaps_total : late_kernarg<double*>(AD + offsetof(AdaptArgs, swaps_total));
            if (e0 < T) wt_store(&betas_out_l[e0], bnew0);
            if (e1 < T) wt_store(&betas_out_l[e1], bnew1);
            if (e0 < T - 1 && !ad_x) {             // (ad_x: wave ADX, the only reader of the counts then, keeps these books)
                wt_store(&swaps_last_l[e0], cnt0);
                // (a pipeline rank: an atomic without a return value - `+=` is a load this wave, in front of workgroup (0,0)'s first
                //  barrier there, waits a memory round trip for: the rank's first launch 9.4 -> 8.8 us at 16 x 4096 x 32; the counts are
                //  integers, the sum is the same double.  One GPU keeps `+=`: in the gathers' shadow it costs nothing, and the atomic made
                //  config 2 0.1 us SLOWER - 17.40 -> 17.50, four alternations)
                if constexpr (PIPE) atomicAdd(&swaps_total_l[e0], cnt0);
                else wt_store(&swaps_total_l[e0], swaps_total_l[e0] + cnt0);
            }
            if (e1 < T - 1) {
                wt_store(&swaps_last_l[e1], cnt1);
                if constexpr (PIPE) atomicAdd(&swaps_total_l[e1], cnt1);
                else wt_store(&swaps_total_l[e1], swaps_total_l[e1] + cnt1);
            }
        }
    };
    unsigned ad_u0[8], ad_u1[8];
    double ad_bi0 = 1.0, ad_bi1 = 1.0;
    // (column-ordered records, measured at config 2 in workgroup cycles up to the second barrier: both parts in front of the
    //  first barrier 13 070, both in the gathers' shadow 11 520, the split as it is 9 900)
    constexpr bool ad_defer_all = false;
    // Round 3: the first part was the last to reach the first barrier (4 460 cycles after the workgroup's start; the complement
    // rows' wave 2 830, the rest < 2 000): its two independent chains run on two waves - ratios -> dS -> exp on wave ADX,
    // reciprocals of the ladder on wave ADW - and meet through LDS after the barrier (ladders of up to 64 rungs).
    constexpr int ADX = 3;
    double* s_exp = s_part;                  // [64] exp(dS) per rung (phase C overwrites it after the second barrier)
    if (ad_x && wv == ADX) {
        const int T = A.ad.T, NR = A.ad.nblocks;
        const int G = A.ad.row_groups, P2 = 64 / G, p = lane & (P2 - 1), g = lane / P2;
        unsigned u[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) u[r] = (r * G + g < NR && p < T - 1) ? A.ad.swap_part[(size_t)(r * G + g) * (T - 1) + p] : 0u;
        __builtin_amdgcn_s_waitcnt(0x0F70);
        unsigned s0 = 0;
#pragma unroll
        for (int r = 0; r < 8; ++r) s0 += u[r];
        for (int m = P2; m < 64; m <<= 1) s0 += __shfl_xor(s0, m);
        const double r0 = (double)s0 / (double)A.ad.W;                             // :587
        const double r0n = __shfl_down(r0, 1);
        s_exp[lane] = (lane + 2 < T) ? exp(A.ad.kappa * (r0 - r0n)) : 1.0;        // :575
        // This wave is the ONLY reader of the count rows (the reciprocal chain on wave ADW needs none of them: every workgroup
        // reading the same 15 cache lines twice made those loads ~3 000 cycles long and wave ADW the last at the first
        // barrier, 4 150 cycles after the start against the complement wave's 3 000), so workgroup (0,0)'s books are kept here.
        if (blockIdx.x == 0 && blockIdx.y == 0) {
            if (g == 0 && p < T - 1) {
                constexpr size_t AD = offsetof(StretchArgs, ad);
                wt_store(&late_kernarg<double*>(AD + offsetof(AdaptArgs, swaps_last))[p], (double)s0);
                double* const tot = late_kernarg<double*>(AD + offsetof(AdaptArgs, swaps_total));
                wt_store(&tot[p], tot[p] + (double)s0);
            }
            if (A.ad.zero_after) {
#pragma unroll
                for (int r = 0; r < 8; ++r)
                    if (r * G + g < NR && p < T - 1 && u[r]) wt_store(&A.ad.swap_part[(size_t)(r * G + g) * (T - 1) + p], 0u);
            }
            if (A.ad.zero_rows)
                for (int e = lane; e < NR * (T - 1); e += 64) wt_store(&A.ad.zero_rows[e], 0u);
        }
    }
    auto adapt_early = [&]() {
        const int T = A.ad.T, NR = A.ad.nblocks;
        unsigned s0 = 0, s1 = 0;
#pragma unroll
        for (int r = 0; r < 8; ++r) { s0 += ad_u0[r]; s1 += ad_u1[r]; }
        const int G = A.ad.row_groups, P2 = 64 / G;
        for (int m = P2; m < 64; m <<= 1) s0 += __shfl_xor(s0, m);       // (the lane groups' partial sums)
        if (blockIdx.x == 0 && blockIdx.y == 0 && !ad_x) {
            if (A.ad.zero_after) {                   // sole reader (mode 2 / a pipeline rank): clear what was read
                const int p = lane & (P2 - 1), g = lane / P2;
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    if (r * G + g < NR && p < T - 1 && ad_u0[r]) wt_store(&A.ad.swap_part[(size_t)(r * G + g) * (T - 1) + p], 0u);
                    if (r < NR && lane + 64 < T - 1 && ad_u1[r]) wt_store(&A.ad.swap_part[(size_t)r * (T - 1) + lane + 64], 0u);
                }
            }
            if (A.ad.zero_rows)                      // every workgroup reads the rows: clear the buffer of the NEXT sweep
                for (int e = lane; e < NR * (T - 1); e += 64) wt_store(&A.ad.zero_rows[e], 0u);
        }
        adapt_part1((double)s0, (double)s1, ad_bi0, ad_bi1, ad_x);
    };
    // A pipeline rank's adapting wave (workgroup (0,0)), k_stretch_fast's in two pieces: in front of the first barrier only the LOADS
    // (ladder, one look at the other ranks' count flags, its own rank's counts summed out of the accumulation rows, the mailbox's
    // row) - one round trip in the shadow of phase A; the ~4 000-cycle chain behind the barrier, in the shadow of tile 0's first
    // gathers (pipe_chain below).  A persistent workgroup that starts its first tile 2 us late ends 2 us late, and the launch with
    // it: measured with the whole chain in front of the barrier (k_stretch_fast's D = 64 place), a lone rank's first launch at
    // 8 x 16384 x 64 22.8 us against 19.9 for the same shape as a ladder of its own.  What crosses the barrier waits in LDS
    // (s_part: nobody touches it before tile 0's likelihood phase).
    unsigned* const s_late = reinterpret_cast<unsigned*>(s_part);      // [l] late, [64 + l] own sums, [128 + l] / [192 + l] counts of rungs l / l + 64
    double* const s_lateb = s_part + 128;                              // [l] / [64 + l] the ladder
    auto pipe_chain = [&]() {
        const int T = A.ad.T, rb = A.rung_begin, np = A.cp_np;
        const bool own_acc = A.cnt_push == 2;
        unsigned m0 = s_late[128 + lane], m1 = s_late[192 + lane];
        const unsigned own_sum = s_late[64 + lane];
        const double b0 = s_lateb[lane], b1 = s_lateb[64 + lane];
        if (s_late[lane] != 0u) {
            // a rank's counts were not there at the first look: wait for them now
            if (lane >= PF_CNT0 && lane != PF_CNT0 + A.cp_rank && ((A.wmask >> lane) & 1ull))
                pipe_spin(A.wflags + lane, A.wtarget_cnt, A.wbudget, A.flags, A.wstats ? A.wstats + 2 : nullptr);
            if (!(own_acc && A.cp_nranks == 1)) {
                if (lane < T - 1) m0 = A.ad.swap_part[lane];
                if (lane + 64 < T - 1) m1 = A.ad.swap_part[lane + 64];
            }
        }
        if (own_acc) {                           // my pairs out of my own sums: global pair e = rung_begin + local pair
            const int j0 = lane - rb, j1 = lane + 64 - rb;
            const unsigned o0 = (unsigned)__shfl((int)own_sum, (j0 >= 0 && j0 < np) ? j0 : 0);
            const unsigned o1 = (unsigned)__shfl((int)own_sum, (j1 >= 0 && j1 < np) ? j1 : 0);
            if (j0 >= 0 && j0 < np) m0 = o0;
            if (j1 >= 0 && j1 < np) m1 = o1;
        }
        adapt_part1((double)m0, (double)m1, b0, b1);
        adapt_part2();
    };
    if constexpr (PIPE) {
        if (ad_early && ad_lead && wv == ADW) {
            const int T = A.ad.T;
            const double b0 = (lane < T) ? A.ad.betas_in[lane] : 1.0, b1 = (lane + 64 < T) ? A.ad.betas_in[lane + 64] : 1.0;
            uint32_t fl = 0xFFFFFFFFu;
            if (cnt_wait && lane >= PF_CNT0 && lane != PF_CNT0 + A.cp_rank && ((A.wmask >> lane) & 1ull))
                fl = __hip_atomic_load(A.wflags + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            unsigned own_sum = 0;                    // lane p: this rank's count of local pair p (cnt_push == 2)
            if (A.cnt_push == 2) own_sum = acc_rows_sum(A.cp_rows, A.cp_nblocks, A.cp_np, lane);
            bool late = false;
            if (cnt_wait) {
                const bool here = fl >= A.wtarget_cnt;
                late = __ballot(!here) != 0ull;
                if (A.inject_c64 < 0) late = true;       // (HENS_PIPE_FORCE_LATE=1, tests: every adaptation takes the late path)
                __atomic_signal_fence(__ATOMIC_SEQ_CST);          // (acquire side: see pipe_spin)
            }
            unsigned m0 = 0, m1 = 0;                 // (row_groups = 1: the mailbox's reduced counts, one row)
            if (!late && !(A.cnt_push == 2 && A.cp_nranks == 1)) {
                if (lane < T - 1) m0 = A.ad.swap_part[lane];
                if (lane + 64 < T - 1) m1 = A.ad.swap_part[lane + 64];
            }
            s_late[lane] = late ? 1u : 0u;           // (every lane its own word: see k_stretch_fast)
            s_late[64 + lane] = own_sum; s_late[128 + lane] = m0; s_late[192 + lane] = m1;
            s_lateb[lane] = b0; s_lateb[64 + lane] = b1;
#if !HENS_T2_PIPE_SHADOW
            pipe_chain();
#endif
        }
    }
    if (ad_early && wv == ADW && !(PIPE && ad_lead)) {
        const int T = A.ad.T, NR = A.ad.nblocks;
        // (row_groups G > 1 - ladders of at most 64 / G pairs: lane = (group g, pair p), group g sums rows g, g + G, ...)
        const int G = A.ad.row_groups, P2 = 64 / G, p = lane & (P2 - 1), g = lane / P2;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            ad_u0[r] = (!ad_x && r * G + g < NR && p < T - 1) ? A.ad.swap_part[(size_t)(r * G + g) * (T - 1) + p] : 0u;
            ad_u1[r] = (!ad_x && r < NR && lane + 64 < T - 1) ? A.ad.swap_part[(size_t)r * (T - 1) + lane + 64] : 0u;
        }
        if (lane < T) ad_bi0 = A.ad.betas_in[lane];
        if (lane + 64 < T) ad_bi1 = A.ad.betas_in[lane + 64];
        // Wait for these loads HERE (the wave has nothing else to do before the first barrier), with the builtin the
        // compiler's wait-count pass understands: otherwise it guards the deferred computation with a wait that also
        // covers the row gathers issued in between, and the adaptation no longer overlaps them.
        __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0), expcnt / lgkmcnt untouched
        // first part now: this wave has nothing else to do before the barrier - unless the barrier comes early (column-ordered
        // records: phase A is one coalesced load), then everything waits for the shadow of the row gathers
        if (!ad_defer_all) adapt_early();
        if (!ad_defer) adapt_part2();
    }

    // ---- what an accept wave keeps of its tile from phase A to phase D (k_stretch_fast: wave 0's registers) ----------------------
    double factors = 0.0, lu = 0.0, Lold = 0.0, Pold = 0.0, beta_pre = 1.0, zz_mine = 1.0;
    int own = 0;
    uint32_t acc_old = 0;
    bool valid = false;
    int32_t rs_mine = 0, rc_mine = 0;
    int32_t ghome_row = 0;                   // pipeline rank: the home row of a walker that sits in a guest row (rs_mine < 0)
    if (A.tempered && !ad_on) beta_pre = A.betas[A.rung_begin + tl];

    // phase A of tile j, first part: requests and draws (accept wave: the walker's record, one Philox call -> zz, log u, the Hastings
    // factor; complement wave: the same call -> the complement's column -> its row out of the rung's compact table)
    auto phaseA_request = [&](const int j) {
        const int k0 = (bx + j * GX) * TILE;
        const int aw = (j & 1) * 4, cw = aw + 2;
        if (wv == aw) {
            const int k = k0 + lane;
            valid = k < Ns;
            zz_mine = 1.0; rs_mine = 0;
            if (valid) {
                own = place_column(A.split, k, A.hb_shift);
                const WalkerRec* o = A.wrec + (tl * W + own);
                const double2 lp = *reinterpret_cast<const double2*>(&o->L);
                const int2 la = *reinterpret_cast<const int2*>(&o->loc);
                const StretchDraw sd = stretch_draw(A.iseed, A.iiter, (uint32_t)(A.rung_begin + tl) * (uint32_t)W + (uint32_t)(s_off + k));
                zz_mine = draw_zz(sd.uz, A.ia);
                lu = log(sd.ua);                                 // red_blue.py:294
                factors = ((double)A.ndim_active - 1.0) * log(zz_mine);           // stretch.py:223
                rs_mine = la.x;
                acc_old = (uint32_t)la.y;
                if (PIPE && A.ghome) ghome_row = A.ghome[la.x < 0 ? ~la.x : 0];   // (tile 0: consumed in phase D - no wait in front of the barrier)
                Lold = lp.x; Pold = lp.y;
            }
        } else if (wv == cw) {
            const int k = k0 + lane;
            rc_mine = 0;
            if (k < Ns) {
                const StretchDraw sd = stretch_draw(A.iseed, A.iiter, (uint32_t)(A.rung_begin + tl) * (uint32_t)W + (uint32_t)(s_off + k));
                const int colc = place_column(1 - A.split, stretch_index(sd.r22, W >> 1), A.hb_shift);
                rc_mine = A.loc[tl * W + colc];
            }
        }
    };
    // ... second part: into the tile's LDS arrays (buffer j & 1), where every wave's gather passes read them
    auto phaseA_publish = [&](const int j) {
        const int b = j & 1, aw = b * 4, cw = aw + 2;
        if (wv == aw) {
            s_zz2[b * TILE + lane] = zz_mine;
            S_RS(b)[lane] = rs_mine;
            // (in place: an accepted proposal overwrites the walker's row.  A guest of a pipeline rank goes home, accepted or not:
            //  tiles behind the first have had its home row for a whole tile - flag 16, the proposal phase stores the old row there
            //  and phase E only accepted rows as everywhere; tile 0 learns it in phase D as in k_stretch_fast - flag 8, phase E)
            const bool guest_known = PIPE && j > 0 && A.ghome && rs_mine < 0 && valid;
            S_DST(b)[lane] = guest_known ? ghome_row : rs_mine;
            S_FLAG(b)[lane] = (valid ? 4 : 0) | (guest_known ? 16 : 0);
        } else if (wv == cw) {
            S_RC(b)[lane] = rc_mine;
        }
    };

    phaseA_request(0);
    phaseA_publish(0);
    lds_barrier();
    // (the accumulation rows the publishing and the adapting wave have read: cleared behind the first barrier)
    if (PIPE && A.cnt_push >= 2 && A.cp_zero && wv == PUSHW && blockIdx.x == 0 && blockIdx.y == 0)
        acc_rows_clear(const_cast<uint32_t*>(A.cp_rows), A.cp_nblocks, A.cp_np, lane);

    // ---- the tile loop -------------------------------------------------------------------------------------------------------------
    const int jl = tid & (LPR - 1);
    const int rsub = tid / LPR;
    const double* __restrict__ pool_r = A.pool;
    double* __restrict__ pool_w = A.pool;
    const bool sysw = PIPE && (tl == A.sys_rung || A.sys_all);           // (rows a peer may pull: a property of the launch / the rung)
    double2 sreg[NPASS], creg[NPASS];
    bool rv[NPASS];
#define T2_GATHER(p, b, k0)                                                                                                     \
    {                                                                                                                           \
        const int r = p * RPP + rsub;                                                                                           \
        rv[p] = (r < TILE) && (k0 + r < Ns);                                                                                    \
        sreg[p] = double2{0.0, 0.0};                                                                                            \
        creg[p] = double2{0.0, 0.0};                                                                                            \
        if (rv[p]) {                                                                                                            \
            sreg[p] = *reinterpret_cast<const double2*>(pool_r + (PIPE ? row_off(rs_i[p], D, A.guest_delta) : (int64_t)rs_i[p] * D) + jl * 2); \
            creg[p] = *reinterpret_cast<const double2*>(pool_r + (PIPE ? row_off(rc_i[p], D, A.guest_delta) : (int64_t)rc_i[p] * D) + jl * 2); \
        }                                                                                                                       \
    }
#define T2_PROPOSE(p, b)                                                                                                        \
    {                                                                                                                           \
        const int r = p * RPP + rsub;                                                                                           \
        bool ok = true, finite = true;                                                                                          \
        if (PIPE && b > 0 && rv[p] && rs_i[p] < 0 && A.ghome) { /* a guest whose home row is known (flag 16): the old row goes home now */ \
            double* const hd = pool_w + (size_t)S_DST(b)[r] * D + jl * 2;                                                       \
            if (sysw) { sys_store(hd, sreg[p].x); sys_store(hd + 1, sreg[p].y); }                                               \
            else { wt_store(hd, sreg[p].x); wt_store(hd + 1, sreg[p].y); }                                                      \
        }                                                                                                                       \
        if (rv[p]) {                                                                                                            \
            const double zz = s_zz2[b * TILE + r];                                                                              \
            double2 qv;                                                                                                         \
            qv.x = creg[p].x - (creg[p].x - sreg[p].x) * zz; /* stretch.py:143,145 */                                           \
            qv.y = creg[p].y - (creg[p].y - sreg[p].y) * zz;                                                                    \
            ok = (qv.x >= lov.x) && (qv.x <= hiv.x) && (qv.y >= lov.y) && (qv.y <= hiv.y);                                      \
            finite = (fabs(qv.x) < INFINITY) && (fabs(qv.y) < INFINITY);                                                        \
            *reinterpret_cast<double2*>(qt + (size_t)b * TILE * RS + r * RS + jl * 2) = qv;                                     \
        }                                                                                                                       \
        const unsigned long long bad = __ballot(!ok); /* prior.py:80-88, row-wide AND */                                        \
        const unsigned long long nonfin = __ballot(!finite);                                                                    \
        const int gshift = lane & ~(LPR - 1);                                                                                   \
        const unsigned long long gmask = (LPR == 64) ? ~0ull : (((1ull << (LPR & 63)) - 1ull) << gshift);                       \
        if (jl == 0 && rv[p]) {                                                                                                 \
            if ((bad & gmask) == 0ull) atomicOr(&S_FLAG(b)[r], 1);                                                              \
            if ((nonfin & gmask) != 0ull) atomicOr(A.flags, FLAG_NONFINITE_X);                                                  \
        }                                                                                                                       \
    }
    // phase E of tile j: accepted rows only, in place, out of the tile (k_stretch_fast's phase E)
    auto phaseE = [&](const int j) {
        const int b = j & 1, k0 = (bx + j * GX) * TILE;
        if constexpr (PIPE) {
            // A pipeline rank: rows a peer may pull are written at system scope; a guest of tile 0 (flag 8, set in phase D) goes home
            // accepted or not - its old row is read again (the registers it was gathered into carry the next tile's rows by now;
            // this phase runs in the shadow of that tile's gathers).  Values and addresses of all passes first, then the stores back
            // to back (see k_stretch_fast's phase E).
            double2 val[NPASS];
            double* dstp[NPASS];
            bool on[NPASS];
#pragma unroll
            for (int p = 0; p < NPASS; ++p) {
                const int r = p * RPP + rsub;
                on[p] = false;
                val[p] = double2{0.0, 0.0};
                dstp[p] = pool_w;
                if (!((r < TILE) && (k0 + r < Ns))) continue;
                const int fl = S_FLAG(b)[r];
                val[p] = *reinterpret_cast<const double2*>(qt + (size_t)b * TILE * RS + r * RS + jl * 2);
                if ((fl & (2 | 8)) == 8) val[p] = *reinterpret_cast<const double2*>(pool_r + row_off(S_RS(b)[r], D, A.guest_delta) + jl * 2);
                on[p] = (fl & (2 | 8)) != 0;
                dstp[p] = pool_w + (size_t)S_DST(b)[r] * D + jl * 2;
            }
            if (sysw) {
#pragma unroll
                for (int p = 0; p < NPASS; ++p) if (on[p]) store_row16_sys(dstp[p], val[p]);
            } else {
#pragma unroll
                for (int p = 0; p < NPASS; ++p) if (on[p]) store_row16(dstp[p], val[p]);
            }
        } else {
#pragma unroll
            for (int p = 0; p < NPASS; ++p) {
                const int r = p * RPP + rsub;
                if (!((r < TILE) && (k0 + r < Ns))) continue;
                if ((S_FLAG(b)[r] & 2) == 0) continue;
                const double2 qv = *reinterpret_cast<const double2*>(qt + (size_t)b * TILE * RS + r * RS + jl * 2);
                store_row16(pool_w + (size_t)S_DST(b)[r] * D + jl * 2, qv);
            }
        }
    };

    int rs_i[NPASS], rc_i[NPASS];
    {   // tile 0: the first half of its gathers, and in their shadow the adaptation's second part (once per workgroup)
        const int k00 = bx * TILE;
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {
            const int rr = p * RPP + rsub < TILE ? p * RPP + rsub : 0;
            rs_i[p] = S_RS(0)[rr];
            rc_i[p] = S_RC(0)[rr];
        }
#pragma unroll
        for (int p = 0; p < HP; ++p) T2_GATHER(p, 0, k00)
        if (ad_defer && wv == ADW) {
            if (ad_x && lane + 2 < A.ad.T) ad_dT0 *= s_exp[lane];
            adapt_part2();
        }
#if HENS_T2_PIPE_SHADOW
        if constexpr (PIPE) {
            if (ad_early && ad_lead && wv == ADW) pipe_chain();
        }
#endif
    }
#pragma unroll
    for (int j = 0; j < TP; ++j) {
        const int b = j & 1, k0 = (bx + j * GX) * TILE;
        const bool more = j + 1 < TP;
        // ---- phase B of tile j: the first half of its gathers has been in flight since before tile j - 1's accept phase (tile 0: since
        // the prologue).  In their shadow: phase A of tile j + 1 (requests + draws) and the stores of tile j - 1's accepted rows.
#pragma unroll
        for (int p = HP; p < NPASS; ++p) {                       // (the second half's row indices)
            const int rr = p * RPP + rsub < TILE ? p * RPP + rsub : 0;
            rs_i[p] = S_RS(b)[rr];
            rc_i[p] = S_RC(b)[rr];
        }
        // (the box of this lane's two coordinates: requested per tile - across the likelihood phase they cost the dense instantiation
        //  eight registers it does not have)
        const double2 lov = *reinterpret_cast<const double2*>(A.lo + jl * 2);
        const double2 hiv = *reinterpret_cast<const double2*>(A.hi + jl * 2);
        if (more) phaseA_request(j + 1);
        if (j > 0) phaseE(j - 1);
#pragma unroll
        for (int p = 0; p < HP; ++p) T2_PROPOSE(p, b)
#pragma unroll
        for (int p = HP; p < NPASS; ++p) T2_GATHER(p, b, k0)
#pragma unroll
        for (int p = HP; p < NPASS; ++p) T2_PROPOSE(p, b)
        mfr = like_prefetch<DT, LIKE, NW>(lane, wv, A.prec_sym);
        lds_barrier();
        // (tile j - 1's phase E has read the other buffer's flags / destinations in front of this barrier: they are free now)
        if (more) phaseA_publish(j + 1);
        // a pipeline rank: the rung's new beta out of the lead workgroup's ring, requested now and consumed behind the likelihood
        double beta_ring = -1.0;
        const double* ring_slot = nullptr;
        const bool ring_me = PIPE && ad_lead && !ad_here;
        if (ring_me && wv == b * 4) {
            ring_slot = A.ad_ring + (size_t)(A.ad_serial & 3u) * A.ad.T + (A.rung_begin + tl);
            beta_ring = __hip_atomic_load(ring_slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#if HENS_T2_PREC > 0
        if (more) {
            lds_barrier();
            const int b1 = b ^ 1, k1 = (bx + (j + 1) * GX) * TILE;
#pragma unroll
            for (int p = 0; p < HP; ++p) {
                const int rr = p * RPP + rsub < TILE ? p * RPP + rsub : 0;
                rs_i[p] = S_RS(b1)[rr];
                rc_i[p] = S_RC(b1)[rr];
            }
#pragma unroll
            for (int p = 0; p < HENS_T2_PREC; ++p) T2_GATHER(p, b1, k1)
        }
#endif
        // ---- phase C: the likelihood of tile j (matrix pipe at dense D = 64 / 128), every wave its partial sums
        {
            const bool inbox = (S_FLAG(b)[lane] & 1) != 0;
            like_partials<DT, LIKE, NW, CEN>(qt + (size_t)b * TILE * RS, s_part, lane, wv, inbox, like_mf<DT, LIKE, NW>() ? s_mu : A.mu, A.prec, A.prec_sym, A.rosen_a, A.rosen_b, mfr);
        }
        lds_barrier();
        // ---- the first half of tile j + 1's gathers goes out in front of tile j's accept phase
        if (more) {
            const int b1 = b ^ 1, k1 = (bx + (j + 1) * GX) * TILE;
#pragma unroll
            for (int p = 0; p < HP; ++p) {
                const int rr = p * RPP + rsub < TILE ? p * RPP + rsub : 0;
                rs_i[p] = S_RS(b1)[rr];
                rc_i[p] = S_RC(b1)[rr];
            }
#if HENS_T2_PREC > 0
#pragma unroll
            for (int p = HENS_T2_PREC; p < HP; ++p) T2_GATHER(p, b1, k1)
#else
#pragma unroll
            for (int p = 0; p < HP; ++p) T2_GATHER(p, b1, k1)
#endif
        }
        // ---- phase D: accept / update of tile j, on its accept wave (k_stretch_fast's phase D, record mode, in place)
        if (wv == b * 4 && valid) {
            const bool inbox = (S_FLAG(b)[lane] & 1) != 0;
            double acc = 0.0;
#pragma unroll
            for (int w2 = 0; w2 < like_nparts<DT, LIKE, NW>(); ++w2) acc += s_part[w2 * TILE + lane];
            double logl = inbox ? -0.5 * acc : A.fill;             // ensemble.py:1486-1513
            if (logl != logl) {                                    // red_blue.py:279-281
                logl = -1e300;
                atomicOr(A.flags, FLAG_NAN_LOGL);
            }
            const double logp = inbox ? A.logp_in : -INFINITY;     // prior.py:80-88
            const size_t gi = (size_t)tl * W + own;
            double logP, prevP;
            if (A.tempered) {                                      // tempering.py:304-306,343-349
                double beta = beta_pre;
                if (ring_me) {                                 // workgroup (0,0) may still be adapting: wait for the value
                    if (beta_ring < 0.0) {                     // (the budget runs on the shader clock: see spin_expired)
                        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
                        while (beta_ring < 0.0) {
                            __builtin_amdgcn_s_sleep(1);
                            beta_ring = __hip_atomic_load(ring_slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (spin_expired(t0, 200000000LL)) { atomicOr(A.flags, FLAG_PIPE_TIMEOUT); break; }
                        }
                    }
                    beta = beta_ring;
                } else if (ad_on) {
                    beta = s_beta[A.rung_begin + tl];
                }
                double lt = logl * beta;
                if (lt != lt) lt = -INFINITY;
                logP = lt + logp;
                double lo_ = Lold * beta;
                if (lo_ != lo_) lo_ = -INFINITY;
                prevP = lo_ + Pold;
            } else {                                               // move.py:443-457
                logP = logl + logp;
                prevP = Lold + Pold;
            }
            const double lnpdiff = factors + logP - prevP;         // red_blue.py:292
            const bool keep = lnpdiff > lu;                        // red_blue.py:294
            const double newP = (fabs(logp) == INFINITY) ? 0.0 : logp;
            if (keep) {                                            // move.py:513-532
                if (PIPE || late_kernarg<int32_t>(offsetof(StretchArgs, norel))) {
                    store_row16(&A.wrec[gi].L, double2{logl, newP});
                    wt_store(&A.wrec[gi].acc, acc_old + 1u);
                } else {
                    *reinterpret_cast<double2*>(&A.wrec[gi].L) = double2{logl, newP};
                    A.wrec[gi].acc = acc_old + 1u;
                }
                atomicOr(&S_FLAG(b)[lane], 2);
            }
            if (PIPE && A.ghome && rs_mine < 0) {              // a guest: its row - new or old - goes to its home row
                A.wrec[gi].loc = ghome_row;
                A.loc[gi] = ghome_row;
                if (j == 0) {                                  // (tile 0: phase E moves the row; later tiles' old rows are home already)
                    S_DST(b)[lane] = ghome_row;
                    atomicOr(&S_FLAG(b)[lane], 8);
                }
            }
            if (A.keep_out) A.keep_out[(size_t)tl * Ns + k0 + lane] = keep ? 1 : 0;
        }
        lds_barrier();
    }
    phaseE(TP - 1);
    if constexpr (!PIPE) if (late_kernarg<int32_t>(offsetof(StretchArgs, norel))) launch_end_wait();
#undef T2_GATHER
#undef T2_PROPOSE
}

}  // namespace hens

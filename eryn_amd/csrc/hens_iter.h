// hens_iter.h - one launch per iteration for latency-bound shapes (k_iter).
//
// At config-2 size (16 x 4096 x 32: 65 536 walkers, 26 MB of rows per half-step) an iteration is not bound by bytes but
// by its chain of dependent phases: two launches, each {indices -> rows -> likelihood -> accept -> store} with every
// workgroup of the chip in lockstep, plus two launch boundaries.  k_split1_pt already merged the second
// half-step with the cascade by giving a workgroup a block of cascade columns.  The first half-step could not join it:
// a second-half walker's complement is a random FIRST-half walker of its rung (red_blue.py:183-193, stretch.py:93-99),
// whose update belongs to some other workgroup.
//
// But that dependency is ONE walker deep: the complement c1 of a second-half walker m2 moves in the first half-step
// against a complement c2 that is again a second-half walker - untouched by the first half-step.  So the workgroup that
// owns m2 REPLAYS c1's first half-step itself from the pre-iteration state (rows of c1 and c2, c1's draws, c1's
// {L, P}): the same instructions on the same inputs give the same proposal, the same likelihood bits and the same
// accept decision as the workgroup that owns c1 computes.  With that every workgroup needs only the state as it was
// BEFORE the launch:
//   stage 1   first half-step of the block's 64 first-half walkers (tile A), replay for the complements of its 64
//             second-half walkers (tile X: results used, nothing stored)
//   stage 2   second half-step of its 64 second-half walkers (tile B) against the replayed rows
//   cascade   of its columns, permuted records out (as k_split1_pt)
// = 1.25 x the row gathers and 1.5 x the likelihood evaluations of the two-launch iteration, for one launch boundary,
// one ramp / drain and one chain of index hops instead of two.  A trade for shapes whose launches are a single round of
// workgroups; larger shapes (config 3's shards), where launches overlap their own phases, keep the two launches.
//
// Reading only pre-launch state while other workgroups write needs versioned rows: every walker owns the pool rows
// {loc, loc ^ half}; an accepted proposal goes to the OTHER row and the walker's record points there afterwards, a
// rejected one writes nothing.  Nobody reads the other row during the launch (the records every workgroup reads are the
// pre-launch buffer), and the next launch reads through the new records.  The pool always had the second half (the
// copying launches' homes); leaving record mode folds the rows back into one half (k_fold_rows).
#pragma once

namespace hens {

struct IterArgs {
    double* pool;
    const WalkerRec* wrec;                                    // pre-iteration state (read-only for the launch)
    WalkerRec* wrecnew;                                       // after the cascade
    const int32_t* loc;                                       // [T][W] rows once more, compact (complement lookups)
    int32_t* locnew;
    const DrawRec* rec1;                                      // [W / cb][64] first half-step draws in block order
    const DrawRec* rec2;                                      // [W / cb][64] second half-step draws in block order
    const DrawRec* rec3;                                      // [W / cb][64] first half-step draws of rec2[i].cw
    const uint32_t* keys;                                     // [T][8]
    uint32_t* accepted;                                       // [T][W]
    uint32_t* swap_acc;                                       // [acc_rows][T-1] this cascade's counts (clean on entry)
    const double* betas;                                      // [T] (ad_on == 0)
    const double* lo; const double* hi; const double* mu; const double* prec; const double* prec_sym;
    const double* period;                                     // [D] periodic parameters (StretchArgs::period), PER instantiations
    unsigned* flags;
    unsigned long long* trace;
    double logp_in, fill, rosen_a, rosen_b;
    uint64_t iter, seed;
    int32_t T, W, idx_bits, cb, cb_shift;
    int32_t acc_rows;                                         // rows of swap_acc (FusedArgs::acc_rows)
    int32_t norel;                                            // see StretchArgs::norel
    int32_t half_rows;                                        // T * W: rows r and r ^ half belong to the same walker
    int32_t ad_on;                                            // the previous cascade's ladder adaptation rides in this launch
    AdaptArgs ad;                                             //   (every workgroup recomputes it from the accumulated counts)
};

__host__ __device__ inline size_t iter_lds_bytes(int D, int NW) {
    return ((size_t)3 * TILE * (D + 2) + (size_t)2 * NW * TILE + 3 * TILE + 3 * 2 * TILE + 128 + 32) * 8 +
           ((size_t)2 * 2 * TILE + 3 * TILE + 2 * TILE + 3 * TILE + 2 * TILE + 64) * 4;
}

// The ladder adaptation of one wavefront (tempering.py:563-596; the arithmetic of k_stretch_fast's folded form), T <= 128:
// lane l owns rungs l and l + 64.  part1 (ratios, dS, exp, deltaT) needs the counts only; part2 (cumulative sum,
// reciprocals, update) runs while the row gathers are in flight.
struct LadderFold {
    double c0 = 0.0, c1 = 0.0, dT0 = 0.0, dT1 = 0.0, b0n = 1.0, b1n = 1.0, bb0 = 1.0, bb1 = 1.0;
    __device__ __forceinline__ void part1(const AdaptArgs& ad, int lane, double cnt0, double cnt1, double b, double b1) {
        const int T = ad.T;
        const int e0 = lane, e1 = lane + 64;
        c0 = cnt0; c1 = cnt1; bb0 = b; bb1 = b1;
        if (!ad.moving) return;
        const bool two = T > 64;
        const double r0 = cnt0 / (double)ad.W, r1 = two ? cnt1 / (double)ad.W : 0.0;     // :587
        const double decay = ad.lag / ((double)ad.time + ad.lag);                        // :571
        const double kappa = decay / ad.nu;                                              // :572
        const double r0d = __shfl_down(r0, 1), r1first = __shfl(r1, 0);
        const double b0d = __shfl_down(b, 1), b1first = __shfl(b1, 0);
        const double r0n = lane < 63 ? r0d : r1first, b0n_ = lane < 63 ? b0d : b1first;
        const double r1n = __shfl_down(r1, 1), b1n_ = __shfl_down(b1, 1);
        b0n = b0n_; b1n = b1n_;
        double d0 = 0.0, d1 = 0.0;
        if (e0 + 2 < T) {
            const double dS = kappa * (r0 - r0n);                                        // :575
            d0 = 1.0 / b0n_ - 1.0 / b;                                                   // :578
            d0 *= exp(dS);
        }
        if (two && e1 + 2 < T) {
            const double dS = kappa * (r1 - r1n);
            d1 = 1.0 / b1n_ - 1.0 / b1;
            d1 *= exp(dS);
        }
        dT0 = d0; dT1 = d1;
    }
    __device__ __forceinline__ void part2(const AdaptArgs& ad, int lane, double* s_beta, bool lead) {
        const int T = ad.T;
        const int e0 = lane, e1 = lane + 64;
        double bnew0 = bb0, bnew1 = bb1;
        if (ad.moving) {
            double cs0 = 0.0, cs1 = 0.0;                                                 // np.cumsum: left-to-right
            for (int i = 0; i + 2 < T; ++i) {
                const double v = i < 64 ? readlane_f64(dT0, i) : readlane_f64(dT1, i - 64);
                if (i == 0) { cs0 = v; cs1 = v; }
                else {
                    if (i <= e0) cs0 = cs0 + v;
                    if (i <= e1) cs1 = cs1 + v;
                }
            }
            const double inv0 = 1.0 / __shfl(bb0, 0);
            const double bn0 = 1.0 / (cs0 + inv0), bn1 = T > 64 ? 1.0 / (cs1 + inv0) : 0.0;   // :580
            const double upd0 = b0n + (bn0 - b0n), upd1 = b1n + (bn1 - b1n);             // :583,:593
            const double up0 = __shfl_up(upd0, 1), up1 = __shfl_up(upd1, 1), upd0last = __shfl(upd0, 63);
            if (e0 >= 1 && e0 + 1 < T) bnew0 = up0;
            if (e1 + 1 < T) bnew1 = lane >= 1 ? up1 : upd0last;
        }
        if (e0 < T) s_beta[e0] = bnew0;
        if (e1 < T) s_beta[e1] = bnew1;
        if (lead) {
            if (e0 < T) wt_store(&ad.betas_out[e0], bnew0);        // (written through: see wt_store, hens_kernels.h)
            if (e1 < T) wt_store(&ad.betas_out[e1], bnew1);
            if (e0 < T - 1) {
                wt_store(&ad.swaps_last[e0], c0);
                wt_store(&ad.swaps_total[e0], ad.swaps_total[e0] + c0);
            }
            if (e1 < T - 1) {
                wt_store(&ad.swaps_last[e1], c1);
                wt_store(&ad.swaps_total[e1], ad.swaps_total[e1] + c1);
            }
        }
    }
};

// PER: periodic parameters (hens_set_periodic), an instantiation of its own as in k_stretch_fast / k_split1_pt
template <int DT, int LIKE, int NW, bool PER = false>
__global__ __launch_bounds__(NW * 64) void k_iter(const IterArgs A) {
    static_assert(DT == 16 || DT == 32, "row widths with three tiles in half a CU's LDS");
    static_assert(NW == 8, "one wave per role in the index phase");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int D = DT, RS = DT + 2, NT = NW * 64, LPR = DT / 2, RPP = NT / LPR, NPASS = (TILE + RPP - 1) / RPP;
    constexpr int NE = 2 * TILE;
    // the likelihood's rows are dealt to the waves exactly as in k_stretch_fast / k_split1_pt for this row width (4 waves at
    // D = 16): the same partial sums in the same order, so an iteration gives the same bits whichever path runs it
    constexpr int LNW = DT >= 32 ? NW : 4;
    double* tileA = reinterpret_cast<double*>(smem_raw);                 // [TILE][RS] first half-step proposals
    double* tileX = tileA + TILE * RS;                                   // replayed first half-step of the complements
    double* tileB = tileX + TILE * RS;                                   // second half-step proposals
    double* s_part = tileB + TILE * RS;                                  // [2][NW][TILE]
    double* s_zz = s_part + 2 * NW * TILE;                               // [3][TILE]  A, X, B
    double* Lc = s_zz + 3 * TILE;                                        // [NE] cascade tables, element e = t * cb + cc
    double* Pc = Lc + NE;
    double* lupt = Pc + NE;                                              // [NE]
    double* sbeta = lupt + NE;                                           // [128]
    double* s_mu = sbeta + 128;                                          // [32] D = 32 dense: mu for the matrix-pipe likelihood (like_tile_mf32)
    int32_t* locc = reinterpret_cast<int32_t*>(s_mu + 32);               // [NE]
    int32_t* scol = locc + NE;                                           // [NE]
    int32_t* s_rs = scol + NE;                                           // [3][TILE] own row: A, X, B
    int32_t* s_rc = s_rs + 3 * TILE;                                     // [2][TILE] complement row: A, X
    int32_t* s_flag = s_rc + 2 * TILE;                                   // [3][TILE] bit0 inbox, bit1 keep
    int32_t* s_el = s_flag + 3 * TILE;                                   // [NE] accept counter of the slot of element e
    uint32_t* smask = reinterpret_cast<uint32_t*>(s_el + 2 * TILE);      // [cb][MW]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int T = A.T, W = A.W, CB = A.cb, CS = A.cb_shift, HB = CB >> 1;
    const int c0 = blockIdx.x * CB;
    const int MW = (T + 31) >> 5;
    const int H = A.half_rows;
    const bool lead = blockIdx.x == 0;
    // ladders whose length does not divide 128: NEr = cb T <= 128 slots, NM = NEr / 2 <= 64 walkers per half-step; the
    // lanes / rows beyond them idle (see k_split1_pt)
    const int NEr = T << CS, NM = NEr >> 1;
#define ITER_TRACE(i) do { if (A.trace && tid == 0) A.trace[(size_t)blockIdx.x * 8 + (i)] = trace_stamp(); } while (0)
    ITER_TRACE(0);
    const MfRegs mfr = like_prefetch<DT, LIKE, NW>(lane, wv, A.prec_sym);   // (D = 32 dense: the matrix operands of the three likelihood phases, requested first)
    if constexpr (like_mf<DT, LIKE, NW>())
        if (tid >= NT - DT / 2) *reinterpret_cast<double2*>(s_mu + 2 * (tid - (NT - DT / 2))) = *reinterpret_cast<const double2*>(A.mu + 2 * (tid - (NT - DT / 2)));

    // ---- phase A: one wave per role ---------------------------------------------------------------------------
    //   waves 0 / 1 / 2 : lane m = the block's m-th first-half walker / the complement of its m-th second-half walker /
    //                     its m-th second-half walker: draw record (coalesced, block order) -> walker record, complement row
    //   waves 3, 4      : the cascade's log-uniforms
    //   wave 5          : ladder (adaptation of the previous cascade, first part)
    //   waves 6, 7      : column map of the 128 slots (where phase G writes, which element a walker is)
    double Lold = 0.0, Pold = 0.0, fac = 0.0, lu = 0.0;                  // waves 0-2, lane m
    int32_t gi_m = 0;
    uint32_t acc_m = 0;                                                  // the slot's accept counter (WalkerRec::acc)
    const int tm = lane >> (CS - 1);                                     // rung of the m-th walker of a half
    LadderFold fold;
    if (wv < 3) s_flag[wv * TILE + lane] = 0;
    if (wv < 3 && lane < NM) {
        const DrawRec* src = wv == 0 ? A.rec1 : (wv == 1 ? A.rec3 : A.rec2);
        const DrawRec rc = src[(size_t)blockIdx.x * TILE + lane];
        gi_m = tm * W + rc.own;
        const int32_t rs = A.wrec[gi_m].loc;
        acc_m = A.wrec[gi_m].acc;
        int32_t rcw = 0;
        if (wv < 2) rcw = A.loc[tm * W + rc.cw];
        const double2 lp = *reinterpret_cast<const double2*>(&A.wrec[gi_m].L);
        Lold = lp.x; Pold = lp.y; fac = rc.fac; lu = rc.lu;
        s_rs[wv * TILE + lane] = rs;
        if (wv < 2) s_rc[wv * TILE + lane] = rcw;
        s_zz[wv * TILE + lane] = rc.zz;
    } else if (wv < 3) {
        // (idle lanes of a short tile)
    } else if (wv < 5) {
        const int e = tid - 3 * 64, t = e >> CS, c = c0 + (e & (CB - 1));
        if (t < T - 1) lupt[e] = log(pt_uniform(A.seed, A.iter, t, W, c));   // tempering.py:535 (row j = t: pair T-1-t)
    } else if (wv == 5) {
        if (A.ad_on) {
            const int NR = A.ad.nblocks;
            const int G = A.ad.row_groups, P2 = 64 / G, p = lane & (P2 - 1), g = lane / P2;   // (see k_stretch_fast)
            unsigned u0[8], u1[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                u0[r] = (r * G + g < NR && p < T - 1) ? A.ad.swap_part[(size_t)(r * G + g) * (T - 1) + p] : 0u;
                u1[r] = (r < NR && lane + 64 < T - 1) ? A.ad.swap_part[(size_t)r * (T - 1) + lane + 64] : 0u;
            }
            double b0 = 1.0, b1 = 1.0;
            if (lane < T) b0 = A.ad.betas_in[lane];
            if (lane + 64 < T) b1 = A.ad.betas_in[lane + 64];
            __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0) here: the deferred part must not wait for the row gathers
            unsigned s0 = 0, s1 = 0;
#pragma unroll
            for (int r = 0; r < 8; ++r) { s0 += u0[r]; s1 += u1[r]; }
            for (int m = P2; m < 64; m <<= 1) s0 += __shfl_xor(s0, m);
            if (lead && A.ad.zero_rows)              // the buffer nobody reads or writes during this launch
                for (int e = lane; e < NR * (T - 1); e += 64) wt_store(&A.ad.zero_rows[e], 0u);
            fold.part1(A.ad, lane, (double)s0, (double)s1, b0, b1);
        } else {
            if (lane < T) sbeta[lane] = A.betas[lane];
            if (lane + 64 < T) sbeta[lane + 64] = A.betas[lane + 64];
        }
    } else if (tid - 6 * 64 < NEr) {
        const int e = tid - 6 * 64, t = e >> CS, cc = e & (CB - 1), c = c0 + cc;
        const uint4* kp = reinterpret_cast<const uint4*>(A.keys) + (size_t)t * 2;
        const uint4 ka = kp[0], kb = kp[1];
        const uint32_t key[8] = {ka.x, ka.y, ka.z, ka.w, kb.x, kb.y, kb.z, kb.w};
        scol[e] = (int)prp((uint32_t)c, key, A.idx_bits, (uint32_t)W);
    }
    ITER_TRACE(1);
    lds_barrier();

    // ---- phase B1: lanes over d, every row the launch reads is requested here -------------------------------------
    const int jl = tid & (LPR - 1);
    const int rsub = tid / LPR;
    const double* __restrict__ pool_r = A.pool;
    double2 sA[NPASS], cA[NPASS], sX[NPASS], cX[NPASS], sB[NPASS];
    bool rv[NPASS];
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
        const int r = p * RPP + rsub;
        rv[p] = r < NM;
        sA[p] = cA[p] = double2{0.0, 0.0};
        if (rv[p]) {
            sA[p] = *reinterpret_cast<const double2*>(pool_r + (int64_t)s_rs[r] * D + jl * 2);
            cA[p] = *reinterpret_cast<const double2*>(pool_r + (int64_t)s_rc[r] * D + jl * 2);
        }
    }
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
        const int r = p * RPP + rsub;
        sX[p] = cX[p] = double2{0.0, 0.0};
        if (rv[p]) {
            sX[p] = *reinterpret_cast<const double2*>(pool_r + (int64_t)s_rs[TILE + r] * D + jl * 2);
            cX[p] = *reinterpret_cast<const double2*>(pool_r + (int64_t)s_rc[TILE + r] * D + jl * 2);
        }
    }
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
        const int r = p * RPP + rsub;
        sB[p] = double2{0.0, 0.0};
        if (rv[p]) sB[p] = *reinterpret_cast<const double2*>(pool_r + (int64_t)s_rs[2 * TILE + r] * D + jl * 2);
    }
    const double2 lov = *reinterpret_cast<const double2*>(A.lo + jl * 2);
    const double2 hiv = *reinterpret_cast<const double2*>(A.hi + jl * 2);
    if (A.ad_on && wv == 5) fold.part2(A.ad, lane, sbeta, lead);        // the row gathers are in flight
    const int gshift = lane & ~(LPR - 1);
    const unsigned long long gmask = (LPR == 64) ? ~0ull : (((1ull << (LPR & 63)) - 1ull) << gshift);
    // q = c - (c - s) zz (stretch.py:143,145), box test by ballot over the row's lanes (prior.py:80-88)
    auto propose = [&](const double2 s, const double2 c, const double zz, double* tile, int32_t* flag, const int r, const bool on) {
        bool ok = true, finite = true;
        if (on) {
            double2 qv;
            if (PER) {                                                   // periodic parameters: stretch.py:136-154
                const double2 pv = *reinterpret_cast<const double2*>(A.period + jl * 2);
                qv.x = periodic_wrap(c.x - periodic_diff(s.x, c.x, pv.x) * zz, pv.x);
                qv.y = periodic_wrap(c.y - periodic_diff(s.y, c.y, pv.y) * zz, pv.y);
            } else {
                qv.x = c.x - (c.x - s.x) * zz;
                qv.y = c.y - (c.y - s.y) * zz;
            }
            ok = (qv.x >= lov.x) && (qv.x <= hiv.x) && (qv.y >= lov.y) && (qv.y <= hiv.y);
            finite = (fabs(qv.x) < INFINITY) && (fabs(qv.y) < INFINITY);
            *reinterpret_cast<double2*>(tile + r * RS + jl * 2) = qv;
        }
        const unsigned long long bad = __ballot(!ok);
        const unsigned long long nonfin = __ballot(!finite);
        if (jl == 0 && on) {
            if ((bad & gmask) == 0ull) atomicOr(&flag[r], 1);
            if ((nonfin & gmask) != 0ull) atomicOr(A.flags, FLAG_NONFINITE_X);
        }
    };
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
        const int r = p * RPP + rsub;
        propose(sA[p], cA[p], rv[p] ? s_zz[r] : 1.0, tileA, s_flag, r, rv[p]);
    }
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
        const int r = p * RPP + rsub;
        propose(sX[p], cX[p], rv[p] ? s_zz[TILE + r] : 1.0, tileX, s_flag + TILE, r, rv[p]);
    }
    ITER_TRACE(2);
    lds_barrier();

    // ---- phase C1: likelihood of the first half-step proposals and of the replayed ones ---------------------------
    if constexpr (like_mf<DT, LIKE, NW>()) {
        like_tile_mf32<false>(tileA, s_part, lane, wv, s_mu, mfr);
        like_tile_mf32<false>(tileX, s_part + NW * TILE, lane, wv, s_mu, mfr);
    } else {
        const bool inA = (s_flag[lane] & 1) != 0, inX = (s_flag[TILE + lane] & 1) != 0;
        if (wv < LNW) {
            s_part[wv * TILE + lane] = like_partial<DT, LIKE, LNW, false>(tileA, lane, wv, inA, A.mu, A.prec, A.prec_sym, A.rosen_a, A.rosen_b);
            s_part[(NW + wv) * TILE + lane] = like_partial<DT, LIKE, LNW, false>(tileX, lane, wv, inX, A.mu, A.prec, A.prec_sym, A.rosen_a, A.rosen_b);
        }
    }
    ITER_TRACE(3);
    lds_barrier();

    // ---- phase D: tempered accept test (red_blue.py:285-308, move.py:513-532), lane per walker ----------------------
    auto accept = [&](const double* part, const int32_t flagw, double& logl_out, double& newP_out) -> bool {
        const bool inbox = (flagw & 1) != 0;
        double acc = 0.0;
#pragma unroll
        for (int w2 = 0; w2 < like_nparts<DT, LIKE, LNW>(); ++w2) acc += part[w2 * TILE + lane];
        double logl = inbox ? -0.5 * acc : A.fill;                      // ensemble.py:1486-1513
        if (logl != logl) {                                             // red_blue.py:279-281
            logl = -1e300;
            atomicOr(A.flags, FLAG_NAN_LOGL);
        }
        const double logp = inbox ? A.logp_in : -INFINITY;              // prior.py:80-88
        const double beta = sbeta[tm];
        double lt = logl * beta;                                        // tempering.py:304-306,343-349
        if (lt != lt) lt = -INFINITY;
        const double logP = lt + logp;
        double lo_ = Lold * beta;
        if (lo_ != lo_) lo_ = -INFINITY;
        const double prevP = lo_ + Pold;
        const double lnpdiff = fac + logP - prevP;                      // red_blue.py:292
        logl_out = logl;
        newP_out = (fabs(logp) == INFINITY) ? 0.0 : logp;               // move.py:513-532
        return lnpdiff > lu;                                            // red_blue.py:294
    };
    auto alt_row = [&](const int32_t r) -> int32_t { return r < H ? r + H : r - H; };
    if (wv == 0 && lane < NM) {                                          // the block's own first-half walkers: results count
        double logl, newP;
        const bool keep = accept(s_part, s_flag[lane], logl, newP);
        const int e = (tm << CS) + (lane & (HB - 1));                   // first-half walkers: the block's first cb/2 columns
        const int32_t rs = s_rs[lane];
        Lc[e] = keep ? logl : Lold;
        Pc[e] = keep ? newP : Pold;
        locc[e] = keep ? alt_row(rs) : rs;
        s_el[e] = (int32_t)(acc_m + (keep ? 1u : 0u));                  // (every slot moves once per iteration)
        if (keep) s_flag[lane] |= 2;
    } else if (wv == 1 && lane < NM) {                                   // replay: only the decision is needed
        double logl, newP;
        if (accept(s_part + NW * TILE, s_flag[TILE + lane], logl, newP)) s_flag[TILE + lane] |= 2;
    }
    lds_barrier();

    // ---- phase B2: second half-step proposals against the complements as the first half-step left them ------------
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
        const int r = p * RPP + rsub;
        double2 c = sX[p];
        if (rv[p] && (s_flag[TILE + r] & 2)) c = *reinterpret_cast<const double2*>(tileX + r * RS + jl * 2);
        propose(sB[p], c, rv[p] ? s_zz[2 * TILE + r] : 1.0, tileB, s_flag + 2 * TILE, r, rv[p]);
    }
    ITER_TRACE(4);
    lds_barrier();

    // ---- phase C2 / D2 ------------------------------------------------------------------------------------------
    if constexpr (like_mf<DT, LIKE, NW>()) {
        like_tile_mf32<false>(tileB, s_part, lane, wv, s_mu, mfr);
    } else {
        const bool inB = (s_flag[2 * TILE + lane] & 1) != 0;
        if (wv < LNW) s_part[wv * TILE + lane] = like_partial<DT, LIKE, LNW, false>(tileB, lane, wv, inB, A.mu, A.prec, A.prec_sym, A.rosen_a, A.rosen_b);
    }
    ITER_TRACE(5);
    lds_barrier();
    if (wv == 2 && lane < NM) {
        double logl, newP;
        const bool keep = accept(s_part, s_flag[2 * TILE + lane], logl, newP);
        const int e = (tm << CS) + HB + (lane & (HB - 1));
        const int32_t rs = s_rs[2 * TILE + lane];
        Lc[e] = keep ? logl : Lold;
        Pc[e] = keep ? newP : Pold;
        locc[e] = keep ? alt_row(rs) : rs;
        s_el[e] = (int32_t)(acc_m + (keep ? 1u : 0u));
        if (keep) s_flag[2 * TILE + lane] |= 2;
    }
    lds_barrier();
    ITER_TRACE(6);

    // ---- phase F: one lane per column walks hot -> cold (tempering.py:515-541); phase E in its shadow ----------------
    auto walk = [&](auto tt) {
        constexpr int TT = decltype(tt)::value;                          // 0: runtime ladder length
        const int Tn = TT ? TT : T;
        const int cc = lane;
        double cL = Lc[((Tn - 1) << CS) + cc];
        uint32_t m = 0;
#pragma unroll
        for (int i0 = Tn - 1; i0 >= 1; i0 -= 8) {
            double Lb[8], lv[8], db[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int i = (i0 - q >= 1) ? i0 - q : 1;
                Lb[q] = Lc[((i - 1) << CS) + cc];
                lv[q] = lupt[((Tn - 1 - i) << CS) + cc];
                db[q] = sbeta[i - 1] - sbeta[i];                         // tempering.py:518-522
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int i = i0 - q;
                if (i >= 1) {
                    const double pacc = db[q] * (cL - Lb[q]);            // tempering.py:538
                    const bool sw = pacc > lv[q];                        // tempering.py:541
                    m |= sw ? (1u << (i & 31)) : 0u;
                    cL = sw ? cL : Lb[q];
                    if ((i & 31) == 0 || i == 1) {
                        smask[cc * MW + (i >> 5)] = m;
                        m = 0;
                    }
                }
            }
        }
    };
    const bool walking = wv == 1;
    auto store_accepted = [&]() {                                        // accepted rows go to the walker's OTHER row
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {
            const int r = p * RPP + rsub;
            if (!rv[p]) continue;
            if (s_flag[r] & 2)
                store_row16(A.pool + (size_t)alt_row(s_rs[r]) * D + jl * 2, *reinterpret_cast<const double2*>(tileA + r * RS + jl * 2));
            if (s_flag[2 * TILE + r] & 2)
                store_row16(A.pool + (size_t)alt_row(s_rs[2 * TILE + r]) * D + jl * 2, *reinterpret_cast<const double2*>(tileB + r * RS + jl * 2));
        }
    };
    if (!walking) store_accepted();
    if (walking && lane < CB) {
        if (T == 16) walk(std::integral_constant<int, 16>{});
        else if (T == 8) walk(std::integral_constant<int, 8>{});
        else if (T == 32) walk(std::integral_constant<int, 32>{});
        else walk(std::integral_constant<int, 0>{});
    }
    lds_barrier();

    // ---- phase G: permuted records of the 128 slots, swap counts ------------------------------------------------
    auto bit = [&](int cc, int i) -> bool { return (i >= 1 && i < T) && ((smask[cc * MW + (i >> 5)] >> (i & 31)) & 1u); };
    const bool paired = A.norel != 0;            // (two lanes per record, one 32-byte sector: see k_split1_pt's phase G)
    if (paired ? tid < 2 * NEr : tid < NEr) {
        const int e = paired ? tid >> 1 : tid, t = e >> CS, cc = e & (CB - 1);
        int st;
        if (MW == 1) {
            const uint32_t mw = smask[cc];
            if ((mw >> t) & 1u) st = t - 1;
            else st = t + __builtin_ctz(~(mw >> 1 >> t));
        } else if (bit(cc, t)) {
            st = t - 1;
        } else {
            st = t;
            while (bit(cc, st + 1)) ++st;
        }
        const int se = (st << CS) + cc;
        const size_t di = (size_t)t * W + scol[e];
        if (paired) {
            const int h = tid & 1;
            const double2 v = h == 0 ? double2{Lc[se], Pc[se]} : double2{__hiloint2double(s_el[e], locc[se]), 0.0};
            store_row16(reinterpret_cast<double*>(&A.wrecnew[di]) + 2 * h, v);
            if (h == 1) wt_store(&A.locnew[di], locc[se]);
        } else {
            A.wrecnew[di] = make_wrec(Lc[se], Pc[se], locc[se], (uint32_t)s_el[e]);
            A.locnew[di] = locc[se];
        }
    }
    for (int i = 1 + tid; i < T; i += NT) {
        unsigned n = 0;
        for (int cc = 0; cc < CB; ++cc) n += bit(cc, i) ? 1u : 0u;
        if (n) atomicAdd(&A.swap_acc[(size_t)(blockIdx.x & (A.acc_rows - 1)) * (T - 1) + (i - 1)], n);
    }
    if (walking) store_accepted();
    if (A.norel) launch_end_wait();          // (this launch's packet carries no release fence - see wt_store)
    ITER_TRACE(7);
#undef ITER_TRACE
}

// Leaving record mode after k_iter launches: walkers whose current row sits in the half the copying launches write next
// (`free_lo` .. `free_lo + half`) move back to their other row, so that all rows live in one half again.
inline __global__ void k_fold_rows(double* __restrict__ pool, int32_t* __restrict__ loc, int64_t n, int D, int32_t half, int32_t free_lo) {
    const int lpr = D / 2;
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t i = g / lpr;
    const int j = (int)(g - i * lpr);
    if (i >= n) return;
    const int32_t r = loc[i];
    if (r < free_lo || r >= free_lo + half) return;
    const int32_t o = r < half ? r + half : r - half;
    // (the D / 2 lanes of a row sit in one wavefront - D / 2 divides 64 - and a wavefront runs in program order: every lane
    //  has loaded loc[i] before lane 0 rewrites it)
    *reinterpret_cast<double2*>(pool + (int64_t)o * D + j * 2) = *reinterpret_cast<const double2*>(pool + (int64_t)r * D + j * 2);
    if (j == 0) loc[i] = o;
}

}  // namespace hens
